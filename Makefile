# Convenience targets; the driver uses __graft_entry__.py, bench.py and pytest directly.
PY ?= python

.PHONY: build test test-gpu bench golden clean

build:            ## hipcc (gfx950) -> gym-lmaze_amd/liblmaze_hip.so ; gcc -> oracle/liblmaze_oracle.so
	$(PY) -c "import __graft_entry__ as g; g.build()"

test:             ## CPU suite: oracle == reference fixtures, host logic, ABI export, gloo world_size 2
	$(PY) -m pytest tests -x -q -m "not gpu"

test-gpu:         ## MI355X: parity of the HIP path against fixtures and oracle, through the C ABI
	$(PY) -m pytest tests -x -q -m gpu

bench:            ## the metric: env-steps/s + HBM roofline fraction + CPU baseline, one JSON line
	$(PY) bench.py

golden:           ## regenerate tests/golden/*.npz from the reference's own step() (needs /root/reference)
	$(PY) oracle/gen_golden.py

clean:
	$(MAKE) -C gym-lmaze_amd/csrc clean
	$(MAKE) -C oracle clean
