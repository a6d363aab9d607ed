# Convenience targets; the driver uses __graft_entry__.py, bench.py and pytest directly.
PY ?= python

.PHONY: build test test-gpu bench golden golden-check install profile clean

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

golden-check:     ## the committed fixtures are what the committed generator emits (byte for byte, any generator order)
	$(PY) -m pytest tests/test_golden_provenance.py -q

install:          ## as the reference: pip install -e . (builds liblmaze_hip.so, installs gym_lmaze + gym_lmaze_amd)
	$(PY) -m pip install --no-build-isolation --no-deps -e .

profile:          ## on the GPU box: re-record profiles/rNN (rocprofv3 stats + PMC passes); R=2 make profile
	bash tools/profile_round.sh $(or $(R),2)

clean:
	$(MAKE) -C gym-lmaze_amd/csrc clean
	$(MAKE) -C oracle clean
