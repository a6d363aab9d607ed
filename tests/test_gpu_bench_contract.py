"""GPU: the one JSON line `bench.py` owes the driver -- keys, types and internal consistency -- on small runs of
every workload (seconds each).  stdout must hold that line and nothing else."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[:2000]
    return json.loads(lines[0])


@pytest.mark.parametrize("flags", [
    ("--steps", "12", "--warmup", "3", "--cpu-baseline-seconds", "0.5"),                      # the metric's config, short
    ("--workload", "c2", "--steps", "12", "--warmup", "3", "--cpu-baseline-seconds", "0.5"),
    ("--workload", "c5", "--envs", "65536", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"),
    ("--workload", "v2", "--envs", "262144", "--steps", "8", "--warmup", "2", "--cpu-baseline-seconds", "0.5"),
    ("--workload", "c2", "--graph", "--auto-reset", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"),
])
def test_bench_line(flags):
    d = _run(*flags)
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "env-steps/s" and d["n_gpus"] == 1 and d["data"] == "synthetic"
    assert d["steps"] == int(flags[flags.index("--steps") + 1]) and isinstance(d["config"]["workload"], str)
    envs = d["config"]["global_envs"]
    assert abs(d["value"] - envs * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - envs * r["bytes_per_env_step"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * envs * r["bytes_per_env_step"]
    assert set(r["measured_ceiling"]) >= {"fill", "copy", "unit"}
    if "--no-cpu-baseline" in flags:
        assert "cpu_baseline" not in d
    else:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    if "--workload" not in flags:
        assert d["metric"].startswith("env steps/sec (whole node), 1M parallel 11x11 mazes")
        assert d["config"]["envs_per_gpu"] == 1 << 20 and d["config"]["grid"] == 11 and r["bytes_per_env_step"] == 521
