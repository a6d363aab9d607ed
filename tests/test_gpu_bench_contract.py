"""GPU: the one JSON line `bench.py` owes the driver -- keys, types and internal consistency -- on small runs of
every workload (seconds each).  stdout must hold that line and nothing else."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
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
    ("--workload", "v5", "--envs", "262144", "--steps", "30", "--warmup", "12", "--cpu-baseline-seconds", "0.5"),
    ("--workload", "c2", "--one-launch", "--auto-reset", "--steps", "64", "--warmup", "4", "--no-cpu-baseline"),
    ("--obs-dtype", "u8", "--envs", "262144", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"),
    ("--workload", "v4", "--envs", "262144", "--steps", "12", "--warmup", "4", "--no-cpu-baseline"),
    ("--workload", "v4", "--envs", "262144", "--steps", "12", "--warmup", "60", "--auto-reset", "--no-cpu-baseline"),
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
    # (a rollout whose planes never leave L2 can exceed the HBM "roofline": the line says so)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < (1.5 if "--one-launch" in flags else 1.0)
    assert ("on-die" in r.get("note", "")) == ("--one-launch" in flags)
    # ONE clock: the roofline fraction follows from the line's own ms_per_step ...
    per_gpu = envs // d["n_gpus"]
    assert abs(r["frac"] - per_gpu * r["bytes_per_env_step"] / (d["ms_per_step"] * 1e-3) / 1e9 / r["peak"]) <= 0.005 * r["frac"]
    # ... and the HIP-event figure sits beside it under its own name
    assert abs(r["achieved_events"] - per_gpu * r["bytes_per_env_step"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) <= 1e-6 * r["achieved_events"]
    assert abs(r["frac_events"] - r["achieved_events"] / r["peak"]) < 1e-12 and "ms_per_step" in r["clock"]
    assert r["traffic"] is None or r["traffic"] >= 0.9 * (envs // d["n_gpus"]) * r["bytes_per_env_step"]
    assert set(r["measured_ceiling"]) >= {"fill", "copy", "unit"}
    if "--no-cpu-baseline" in flags:
        assert "cpu_baseline" not in d
    else:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "env-steps/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
        assert c["single_thread"]["cores"] == 1 and c["single_thread"]["value"] > 0 and c["single_thread"]["kind"] == "port"
        ri = c["reference_interpreter"]
        assert ri["kind"] == "reference" and ri["value"] > 0 and "NOT this box" in ri["sample"]
        if "--workload" not in flags or flags[flags.index("--workload") + 1].startswith("c"):
            assert c["numpy_vectorised"]["value"] > 0
    # the regime does not depend on --steps: the action ring is larger than the Infinity Cache (c2: SURVEY's 256 rows)
    rows = int(d["config"]["actions"].split("int32[")[1].split(",")[0])
    wl = flags[flags.index("--workload") + 1] if "--workload" in flags else "c3"
    assert rows == 256 if wl == "c2" else rows * d["config"]["envs_per_gpu"] * 4 >= (320 << 20)
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"] and len(d["per_rank_ms_per_step"]["all"]) == 1
    assert d["config"]["world_size_seen"] == 1 and "traffic_source" in r and "perenv_kernel" in d["config"]
    assert (r["traffic"] is None) == (r["traffic_source"] is None)
    if wl in ("v2", "v5"):          # the foveal workloads pick their launch policy before the timed region, as C3 does
        assert len(d["config"]["autotune_ms"]) >= 5 and ("0x%02x" % d["config"]["launch_hint"]) in d["config"]["autotune_ms"]
    # roofline.kernel is what the launcher picked (lmaze_describe_step), and config.workload says what ran
    k = r["kernel"]
    assert k.startswith("lmaze::") and " grid=" in k and "envs_per_workgroup=" in k
    if "--one-launch" in flags:
        assert "ONE lmaze_rollout call" in k and d["config"]["one_launch_rollout"] is True
    if "--obs-dtype" in flags:
        assert d["dtype"] == "u8" and r["bytes_per_env_step"] == 37 + 121 and "step_shared_u8_kernel" in k and "NOT the configuration" in d["metric"]
    if wl == "c2":
        assert "step_shared_wave8_kernel<v0" in k and "8x8 literal" in d["config"]["workload"] and "open" not in d["config"]["workload"]
    if wl == "c3" and "--obs-dtype" not in flags:
        assert "step_shared_kernel<11, v0, step" in k and "open room" in d["config"]["workload"]
    if wl == "c5":
        assert "wave/register-tiled" in d["config"]["perenv_kernel"] and "step_perenv_wave_kernel<32" in k
        assert "uniformly chosen free cell" in d["config"]["workload"]
    if wl in ("v2", "v5"):
        assert "foveal_kernel<%s, step" % wl in k
    if wl == "v5":
        ev = d["config"]["v5_events_in_timed_steps"]
        total = d["config"]["envs_per_gpu"] * d["steps"]
        assert ev["resets"] + ev["visit_updates"] + ev["window_gathers"] == total and 0 < ev["resets"] < ev["planner_steps"] < total
        # round 3: a step touches only the window -- about 1.46 KB per env-step on these events, not round 2's 1.7 KB
        assert 1432 < r["bytes_per_env_step"] < 1560
        assert 0.05 < d["config"]["local_done_rate"] < 0.6           # the natural rate, not 1.0
    if wl == "v4":
        ev = d["config"]["v4_events_in_timed_steps"]
        total = d["config"]["envs_per_gpu"] * d["steps"]
        assert 5 * total < ev["previous_only_cells"] < 16 * total      # a uniform 25-way teleport leaves ~10 cells outside the overlap
        assert (ev["resets"] > 0) == ("--auto-reset" in flags)
        assert 953 + 20 < r["bytes_per_env_step"] < 953 + 64 + (60 if "--auto-reset" in flags else 0)
    if "--workload" not in flags and "--obs-dtype" not in flags:
        assert d["metric"].startswith("env steps/sec (whole node), 1M parallel 11x11 mazes")
        assert d["config"]["envs_per_gpu"] == 1 << 20 and d["config"]["grid"] == 11 and r["bytes_per_env_step"] == 521


def test_plain_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell (no torchrun): the parent spawns two ranks before touching the
    GPU; here they share the one visible device and meet over gloo (the rehearsal backend), on a real node each
    takes its own device over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LMAZE_BENCH_BACKEND"] = "gloo"
    d = _run("--gpus", "2", "--steps", "20", "--warmup", "5", "--envs", "262144", "--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 2 and d["config"]["world_size_seen"] == 2 and d["config"]["global_envs"] == 2 * 262144
    assert "self" in d["config"]["launcher"] and "REHEARSAL" in d["config"]["collective_backend"]
    pr = d["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) < 1e-9
    # the N > 1 line explains its own stragglers: every rank's event time, launch policy and roofline fraction
    q = d["per_rank"]
    assert all(len(q[k]) == 2 for k in ("ms_per_step", "kernel_ms_avg", "launch_hint", "roofline_frac", "roofline_frac_events"))
    assert abs(min(q["roofline_frac"]) - d["roofline"]["frac"]) <= 2e-4 and all(0 < f < 1 for f in q["roofline_frac_events"])
    assert all(k > 0 for k in q["kernel_ms_avg"]) and all(isinstance(h, int) for h in q["launch_hint"])
    assert abs(d["value"] - d["config"]["global_envs"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-6 * d["value"]
    assert "cpu_baseline" not in d


def test_torchrun_entry_still_works():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "10", "--warmup", "3", "--envs", "262144", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["collective_backend"] == "rccl" and "external" in d["config"]["launcher"]
