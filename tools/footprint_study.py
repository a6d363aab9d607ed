import importlib, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("gym-lmaze_amd")
N, G, T = 1 << 20, 11, 300
lay = pkg.layouts.to_codes(pkg.layouts.open_room(G, (5, 5)))
envs = [pkg.LmazeVecEnv(N, variant="v0", layout=lay, seed=s, online_autotune=False) for s in range(4)]
acts = torch.randint(0, 4, (T, N), dtype=torch.int32, device="cuda")
def timed(fn, reps):
    fn(30); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(reps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for _ in range(200): envs[0].step_raw(acts[0].data_ptr())
for hint in (0x23, 0x28, 0x18):
    for e in envs: e.params.launch_hint = hint
    for k in (1, 2, 4):
        us = timed(lambda r: [envs[t % k].step_raw(acts[t % T].data_ptr()) for t in range(r)], 300)
        print(json.dumps({"hint": hex(hint), "envs_round_robin": k, "footprint_GB": round(k * 0.57, 2), "us_per_step": round(us, 2)}), flush=True)
