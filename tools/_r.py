import importlib, sys, torch, statistics
sys.path.insert(0, ".")
lm = importlib.import_module("gym-lmaze_amd")
def run(env, acts, ar, hint, steps=24):
    env.params.launch_hint = hint
    for i in range(6): env.step(acts[i % len(acts)], auto_reset=ar)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step(acts[i % len(acts)], auto_reset=ar)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3
H = lm.LmazeVecEnv.launch_hint_of
for G, var, N in ((11, "v0", 1 << 20), (11, "v3", 1 << 20), (18, "v3", 1 << 19), (14, "v3", 1 << 20)):
    acts = torch.randint(0, 4, (48, N), dtype=torch.int32, device="cuda")
    env = lm.LmazeVecEnv(N, variant=var, layout=lm.layouts.open_room(G))
    for ar in (False, True):
        cands = [("default", 0)]
        for sel in (1, 2, 3) if G in (11, 12) else (1, 2):
            for c, m in ((8,1),(6,1),(5,1),(4,1),(3,1),(2,1),(8,2),(5,2),(4,2),(3,2),(2,2),(4,3)):
                cands.append(("s%d %dx%d" % (sel, c, m), H(c, m, sel)))
        res = {k: [] for k, _ in cands}
        for r in range(3):
            for name, h in cands:
                res[name].append(run(env, acts, ar, h))
        med = {k: round(statistics.median(v), 1) for k, v in res.items()}
        best = sorted(med.items(), key=lambda kv: kv[1])[:8]
        print("G", G, var, "AR", ar, "default", med["default"], "best", best, flush=True)
    del env, acts
