"""What the per-step outputs other than the rows cost the c2 launch: the same env, the same observation buffer, 150-step launches with
(a) every output, (b) no assignment rows, (c) no [K,E] scalars (reward / done / connect / connect_s / coverage), (d) rows only.
usage: python tools/small_outputs_cost.py [rounds]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch, dcc_hip
N, M, T, E = 8, 64, 150, 4096
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ["DCC_AUTOTUNE"] = "0"
poi = np.load(R + "/dynamic-coverage-control_amd/envs/mpe/pos_pois.npy")[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi); env.reset()
full = env.alloc_out(T, placed=6)
sc = ("reward", "done", "connect", "connect_s", "coverage")
outs = {"all": full, "no assign": {k: v for k, v in full.items() if k != "assign"},
        "no scalars": {k: v for k, v in full.items() if k not in sc}, "rows only": {"obs": full["obs"]}}
acts = torch.rand(T, E, N, 2, device="cuda") * 2 - 1


def timed(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


for hbm in (False, True):
    res = {k: [] for k in outs}
    for r in range(rounds):
        for k, o in outs.items():
            res[k].append(timed(lambda: env.rollout(T, actions=acts if hbm else None, seed=0, step0=0, env0=0, env_total=E, out=o)))
    print("actions %s: " % ("hbm" if hbm else "rng") + "   ".join("%s %.4f ms" % (k, sorted(v)[len(v) // 2]) for k, v in res.items()), flush=True)
