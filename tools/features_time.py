"""dcc_obs_features at a given shape, by output subset (which part of the kernel costs what).
Run on the GPU box: python tools/features_time.py [N M E]"""
import sys, numpy as np, torch
sys.path.insert(0, "dynamic-coverage-control_amd")
import dcc_hip
N, M, E = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 1024, 2048)
poi = np.random.RandomState(0).uniform(-1, 1, (M, 2))
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.1, 0.95, 0.0)
env.reset()
st = env.alloc_state_out(2)
env.rollout(2, seed=1, out=dict(st, reward=torch.empty(2, E, device="cuda")))
state = [st[k][-1].contiguous() for k in ("state_pos", "state_vel", "state_energy", "state_done")]
full = env.obs_features(*state)


def t(keys):
    out = {k: full[k] for k in keys}
    for _ in range(5):
        env.obs_features(*state, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        env.obs_features(*state, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


for keys in (("head", "poi_feat", "stats", "cstats", "xa", "xc"), ("head", "stats", "cstats", "xa", "xc"), ("stats",), ("cstats",),
             ("head",), ("xa",), ("xc",), ("head", "xa", "xc"), ("poi_feat",)):
    print("%-50s %.1f us" % ("+".join(keys), t(keys)))
