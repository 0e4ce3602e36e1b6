"""Do D2D memcpy nodes (dcc_env_get_state inside the captured rollout warm-up) survive many replays?  A graph holding one
env step + get_state (4 hipMemcpyAsync nodes) is replayed N times next to an eagerly driven twin env; states must stay equal.
usage: python tools/graph_memcpy_soak.py [replays]"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import dcc_hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
dev = torch.device("cuda", 0)
E, N, M = 64, 4, 20
poi = np.random.RandomState(0).uniform(-1, 1, (M, 2))
a_env, b_env = dcc_hip.HipCoverageEnv(E, N, M, poi), dcc_hip.HipCoverageEnv(E, N, M, poi)
a_env.reset(); b_env.reset()
act = torch.zeros(E, N, 2, device=dev)
out_a, out_b = a_env.alloc_out(), b_env.alloc_out()
a_env.step(act, out_a); b_env.step(act, out_b)          # warm-up
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    a_env.step(act, out_a)
    st = a_env.get_state()                               # 4 D2D memcpy nodes
gen = torch.Generator(device=dev).manual_seed(1)
bad = 0
for i in range(n):
    act.uniform_(-1, 1, generator=gen)
    g.replay()
    b_env.step(act, out_b)
    if i % 500 == 499 or i == n - 1:
        sb = b_env.get_state()
        ok = all(torch.equal(st[k], sb[k]) for k in st) and torch.equal(out_a["reward"], out_b["reward"])
        bad += (not ok)
        if not ok and bad < 4:
            print("MISMATCH at replay", i)
print("memcpy-node soak: %d replays (4 memcpy nodes each), %d mismatching checks" % (n, bad))
