"""Map of fast / slow placements: n buffers of K steps of c2 observation rows each (K = 24: 1.06 GB), allocated back to back and all
kept; the observation producer alone is timed into each (dcc_env_obs_write_probe).  Prints GB/s per buffer in allocation order.
usage: python tools/placement_map.py [n] [K]"""
import ctypes, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import dcc_hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
E, N, M = 4096, 8, 64
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
bufs = [torch.empty((K, E, N, env.D), dtype=torch.float32, device="cuda") for _ in range(n)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def probe(t):
    best = 1e9
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.lib.dcc_env_obs_write_probe(env._h, K, ctypes.c_void_p(t.data_ptr()), st); b.record(); b.synchronize()
        if rep: best = min(best, a.elapsed_time(b))
    return t.numel() * 4 / best / 1e6
for rnd in range(2):
    g = [probe(t) for t in bufs]
    print("round %d GB/s: " % rnd + " ".join("%4.0f" % x for x in g))
print("VA (GiB): " + " ".join("%.1f" % (t.data_ptr() / 2 ** 30 % 1000) for t in bufs))
