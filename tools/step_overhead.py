"""Host-side cost of one HipCoverageEnv.step() call (Python + ctypes + launch), and the GPU time per single-step launch.
usage: python tools/step_overhead.py [E N M]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch, dcc_hip
E, N, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 8, 64)
poi = np.load(R + "/dynamic-coverage-control_amd/envs/mpe/pos_pois.npy")[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi)
env.reset()
out = env.alloc_out()
a = torch.zeros(E, N, 2, device="cuda")
for _ in range(50): env.step(a, out)
torch.cuda.synchronize()
n = 2000
t0 = time.perf_counter()
for _ in range(n): env.step(a, out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue time per step() call: %.1f us; wall per step incl. drain: %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
