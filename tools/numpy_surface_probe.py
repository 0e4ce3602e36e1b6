"""The numpy drop-in surface (HipCoverageVecEnv.step: what the reference's host-side learner calls) is PCIe-bound: time it, and
the same data movement through pinned staging buffers.  usage: python tools/numpy_surface_probe.py [E N M]"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch
from envs.hip_vec_env import HipCoverageVecEnv
E, N, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 8, 64)
env = HipCoverageVecEnv(E, num_agents=N, num_pois=M)
env.reset()
a = np.random.uniform(-1, 1, (E, N, 2)).astype(np.float32)
for _ in range(5): env.step(a)
n = 30
t0 = time.perf_counter()
for _ in range(n): obs, rew, done, infos = env.step(a)
dt = (time.perf_counter() - t0) / n
print("HipCoverageVecEnv.step (numpy in / numpy out): %.2f ms per step = %.1f M agent-env-steps/s; obs %.1f MB per step"
      % (dt * 1e3, E * N / dt / 1e6, obs.nbytes / 1e6))
# raw D2H of the same obs tensor: pageable vs pinned
out = env.step_device(torch.from_numpy(a).cuda())
o = out["obs"]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): x = o.cpu()
dt1 = (time.perf_counter() - t0) / n
pin = torch.empty(o.shape, dtype=o.dtype, pin_memory=True)
t0 = time.perf_counter()
for _ in range(n):
    pin.copy_(o, non_blocking=True); torch.cuda.current_stream().synchronize()
dt2 = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    pin.copy_(o, non_blocking=True); torch.cuda.current_stream().synchronize(); y = pin.numpy().copy()
dt3 = (time.perf_counter() - t0) / n
print("obs D2H: pageable .cpu() %.2f ms (%.1f GB/s); pinned %.2f ms (%.1f GB/s); pinned + fresh numpy copy %.2f ms"
      % (dt1 * 1e3, o.numel() * 4 / dt1 / 1e9, dt2 * 1e3, o.numel() * 4 / dt2 / 1e9, dt3 * 1e3))
