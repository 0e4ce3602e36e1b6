"""Does the PLACEMENT of the output buffer decide which of the two speeds a process sees?  One process, one env, N separately
allocated output sets (all kept alive, so they occupy different physical ranges); the same 150-step launch into each, twice round.
usage: python tools/placement_probe.py [n_buffers]"""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import dcc_hip
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10
E, N, M, T = 4096, 8, 64, 150
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
env.reset()
acts = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (T, E, N, 2)).astype(np.float32)).cuda()
outs = [env.alloc_out(T) for _ in range(nb)]
def run(out, n=24):
    ev = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.rollout(T, actions=acts, out=out); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([x.elapsed_time(y) for x, y in ev][4:]))
import ctypes
L = env.lib; L.dcc_env_obs_write_probe.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
def timed(f, n=6):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for rnd in range(2):
    print("round %d env rollout ms: " % rnd + "  ".join("%.4f" % run(o) for o in outs), flush=True)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print("obs-only launch ms     : " + "  ".join("%.4f" % timed(lambda o=o: L.dcc_env_obs_write_probe(env._h, T, ctypes.c_void_p(o["obs"].data_ptr()), st)) for o in outs))
print("memset of obs, ms      : " + "  ".join("%.4f" % timed(lambda o=o: o["obs"].zero_()) for o in outs))
small = [o["obs"][:8] for o in outs]      # the first 8 steps only: 354 MB
print("memset of 8 steps, ms  : " + "  ".join("%.4f" % timed(lambda o=o: o.zero_(), 20) for o in small))
print("obs base addresses (mod 1 GiB, MiB): " + " ".join("%d" % ((o["obs"].data_ptr() % (1 << 30)) >> 20) for o in outs))
