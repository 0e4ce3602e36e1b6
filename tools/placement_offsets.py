"""Is the placement effect a matter of the buffer's BASE OFFSET (address-bit hashing) or of the physical region?  For each of a few
separately allocated buffers, the observation producer (K = 150) is timed at base offsets 0 ... 2 MiB inside the same allocation."""
import ctypes, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
os.environ["DCC_AUTOTUNE"] = "0"
import dcc_hip
E, N, M, K = 4096, 8, 64, 150
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
nfl = K * E * N * env.D
pad = (8 << 20) // 4
bufs = [torch.empty(nfl + pad, dtype=torch.float32, device="cuda") for _ in range(6)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
offs = [0, 128, 1024, 4096, 16384, 65536, 262144, 1 << 20, 2 << 20, 3 << 20, 4 << 20]
def probe(ptr):
    best = 1e9
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.lib.dcc_env_obs_write_probe(env._h, K, ctypes.c_void_p(ptr), st); b.record(); b.synchronize()
        if rep: best = min(best, a.elapsed_time(b))
    return best
print("offset B : " + " ".join("%7d" % o for o in offs))
for i, t in enumerate(bufs):
    print("buffer %d : " % i + " ".join("%7.4f" % probe(t.data_ptr() + o) for o in offs))
