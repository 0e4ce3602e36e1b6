"""Placement experiment 2: does hipExtMallocWithFlags(hipDeviceMallocContiguous) always give the fast mode?
Alternates regular torch allocations and contiguous allocations of the obs buffer; same launch into each."""
import ctypes, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import dcc_hip
E, N, M, T = 4096, 8, 64, 150
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
env.reset()
acts = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (T, E, N, 2)).astype(np.float32)).cuda()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
class Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}
def contiguous_obs(flag):
    nbytes = T * E * N * env.D * 4
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), nbytes, flag)
    if rc != 0:
        return None, rc
    return torch.as_tensor(Raw(p.value, (T, E, N, env.D)), device="cuda"), p
def run(out, n=24):
    ev = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.rollout(T, actions=acts, out=out); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([x.elapsed_time(y) for x, y in ev][4:]))
res = []
keep = []
for i in range(8):
    o = env.alloc_out(T)
    res.append(("torch", run(o))); keep.append(o)
    t, p = contiguous_obs(0x4)
    if t is None:
        res.append(("contig-failed rc=%s" % p, 0.0)); continue
    o2 = dict(env.alloc_out(T, obs=False)); o2["obs"] = t
    res.append(("contig", run(o2))); keep.append((o2, p))
print("  ".join("%s %.4f" % r for r in res))
