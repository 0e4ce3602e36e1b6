"""Helper of tools/placement_pmc.sh: the observation producer (dcc_env_obs_write_probe, K = 150) into 10 separately allocated
buffers of one process, one warm-up + 3 launches each, in allocation order."""
import ctypes, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
os.environ["DCC_AUTOTUNE"] = "0"
import dcc_hip
E, N, M, K = 4096, 8, 64, 150
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
bufs = [torch.empty((K, E, N, env.D), dtype=torch.float32, device="cuda") for _ in range(10)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for t in bufs:
    t.zero_()
torch.cuda.synchronize()
for t in bufs:
    for rep in range(3):
        env.lib.dcc_env_obs_write_probe(env._h, K, ctypes.c_void_p(t.data_ptr()), st)
    torch.cuda.synchronize()
