import sys, os, ctypes, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "dynamic-coverage-control_amd"))
import numpy as np, torch, dcc_hip
E,N,M,K=4096,8,64,150
poi=np.load(os.path.join(os.path.dirname(dcc_hip.__file__),"envs","mpe","pos_pois.npy"))[:M]
env=dcc_hip.HipCoverageEnv(E,N,M,poi)
obs=torch.empty((K,E,N,env.D),dtype=torch.float32,device="cuda")
L=env.lib; L.dcc_env_obs_write_probe.argtypes=[ctypes.c_void_p,ctypes.c_int32,ctypes.c_void_p,ctypes.c_void_p]
st=ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for r in range(2): L.dcc_env_obs_write_probe(env._h,K,ctypes.c_void_p(obs.data_ptr()),st)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(5): L.dcc_env_obs_write_probe(env._h,K,ctypes.c_void_p(obs.data_ptr()),st)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/5
print("obs-only: %.2f us/step  %.0f GB/s"%(ms/K*1e3, obs.numel()*4/ms/1e6))
