"""What the ~1.8 us per-step latency chain of a small c2 batch (256 / 512 envs, 150-step launches) is NOT made of: the same launch
without the connectivity flags, without the PoI-assignment output, without observation rows (state only: the fused kernel).
Round 5: connectivity 1-3 %, assignment output 3-5 %, and the state-only launch is no faster -- the chain is the float64 physics of
one wavefront (in-kernel RNG, IEEE sqrt / divisions of the speed clamp, the 8-agent energy pass, LDS fences), not any one output.
usage: python tools/chain_ablation.py"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch, dcc_hip
N, M, T = 8, 64, 150
poi = np.load(R + "/dynamic-coverage-control_amd/envs/mpe/pos_pois.npy")[:M]
def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]
os.environ["DCC_AUTOTUNE"] = "0"
for E in (256, 512):
    for label, crs, keys in (("full", 0.95, None), ("no connect flags", 0.0, None), ("no assign", 0.95, "assign"), ("no obs (state only)", 0.95, "obs")):
        env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, crs, 0.0); env.reset()
        out = env.alloc_out(T, obs=(keys != "obs"), assign=(keys != "assign"))
        med = timed(lambda: env.rollout(T, seed=0, step0=0, env0=0, env_total=E, out=out), 20)
        print("E = %4d  %-20s %.3f us/step" % (E, label, med / T * 1e3), flush=True)
        env.close()
