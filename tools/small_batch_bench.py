"""Latency-bound shapes of the c2 env kernel (VERDICT r03 task 5): us per step of a 150-step fused launch at small env counts (what
`c2_strong` runs at 8 GPUs: 4096 / 8 = 512 envs per GPU) with one and with two envs per role-specialised workgroup, and of the
single-step launch with observation rows at 4096 envs (the numpy drop-in surface / the learner's non-graph path).
usage: python tools/small_batch_bench.py [N M]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch, dcc_hip
N, M = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (8, 64)
poi = np.load(R + "/dynamic-coverage-control_amd/envs/mpe/pos_pois.npy")[:M]
bstep = dcc_hip.bytes_per_step(N, M, with_actions=False, with_obs=True)


def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2], ms[0]


T = 150
for E in (256, 512, 1024, 2048, 4096):
    row = []
    for envs_per_wg in (1, 2):
        os.environ["DCC_ROLES_ENVS"] = str(envs_per_wg)
        os.environ["DCC_AUTOTUNE"] = "0"
        env = dcc_hip.HipCoverageEnv(E, N, M, poi); env.reset()
        out = env.alloc_out(T)
        med, mn = timed(lambda: env.rollout(T, seed=0, step0=0, env0=0, env_total=E, out=out), 20)
        row.append("%d env/wg: %.2f us/step (min %.2f) = %.3f of 8 TB/s" % (envs_per_wg, med / T * 1e3, mn / T * 1e3, bstep * E * T / (med * 1e-3) / 8e12))
        env.close(); del env, out
    print("E = %4d, K = %d:  " % (E, T) + "   |   ".join(row))
os.environ.pop("DCC_ROLES_ENVS")


def back_to_back(fn, n=400):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


one = torch.zeros(64, device="cuda")
med, mn = timed(lambda: one.add_(1.0), 200)
print("a 64-element torch kernel between two events: %.2f us (min %.2f); back to back %.2f us per launch  <- the floor of any single launch" % (med * 1e3, mn * 1e3, back_to_back(lambda: one.add_(1.0))))
for E in (512, 4096):
    env = dcc_hip.HipCoverageEnv(E, N, M, poi); env.reset()
    a = torch.zeros(E, N, 2, device="cuda")
    for label, out in (("rows", env.alloc_out()), ("state only", {**{k: v for k, v in env.alloc_out(obs=False).items()}, **env.alloc_state_out()})):
        med, mn = timed(lambda: env.step(a, out), 200)
        b2b = back_to_back(lambda: env.step(a, out))          # ONE measurement: the fraction below is of THIS time (r04 printed the
        nbytes = dcc_hip.bytes_per_step(N, M, with_actions=True, with_obs=(label == "rows")) * E      # fraction of a second, slower run)
        print("E = %4d, K = 1, %-10s: %.2f us per step between events (min %.2f); %.2f us per step back to back (%.3f of 8 TB/s)"
              % (E, label, med * 1e3, mn * 1e3, b2b, nbytes / (b2b * 1e-6) / 8e12))
    env.close()
