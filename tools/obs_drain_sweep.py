"""Where the observation wave of the role-specialised c2 kernel waits for its own stores (DCC_OBS_DRAIN, csrc/dcc_env.hip):
-1 never, 0 at the start of every env-step, 1 once per workgroup-step, 2 also before every staging-window flush -- by batch size,
envs per workgroup and action source.  us per step of a 150-step fused launch with observation rows, all variants into the
SAME output buffers of one process (placement held fixed), interleaved rounds, median of the per-round medians.
usage: python tools/obs_drain_sweep.py [rounds]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R + "/dynamic-coverage-control_amd")
import numpy as np, torch, dcc_hip
N, M, T = 8, 64, 150
CFS, RCOMM = 0.0, 0.4
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ["DCC_AUTOTUNE"] = "0"
VAR = os.environ.get("SWEEP_ENVVAR", "DCC_OBS_DRAIN")      # which library knob the columns vary (e.g. DCC_ROLES_PAIRS)
DR = tuple(int(v) for v in os.environ.get("SWEEP_MODES", "-1,0,2,5").split(","))


def timed(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def case(E, epw, hbm):
    from envs.hip_vec_env import load_pois
    poi = load_pois(M)
    bstep = dcc_hip.bytes_per_step(N, M, with_actions=False, with_obs=True)
    os.environ["DCC_ROLES_ENVS"] = str(epw)
    envs, out = {}, None
    for d in DR:
        os.environ[VAR] = str(d)
        envs[d] = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, RCOMM, 0.95, CFS); envs[d].reset()
        if out is None:
            out = envs[d].alloc_out(T, placed=(6 if E >= 4096 else 0))
    acts = (torch.rand(T, E, N, 2, device="cuda") * 2 - 1) if hbm else None
    res = {d: [] for d in DR}
    for r in range(rounds):
        for d in DR:
            e = envs[d]
            res[d].append(timed(lambda: e.rollout(T, actions=acts, seed=0, step0=0, env0=0, env_total=E, out=out)))
    line = []
    for d in DR:
        med = sorted(res[d])[len(res[d]) // 2]
        line.append("%2d: %.3f us (%.3f)" % (d, med / T * 1e3, bstep * E * T / (med * 1e-3) / 8e12))
    print("E = %4d, %d env/wg, actions %s:  " % (E, epw, "hbm" if hbm else "rng") + "  ".join(line), flush=True)
    for e in envs.values(): e.close()
    del envs, out, acts
    torch.cuda.empty_cache()


SHAPE = os.environ.get("SWEEP_SHAPE", "c2")
if SHAPE == "mid":
    for E in (1152, 1280, 1536, 1792, 2048, 2304):
        for epw in (1, 2):
            case(E, epw, False)
elif SHAPE == "small":
    for E in (288, 320, 384, 448, 512, 576, 640, 768, 896, 1024, 1280, 1536):
        case(E, 1 if E <= 1024 else 2, False)
elif SHAPE == "c4":
    N, M, T = 16, 256, 30
    for E in (int(v) for v in os.environ.get("SWEEP_ES", "1024,3584,4096,6144,8192").split(",")):
        case(E, 2, False)
elif SHAPE == "c5":
    N, M, T, CFS, RCOMM = 32, 1024, 4, 0.5, 0.1
    for E in (2048, 16384):
        case(E, 2, False)
elif SHAPE == "c2":
    for E, epw, hbm in ((256, 1, False), (512, 1, False), (512, 1, True), (1024, 1, False), (1024, 2, False), (2048, 1, False), (2048, 2, False), (2048, 2, True), (4096, 2, False), (4096, 2, True)):
        case(E, epw, hbm)
