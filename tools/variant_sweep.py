"""Same-process A/B of library variants (csrc/variants/*.so) x DCC_OBS_DRAIN modes at the c2 shapes: every (variant, mode) rolls
150-step launches into the SAME output buffers (placement held fixed), interleaved rounds, median of per-round medians.
usage: python tools/variant_sweep.py "<modes, comma separated>" [rounds] [E ...]"""
import glob, importlib.util, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); PKG = R + "/dynamic-coverage-control_amd"; sys.path.insert(0, PKG)
import numpy as np, torch
N, M, T = (int(v) for v in os.environ.get("SWEEP_NMT", "8,64,150").split(","))      # e.g. SWEEP_NMT=32,1024,4 SWEEP_FORCE=0.5,0.1 for the c5 shapes
CFS, RCOMM = (float(v) for v in os.environ.get("SWEEP_FORCE", "0.0,0.4").split(","))
modes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [-1, 2]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Es = [int(v) for v in sys.argv[3:]] or [4096]
from envs.hip_vec_env import load_pois
poi = load_pois(M)
os.environ["DCC_AUTOTUNE"] = "0"
libs = {}
for so in sorted(glob.glob(PKG + "/csrc/variants/*.so")):
    name = os.path.basename(so)[:-3]
    os.environ["DCC_HIP_LIB"] = so
    spec = importlib.util.spec_from_file_location("dcc_hip_" + name, PKG + "/dcc_hip.py")
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.LIB_PATH == so
    libs[name] = mod
bstep = next(iter(libs.values())).bytes_per_step(N, M, with_actions=False, with_obs=True)


def timed(fn, n=12):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


for E in Es:
    for hbm in ((False, True) if (N, M) == (8, 64) else (False,)):
        envs, out = {}, None
        for name, mod in libs.items():
            for d in modes:
                os.environ["DCC_OBS_DRAIN"] = str(d)
                try:
                    e = mod.HipCoverageEnv(E, N, M, poi, 0.2, RCOMM, 0.95, CFS); e.reset()
                except Exception as ex:  # noqa: BLE001
                    print("  (%s mode %d: %s)" % (name, d, str(ex)[:100])); continue
                envs[(name, d)] = e
                if out is None:
                    out = e.alloc_out(T, placed=(6 if E * N * M >= 4096 * 512 else 0))
        acts = (torch.rand(T, E, N, 2, device="cuda") * 2 - 1) if hbm else None
        res = {k: [] for k in envs}
        for r in range(rounds):
            for k, e in envs.items():
                try:
                    res[k].append(timed(lambda: e.rollout(T, actions=acts, seed=0, step0=0, env0=0, env_total=E, out=out)))
                except Exception as ex:  # noqa: BLE001
                    res[k].append(float("nan"))
        print("E = %d, actions %s" % (E, "hbm" if hbm else "rng"))
        for name in libs:
            line = []
            for d in modes:
                if (name, d) in res:
                    med = sorted(res[(name, d)])[len(res[(name, d)]) // 2]
                    line.append("%2d: %.3f us (%.3f)" % (d, med / T * 1e3, bstep * E * T / (med * 1e-3) / 8e12))
            print("  %-14s " % name + "  ".join(line), flush=True)
        for e in envs.values(): e.close()
        del envs, out, acts
        torch.cuda.empty_cache()
