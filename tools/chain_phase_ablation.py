"""Where the ~1.8 us per-step chain of a small c2 batch goes, by PHASE: timing-only variants of csrc/dcc_env.hip with one phase of
env_physics_step (or one role of the roles kernel) removed -- their results are WRONG on purpose, they are built into /tmp and never
shipped.  Round 6 (VERDICT r05 task 6): which phase, if any, is worth spreading over lanes at 512 envs per GPU (`c2_strong` at 8 GPUs).

  build (container):  python tools/chain_phase_ablation.py build      -> dynamic-coverage-control_amd/csrc/variants/abl_<name>.so
  run (GPU box):      python tools/chain_phase_ablation.py run        -> us per step of a 150-step launch at E = 256 / 512, per variant
"""
import os
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(R, "dynamic-coverage-control_amd", "csrc")
VAR = os.path.join(CSRC, "variants")

EDITS = {   # name -> [(old, new)]: each `old` must occur exactly once in dcc_env.hip
    "base": [],
    "no_connect": [("    if (p.use_connect) {\n        const int npairs = N * N;", "    if (false) {\n        const int npairs = N * N;")],
    "no_energy_loop": [("#pragma unroll UNR_F\n        for (int i = 0; i < N; ++i) {", "#pragma unroll UNR_F\n        for (int i = 0; i < 1; ++i) {")],
    "no_clamp": [("        if (s2 > p.sq_speed) {                        // sqrt(s2) > max_speed", "        if (false) {")],
    "no_rng": [("            const unsigned long long z = splitmix64(p.seed + 0x9E3779B97F4A7C15ULL * (idx + 1ULL));", "            const unsigned long long z = idx * 0x12345677ULL;")],
    "no_wave_sum": [("    double base = wave_sum_f64_dpp(part);", "    double base = part;")],
    "no_sqrt": [("        if (valid && !dn) part -= __builtin_sqrt(smin);", "        if (valid && !dn) part -= smin;")],
    "no_obs_rows": [("""                    produce_obs<PPL, FORCE, NC, MC>(p, st, reinterpret_cast<const double*>(h.apos), en, dmask, poi, lane,
                                                    s * L, (s == 1) || epw == 1 || (env_base + 1 >= p.E));""", "                    (void)st;")],
    "no_physics": [("""                if (p.mode == 0) {
                    env_physics_step<PPL, ACT, FORCE, NC, MC>(p, env, k, lane, r[s], af[s], poi, in.apos, out.apos, out.avel, out.rec);
                } else if (lane < N) {""", """                if (false) {
                } else if (lane < N) {""")],
}


def build():
    src = open(os.path.join(CSRC, "dcc_env.hip")).read()
    os.makedirs(VAR, exist_ok=True)
    procs = []
    for name, edits in EDITS.items():
        d = "/tmp/abl/" + name
        os.makedirs(d, exist_ok=True)
        s = src
        for old, new in edits:
            assert s.count(old) == 1, (name, s.count(old), old[:60])
            s = s.replace(old, new)
        open(os.path.join(d, "dcc_env.hip"), "w").write(s)
        subprocess.check_call(["cp", os.path.join(CSRC, "dcc_internal.h"), d])
        cmd = ("cd %s && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden "
               "-I%s/include -I%s -DDCC_BUILDING=1 -c -o dcc_env.o dcc_env.hip && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "
               "-o %s/abl_%s.so dcc_env.o %s/dcc_gae.o %s/dcc_mlp.o %s/dcc_optim.o" % (d, R, CSRC, VAR, name, CSRC, CSRC, CSRC))
        procs.append((name, subprocess.Popen(cmd, shell=True)))
        if len(procs) % 5 == 0:
            for n, p in procs[-5:]:
                assert p.wait() == 0, n
    for n, p in procs:
        assert p.wait() == 0, n
    print("built", sorted(os.listdir(VAR)))


def one():
    sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
    import numpy as np
    import torch
    import dcc_hip
    N, M, T = 8, 64, 150
    poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
    os.environ["DCC_AUTOTUNE"] = "0"
    res = []
    for E in [int(v) for v in os.environ.get("AB_ENVS", "256,512").split(",")]:
        for act in ("rng", "hbm"):
            env = dcc_hip.HipCoverageEnv(E, N, M, poi)
            env.reset()
            out = env.alloc_out(T)
            acts = torch.rand(T, E, N, 2, device="cuda") * 2 - 1 if act == "hbm" else None
            fn = lambda: env.rollout(T, actions=acts, seed=0, step0=0, env0=0, env_total=E, out=out)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            res.append("E=%d %s %.3f" % (E, act, ms[len(ms) // 2] / T * 1e3))
            env.close()
    print("  ".join(res), flush=True)


def run():
    names = sorted(f[4:-3] for f in os.listdir(VAR) if f.startswith("abl_") and f.endswith(".so"))
    names = ["base"] + [n for n in names if n != "base"]
    for rnd in range(2):
        for n in names:
            sys.stdout.write("%-16s " % n); sys.stdout.flush()
            subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, DCC_HIP_LIB=os.path.join(VAR, "abl_%s.so" % n)))


if __name__ == "__main__":
    {"build": build, "run": run, "one": one}[sys.argv[1]]()
