"""Same-box A/B of kernel SHAPES for small c2 batches (round 6): us per step of a 150-step fused launch with observation rows at
E = 256 ... 4096, through the role-specialised kernel (DCC_SPLIT1_MAX=0) and through the split kernel (one env per workgroup: a
physics wave + DCC_SPLIT_OBS row-producing waves), per library under csrc/variants/ (or the shipped one), interleaved.
needs tools/attic/split1_small_batch.patch applied (DCC_SPLIT1_MAX).  usage (GPU box): python tools/attic/small_batch_ab.py [rounds]      |  one measurement: python tools/small_batch_ab.py one"""
import os
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VAR = os.path.join(R, "dynamic-coverage-control_amd", "csrc", "variants")


def one():
    sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
    import numpy as np
    import torch
    import dcc_hip
    N, M, T = 8, 64, 150
    poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
    os.environ["DCC_AUTOTUNE"] = "0"
    bstep = dcc_hip.bytes_per_step(N, M, with_actions=False, with_obs=True)
    res = []
    for E in [int(v) for v in os.environ.get("AB_ENVS", "256,512,1024,2048").split(",")]:
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
            us = ms[len(ms) // 2] / T * 1e3
            res.append("E=%d %s %.3f us (%.3f)" % (E, act, us, bstep * E / (us * 1e-6) / 8e12))
            env.close()
            del env, out, acts
    print("  ".join(res), flush=True)


def main():
    libs = [("shipped", None)] + [(f[:-3], os.path.join(VAR, f)) for f in sorted(os.listdir(VAR)) if f.endswith(".so")] if os.path.isdir(VAR) else [("shipped", None)]
    for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
        for name, lib in libs:
            for smax in ("0", "100000"):
                sys.stdout.write("%-10s %-6s " % (name, "roles" if smax == "0" else "split")); sys.stdout.flush()
                env = dict(os.environ, DCC_SPLIT1_MAX=smax)
                if lib:
                    env["DCC_HIP_LIB"] = lib
                subprocess.call([sys.executable, os.path.abspath(__file__), "one"], env=env, stderr=subprocess.DEVNULL)


if __name__ == "__main__":
    one() if sys.argv[1:2] == ["one"] else main()
