"""Container-only (needs /root/reference): time the REFERENCE env (pure Python, imported through tools/ref_harness.py) and
the C restatement (oracle/dcc_oracle.c, what bench.py's cpu_baseline leg runs on the GPU box) on the same workload, one
thread each, and print the ratio -- so that the cpu_baseline of a bench line can be related to "the reference's Python on
that host" (CPU-reference item 3).  usage: python tools/cpu_ratio.py"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from ref_harness import make_reference_env  # noqa: E402
from oracle import oracle  # noqa: E402

for (N, M, steps_ref) in ((4, 20, 600), (8, 64, 300), (16, 256, 60)):
    rs = np.random.RandomState(0)
    env, world, sc = make_reference_env(N, M, 0.2, 0.4, 0.95, 0.0, None)
    env.reset()
    acts = rs.uniform(-1, 1, (steps_ref, N, 2)).astype(np.float32)
    t0 = time.perf_counter()
    for t in range(steps_ref):
        ob, rew, dn, info = env.step(acts[t].copy())
        if np.all(dn):
            env.reset()
    t_ref = (time.perf_counter() - t0) / steps_ref
    poi = np.array(sc.pos_pois[:M], np.float64)
    E, K = 64, 200
    orc = oracle.OracleEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    orc.reset()
    orc.rollout_rng(20, 1)
    t0 = time.perf_counter()
    orc.rollout_rng(K, 1, 20)
    t_c = (time.perf_counter() - t0) / (E * K)
    print("N=%d M=%d: reference Python %.1f env-steps/s (%.0f agent-env-steps/s); C restatement %.0f env-steps/s (%.0f agent-env-steps/s); "
          "ratio C / reference = %.0fx" % (N, M, 1 / t_ref, N / t_ref, 1 / t_c, N / t_c, t_ref / t_c))
