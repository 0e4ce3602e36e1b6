"""Generate tests/golden/returns_modes.npz: every branch of the REFERENCE's SharedReplayBuffer.compute_returns
(/root/reference/uav_dcc_control/buffer/shared_buffer.py:160-217) on one synthetic buffer -- use_gae x use_proper_time_limits x
use_valuenorm = 8 runs of the reference's own method.  Container-only; the outputs are data.  Re-run: python tools/gen_golden_returns.py

N = 4 agents, E = 3 envs, T = 37 steps (crosses two 16-step look-ahead windows of the scan kernel).  Contents:
  rewards [T,E,N,1], value_preds [T+1,E,N,1] (row T as allocated: zeros), masks / bad_masks [T+1,E,N,1] (episode ends /
  time-limit cuts), next_value [E,N,1], vn_mean / vn_mean_sq / vn_debias (ValueNorm state), gamma, gae_lambda,
  returns_g<g>_p<p>_v<v> [T+1,E,N,1] and value_preds_after_g<g>_p<p>_v<v> for the 8 flag combinations.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

REF = "/root/reference/uav_dcc_control"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "returns_modes.npz")


class Box:
    def __init__(self, n):
        self.shape = (n,)


def main():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from buffer.shared_buffer import SharedReplayBuffer
    from utils.valuenorm import ValueNorm
    base = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml"):
        base.update(yaml.safe_load(open(os.path.join(REF, f))))
    N, E, T, D, A = 4, 3, 37, 6, 2
    base.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=8)
    rs = np.random.RandomState(5)
    rewards = np.repeat(rs.normal(-50, 30, (T, E, 1, 1)).astype(np.float32), N, axis=2)
    vp = rs.normal(0, 1, (T + 1, E, N, 1)).astype(np.float32)
    vp[-1] = 0
    masks = (rs.uniform(0, 1, (T + 1, E, 1, 1)) > 0.08).astype(np.float32).repeat(N, axis=2)
    bad = (rs.uniform(0, 1, (T + 1, E, 1, 1)) > 0.10).astype(np.float32).repeat(N, axis=2)
    next_value = rs.normal(0, 1, (E, N, 1)).astype(np.float32)
    vn = ValueNorm(1, device=torch.device("cpu"))
    vn.update(rs.normal(-300, 120, (200, 1)).astype(np.float32))
    out = dict(rewards=rewards, value_preds=vp, masks=masks, bad_masks=bad, next_value=next_value,
               vn_mean=vn.running_mean.numpy().copy(), vn_mean_sq=vn.running_mean_sq.numpy().copy(),
               vn_debias=vn.debiasing_term.numpy().copy(), gamma=np.array(base["gamma"]), gae_lambda=np.array(base["gae_lambda"]))
    for g in (0, 1):
        for p in (0, 1):
            for v in (0, 1):
                cfg = Namespace(**dict(base, use_gae=bool(g), use_proper_time_limits=bool(p), use_valuenorm=bool(v)))
                buf = SharedReplayBuffer(cfg, Box(D), Box(N * D), Box(A))
                buf.rewards[:] = rewards; buf.value_preds[:] = vp; buf.masks[:] = masks; buf.bad_masks[:] = bad
                buf.compute_returns(next_value, vn if v else None)
                key = "g%d_p%d_v%d" % (g, p, v)
                out["returns_" + key] = buf.returns.copy()
                out["value_preds_after_" + key] = buf.value_preds.copy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB; episode ends", int((masks[1:, :, 0, 0] == 0).sum()), "time-limit cuts",
          int((bad[1:, :, 0, 0] == 0).sum()))


if __name__ == "__main__":
    main()
