"""Write tests/golden/ref_agent_small/agent.pkl: a checkpoint produced by the REFERENCE's own
MAPPOTrainer.save_model (uav_dcc_control/algos/mappo.py:236-239: pickle.dump of the policy object), with the same
seed / sizes as tools/gen_golden_mappo.py, so its parameters equal the actor/..., critic/... arrays of
tests/golden/mappo_small.npz.  Container-only; the file is data (pickled parameter tensors + class paths).
Re-run: python tools/gen_golden_ref_checkpoint.py"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

REF = "/root/reference/uav_dcc_control"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_agent_small")


class Box:
    def __init__(self, n):
        self.shape = (n,)


def main():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(REF, f))))
    for k in ("actor_lr", "critic_lr", "opti_eps"):
        cfg[k] = float(cfg[k])
    N, E, T, D, A, H = 4, 3, 16, 20, 2, 32
    cfg.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2)
    torch.manual_seed(7); np.random.seed(7)
    policy = MAPPOPolicy(Namespace(**cfg), Box(D), Box(N * D), Box(A))
    trainer = MAPPOTrainer(Namespace(**cfg), policy)
    os.makedirs(OUT, exist_ok=True)
    trainer.save_model(OUT)
    print("wrote", os.path.join(OUT, "agent.pkl"), os.path.getsize(os.path.join(OUT, "agent.pkl")) // 1024, "KB")


if __name__ == "__main__":
    main()
