"""Launcher: `python train.py <gpu_id> [key=value ...]` from this directory (like the reference's
`cd uav_dcc_control && python train.py 0`), or one process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py 0 n_rollout_threads=32768
The reference's own train.py also runs unchanged against this package (it only needs
utils.pytorch_utils.set_gpu_mode, learner.Learner and the three YAML files); this file differs by
accepting overrides, by working without omegaconf and by the one-process-per-GPU launch.
"""
import os
import sys

import torch
import yaml

import utils.pytorch_utils as ptu
from learner import Learner


def load_cfg(overrides):
    here = os.path.dirname(os.path.abspath(__file__))
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):  # later wins
        cfg.update(yaml.safe_load(open(os.path.join(here, f))))
    for kv in overrides:
        k, v = kv.split("=", 1)
        cfg[k] = yaml.safe_load(v)
    for k in ("actor_lr", "critic_lr", "opti_eps", "lr"):
        cfg[k] = float(cfg[k])
    return cfg


if __name__ == "__main__":
    gpu_id = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 0
    overrides = [a for a in sys.argv[1:] if "=" in a]
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        gpu_id = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("DCC_DIST_BACKEND") == "gloo" and torch.cuda.is_available():   # test hook: the ranks may share a GPU
            gpu_id %= torch.cuda.device_count()
    ptu.set_gpu_mode(torch.cuda.is_available(), gpu_id=gpu_id)
    cfg = load_cfg(overrides)
    print("cuda is available: ", torch.cuda.is_available())
    torch.set_num_threads(min(int(cfg["n_training_threads"]), os.cpu_count() or 1))
    os.makedirs(cfg["main_save_path"], exist_ok=True)
    cfg["log_wandb"] = False
    from argparse import Namespace
    Learner(Namespace(**cfg)).train()
