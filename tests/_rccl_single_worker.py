"""Worker of tests/test_distributed_gpu.py::test_single_rank_group_over_rccl: ONE rank, DCC_DIST_SINGLE=1, backend nccl (= RCCL).
Every distributed code path of the learner runs against a real RCCL communicator on the one GPU of the box: communicator
creation with device_id, parameter broadcast, all-reduced rollout statistics / advantage and ValueNorm moments / logged metrics,
the asynchronous all-reduce of the flat gradients overlapped with the actor backward, the collective checkpoint
(gather_object) and the per-rank restore.  With one rank every collective is the identity, so the run must equal a
non-distributed run of the same seed bit for bit."""
import os
import sys
import tempfile

import torch
import yaml
from argparse import Namespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-coverage-control_amd")
sys.path.insert(0, PKG)
os.chdir(PKG)
import utils.pytorch_utils as ptu  # noqa: E402

cfg = {}
for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
    cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
cfg.update(n_rollout_threads=64, n_eval_rollout_threads=0, num_agents=4, num_pois=20, max_ep_len=25, n_iters=3, ppo_epoch=3,
           algo_hidden_size=64, save_model=False, seed=3)
from learner import Learner  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run(n_iters=3):
    lr = Learner(Namespace(**cfg))
    infos = []
    for it in range(1, n_iters + 1):
        lr.policy.lr_decay(it, 3)
        r = lr.rollout(lr.rl_buffer, lr.train_envs)
        infos.append((r, lr.rl_update()))
    flat = torch.cat([p.detach().reshape(-1) for p in list(lr.policy.actor.parameters()) + list(lr.policy.critic.parameters())])
    return lr, infos, flat.clone()


assert os.environ.get("DCC_DIST_SINGLE") == "1"
lr, infos, flat = run()
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1 and lr.dist_on
ck = os.path.join(os.environ.get("DCC_TEST_CKPT_DIR") or tempfile.gettempdir(), "rccl_single_test.pt")
lr.cur_iter = 3
lr.save_checkpoint(ck)                      # gather_object over RCCL
nxt = lr.rollout(lr.rl_buffer, lr.train_envs)
lr2 = Learner(Namespace(**dict(cfg, seed=77)))
lr2.load_checkpoint(ck)
assert lr2.rollout(lr2.rl_buffer, lr2.train_envs) == nxt
os.remove(ck)
dist.barrier()
dist.destroy_process_group()
# the same seed without any process group: identical parameters and statistics
os.environ["DCC_DIST_SINGLE"] = "0"
lr3, infos3, flat3 = run()
assert not lr3.dist_on
assert torch.equal(flat, flat3), "a one-rank RCCL job must equal the non-distributed job"
assert infos == infos3
print("RCCL_SINGLE_OK reward=%.3f" % infos[-1][0]["reward"])
