"""Worker of tests/test_distributed_gpu.py: one rank of a 2-rank MAPPO job on the GPU (launched by torch.distributed.run;
DCC_DIST_BACKEND=gloo lets both ranks share the one device).  Exercises the shipped config end to end -- env shards with
global env offsets, broadcast initial parameters, all-reduced advantage / ValueNorm moments and gradients, hipGraph rollouts,
structured input, fused kernels -- and checks that the replicas stay bit-identical."""
import os
import sys

import numpy as np
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
           algo_hidden_size=64, save_model=False, seed=3,
           num_mini_batch=int(os.environ.get("DCC_TEST_MINI_BATCH", "1")))     # > 1: every rank permutes its own rows (row mini-batches)
from learner import Learner  # noqa: E402
import torch.distributed as dist  # noqa: E402

lr = Learner(Namespace(**cfg))
assert lr.world == 2 and lr.train_envs.n_envs == 32 and lr.train_envs.env_total == 64 and lr.train_envs.env0 == 32 * lr.rank
infos = []
for it in range(1, 4):
    lr.policy.lr_decay(it, 3)
    r = lr.rollout(lr.rl_buffer, lr.train_envs)
    infos.append((r, lr.rl_update()))
flat = torch.cat([p.detach().reshape(-1) for p in list(lr.policy.actor.parameters()) + list(lr.policy.critic.parameters())])
vn = lr.trainer.value_normalizer
digest = torch.stack([flat.double().sum(), flat.double().abs().sum(), vn.running_mean.double().sum(), vn.running_mean_sq.double().sum()])
both = [torch.zeros_like(digest) for _ in range(2)]
dist.all_gather(both, digest)
assert torch.equal(both[0], both[1]), ("replicas diverged", both)
assert all(np.isfinite(v) for _, i in infos for v in i.values())
assert infos[0][0] == infos[0][0]      # rollout statistics are all-reduced: same on both ranks
rs = [torch.tensor([infos[-1][0]["reward"]], dtype=torch.float64, device=ptu.device) for _ in range(2)]
dist.all_gather(rs, rs[0].clone())
assert torch.equal(rs[0], rs[1])
# per-rank resume state (advisor finding of round 1: every rank used to restore rank 0's RNG streams): the checkpoint is
# written collectively, a fresh Learner of each rank restores ITS OWN streams and env shard, and the next rollout of the
# resumed job equals the next rollout of the original one on every rank
import tempfile  # noqa: E402
ckdir = os.environ.get("DCC_TEST_CKPT_DIR") or tempfile.gettempdir()
ck = os.path.join(ckdir, "dist_resume_test.pt")
lr.cur_iter = 3
lr.save_checkpoint(ck)
dist.barrier()
rng_here = torch.cuda.get_rng_state(ptu.device).clone()
nxt = lr.rollout(lr.rl_buffer, lr.train_envs)
acts = lr.rl_buffer.actions.clone()
lr2 = Learner(Namespace(**dict(cfg, seed=77)))
lr2.load_checkpoint(ck)
assert lr2.start_iter == 4 and torch.equal(torch.cuda.get_rng_state(ptu.device), rng_here)
nxt2 = lr2.rollout(lr2.rl_buffer, lr2.train_envs)
assert nxt2 == nxt and torch.equal(lr2.rl_buffer.actions, acts)
a_sum = [torch.zeros(1, dtype=torch.float64, device=ptu.device) for _ in range(2)]
dist.all_gather(a_sum, acts.double().sum().reshape(1))
assert not torch.equal(a_sum[0], a_sum[1]), "the ranks must keep decorrelated action noise after a resume"
if lr.rank == 0:
    os.remove(ck)
    print("DIST_GPU_OK graphs=%d reward=%.3f" % (len(lr._graphs), infos[-1][0]["reward"]))
dist.barrier()
dist.destroy_process_group()
