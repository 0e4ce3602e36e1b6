"""Guards of the PPO update (round-5 advisor findings): the 2^31-element activation limit per storage mode of a row mini-batch,
repeated rows in an injected sampler, and chunked / state-only updates with a recurrent policy."""
import os
from argparse import Namespace
from types import SimpleNamespace

import pytest
import torch
import yaml

from conftest import PKG


class Box:
    def __init__(self, n):
        self.shape = (n,)


def _cfg(**over):
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    cfg.update(num_agents=3, n_rollout_threads=4, max_ep_len=6, algo_hidden_size=16, structured_input=False, compact_obs=False)
    cfg.update(over)
    return Namespace(**cfg)


def _trainer(**over):
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    cfg = _cfg(**over)
    return MAPPOTrainer(cfg, MAPPOPolicy(cfg, Box(5), Box(15), Box(2))), cfg


@pytest.mark.parametrize("mode,kw,mb,refused", [
    # 8 UAV x 64 PoI (D = 338, S = 2704), 150 steps x 65,536 envs = 78.6 M agent rows
    ("rows-dedup-all-pairs", dict(compact=False), 16, True),      # critic on ALL 9.8 M pairs: 2.7e10 elements (the old guard saw 1.7e9)
    ("rows-per-row-critic", dict(compact=False, dedup=False), 64, True),   # [mb_rows, S] = 3.3e9
    ("compact-touched-pairs", dict(compact=True), 16, True),      # regenerated rows of ~all pairs
    ("rows-dedup-small-mb", dict(compact=False), 4096, False),    # 19,200 rows per mini-batch: fine
])
def test_activation_guard_counts_the_widest_tensor_of_each_storage_mode(mode, kw, mb, refused):
    tr, _ = _trainer(num_mini_batch=mb, algo_hidden_size=256)
    tr.dedup_critic = kw.get("dedup", True)
    fake = SimpleNamespace(episode_length=150, n_rollout_threads=65536, num_agents=8, obs_dim=338, share_obs_dim=2704,
                           structured=False, compact=kw["compact"], decentralized=False)
    if refused:
        with pytest.raises(ValueError, match="num_mini_batch"):
            tr._train_mini_batches(fake, None, True, {})
    else:
        with pytest.raises(AttributeError):        # past the guard: the fake has no generator
            tr._train_mini_batches(fake, None, True, {})


def test_injected_sampler_must_not_repeat_rows():
    from buffer.shared_buffer import SharedReplayBuffer
    buf = SharedReplayBuffer(_cfg(), Box(5), Box(15), Box(2))
    adv = torch.zeros(6, 4, 3, 1)
    perm = torch.arange(72)
    perm[5] = perm[6]
    with pytest.raises(ValueError, match="repeats rows"):
        next(buf.feed_forward_generator(adv, 2, perm=perm))
    assert len(list(buf.feed_forward_generator(adv, 2, perm=torch.arange(72)))) == 2


def test_chunked_update_refuses_a_recurrent_policy():
    from buffer.shared_buffer import SharedReplayBuffer
    tr, cfg = _trainer(use_recurrent_policy=True, update_chunk_steps=3)
    buf = SharedReplayBuffer(cfg, Box(5), Box(15), Box(2))
    with pytest.raises(ValueError, match="recurrent"):
        tr.train(buf, update_actor=True)
