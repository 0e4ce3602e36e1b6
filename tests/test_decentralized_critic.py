"""`use_centralized_V: false` (uav_dcc_control/learner.py:43-46,218-222,269-273): the critic's input is each agent's own
observation row.  Host-side storage logic on CPU tensors; the end-to-end check against the reference's own Learner is
tests/test_learner_reference_replay.py [e2_decv]."""
from argparse import Namespace
import os

import numpy as np
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
    cfg.update(num_agents=3, n_rollout_threads=4, max_ep_len=6, use_centralized_V=False, structured_input=False, compact_obs=False)
    cfg.update(over)
    return Namespace(**cfg)


def _buffer(**over):
    from buffer.shared_buffer import SharedReplayBuffer
    buf = SharedReplayBuffer(_cfg(**over), Box(5), Box(5), Box(2))
    g = torch.Generator().manual_seed(1)
    buf.obs.copy_(torch.randn(buf.obs.shape, generator=g))
    return buf


def test_share_obs_is_obs_and_every_generator_feeds_the_critic_rows():
    buf = _buffer()
    T, E, N, D = 6, 4, 3, 5
    assert buf.decentralized and buf.share_obs is buf.obs and buf._share_obs is None
    with pytest.raises(RuntimeError):
        buf.share_obs_env
    adv = torch.zeros(T, E, N, 1)
    full = next(buf.feed_forward_generator(adv, 1, dedup_critic=False))
    assert torch.equal(full[0], buf.obs[:-1].reshape(T * E * N, D)) and torch.equal(full[0], full[1])
    with pytest.raises(ValueError):
        next(buf.feed_forward_generator(adv, 1, dedup_critic=True))
    perm = torch.randperm(T * E * N, generator=torch.Generator().manual_seed(2))
    got = list(buf.feed_forward_generator(adv, 2, dedup_critic=False, perm=perm))
    assert len(got) == 2 and all(len(s) == 12 for s in got)                     # per-row entries: no (row_sel, pair_sel)
    rows = buf.obs[:-1].reshape(T * E * N, D)
    for i, s in enumerate(got):
        want = rows[perm[i * 36:(i + 1) * 36]]
        assert torch.equal(s[0], want) and torch.equal(s[1], want)
    ch = buf.chunk_sample(adv, 2, 5, dedup_critic=False)
    assert torch.equal(ch[0], buf.obs[2:5].reshape(3 * E * N, D)) and torch.equal(ch[0], ch[1])
    # the reference's insert() hands share_obs == obs in (learner.py:272-276): accepted, the rows are what is stored
    z = torch.zeros
    buf.insert(torch.ones(E, N, D), torch.ones(E, N, D), None, None, z(E, N, 2), z(E, N, 1), z(E, N, 1), z(E, N, 1), torch.ones(E, N, 1))
    assert float(buf.obs[1].min()) == 1.0 and buf.share_obs[1].shape == (E, N, D)


def test_recurrent_generators_gather_the_agents_own_rows():
    buf = _buffer(use_recurrent_policy=True, data_chunk_length=3)
    T, E, N, D = 6, 4, 3, 5
    adv = torch.zeros(T, E, N, 1)
    for s in list(buf.recurrent_generator(adv, 2, 3)) + list(buf.naive_recurrent_generator(adv, 2)):
        assert s[0].shape == s[1].shape and torch.equal(s[0], s[1])


def test_storage_modes_that_share_a_value_per_env_are_refused():
    from buffer.shared_buffer import SharedReplayBuffer
    with pytest.raises(ValueError):
        SharedReplayBuffer(_cfg(), Box(5), Box(15), Box(2))                      # centralised space with the decentralised flag
    with pytest.raises(ValueError):
        SharedReplayBuffer(_cfg(), Box(5), Box(5), Box(2), compact=True, n_pois=2, expander=lambda *a: None)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    tr = MAPPOTrainer(_cfg(dedup_critic=True), MAPPOPolicy(_cfg(), Box(5), Box(5), Box(2)))
    assert not tr.dedup_critic
