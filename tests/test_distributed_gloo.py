"""world_size-2 `gloo` tests on CPU of the one exchange step of the path: the MAPPO learner replicas.
Two ranks, each holding half of the envs of a rollout buffer, must end a `train()` call with the same
parameters as ONE process holding all envs: gradients are all-reduced before clipping, advantage
moments and ValueNorm batch moments are all-reduced too (algos/mappo.py, utils/valuenorm.py)."""
import os
import socket
import sys
from argparse import Namespace

import numpy as np
import pytest
import torch
import multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(E, seed_w, data, lo, hi, cfg_over=None):
    for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from test_mappo_golden import make_cfg, Box
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer
    N, T, D, A = 4, 16, 20, 2
    cfg = make_cfg(n_rollout_threads=hi - lo, ppo_epoch=2, **(cfg_over or {}))
    torch.manual_seed(seed_w)
    pol = MAPPOPolicy(cfg, Box(D), Box(N * D), Box(A))
    tr = MAPPOTrainer(cfg, pol)
    buf = SharedReplayBuffer(cfg, Box(D), Box(N * D), Box(A))
    for k in ("obs", "actions", "action_log_probs", "rewards", "value_preds", "masks", "returns"):
        getattr(buf, k).copy_(torch.from_numpy(data[k][:, lo:hi]))
    return pol, tr, buf


def _data(E):
    rs = np.random.RandomState(5)
    N, T, D, A = 4, 16, 20, 2
    vp = np.repeat(rs.normal(0, 1, (T + 1, E, 1, 1)), N, 2)
    return dict(obs=rs.normal(0, 1, (T + 1, E, N, D)).astype(np.float32),
                actions=rs.uniform(-1, 1, (T, E, N, A)).astype(np.float32),
                action_log_probs=np.repeat(rs.normal(-2.5, 0.3, (T, E, N, 1)), A, -1).astype(np.float32),
                rewards=np.repeat(rs.normal(-50, 30, (T, E, 1, 1)), N, 2).astype(np.float32),
                value_preds=vp.astype(np.float32), masks=np.ones((T + 1, E, N, 1), np.float32),
                returns=(vp * 90 - 250 + rs.normal(0, 20, vp.shape)).astype(np.float32))


def _local_perms(rank, per, epochs=2):
    """The permutations rank `rank` draws over its own T * per * N agent rows (mini-batch test)."""
    g = torch.Generator().manual_seed(1000 + rank)
    return [torch.randperm(16 * per * 4, generator=g) for _ in range(epochs)]


def _worker(rank, world, port, E, q, mini_batches=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    per = E // world
    pol, tr, buf = _build(E, 100 + rank, _data(E), rank * per, (rank + 1) * per,      # different init per rank ...
                          dict(num_mini_batch=mini_batches))
    pol.broadcast_parameters(0)                                                       # ... made identical here
    if mini_batches > 1:
        tr.minibatch_perms = _local_perms(rank, per)
    tr.prep_training()
    info = tr.train(buf)
    sd = {k: v.numpy().copy() for k, v in list(pol.actor.state_dict().items()) + list(pol.critic.state_dict().items())}
    q.put((rank, info, sd, tr.value_normalizer.running_mean.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_update_equals_single_process_full_batch():
    E = 4
    pol, tr, buf = _build(E, 100, _data(E), 0, E)
    tr.prep_training()
    info1 = tr.train(buf)
    ref = {k: v.clone() for k, v in list(pol.actor.state_dict().items()) + list(pol.critic.state_dict().items())}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, info, sd, vmean in res:
        for k in info1:   # losses are all-reduced means over the equally sized shards, the gradient norms are global: every
            np.testing.assert_allclose(info[k], info1[k], rtol=1e-4, atol=1e-6, err_msg=k)   # rank logs the full-batch numbers
        for k in ref:
            np.testing.assert_allclose(sd[k], ref[k].numpy(), rtol=2e-4, atol=2e-6, err_msg="rank%d %s" % (rank, k))
        np.testing.assert_allclose(vmean, tr.value_normalizer.running_mean.numpy(), rtol=1e-5)
    # replicas stay in lock-step
    for k in ref:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k


@pytest.mark.timeout(300)
def test_two_rank_mini_batches_equal_single_process_on_the_union():
    """num_mini_batch = 2 on two ranks: every rank permutes ITS OWN rows; mini-batch i of the job is the union of the ranks' i-th
    row sets, its loss the mean of the two equally sized per-rank means, ValueNorm sees the all-reduced batch moments.  One
    process fed the union of those row sets (in its own row numbering) must end with the same parameters."""
    E, N, T, MB = 4, 4, 16, 2
    per = E // 2
    locs = [_local_perms(r, per) for r in range(2)]
    mb = T * per * N // MB

    def to_global(rows, rank):          # local (t, e_l, n) -> global (t, rank * per + e_l, n)
        t, e, n = rows // (per * N), (rows // N) % per, rows % N
        return (t * E + rank * per + e) * N + n

    perms = [torch.cat([torch.cat([to_global(locs[r][ep][i * mb:(i + 1) * mb], r) for r in range(2)]) for i in range(MB)])
             for ep in range(2)]
    assert all(p.unique().numel() == T * E * N for p in perms)
    pol, tr, buf = _build(E, 100, _data(E), 0, E, dict(num_mini_batch=MB))
    tr.minibatch_perms = perms
    tr.prep_training()
    info1 = tr.train(buf)
    ref = {k: v.clone() for k, v in list(pol.actor.state_dict().items()) + list(pol.critic.state_dict().items())}
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, E, q, MB)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, info, sd, vmean in res:
        for k in info1:
            np.testing.assert_allclose(info[k], info1[k], rtol=1e-4, atol=1e-6, err_msg=k)
        for k in ref:
            np.testing.assert_allclose(sd[k], ref[k].numpy(), rtol=2e-4, atol=2e-6, err_msg="rank%d %s" % (rank, k))
        np.testing.assert_allclose(vmean, tr.value_normalizer.running_mean.numpy(), rtol=1e-5)
    for k in ref:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k


def test_make_env_shards_global_env_count(monkeypatch):
    """n_rollout_threads is the job-wide env count: each rank builds n/world envs at offset rank*n/world."""
    import utils.pytorch_utils as ptu
    import envs.make_env as me
    seen = {}

    class FakeEnv:
        def __init__(self, E, **kw):
            seen.update(E=E, **kw)

    monkeypatch.setattr(me, "HipCoverageVecEnv", FakeEnv)
    monkeypatch.setattr(ptu, "world_size", lambda: 4)
    import torch.distributed as dist
    monkeypatch.setattr(dist, "get_rank", lambda: 3)
    cfg = Namespace(env_file="mpe.uav_dcc", n_rollout_threads=4096, num_agents=8, num_pois=64, r_cover=0.2, r_comm=0.4,
                    comm_r_scale=0.95, comm_force_scale=0.0, max_ep_len=150)
    me.make_env(cfg)
    assert seen["E"] == 1024 and seen["env0"] == 3072 and seen["env_total"] == 4096
    cfg.n_rollout_threads = 10
    with pytest.raises(ValueError):
        me.make_env(cfg)
