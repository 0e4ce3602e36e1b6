"""Every branch of the reference's SharedReplayBuffer.compute_returns (uav_dcc_control/buffer/shared_buffer.py:160-217):
use_gae x use_proper_time_limits x use_valuenorm.  tests/golden/returns_modes.npz holds the outputs of the reference's own
method for the 8 combinations on one buffer (tools/gen_golden_returns.py).  Bit-exact float32 everywhere:
  CPU   the numpy restatement (oracle/mappo_oracle.compute_returns) and the C twin dcc_returns_compute_cpu == the fixture;
        this package's SharedReplayBuffer.compute_returns over the twin == the fixture
  GPU   dcc_returns_compute == the fixture and == the twin on random shapes around the 16-step look-ahead; the buffer method."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG

COMBOS = [(g, p, v) for g in (0, 1) for p in (0, 1) for v in (0, 1)]
Z = np.load(os.path.join(GOLDEN, "returns_modes.npz"))
T, E, N = Z["rewards"].shape[:3]


def _denorm():
    from oracle import mappo_oracle as mo
    return mo.valuenorm_mean_std(Z["vn_mean"][0], Z["vn_mean_sq"][0], Z["vn_debias"])


def _inputs(g, p, v):
    """flat [., C] inputs of the C-ABI call for one combination: value_preds / returns with the bootstrap where the branch puts it"""
    C = E * N
    vp = Z["value_preds"].reshape(T + 1, C).copy()
    ret = np.zeros((T + 1, C), np.float32)
    (vp if g else ret)[-1] = Z["next_value"].reshape(C)
    dn = np.array(_denorm(), np.float32) if v else None
    return (Z["rewards"].reshape(T, C).copy(), vp, Z["masks"].reshape(T + 1, C).copy(),
            Z["bad_masks"].reshape(T + 1, C).copy() if p else None, dn, ret)


def _want(g, p, v):
    key = "g%d_p%d_v%d" % (g, p, v)
    ret = Z["returns_" + key].reshape(T + 1, E * N)
    vp = Z["value_preds_after_" + key].reshape(T + 1, E * N)
    dv = vp[:-1] * np.float32(_denorm()[1]) + np.float32(_denorm()[0]) if v else vp[:-1]
    return ret, ret[:-1] - dv


@pytest.mark.parametrize("g,p,v", COMBOS)
def test_numpy_restatement_equals_the_reference(g, p, v):
    from oracle import mappo_oracle as mo
    mean, std = _denorm() if v else (None, None)
    ret, vp = mo.compute_returns(Z["rewards"], Z["value_preds"], Z["masks"], Z["bad_masks"], Z["next_value"], float(Z["gamma"]),
                                 float(Z["gae_lambda"]), bool(g), bool(p), mean, std)
    key = "g%d_p%d_v%d" % (g, p, v)
    np.testing.assert_array_equal(ret, Z["returns_" + key])
    np.testing.assert_array_equal(vp, Z["value_preds_after_" + key])


def test_fixture_branches_differ():
    """the buffer holds episode ends AND time-limit cuts, so the four recurrences give four different answers"""
    r = [Z["returns_g%d_p%d_v1" % (g, p)][:-1] for g in (0, 1) for p in (0, 1)]
    assert all(not np.array_equal(r[i], r[j]) for i in range(4) for j in range(i))
    assert int((Z["masks"][1:] == 0).sum()) > 0 and int((Z["bad_masks"][1:] == 0).sum()) > 0


@pytest.mark.parametrize("g,p,v", COMBOS)
def test_cpu_twin_equals_the_reference(g, p, v, oracle_mod):
    rew, vp, mk, bad, dn, ret = _inputs(g, p, v)
    adv = np.zeros((T, E * N), np.float32)
    oracle_mod.returns_compute_cpu(rew, vp, mk, bad, dn, float(Z["gamma"]), float(Z["gae_lambda"]), g | (p << 1), ret, adv)
    want_ret, want_adv = _want(g, p, v)
    np.testing.assert_array_equal(ret[:-1], want_ret[:-1])
    np.testing.assert_array_equal(adv, want_adv)
    if not g:
        np.testing.assert_array_equal(ret[-1], want_ret[-1])          # the bootstrap row the caller stored stays


def test_cpu_twin_rejects_bad_arguments(oracle_mod):
    rew, vp, mk, bad, dn, ret = _inputs(1, 1, 1)
    L = oracle_mod.lib()
    p = lambda a: a.ctypes.data
    assert L.dcc_returns_compute_cpu(p(rew), p(vp), p(mk), None, p(dn), 0.99, 0.95, 3, p(ret), None, T, E * N, None) == -1   # PROPER without bad_masks
    assert L.dcc_returns_compute_cpu(p(rew), p(vp), p(mk), p(bad), p(dn), 0.99, 0.95, 4, p(ret), None, T, E * N, None) == -1   # unknown mode
    assert L.dcc_returns_compute_cpu(p(rew), p(vp), p(mk), p(bad), p(dn), 0.99, 0.95, 3, p(ret), None, 0, E * N, None) == -1


def _cfg(g, p, v):
    import yaml
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    cfg.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, use_gae=bool(g), use_proper_time_limits=bool(p), use_valuenorm=bool(v),
               structured_input=False, compact_obs=False)
    return Namespace(**cfg)


class Box:
    def __init__(self, n):
        self.shape = (n,)


def _buffer_case(g, p, v):
    import utils.pytorch_utils as ptu
    from buffer.shared_buffer import SharedReplayBuffer
    from utils.valuenorm import ValueNorm
    buf = SharedReplayBuffer(_cfg(g, p, v), Box(6), Box(N * 6), Box(2))
    for name, key in (("rewards", "rewards"), ("value_preds", "value_preds"), ("masks", "masks"), ("bad_masks", "bad_masks")):
        getattr(buf, name).copy_(torch.from_numpy(Z[key]))
    vn = ValueNorm(1, device=ptu.device)
    for a, k in ((vn.running_mean, "vn_mean"), (vn.running_mean_sq, "vn_mean_sq"), (vn.debiasing_term, "vn_debias")):
        a.copy_(torch.from_numpy(Z[k]))
    buf.compute_returns(torch.from_numpy(Z["next_value"]), vn if v else None)
    key = "g%d_p%d_v%d" % (g, p, v)
    np.testing.assert_array_equal(buf.returns.cpu().numpy()[:-1], Z["returns_" + key][:-1])
    np.testing.assert_array_equal(buf.value_preds.cpu().numpy(), Z["value_preds_after_" + key])
    np.testing.assert_array_equal(buf.advantages_raw.cpu().numpy().reshape(T, E * N), _want(g, p, v)[1])
    if not g:
        np.testing.assert_array_equal(buf.returns.cpu().numpy()[-1], Z["returns_" + key][-1])


@pytest.mark.parametrize("g,p,v", COMBOS)
def test_buffer_compute_returns_over_the_cpu_twin(g, p, v, oracle_mod):
    """this package's SharedReplayBuffer.compute_returns (host logic: where the bootstrap goes, which entry point / mode) on
    torch CPU tensors with the `_cpu` twins behind dcc_hip"""
    from _cpu_twin_backend import cpu_twin_backend
    with cpu_twin_backend():
        _buffer_case(g, p, v)


@pytest.mark.gpu
@pytest.mark.parametrize("g,p,v", COMBOS)
def test_hip_returns_equal_the_reference(g, p, v):
    import dcc_hip
    rew, vp, mk, bad, dn, ret = _inputs(g, p, v)
    c = lambda a: None if a is None else torch.from_numpy(a).cuda()
    ret_d, adv_d = c(ret), torch.zeros(T, E * N, device="cuda")
    dcc_hip.returns_compute(c(rew), c(vp), c(mk), c(bad), c(dn), float(Z["gamma"]), float(Z["gae_lambda"]), g | (p << 1), ret_d, adv_d)
    want_ret, want_adv = _want(g, p, v)
    np.testing.assert_array_equal(ret_d.cpu().numpy()[:-1], want_ret[:-1])
    np.testing.assert_array_equal(adv_d.cpu().numpy(), want_adv)


@pytest.mark.gpu
@pytest.mark.parametrize("g,p,v", COMBOS)
def test_buffer_compute_returns_gpu(g, p, v):
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    try:
        _buffer_case(g, p, v)
    finally:
        ptu.set_gpu_mode(False)


@pytest.mark.gpu
@pytest.mark.parametrize("Tn,C", [(150, 4096 * 8), (1, 1), (16, 64), (17, 65), (33, 7), (400, 129)])
@pytest.mark.parametrize("mode", [0, 2, 3])
def test_hip_returns_equal_the_twin_on_random_shapes(Tn, C, mode, oracle_mod):
    """one binding, two libraries: dcc_returns_compute (device) == dcc_returns_compute_cpu (host), with and without ValueNorm"""
    import dcc_hip
    for use_vn in (False, True):
        rs = np.random.RandomState(Tn * 7 + C + mode)
        rew = rs.normal(-40, 30, (Tn, C)).astype(np.float32)
        vp = rs.normal(0, 1, (Tn + 1, C)).astype(np.float32)
        mk = (rs.uniform(0, 1, (Tn + 1, C)) > 0.03).astype(np.float32)
        bad = (rs.uniform(0, 1, (Tn + 1, C)) > 0.05).astype(np.float32) if mode & 2 else None
        ret = np.zeros((Tn + 1, C), np.float32)
        ret[-1] = rs.normal(0, 1, C)
        dn = np.array([-250.0, 97.5], np.float32) if use_vn else None
        ret_c, adv_c = ret.copy(), np.zeros((Tn, C), np.float32)
        oracle_mod.returns_compute_cpu(rew, vp, mk, bad, dn, 0.99, 0.95, mode, ret_c, adv_c)
        c = lambda a: None if a is None else torch.from_numpy(a).cuda()
        ret_d, adv_d = c(ret.copy()), torch.zeros(Tn, C, device="cuda")
        dcc_hip.returns_compute(c(rew), c(vp), c(mk), c(bad), c(dn), 0.99, 0.95, mode, ret_d, adv_d)
        np.testing.assert_array_equal(ret_d.cpu().numpy(), ret_c)
        np.testing.assert_array_equal(adv_d.cpu().numpy(), adv_c)


@pytest.mark.gpu
def test_hip_returns_reject_bad_arguments():
    import dcc_hip
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.returns_compute(z(4, 8), z(5, 8), z(5, 8), None, None, 0.99, 0.95, dcc_hip.RETURNS_PROPER, z(5, 8))      # no bad_masks
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.returns_compute(z(4, 8), z(5, 8), z(5, 8), z(5, 8), None, 0.99, 0.95, 7, z(5, 8))                         # unknown mode
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.returns_compute(torch.zeros(4, 8), z(5, 8), z(5, 8), None, None, 0.99, 0.95, 0, z(5, 8))                  # host tensor
