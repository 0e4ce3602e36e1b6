"""-m gpu: the HIP GAE scan (include/dcc_gae.h) against the reference's golden returns and the
numpy oracle -- bit-exact float32."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_gae_matches_reference_golden_bit_exact():
    import dcc_hip
    from oracle import mappo_oracle as mo
    Z = np.load(os.path.join(GOLDEN, "mappo_small.npz"))
    T, E, N = 16, 3, 4
    dev = torch.device("cuda")
    mean, std = mo.valuenorm_mean_std(Z["vn0_mean"][0], Z["vn0_mean_sq"][0], Z["vn0_debias"])
    t = lambda a, s: torch.from_numpy(np.ascontiguousarray(a)).to(dev).reshape(s)
    rew = t(Z["buf_rewards"], (T, E * N)); vp = t(Z["buf_value_preds_after"], (T + 1, E * N))
    mk = t(Z["buf_masks"], (T + 1, E * N))
    ret = torch.zeros(T + 1, E * N, device=dev); adv = torch.zeros(T, E * N, device=dev)
    dn = torch.tensor([mean, std], dtype=torch.float32, device=dev)
    dcc_hip.gae_compute(rew, vp, mk, dn, 0.99, 0.95, ret, adv)
    np.testing.assert_array_equal(ret.cpu().numpy().reshape(T + 1, E, N, 1)[:-1], Z["returns"][:-1])
    np.testing.assert_array_equal(adv.cpu().numpy().reshape(T, E, N, 1), Z["adv_raw"])


@pytest.mark.parametrize("T,C,use_vn", [(150, 4096 * 8, True), (150, 1000, False), (7, 33, True), (1, 1, True),
                                          (16, 64, True), (17, 65, False), (32, 100, True), (33, 7, True), (400, 129, True)])   # around the 16-step look-ahead
def test_gae_matches_oracle_random(T, C, use_vn):
    import dcc_hip
    from oracle import mappo_oracle as mo
    rs = np.random.RandomState(T * 7 + C)
    rew = rs.normal(-40, 30, (T, C)).astype(np.float32)
    vp = rs.normal(0, 1, (T + 1, C)).astype(np.float32)
    mk = (rs.uniform(0, 1, (T + 1, C)) > 0.03).astype(np.float32)
    mean, std = (np.float32(-250.0), np.float32(97.5)) if use_vn else (None, None)
    ref, _ = mo.compute_returns_gae(rew, vp, mk, vp[-1], 0.99, 0.95, mean, std)
    dev = torch.device("cuda")
    ret = torch.zeros(T + 1, C, device=dev); adv = torch.zeros(T, C, device=dev)
    dn = torch.tensor([mean, std], dtype=torch.float32, device=dev) if use_vn else None
    dcc_hip.gae_compute(torch.from_numpy(rew).to(dev), torch.from_numpy(vp).to(dev), torch.from_numpy(mk).to(dev), dn,
                        0.99, 0.95, ret, adv)
    np.testing.assert_array_equal(ret.cpu().numpy()[:-1], ref[:-1])
    dv = vp[:-1] * std + mean if use_vn else vp[:-1]
    np.testing.assert_array_equal(adv.cpu().numpy(), ref[:-1] - dv)


def test_buffer_compute_returns_uses_the_kernel():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from test_mappo_golden import make_cfg, Box, Z, T, E, N, D, S, A
    from buffer.shared_buffer import SharedReplayBuffer
    from utils.valuenorm import ValueNorm
    buf = SharedReplayBuffer(make_cfg(), Box(D), Box(S), Box(A))
    for name, key in (("rewards", "buf_rewards"), ("value_preds", "buf_value_preds"), ("masks", "buf_masks")):
        getattr(buf, name).copy_(torch.from_numpy(Z[key]))
    vn = ValueNorm(1, device=ptu.device)
    for a, k in ((vn.running_mean, "vn0_mean"), (vn.running_mean_sq, "vn0_mean_sq"), (vn.debiasing_term, "vn0_debias")):
        a.copy_(torch.from_numpy(Z[k]))
    buf.compute_returns(torch.from_numpy(Z["next_value"]), vn)
    np.testing.assert_array_equal(buf.returns.cpu().numpy()[:-1], Z["returns"][:-1])
    np.testing.assert_array_equal(buf.advantages_raw.cpu().numpy(), Z["adv_raw"])
    ptu.set_gpu_mode(False)
