"""The structured first layer (algos/algo_utils/structured.py) is the same function of the parameters as
LayerNorm -> Linear on the materialised observation rows: forward values and every parameter gradient agree.
Rows come from the oracle (random states, one step), i.e. they have the real column structure."""
from argparse import Namespace

import numpy as np
import pytest
import torch

CASES = [(8, 64), (4, 20), (3, 7), (1, 9), (5, 37)]


def _rows(oracle_mod, N, M, n, seed):
    rs = np.random.RandomState(seed)
    poi = rs.uniform(-1, 1, (M, 2))
    env = oracle_mod.OracleEnv(n, N, M, poi, 0.25, 0.4, 0.95, 0.0)
    env.reset()
    env.set_state(pos=rs.uniform(-1.2, 1.2, (n, N, 2)), vel=rs.uniform(-1, 1, (n, N, 2)),
                  energy=rs.randint(0, 5, (n, M)).astype(np.float64), done=(rs.uniform(size=(n, M)) < 0.3).astype(np.uint8))
    obs = env.step(rs.uniform(-1, 1, (n, N, 2)).astype(np.float32))["obs"]      # float64 rows, as the reference builds them
    env.close()
    return torch.from_numpy(obs), poi


def _base(in_dim, feature_norm=True, hidden=48, seed=0):
    from algos.algo_utils.mlp import MLPBase
    torch.manual_seed(seed)
    cfg = Namespace(use_feature_normalization=feature_norm, algo_hidden_size=hidden, layer_N=1, use_orthogonal=True,
                    use_ReLU=True)
    base = MLPBase(cfg, (in_dim,))
    if feature_norm:   # non-trivial affine so that the folding is exercised
        with torch.no_grad():
            base.feature_norm.weight.uniform_(0.5, 1.5)
            base.feature_norm.bias.uniform_(-0.3, 0.3)
    return base


def _compare(base, obs64, lay, width, structured_fn):
    """float32 (rows cast like the rollout buffer stores them): forward to 2e-5.  float64 on the unrounded rows:
    the algebra is exact -- values and all parameter gradients to 1e-9."""
    from algos.algo_utils import structured as S
    obs32 = obs64.float()
    out32_d = base(obs32.view(-1, width))
    out32_s = structured_fn(base, S.features_from_obs(obs32, lay))
    np.testing.assert_allclose(out32_s.detach().numpy(), out32_d.detach().numpy(), rtol=2e-4, atol=2e-5)
    base = base.double()
    feats = S.features_from_obs(obs64, lay)
    out_d = base(obs64.view(-1, width))
    gd = torch.autograd.grad(out_d.square().sum() + out_d.sum(), list(base.parameters()))
    out_s = structured_fn(base, feats)
    gs = torch.autograd.grad(out_s.square().sum() + out_s.sum(), list(base.parameters()))
    np.testing.assert_allclose(out_s.detach().numpy(), out_d.detach().numpy(), rtol=1e-9, atol=1e-10)
    for (name, _), a, b in zip(base.named_parameters(), gd, gs):
        scale = float(a.abs().max()) + 1e-300
        assert float((a - b).abs().max()) <= 1e-8 * scale + 1e-10, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("feature_norm", [True, False])
@pytest.mark.parametrize("N,M", CASES)
def test_actor_first_layer_from_features_equals_dense(oracle_mod, N, M, feature_norm):
    from algos.algo_utils import structured as S
    obs, poi = _rows(oracle_mod, N, M, 24, 3 * N + M)
    lay = S.ObsLayout(N, M, poi, 5.0)
    assert lay.D == obs.shape[-1]
    base = _base(lay.D, feature_norm)
    _compare(base, obs, lay, lay.D, lambda b, f: S.actor_trunk(b, lay, f))


@pytest.mark.parametrize("feature_norm", [True, False])
@pytest.mark.parametrize("N,M", CASES)
def test_critic_first_layer_from_features_equals_dense(oracle_mod, N, M, feature_norm):
    from algos.algo_utils import structured as S
    obs, poi = _rows(oracle_mod, N, M, 24, 5 * N + M)
    lay = S.ObsLayout(N, M, poi, 5.0)
    base = _base(N * lay.D, feature_norm, seed=1)
    _compare(base, obs, lay, N * lay.D, lambda b, f: S.critic_trunk(b, lay, f))


def test_features_from_obs_layout(oracle_mod):
    from algos.algo_utils import structured as S
    obs, poi = _rows(oracle_mod, 4, 20, 6, 1)
    obs = obs.float()
    lay = S.ObsLayout(4, 20, poi, 5.0)
    f = S.features_from_obs(obs, lay)
    assert f["head"].shape == (6, 4, 10) and f["poi_feat"].shape == (6, 40) and f["stats"].shape == (6, 4, 2)
    assert f["stats"].dtype == torch.float64
    blk = obs[:, :, lay.HD:].view(6, 4, 20, 5)
    for i in range(4):      # energy / m_energy / done columns are the same for every agent of an env
        assert torch.equal(blk[:, i, :, 2], f["poi_feat"][:, :20]) and torch.equal(blk[:, i, :, 4], f["poi_feat"][:, 20:])
        assert bool((blk[:, i, :, 3] == 5.0).all())
    var, mean = torch.var_mean(obs.double(), -1, unbiased=False)
    np.testing.assert_allclose(f["stats"][..., 0].numpy(), mean.numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose((f["stats"][..., 1] / lay.D).numpy(), var.numpy(), rtol=1e-10)


def test_inference_cache_of_folded_weights_tracks_parameter_updates(oracle_mod):
    """Without autograd the weight-side tensors are cached per parameter version and refreshed IN PLACE (a captured
    rollout graph keeps reading the same addresses)."""
    from algos.algo_utils import structured as S
    obs, poi = _rows(oracle_mod, 4, 20, 6, 2)
    lay = S.ObsLayout(4, 20, poi, 5.0)
    base = _base(lay.D, True, hidden=16)
    feats = S.features_from_obs(obs.float(), lay)
    with torch.no_grad():
        a1 = S.folded_weights(base, lay, 1)
        a2 = S.folded_weights(base, lay, 1)
        assert all(x is y for x, y in zip(a1[:4], a2[:4]))                    # cache hit: the very same tensors
        out1 = S.actor_trunk(base, lay, feats).clone()
        ptrs = [t.data_ptr() for t in a1[:4]]
        base.mlp.fc1[0].weight.mul_(1.5)                                       # in-place update (like an optimizer step)
        base.feature_norm.bias.add_(0.1)
        a3 = S.folded_weights(base, lay, 1)
        assert [t.data_ptr() for t in a3[:4]] == ptrs                          # refreshed in place
        out2 = S.actor_trunk(base, lay, feats)
    ref = S.actor_trunk(base, lay, feats)                                      # autograd path: no cache involved
    np.testing.assert_allclose(out2.numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)
    assert float((out1 - out2).abs().max()) > 1e-3
    dense = base(obs.float().view(-1, lay.D))
    np.testing.assert_allclose(out2.numpy(), dense.detach().numpy(), rtol=2e-4, atol=2e-5)
