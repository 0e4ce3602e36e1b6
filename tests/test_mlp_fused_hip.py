"""-m gpu: the fused element-wise stages of the policy trunks (include/dcc_mlp.h) against the plain PyTorch fp32
formulation of the same ops (forward values, input gradients, parameter gradients), their run-to-run determinism,
and the shapes that must be refused."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = dict(rtol=2e-5, atol=2e-6)


def _close(a, b, name, rtol=2e-5, atol=2e-6):
    scale = float(b.abs().max()) + 1e-20
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=rtol, atol=atol * max(1.0, scale), err_msg=name)


def _close_grad(a, b, name, n_act):
    """Two fp32 evaluations of the same pre-activation differ by ~1e-7 relative, which flips the ReLU of about one
    activation per ten million (|z| below the rounding difference); each flip moves every gradient sum it feeds by
    O(1).  Small cases (no flip expected) are compared tightly; for the large ones no entry may be off by more than
    a few flips' worth."""
    scale = float(b.abs().max()) + 1e-20
    tol = 2e-4 if n_act < 50000 else 3e-2
    assert float((a - b).abs().max()) <= tol * scale, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("R,H", [(4099, 256), (77, 64), (5, 32), (301, 128), (130, 512), (50, 100), (1, 256), (9000, 8)])
def test_relu_ln_matches_torch(R, H):
    import dcc_hip
    from algos.algo_utils import fused
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(R + H)
    z = torch.randn(R, H, device=dev, generator=g) * 1.7 + 0.2
    ln = torch.nn.LayerNorm(H).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.5, 0.5)
    assert dcc_hip.mlp_fused_supported(H)
    dh = torch.randn(R, H, device=dev, generator=g)
    for with_bias in (False, True):
        bias = (torch.randn(H, device=dev, generator=g) * 0.5).requires_grad_(True) if with_bias else None
        ln.zero_grad()
        z1 = z.clone().requires_grad_(True)
        h1 = fused.relu_ln(z1, bias, ln)
        assert h1.grad_fn is not None and "ReluLN" in type(h1.grad_fn).__name__      # the HIP path ran, not the fallback
        h1.backward(dh)
        g1 = (z1.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone(), bias.grad.clone() if with_bias else None)
        ln.zero_grad()
        if with_bias:
            bias.grad = None
        z2 = z.clone().requires_grad_(True)
        h2 = ln(torch.relu(z2 + bias if with_bias else z2))
        h2.backward(dh)
        _close(h1.detach(), h2.detach(), "h")
        _close(g1[0], z2.grad, "dz")
        _close(g1[1], ln.weight.grad, "dgamma", rtol=1e-4, atol=1e-5)
        _close(g1[2], ln.bias.grad, "dbeta", rtol=1e-4, atol=1e-5)
        if with_bias:
            _close(g1[3], bias.grad, "dbias", rtol=1e-4, atol=1e-5)
            bias.grad = None
        # deterministic: same bits on a second run
        ln.zero_grad()
        z3 = z.clone().requires_grad_(True)
        fused.relu_ln(z3, bias, ln).backward(dh)
        assert torch.equal(z3.grad, g1[0]) and torch.equal(ln.weight.grad, g1[1]) and torch.equal(ln.bias.grad, g1[2])


def test_unsupported_width_falls_back_to_torch_ops():
    import dcc_hip
    from algos.algo_utils import fused
    dev = torch.device("cuda", 0)
    assert not dcc_hip.mlp_fused_supported(1000) and not dcc_hip.mlp_fused_supported(130)
    ln = torch.nn.LayerNorm(130).to(dev)
    z = torch.randn(7, 130, device=dev, requires_grad=True)
    h = fused.relu_ln(z, None, ln)
    assert "ReluLN" not in type(h.grad_fn).__name__
    assert torch.equal(h, ln(torch.relu(z)))


@pytest.mark.parametrize("feature_norm", [True, False])
@pytest.mark.parametrize("N,M,H,n", [(8, 64, 256, 517), (4, 20, 256, 301), (4, 20, 64, 33), (1, 9, 32, 5), (5, 37, 128, 70), (16, 256, 256, 40),
                                     (11, 30, 256, 19), (32, 100, 256, 23), (20, 30, 64, 9), (63, 40, 256, 6)])
def test_actor_first_block_matches_torch(N, M, H, n, feature_norm):
    """fused actor L1 (from dcc_obs_features of random states) == the torch formulation of structured.actor_trunk's
    first block == LayerNorm(ReLU(Linear(LayerNorm(rows)))) on the rows dcc_obs_expand builds."""
    import dcc_hip
    from argparse import Namespace
    from algos.algo_utils import fused, structured as S
    from algos.algo_utils.mlp import MLPBase
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(N * 100 + M)
    poi = rs.uniform(-1, 1, (M, 2))
    env = dcc_hip.HipCoverageEnv(n, N, M, poi, 0.3, 0.3, 0.95, 0.0)
    env.reset()
    K = 6
    st = env.alloc_state_out(K)
    env.rollout(K, seed=3, out=dict(st, reward=torch.empty(K, n, device=dev)))
    state = [st[k][-1].contiguous() for k in ("state_pos", "state_vel", "state_energy", "state_done")]
    feats = env.obs_features(*state)
    rows = env.expand_obs(*state)
    lay = S.ObsLayout(N, M, poi, env.m_energy)
    torch.manual_seed(1)
    cfg = Namespace(use_feature_normalization=feature_norm, algo_hidden_size=H, layer_N=1, use_orthogonal=True, use_ReLU=True)
    base = MLPBase(cfg, (lay.D,)).to(dev)
    with torch.no_grad():
        if feature_norm:
            base.feature_norm.weight.uniform_(0.5, 1.5); base.feature_norm.bias.uniform_(-0.3, 0.3)
        base.mlp.fc1[2].weight.uniform_(0.5, 1.5); base.mlp.fc1[2].bias.uniform_(-0.3, 0.3)
    assert dcc_hip.mlp_fused_supported(H, lay.HD)
    dh = torch.randn(n * N, H, device=dev)
    params = list(base.parameters())

    def run(enabled, dense=False):
        fused.ENABLED = enabled
        try:
            for p in params:
                p.grad = None
            out = base(rows.view(n * N, lay.D)) if dense else S.actor_trunk(base, lay, feats)
            out.backward(dh)
            return out.detach().clone(), [None if p.grad is None else p.grad.clone() for p in params]
        finally:
            fused.ENABLED = True

    o_f, g_f = run(True)
    o_t, g_t = run(False)
    o_d, g_d = run(False, dense=True)
    _close(o_f, o_t, "fused vs torch-structured", rtol=1e-4, atol=1e-5)
    _close(o_f, o_d, "fused vs dense rows", rtol=2e-4, atol=2e-5)
    for (name, _), a, b, d in zip(base.named_parameters(), g_f, g_t, g_d):
        _close_grad(a, b, "grad " + name, n * N * H)
        _close_grad(a, d, "grad(dense) " + name, n * N * H)
    o_f2, g_f2 = run(True)
    assert torch.equal(o_f, o_f2) and all(torch.equal(a, b) for a, b in zip(g_f, g_f2))
    if lay.HD > fused.PRE_GEMM_ABOVE_HD:
        # more than 8 UAVs: the default forms head . Wh^T with a library GEMM (dcc_actor_l1_pre_*); the in-kernel form of the
        # same block (Wh^T in LDS, dcc_actor_l1_*) stays available behind the C-ABI and must agree
        keep = fused.PRE_GEMM_ABOVE_HD
        fused.PRE_GEMM_ABOVE_HD = 10 ** 9
        try:
            o_k, g_k = run(True)
        finally:
            fused.PRE_GEMM_ABOVE_HD = keep
        _close(o_f, o_k, "GEMM-assisted vs in-kernel head term", rtol=1e-4, atol=1e-5)
        for (name, _), a, b in zip(base.named_parameters(), g_f, g_k):
            _close_grad(a, b, "grad(in-kernel) " + name, n * N * H)


def test_shapes_outside_the_compiled_variants_are_refused():
    import dcc_hip
    L = dcc_hip.load_library()
    assert L.dcc_mlp_workspace_floats(256, 18) > 0 and L.dcc_mlp_workspace_floats(256, 0) > 0
    assert L.dcc_mlp_workspace_floats(1000, 0) == 0 and L.dcc_mlp_workspace_floats(256, 129) == 0
    z = torch.zeros(4, 1000, device="cuda")
    v = torch.zeros(1000, device="cuda")
    rc = L.dcc_relu_ln_fwd(z.data_ptr(), None, v.data_ptr(), v.data_ptr(), 1e-5, z.data_ptr(), 4, 1000, None)
    assert rc == -4


@pytest.mark.parametrize("R,H,A", [(4099, 256, 2), (3000, 256, 1), (77, 64, 2), (5, 32, 4), (301, 128, 3), (50, 100, 1)])
def test_relu_ln_head_matches_torch(R, H, A):
    """head(LayerNorm(ReLU(z))) with a narrow head (Gaussian mean A=2, value head A=1): output, dz and every
    parameter gradient against the unfused torch ops; bit-reproducible."""
    from algos.algo_utils import fused
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(R + H + A)
    z = torch.randn(R, H, device=dev, generator=g) * 1.3 + 0.1
    ln = torch.nn.LayerNorm(H).to(dev)
    head = torch.nn.Linear(H, A).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.5, 0.5); head.bias.uniform_(-1, 1)
    dy = torch.randn(R, A, device=dev, generator=g)
    zbias = (torch.randn(H, device=dev, generator=g) * 0.5).requires_grad_(True)
    params = list(ln.parameters()) + list(head.parameters()) + [zbias]

    def run(enabled):
        fused.ENABLED = enabled
        try:
            for p in params:
                p.grad = None
            zz = z.clone().requires_grad_(True)
            y = fused.relu_ln_head(zz, zbias, ln, head)
            if enabled:
                assert "ReluLNHead" in type(y.grad_fn).__name__
            y.backward(dy)
            return y.detach().clone(), zz.grad.clone(), [p.grad.clone() for p in params]
        finally:
            fused.ENABLED = True

    y1, dz1, g1 = run(True)
    y2, dz2, g2 = run(False)
    _close(y1, y2, "y", rtol=1e-4, atol=1e-5)
    _close(dz1, dz2, "dz", rtol=1e-4, atol=1e-5)
    for name, a, b in zip(("ln.weight", "ln.bias", "head.weight", "head.bias", "zbias"), g1, g2):
        _close(a, b, name, rtol=2e-4, atol=2e-5)
    y3, dz3, g3 = run(True)
    assert torch.equal(y1, y3) and torch.equal(dz1, dz3) and all(torch.equal(a, b) for a, b in zip(g1, g3))


@pytest.mark.parametrize("R,K,H", [(4915200 // 8, 256, 256), (65536 + 128 * 3, 18, 64), (70001, 40, 32)])
def test_split_k_linear_matches_plain(R, K, H):
    """The bias-free Linear with the batched (split-K) weight gradient == F.linear / autograd."""
    from algos.algo_utils import fused
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(R % 1000)
    x = torch.randn(R, K, device=dev, generator=g)
    W = torch.randn(H, K, device=dev, generator=g)
    dz = torch.randn(R, H, device=dev, generator=g)
    x1, W1 = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    y1 = fused.linear_w(x1, W1)
    assert "LinearSplitK" in type(y1.grad_fn).__name__
    y1.backward(dz)
    x2, W2 = x.clone().requires_grad_(True), W.clone().requires_grad_(True)
    y2 = torch.nn.functional.linear(x2, W2)
    y2.backward(dz)
    assert torch.equal(y1, y2)
    _close(x1.grad, x2.grad, "dx", rtol=1e-5, atol=1e-6)
    _close(W1.grad, W2.grad, "dW", rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,N,H", [(517, 8, 256), (1003, 4, 256), (33, 4, 64), (8200, 8, 64)])
def test_actor_l1_backward_one_and_two_kernel_variants_agree(n, N, H):
    """dq == NULL (dWh accumulated in registers) and dq != NULL (q stored, dWh = q^T head as a batched GEMM) are the
    same gradients; the second is what long batches use."""
    import dcc_hip
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n + N + H)
    HD = 4 + 2 * (N - 1)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    head, G = rnd(n, N, HD), rnd(n, H)
    stats = torch.stack([rnd(n, N).double() * 0.1 + 1.0, torch.rand(n, N, device=dev, generator=g).double() * 300 + 600], -1).contiguous()
    Wh, s, c = rnd(H, HD) * 0.2, rnd(H), rnd(H)
    gamma, dh = torch.rand(H, device=dev, generator=g) + 0.5, rnd(n * N, H)
    a = dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, gamma, dh, 1e-5, 1e-5, 338, two_kernel=False)
    b = dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, gamma, dh, 1e-5, 1e-5, 338, two_kernel=True)
    for name, x, y in zip(("dG", "dWh", "ds", "dc", "dgamma", "dbeta"), a, b):
        _close(y, x, name, rtol=1e-4, atol=1e-5)
    # (dG: the one-kernel form reads Wh^T from LDS in a k-loop, the q form holds it in registers fully unrolled: same sums, last-ulp
    # differences in the pre-activation; run-to-run each form is bit-reproducible)
    _close(b[0], a[0], "dG", rtol=1e-5, atol=1e-5)
    b2 = dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, gamma, dh, 1e-5, 1e-5, 338, two_kernel=True)
    assert all(torch.equal(x, y) for x, y in zip(b, b2))


@pytest.mark.parametrize("R,A,K,masked", [(100003, 2, 2, True), (5000, 2, 2, False), (777, 3, 1, True), (64, 1, 4, False)])
def test_fused_policy_loss_matches_autograd(R, A, K, masked):
    """dcc_ppo_policy_loss == autograd on the reference's expression (mappo.py:150-160, act.py:165-179): loss, entropy,
    mean ratio, d/d mean and d/d logstd; clipped and unclipped ratios on both sides, partially inactive rows."""
    from algos.algo_utils import fused
    from algos.algo_utils.distributions import FixedNormal
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(R + A)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    mean0 = rnd(R, A) * 0.5
    logstd0 = rnd(A) * 0.2
    actions = mean0 + rnd(R, A) * logstd0.exp()
    adv = rnd(R, 1)
    active = (torch.rand(R, 1, device=dev, generator=g) > 0.2).float()
    with torch.no_grad():       # old log-probs: the current ones perturbed so that some ratios leave [0.8, 1.2]
        lp = FixedNormal(mean0, logstd0.exp().expand(R, A), validate_args=False).log_probs(actions)
        old = (lp + rnd(R, 1) * 0.15).expand(R, K).contiguous()
    clip, coef = 0.2, 0.01

    def run(use_fused):
        mean, logstd = mean0.clone().requires_grad_(True), logstd0.clone().requires_grad_(True)
        if use_fused:
            pl, ent, ratio = fused.policy_loss(mean, logstd, actions, old, adv, active, clip, masked)
        else:
            dist = FixedNormal(mean, logstd.exp().view(1, A).expand(R, A), validate_args=False)
            logp = dist.log_probs(actions)
            imp = torch.exp(logp - old)
            surr = torch.min(imp * adv, torch.clamp(imp, 1 - clip, 1 + clip) * adv).sum(-1, keepdim=True)
            e = dist.entropy()
            if masked:
                pl = (-surr * active).sum() / active.sum()
                ent = (e * active).sum() / active.sum()
            else:
                pl, ent = -surr.mean(), e.mean()
            ratio = imp.mean()
        (pl - ent * coef).backward()
        return pl.detach(), ent.detach(), ratio.detach(), mean.grad.clone(), logstd.grad.clone()

    f, t = run(True), run(False)
    frac_clipped = float(((torch.exp(lp - old[:, :1]) - 1).abs() > clip).float().mean())
    assert 0.05 < frac_clipped < 0.95
    for name, a, b in zip(("policy_loss", "entropy", "ratio", "dmean", "dlogstd"), f, t):
        _close(a, b, name, rtol=2e-5, atol=2e-6)
    f2 = run(True)
    assert all(torch.equal(a, b) for a, b in zip(f, f2))


@pytest.mark.parametrize("n,N,huber,clipped,masked,normed", [(50001, 8, True, True, True, True), (3000, 4, False, True, False, True),
                                                            (777, 1, True, False, True, False), (64, 16, True, True, False, False)])
def test_fused_value_loss_matches_autograd(n, N, huber, clipped, masked, normed):
    """dcc_ppo_value_loss == autograd on MAPPOTrainer.cal_value_loss (mappo.py:103-131) with the one-sided Huber: loss and
    d/d values, errors on both sides of +-delta and of the clip range, partially inactive rows."""
    from algos.algo_utils import fused
    from utils.util import huber_loss, mse_loss
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n + N)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    R = n * N
    v0 = rnd(n, 1) * 6
    vp = v0.repeat_interleave(N, 0) + rnd(R, 1) * 0.4                 # some |v - vp| above the clip of 0.2
    ret = rnd(R, 1) * 300 - 200 if normed else v0.repeat_interleave(N, 0) + rnd(R, 1) * 14   # errors beyond +-10 too
    active = (torch.rand(R, 1, device=dev, generator=g) > 0.25).float()
    norm = torch.tensor([-190.0, 290.0], device=dev) if normed else None
    clip, delta = 0.2, 10.0

    def run(use_fused):
        v = v0.clone().requires_grad_(True)
        if use_fused:
            loss = fused.value_loss(v, vp, ret, active, norm, clip, delta if huber else None, clipped, masked, N)
        else:
            vr = v.unsqueeze(1).expand(-1, N, -1).reshape(-1, 1)
            vpc = vp + (vr - vp).clamp(-clip, clip)
            target = (ret - norm[0]) / norm[1] if normed else ret
            ec, eo = target - vpc, target - vr
            lc, lo = (huber_loss(ec, delta), huber_loss(eo, delta)) if huber else (mse_loss(ec), mse_loss(eo))
            l = torch.max(lo, lc) if clipped else lo
            loss = (l * active).sum() / active.sum() if masked else l.mean()
        loss.backward()
        return loss.detach(), v.grad.clone()

    f, t = run(True), run(False)
    _close(f[0], t[0], "value_loss", rtol=2e-5, atol=1e-6)
    _close(f[1], t[1], "dvalues", rtol=2e-5, atol=2e-6)
    f2 = run(True)
    assert torch.equal(f[0], f2[0]) and torch.equal(f[1], f2[1])


@pytest.mark.parametrize("n,H,with_stats", [(517, 256, True), (64, 64, True), (9000, 256, False), (3, 32, True)])
def test_critic_first_block_tail_no_head_term(n, H, with_stats):
    """dcc_actor_l1 with one row per env and NO per-row head term (N = 1, HD = 0): the centralised critic's first block after
    its per-env GEMM,  h = LayerNorm(ReLU(rstd_in (y - mean_in s) + c)).  Values, dy (= dG), ds, dc, dgamma, dbeta against the
    torch formulation; bit-reproducible."""
    from algos.algo_utils import fused
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(n + H)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    ln = torch.nn.LayerNorm(H).to(dev)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
    stats = None
    if with_stats:
        stats = torch.stack([rnd(n, 1).double() * 0.1 + 1.0, torch.rand(n, 1, device=dev, generator=g).double() * 300 + 600], -1).contiguous()
    D = 10816
    dh = rnd(n, H)

    def run(enabled):
        fused.ENABLED = enabled
        try:
            ln.zero_grad()
            y = (rnd(n, H) * 0 + y0).requires_grad_(True)
            s = s0.clone().requires_grad_(True); c = c0.clone().requires_grad_(True)
            h = fused.actor_l1(None, y, stats, None, s, c, ln, 1e-5, D)
            if enabled:
                assert "ActorL1" in type(h.grad_fn).__name__        # the HIP path ran
            h.backward(dh)
            gs = s.grad if s.grad is not None else torch.zeros_like(s)       # no input LayerNorm: s does not enter
            return h.detach().clone(), [y.grad.clone(), gs.clone(), c.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()]
        finally:
            fused.ENABLED = True

    y0, s0, c0 = rnd(n, H) * 1.5, rnd(H), rnd(H)
    h_f, g_f = run(True)
    h_t, g_t = run(False)
    _close(h_f, h_t, "h", rtol=1e-4, atol=1e-5)
    for name, a, b in zip(("dy", "ds", "dc", "dgamma", "dbeta"), g_f, g_t):
        _close_grad(a, b, name, n * H)
    h_f2, g_f2 = run(True)
    assert torch.equal(h_f, h_f2) and all(torch.equal(a, b) for a, b in zip(g_f, g_f2))
