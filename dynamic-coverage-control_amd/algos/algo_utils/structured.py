"""First layer of actor and critic evaluated from compact features instead of observation rows.

No counterpart in the reference: it builds every observation row (scenarios/coverage.py:99-110), stores it
(buffer/shared_buffer.py:36-44) and feeds it to LayerNorm -> Linear (algos/algo_utils/mlp.py:45-58).  At
8 UAV x 64 PoI a row has D = 1352 columns, and the centralised row N*D = 10816, of which only 4 + 2(N-1) = 18 per
agent depend on the agent in a non-trivial way:

    x_i = [ vel_i, pos_i, (pos_a - pos_i) for a != i  |  (poi_j - pos_i, energy_j, m_energy, done_j) for j ]

With the LayerNorm affine folded into the Linear (W' = W * gamma, c = b + W beta) and (mu_i, rstd_i) the moments of
row i,

    Linear(LayerNorm(x_i)) = rstd_i * (x_i W'^T - mu_i * rowsum(W')) + c

and x_i W'^T splits by linearity into

    head_i W'_head^T  -  pos_i (sum_j W'_xy,j)^T                      per agent   (18 + 2 inputs)
  + energy W'_e^T + done W'_d^T                                      per env     (2M inputs, shared by its agents)
  + sum_j poi_j W'_xy,j^T + m_energy * sum_j W'_m,j                   constant    (once per parameter value)

so the layer costs (2M/N + 2N + 2) multiply-adds per output and agent row instead of D (37x fewer at 8 x 64), reads
~1/37 of the bytes, and the rows never exist.  The centralised critic is the same with per-agent weight blocks
summed over the agents.  It is the same function of the parameters (a re-association of each dot product, 1e-6
relative in fp32) and autograd differentiates it through the slicing/summing of the original parameters, so the
parameters, their gradients and the optimizer state stay those of the reference's layer.  Inputs: the outputs of
dcc_obs_features (include/dcc_env.h) or, for callers that hold rows, features_from_obs().
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused


def _pad8(k):
    return (k + 7) // 8 * 8


class ObsLayout(object):
    """Column layout of an observation row (coverage.py:99-110) + the constants that appear in it."""

    def __init__(self, n_agents, n_pois, poi_xy, m_energy):
        self.N, self.M = int(n_agents), int(n_pois)
        self.HD = 4 + 2 * (self.N - 1)
        self.D = self.HD + 5 * self.M
        self.poi_xy = torch.as_tensor(poi_xy, dtype=torch.float64).reshape(self.M, 2)
        self.m_energy = float(m_energy)
        self._poi32 = {}

    def poi(self, like):
        key = (like.device, like.dtype)
        t = self._poi32.get(key)
        if t is None:
            t = self._poi32[key] = self.poi_xy.to(device=like.device, dtype=like.dtype)
        return t


def features_from_obs(obs, layout):
    """The three feature tensors from materialised rows obs [n,N,D] (any device): what dcc_obs_features computes
    from state.  Used where rows already exist (numpy drop-in callers, tests)."""
    n, N, D = obs.shape
    HD, M = layout.HD, layout.M
    head = obs[..., :HD].contiguous()
    blk = obs[:, 0, HD:].reshape(n, M, 5)
    poi_feat = torch.cat([blk[..., 2], blk[..., 4]], dim=1).contiguous()
    x = obs.double()
    mean = x.mean(-1)
    m2 = ((x - mean.unsqueeze(-1)) ** 2).sum(-1)
    xc = x.reshape(n, N * D)
    cmean = xc.mean(-1)
    cm2 = ((xc - cmean.unsqueeze(-1)) ** 2).sum(-1)
    return dict(head=head, poi_feat=poi_feat, stats=torch.stack([mean, m2], dim=-1),
                cstats=torch.stack([cmean, cm2], dim=-1))


class _FoldedWeights(torch.autograd.Function):
    """All weight-side quantities of the structured layer in one autograd node (the slicing / summing of the
    parameters written with plain torch ops costs >100 tiny kernel launches per network and epoch through autograd;
    here ~15 forward and ~15 backward).  Nb = 1 (actor: W [H, D]) or N (critic: W [H, N*D], per-agent blocks).

        wf = W * gamma,  c = b + W beta,  s = rowsum(wf)
        w_h [H, Nb*HD]  head columns of every block, the pos_i columns minus sum_j wf_xy,j
        w_ed [H, 2M]    energy and done columns, summed over the blocks
        const [H]       sum_j poi_j wf_xy,j + m_energy sum_j wf_m,j   (over all blocks)
        w_env           the weight of the per-env GEMM (env_gemm_inputs): [w_ed | const | 0..] for the actor (Nb = 1, its head
                        term is applied per agent row by the fused kernel), [w_h | w_ed | const | 0..] for the critic
    """

    @staticmethod
    def forward(ctx, W, b, gamma, beta, poi, m_energy, Nb, HD, M, with_head):
        H = W.shape[0]
        D = HD + 5 * M
        wf = W * gamma if gamma is not None else W
        c = b + W @ beta if beta is not None else b.clone()
        s = wf.sum(1)
        w3 = wf.view(H, Nb, D)
        P = w3[:, :, HD:].reshape(H, Nb, M, 5)
        w_h = w3[:, :, :HD].clone()
        w_h[:, :, 2:4] -= P[..., 0:2].sum(2)
        w_h = w_h.reshape(H, Nb * HD)
        const = (P[..., 0:2] * poi).sum((1, 2, 3)) + m_energy * P[..., 3].sum((1, 2))
        k0 = Nb * HD if with_head else 0
        w_env = W.new_zeros(H, _pad8(k0 + 2 * M + 1))
        if k0:
            w_env[:, :k0] = w_h
        w_env[:, k0:k0 + M] = P[..., 2].sum(1)
        w_env[:, k0 + M:k0 + 2 * M] = P[..., 4].sum(1)
        w_env[:, k0 + 2 * M] = const
        ctx.save_for_backward(W, gamma, beta, poi)
        ctx.dims = (Nb, HD, M, float(m_energy), bool(with_head))
        return w_h, w_env, s, c

    @staticmethod
    def backward(ctx, g_wh, g_wenv, g_s, g_c):
        W, gamma, beta, poi = ctx.saved_tensors
        Nb, HD, M, m_energy, with_head = ctx.dims
        H = W.shape[0]
        k0 = Nb * HD if with_head else 0
        g_wh3 = (g_wh + g_wenv[:, :k0] if k0 else g_wh).reshape(H, Nb, HD)
        g_const = g_wenv[:, k0 + 2 * M]
        gP = torch.empty(H, Nb, M, 5, dtype=W.dtype, device=W.device)
        gP[..., 0:2] = g_const.view(H, 1, 1, 1) * poi - g_wh3[:, :, 2:4].unsqueeze(2)
        gP[..., 2] = g_wenv[:, k0:k0 + M].unsqueeze(1)
        gP[..., 3] = (m_energy * g_const).view(H, 1, 1)
        gP[..., 4] = g_wenv[:, k0 + M:k0 + 2 * M].unsqueeze(1)
        g_wf = torch.cat([g_wh3, gP.view(H, Nb, 5 * M)], dim=2).view(H, -1) + g_s.unsqueeze(1)
        if gamma is not None:
            g_W = g_wf * gamma
            g_gamma = (g_wf * W).sum(0)
        else:
            g_W, g_gamma = g_wf, None
        g_beta = None
        if beta is not None:
            g_W = g_W + g_c.unsqueeze(1) * beta
            g_beta = g_c @ W
        return g_W, g_c, g_gamma, g_beta, None, None, None, None, None, None


def folded_weights(base, layout, Nb, with_head=False):
    """(w_h, w_env, s, c, eps) of the structured first layer of `base` (see _FoldedWeights).

    With autograd enabled this is a graph node.  Without (rollout / evaluation) the result only changes when the
    parameters do, so it is cached on `base` keyed by the parameters' version counters and refreshed IN PLACE -- the
    tensors keep their addresses, which lets a captured hipGraph of the rollout keep reading them (the learner calls
    refresh_folded_weights() before every replay, since a replay runs no Python)."""
    lin = base.mlp.fc1[0]
    if base._use_feature_normalization:
        ln = base.feature_norm
        gamma, beta, eps = ln.weight, ln.bias, ln.eps
    else:
        gamma = beta = eps = None
    args = (lin.weight, lin.bias, gamma, beta, layout.poi(lin.weight), layout.m_energy, Nb, layout.HD, layout.M, with_head)
    if torch.is_grad_enabled() or torch.is_autocast_enabled():
        return _FoldedWeights.apply(*args) + (eps,)
    versions = tuple(-1 if t is None else t._version for t in (lin.weight, lin.bias, gamma, beta))
    cache = base.__dict__.setdefault("_folded_cache", {})
    hit = cache.get((Nb, with_head))
    if hit is None or hit[0] != versions or hit[1][0].device != lin.weight.device or hit[1][0].dtype != lin.weight.dtype:
        with torch.no_grad():
            fresh = _FoldedWeights.apply(*args)
        if hit is not None and all(a.shape == b.shape and a.device == b.device and a.dtype == b.dtype
                                   for a, b in zip(hit[1], fresh)):
            for a, b in zip(hit[1], fresh):
                a.copy_(b)
            hit = (versions, hit[1])
        else:
            hit = (versions, tuple(t.clone() for t in fresh))
        cache[(Nb, with_head)] = hit
    return hit[1] + (eps,)


def invalidate_folded_weights(*modules):
    """Mark the inference caches stale but keep their tensors (addresses).  The optimizer step calls this: parameter
    version counters are not enough, because a step replayed inside a hipGraph updates the parameters without running
    the Python that bumps them."""
    for m in modules:
        cache = getattr(m, "base", m).__dict__.get("_folded_cache")
        if cache:
            for k, (_, tensors) in list(cache.items()):
                cache[k] = (None, tensors)


def refresh_folded_weights(actor, critic):
    """Bring the inference caches of both networks up to date (call before replaying a captured rollout)."""
    with torch.no_grad():
        if getattr(actor, "obs_layout", None) is not None:
            folded_weights(actor.base, actor.obs_layout, 1)
        if getattr(critic, "obs_layout", None) is not None:
            folded_weights(critic.base, critic.obs_layout, critic.obs_layout.N, True)


def _tail(blk, z, bias=None):
    """activation + LayerNorm of a `Linear -> act -> LayerNorm` block on its pre-activation z + bias (fused when ReLU)."""
    if isinstance(blk[1], nn.ReLU):
        return fused.relu_ln(z, bias, blk[2])
    return blk[2](blk[1](z if bias is None else z + bias))


def _rest(base, h, head=None):
    """fc2 blocks on the first block's output; with `head` (a narrow nn.Linear) returns head(features), the last
    block's ReLU + LayerNorm and the head in one pass."""
    blocks = list(base.mlp.fc2)
    for i, blk in enumerate(blocks):
        z = fused.linear_nobias(h, blk[0])
        if head is not None and i == len(blocks) - 1 and isinstance(blk[1], nn.ReLU):
            return fused.relu_ln_head(z, blk[0].bias, blk[2], head)
        h = _tail(blk, z, blk[0].bias)
    return h if head is None else head(h)


def env_gemm_inputs(feats, critic):
    """Input matrix of the per-env GEMM of the structured first layer, one row per env state, with a trailing column of ones
    so that the constant term rides in the same GEMM (no broadcast add over the [n, H] result and no column-sum in its
    backward), zero-padded to a multiple of 8 columns (16-byte aligned rows):
        actor   X = [energy | done | 1 | 0..]                      -> G  = X [w_ed | const | 0]^T    shared by the env's agents
        critic  X = [head_0 .. head_{N-1} | energy | done | 1 | 0..] -> y = X [w_h | w_ed | const | 0]^T
    Parameter-free: the rollout buffer builds it once per chunk next to the features (key "xa" / "xc"); built on the fly when
    the caller passed raw features."""
    key = "xc" if critic else "xa"
    x = feats.get(key)
    if x is None:
        head_f, poi_feat = feats["head"], feats["poi_feat"]
        n = poi_feat.shape[0]
        cols = ([head_f.reshape(n, -1)] if critic else []) + [poi_feat, torch.ones(n, 1, dtype=poi_feat.dtype, device=poi_feat.device)]
        k = sum(c.shape[1] for c in cols)
        if _pad8(k) > k:
            cols.append(torch.zeros(n, _pad8(k) - k, dtype=poi_feat.dtype, device=poi_feat.device))
        x = torch.cat(cols, dim=1)
    return x


def actor_trunk(base, layout, feats, head=None, row_sel=None):
    """MLPBase(obs rows) for the n*N agent rows described by feats -> [n*N, H] (or head(.) -> [n*N, A]).
    row_sel (row mini-batches, SharedReplayBuffer.minibatch_rows): indices into the n*N rows; only those rows go on past the
    first block (whose per-env part is shared by an env's agents anyway) -> [len(row_sel), .]."""
    head_f, stats = feats["head"], feats["stats"]
    n, N, HD = head_f.shape
    w_h, w_env, s_w, c, eps = folded_weights(base, layout, 1)            # [H,HD], [H,pad8(2M+1)], [H], [H]
    g = fused.linear_w(env_gemm_inputs(feats, False), w_env)             # [n, H] shared by the agents of an env
    blk = base.mlp.fc1
    if isinstance(blk[1], nn.ReLU):   # one fused pass: the pre-activation never reaches memory (include/dcc_mlp.h)
        h = fused.actor_l1(head_f, g, stats if eps is not None else None, w_h, s_w, c, blk[2], eps, layout.D)
        return _rest(base, h if row_sel is None else fused.select_rows(h, row_sel), head)
    z = F.linear(head_f.reshape(n * N, HD), w_h).view(n, N, -1) + g.unsqueeze(1)
    if eps is not None:
        mean = stats[..., 0]
        rstd = torch.rsqrt(stats[..., 1] / layout.D + eps)
        z = rstd.to(z.dtype).unsqueeze(-1) * (z - mean.to(z.dtype).unsqueeze(-1) * s_w) + c
    else:
        z = z + c
    z = z.reshape(n * N, -1)
    return _rest(base, _tail(blk, z if row_sel is None else fused.select_rows(z, row_sel)), head)


def critic_trunk(base, layout, feats, head=None):
    """MLPBase(centralised rows = concat of the N agent rows of an env) -> [n, H] (or head(.) -> [n, A]).
    ONE GEMM over [head_0..head_{N-1} | energy | done | 1] per env, then the same fused first-block tail as the actor's
    (dcc_actor_l1 with one row per env and no per-row head term): input-LayerNorm correction, bias, ReLU, LayerNorm in one
    pass, the pre-activation never stored."""
    head_f, stats = feats["head"], feats["stats"]
    n, N, HD = head_f.shape
    w_h, w_env, s_w, c, eps = folded_weights(base, layout, N, True)      # [H,N*HD], [H,pad8(N*HD+2M+1)], [H], [H]
    y = fused.linear_w(env_gemm_inputs(feats, True), w_env)              # [n, H]
    cstats = None
    if eps is not None:
        cstats = feats.get("cstats")
        if cstats is None:                                                     # pool the per-agent moments of the N*D-wide row
            mean_i, m2_i = stats[..., 0], stats[..., 1]                        # [n, N] float64
            mean = mean_i.mean(1, keepdim=True)
            m2 = (m2_i + layout.D * (mean_i - mean) ** 2).sum(1, keepdim=True)
            cstats = torch.cat([mean, m2], dim=1)
    blk = base.mlp.fc1
    if isinstance(blk[1], nn.ReLU):
        h = fused.actor_l1(None, y, None if cstats is None else cstats.reshape(n, 1, 2), None, s_w, c, blk[2], eps, N * layout.D)
        return _rest(base, h, head)
    if cstats is not None:
        rstd = torch.rsqrt(cstats[:, 1:2] / (N * layout.D) + eps)
        z = rstd.to(y.dtype) * (y - cstats[:, 0:1].to(y.dtype) * s_w) + c
    else:
        z = y + c
    return _rest(base, _tail(blk, z), head)
