"""autograd wrappers of the fused HIP stages of the policy trunks (include/dcc_mlp.h).

`relu_ln(z, ln)`      LayerNorm(ReLU(z)) -- the tail of every `Linear -> ReLU -> LayerNorm` block of the reference's
                      MLPLayer (uav_dcc_control/algos/algo_utils/mlp.py:13-16) in one pass; the backward recomputes
                      the statistics from z (saved anyway for the ReLU mask) instead of storing them.
`relu_ln_head(...)`   the last block's tail fused with the narrow Linear after it (Gaussian mean, value head).
`actor_l1(...)`       the actor's whole first block from the compact features (structured.py) -- the pre-activation
                      never reaches memory in either direction.
All fall back to the plain torch formulation when the tensors are not float32 CUDA tensors (CPU tests,
autocast) or the shape has no compiled variant; on a GPU box the library itself must load (dcc_hip raises).
"""
import os

import torch
import torch.nn.functional as F

# tests toggle this; DCC_FUSED_MLP=0 (diagnostic, tools/world8_ab.py) runs the plain torch formulations on the GPU as well
ENABLED = os.environ.get("DCC_FUSED_MLP", "1") != "0"


def _usable(t, H, HD=0):
    if not (ENABLED and t.is_cuda and t.dtype == torch.float32) or torch.is_autocast_enabled():
        return False
    import dcc_hip
    return dcc_hip.mlp_fused_supported(H, HD)


class _LinearSplitK(torch.autograd.Function):
    """x @ W^T without bias.  The weight gradient dW = dz^T x reduces over millions of rows into a [H, K] tile;
    hipBLASLt runs that shape at under half its rate for the other two GEMMs (10.2 vs 4.9 ms at 4.9 M x 256 x 256), so
    it is issued as a batched GEMM over row chunks plus one small sum (4.3 ms)."""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return F.linear(x, W)

    @staticmethod
    def backward(ctx, dz):
        x, W = ctx.saved_tensors
        dz = dz.contiguous()
        dx = dz @ W if ctx.needs_input_grad[0] else None
        R = x.shape[0]
        S = 128
        while S > 1 and R % S:
            S //= 2
        if S > 1 and R // S >= 512:
            dW = torch.bmm(dz.view(S, R // S, -1).transpose(1, 2), x.view(S, R // S, -1)).sum(0)
        else:
            dW = dz.t() @ x
        return dx, dW


class _SelectRows(torch.autograd.Function):
    """x[idx] for UNIQUE row indices idx (a row mini-batch): the backward is a plain scatter into zeros (index_copy_) instead of
    index_select's atomic index_add_."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        idx, = ctx.saved_tensors
        out = g.new_zeros((ctx.n,) + tuple(g.shape[1:]))
        out.index_copy_(0, idx, g.contiguous())
        return out, None


def select_rows(x, idx):
    """Rows `idx` (unique) of x [n, .], differentiable."""
    return _SelectRows.apply(x, idx)


def linear_w(x, W):
    """x [R, K] @ W^T for a weight tensor W [H, K] (split-K weight gradient for long batches on the GPU)."""
    if x.dim() == 2 and x.is_cuda and x.shape[0] >= 65536 and x.is_contiguous() and not torch.is_autocast_enabled():
        return _LinearSplitK.apply(x, W)
    return F.linear(x, W)


def linear_nobias(x, lin):
    """lin.weight applied to x [R, K]; the bias is left to the fused tail that follows."""
    return linear_w(x, lin.weight)


class _ReluLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias, gamma, beta, eps):
        import dcc_hip
        z = z.contiguous()
        ctx.save_for_backward(z, bias, gamma)
        ctx.eps = eps
        return dcc_hip.relu_ln_fwd(z, bias, gamma.contiguous(), beta.contiguous(), eps)

    @staticmethod
    def backward(ctx, dh):
        import dcc_hip
        z, bias, gamma = ctx.saved_tensors
        dz, dg, db, dbias = dcc_hip.relu_ln_bwd(z, bias, gamma.contiguous(), dh.contiguous(), ctx.eps)
        return dz, (dbias if bias is not None else None), dg, db, None


def relu_ln(z, bias, ln):
    """LayerNorm `ln` applied to ReLU(z + bias); z [R, H], bias [H] or None (the producing Linear's bias, whose gradient
    the fused backward delivers for free)."""
    if z.dim() == 2 and _usable(z, z.shape[1]):
        return _ReluLN.apply(z, bias, ln.weight, ln.bias, ln.eps)
    return ln(F.relu(z if bias is None else z + bias))


class _ReluLNHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias, gamma, beta, eps, Wo, bo):
        import dcc_hip
        z, gamma, beta, Wo = z.contiguous(), gamma.contiguous(), beta.contiguous(), Wo.contiguous()
        ctx.save_for_backward(z, bias, gamma, beta, Wo)
        ctx.eps, ctx.has_bo = eps, bo is not None
        return dcc_hip.relu_ln_head_fwd(z, bias, gamma, beta, eps, Wo, None if bo is None else bo.contiguous())

    @staticmethod
    def backward(ctx, dy):
        import dcc_hip
        z, bias, gamma, beta, Wo = ctx.saved_tensors
        dy = dy.contiguous()
        dz, dg, db, dbias, dWo = dcc_hip.relu_ln_head_bwd(z, bias, gamma, beta, ctx.eps, Wo, dy)
        return dz, (dbias if bias is not None else None), dg, db, None, dWo, (dy.sum(0) if ctx.has_bo else None)


HEAD_MAX_OUT = 4


def relu_ln_head(z, bias, ln, head):
    """head(LayerNorm_ln(ReLU(z + bias))) for a narrow nn.Linear `head` (action mean / value): the normalised
    activations are produced and consumed in registers."""
    if z.dim() == 2 and head.out_features <= HEAD_MAX_OUT and _usable(z, z.shape[1]):
        return _ReluLNHead.apply(z, bias, ln.weight, ln.bias, ln.eps, head.weight, head.bias)
    return head(relu_ln(z, bias, ln))


class _ActorL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, G, stats, Wh, s, c, gamma, beta, eps_in, eps_ln, D):
        import dcc_hip
        G, s, c = G.contiguous(), s.contiguous(), c.contiguous()
        if head is not None:
            head, Wh = head.contiguous(), Wh.contiguous()
        ctx.save_for_backward(head, G, stats, Wh, s, c, gamma)
        ctx.consts = (eps_in, eps_ln, D)
        return dcc_hip.actor_l1_fwd(head, G, stats, Wh, s, c, gamma.contiguous(), beta.contiguous(), eps_in, eps_ln, D)

    @staticmethod
    def backward(ctx, dh):
        import dcc_hip
        head, G, stats, Wh, s, c, gamma = ctx.saved_tensors
        eps_in, eps_ln, D = ctx.consts
        dG, dWh, ds, dc, dg, db = dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, gamma.contiguous(), dh.contiguous(),
                                                       eps_in, eps_ln, D)
        return None, dG, None, dWh, ds, dc, dg, db, None, None, None


class _ActorL1Pre(torch.autograd.Function):
    """The first block with the per-row term head . Wh^T ready-made by a library GEMM (include/dcc_mlp.h:
    dcc_actor_l1_pre_fwd / _bwd); the gradient w.r.t. `pre` flows back into that GEMM's autograd (dWh = dpre^T head)."""

    @staticmethod
    def forward(ctx, pre, G, stats, s, c, gamma, beta, eps_in, eps_ln, D):
        import dcc_hip
        pre, G, s, c = pre.contiguous(), G.contiguous(), s.contiguous(), c.contiguous()
        ctx.save_for_backward(pre, G, stats, s, c, gamma)
        ctx.consts = (eps_in, eps_ln, D)
        return dcc_hip.actor_l1_pre_fwd(pre, G, stats, s, c, gamma.contiguous(), beta.contiguous(), eps_in, eps_ln, D)

    @staticmethod
    def backward(ctx, dh):
        import dcc_hip
        pre, G, stats, s, c, gamma = ctx.saved_tensors
        eps_in, eps_ln, D = ctx.consts
        dpre, dG, ds, dc, dg, db = dcc_hip.actor_l1_pre_bwd(pre, G, stats, s, c, gamma.contiguous(), dh.contiguous(), eps_in, eps_ln, D)
        return dpre, dG, None, ds, dc, dg, db, None, None, None


# more head columns than this (more than 8 UAVs): the per-row product runs as a library GEMM instead of inside the kernel,
# whose LDS-resident Wh^T costs HD reads per lane and row (16 UAVs, 2.5 M rows: forward 4.2 -> 1.7 ms incl. the GEMM)
PRE_GEMM_ABOVE_HD = 18


def actor_l1(head, G, stats, Wh, s, c, ln, eps_in, D):
    """h1 [n*N, H] = LayerNorm_ln(ReLU(rstd_in * (head Wh^T + G[env] - mean_in s) + c)); stats None = no input LN.
    head = Wh = None: one row per env and no per-row term (the centralised critic's first block: G is its whole GEMM)."""
    if head is None:
        n, N, HD = G.shape[0], 1, 0
    else:
        n, N, HD = head.shape
    if HD > PRE_GEMM_ABOVE_HD and _usable(G, G.shape[1], 0):
        pre = linear_w(head.reshape(n * N, HD), Wh)
        return _ActorL1Pre.apply(pre, G, stats, s, c, ln.weight, ln.bias, float(eps_in or 0.0), ln.eps, int(D))
    if _usable(G, G.shape[1], HD):
        return _ActorL1.apply(head, G, stats, Wh, s, c, ln.weight, ln.bias, float(eps_in or 0.0), ln.eps, int(D))
    z = G.unsqueeze(1)
    if head is not None:
        z = F.linear(head.reshape(n * N, HD), Wh).view(n, N, -1) + G.to(Wh.dtype).unsqueeze(1)
    if stats is not None:
        rstd = torch.rsqrt(stats[..., 1] / D + eps_in).to(G.dtype).unsqueeze(-1)
        z = rstd * (z - stats[..., 0].to(G.dtype).unsqueeze(-1) * s) + c
    else:
        z = z + c
    return ln(F.relu(z.reshape(n * N, -1)))   # (the block's own Linear bias is already folded into c)


class _PolicyLoss(torch.autograd.Function):
    """PPO-clip surrogate + Gaussian entropy from the action mean in one HIP pass (include/dcc_mlp.h:
    dcc_ppo_policy_loss); returns (policy_loss, dist_entropy, mean ratio)."""

    @staticmethod
    def forward(ctx, mean, logstd, actions, old_logp, adv, active, clip, use_active):
        import dcc_hip
        R, A = mean.shape
        act = active.reshape(R).contiguous() if (use_active and active is not None) else None
        dmean_raw, sums = dcc_hip.ppo_policy_loss(mean.contiguous(), logstd.contiguous(), actions.contiguous(),
                                                  old_logp.contiguous(), adv.reshape(R).contiguous(), act, clip)
        denom = sums[1] if act is not None else torch.full((), float(R), device=mean.device)
        ctx.save_for_backward(dmean_raw, sums, denom)
        ctx.A, ctx.masked = A, act is not None
        policy_loss = -sums[0] / denom
        ent = (logstd + (0.5 + LOG_SQRT_2PI)).sum()          # sum over the action dims of the per-dim entropy
        if act is None:
            ent = ent / A                                      # `ent.mean()` averages over the dims as well (act.py:177-179)
        return policy_loss, ent, sums[2] / (R * old_logp.shape[1])

    @staticmethod
    def backward(ctx, g_pl, g_ent, g_ratio):
        dmean_raw, sums, denom = ctx.saved_tensors
        scale = g_pl / denom
        dlogstd = sums[4:4 + ctx.A] * scale + g_ent * (1.0 if ctx.masked else 1.0 / ctx.A)
        return dmean_raw * scale, dlogstd, None, None, None, None, None, None


LOG_SQRT_2PI = 0.91893853320467274178


def policy_loss_usable(mean, old_logp):
    return (ENABLED and mean.is_cuda and mean.dtype == torch.float32 and mean.dim() == 2 and mean.shape[1] <= HEAD_MAX_OUT
            and old_logp.shape[1] <= HEAD_MAX_OUT and not torch.is_autocast_enabled())


def policy_loss(mean, logstd, actions, old_logp, adv, active, clip, use_active):
    """(policy_loss, dist_entropy, ratio_mean) of mappo.py:150-160 / act.py:165-179 from the Gaussian mean."""
    return _PolicyLoss.apply(mean, logstd, actions, old_logp, adv, active, float(clip), bool(use_active))


class _ValueLoss(torch.autograd.Function):
    """Clipped Huber / MSE value loss from the per-env critic output in one HIP pass (dcc_ppo_value_loss)."""

    @staticmethod
    def forward(ctx, values, value_preds, returns, active, norm, clip, delta, use_clipped, use_masks, n_agents):
        import dcc_hip
        n = values.numel()
        R = n * n_agents
        act = active.reshape(R).contiguous() if (use_masks and active is not None) else None
        dv_raw, sums = dcc_hip.ppo_value_loss(values.reshape(n).contiguous(), value_preds.reshape(R).contiguous(),
                                              returns.reshape(R).contiguous(), act, norm, clip, delta, use_clipped, n_agents)
        denom = sums[1] if act is not None else torch.full((), float(R), device=values.device)
        ctx.save_for_backward(dv_raw, denom)
        ctx.shape = values.shape
        return sums[0] / denom

    @staticmethod
    def backward(ctx, g):
        dv_raw, denom = ctx.saved_tensors
        return (dv_raw * (g / denom)).view(ctx.shape), None, None, None, None, None, None, None, None, None


def value_loss_usable(values):
    return ENABLED and values.is_cuda and values.dtype == torch.float32 and not torch.is_autocast_enabled()


def value_loss(values, value_preds, returns, active, norm, clip, huber_delta, use_clipped, use_masks, n_agents):
    """mappo.py:103-131 on [n] critic outputs shared by n_agents rows each; norm = ValueNorm's {mean, std} tensor or None;
    huber_delta None = MSE."""
    return _ValueLoss.apply(values, value_preds, returns, active, norm, float(clip),
                            float(huber_delta) if huber_delta is not None else -1.0, bool(use_clipped), bool(use_masks),
                            int(n_agents))
