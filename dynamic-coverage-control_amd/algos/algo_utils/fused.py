"""autograd wrappers of the fused HIP stages of the policy trunks (include/dcc_mlp.h).

`relu_ln(z, ln)`      LayerNorm(ReLU(z)) -- the tail of every `Linear -> ReLU -> LayerNorm` block of the reference's
                      MLPLayer (uav_dcc_control/algos/algo_utils/mlp.py:13-16) in one pass; the backward recomputes
                      the statistics from z (saved anyway for the ReLU mask) instead of storing them.
`actor_l1(...)`       the actor's whole first block from the compact features (structured.py) -- the pre-activation
                      never reaches memory in either direction.
Both fall back to the plain torch formulation when the tensors are not float32 CUDA tensors (CPU tests, bf16
autocast) or the shape has no compiled variant; on a GPU box the library itself must load (dcc_hip raises).
"""
import torch
import torch.nn.functional as F

ENABLED = True      # cfg.fused_mlp (learner.py) / tests toggle this


def _usable(t, H, HD=0):
    if not (ENABLED and t.is_cuda and t.dtype == torch.float32) or torch.is_autocast_enabled():
        return False
    import dcc_hip
    return dcc_hip.mlp_fused_supported(H, HD)


class _ReluLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, gamma, beta, eps):
        import dcc_hip
        z = z.contiguous()
        ctx.save_for_backward(z, gamma)
        ctx.eps = eps
        return dcc_hip.relu_ln_fwd(z, gamma.contiguous(), beta.contiguous(), eps)

    @staticmethod
    def backward(ctx, dh):
        import dcc_hip
        z, gamma = ctx.saved_tensors
        dz, dg, db = dcc_hip.relu_ln_bwd(z, gamma.contiguous(), dh.contiguous(), ctx.eps)
        return dz, dg, db, None


def relu_ln(z, ln):
    """LayerNorm `ln` applied to ReLU(z); z [R, H]."""
    if z.dim() == 2 and _usable(z, z.shape[1]):
        return _ReluLN.apply(z, ln.weight, ln.bias, ln.eps)
    return ln(F.relu(z))


class _ActorL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, G, stats, Wh, s, c, gamma, beta, eps_in, eps_ln, D):
        import dcc_hip
        head, G, Wh, s, c = head.contiguous(), G.contiguous(), Wh.contiguous(), s.contiguous(), c.contiguous()
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


def actor_l1(head, G, stats, Wh, s, c, ln, eps_in, D):
    """h1 [n*N, H] = LayerNorm_ln(ReLU(rstd_in * (head Wh^T + G[env] - mean_in s) + c)); stats None = no input LN."""
    n, N, HD = head.shape
    if _usable(G, G.shape[1], HD):
        return _ActorL1.apply(head, G, stats, Wh, s, c, ln.weight, ln.bias, float(eps_in or 0.0), ln.eps, int(D))
    z = F.linear(head.reshape(n * N, HD), Wh).view(n, N, -1) + G.to(Wh.dtype).unsqueeze(1)
    if stats is not None:
        rstd = torch.rsqrt(stats[..., 1] / D + eps_in).to(Wh.dtype).unsqueeze(-1)
        z = rstd * (z - stats[..., 0].to(Wh.dtype).unsqueeze(-1) * s) + c
    else:
        z = z + c
    return ln(F.relu(z.reshape(n * N, -1)))
