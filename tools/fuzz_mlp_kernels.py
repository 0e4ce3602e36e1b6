"""Differential fuzz of the fused policy-trunk kernels (include/dcc_mlp.h) against the torch formulation of the same algebra:
the first block in all its forms (register-resident 4 / 8 UAV kernels, LDS-resident generic kernels, GEMM-assisted kernels for
many UAVs, the critic's no-head form) and the head tails, at random row counts / agent counts / widths, values and all
gradients.  Run on the GPU box: python tools/fuzz_mlp_kernels.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-coverage-control_amd"))
from algos.algo_utils import fused  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(seed)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)


def close(a, b, what, tol=3e-4, nsum=1):
    """relative to the largest entry, plus the rounding noise of a sum of nsum O(1) terms (a gradient whose terms cancel has a
    tiny largest entry but the noise of its terms)"""
    scale = float(b.abs().max()) + 1e-20
    err = float((a - b).abs().max())
    assert err <= tol * scale + 2e-6 * nsum ** 0.5, (what, err, scale)


def mask_ambiguous_rows(z, dh, thr=2e-5):
    """Two fp32 evaluations of a pre-activation differ by ~1e-7 relative, so a |z| that small may sit on either side of the
    ReLU in the two implementations, and a flipped element changes its whole row's LayerNorm backward by O(1).  Rows holding
    such an element get a zero upstream gradient: they still check the forward values, but cannot decide the comparison of
    the gradients by luck."""
    amb = (z.abs() < thr * float(z.abs().max())).any(dim=1)
    dh[amb] = 0
    return int(amb.sum())


def run_l1(enabled, head, G, stats, Wh, s, c, ln, eps, D, dh):
    fused.ENABLED = enabled
    try:
        leaves = [t for t in (G, Wh, s, c, ln.weight, ln.bias) if t is not None]
        for t in leaves:
            t.grad = None
        out = fused.actor_l1(head, G, stats, Wh, s, c, ln, eps, D)
        out.backward(dh)
        return out.detach().clone(), [torch.zeros_like(t) if t.grad is None else t.grad.clone() for t in leaves]   # s is unused without input moments
    finally:
        fused.ENABLED = True


t0 = time.time()
cases = {"l1": 0, "head": 0}
AMB = {"rows": 0}
while time.time() - t0 < budget:
    H = int(rs.choice([32, 64, 128, 256, 256, 256]))
    if rs.rand() < 0.7:
        N = int(rs.choice([0, 1, 2, 4, 4, 5, 8, 8, 8, 12, 16, 32, 40]))        # 0: the critic's form (no head term)
        n = int(rs.choice([1, 2, 3, 5, 17, 64, 255, 256, 257, 1000, 4097, 20000]))
        if max(N, 1) * n * H > 6e7:
            continue
        HD = 4 + 2 * (N - 1) if N else 0
        D = HD + 5 * int(rs.randint(1, 300))
        head = rnd(n, N, HD) if N else None
        Wh = (rnd(H, HD) * 0.2).requires_grad_() if N else None
        G = rnd(n, H).requires_grad_()
        s, c = rnd(H).requires_grad_(), rnd(H).requires_grad_()
        with_stats = rs.rand() < 0.7
        rows = n * max(N, 1)
        stats = None
        if with_stats:
            stats = torch.stack([rnd(rows).double() * 0.1 + 1.0, torch.rand(rows, device=dev, generator=g).double() * 300 + 600], -1)
            stats = stats.view(n, max(N, 1), 2).contiguous()
        ln = torch.nn.LayerNorm(H).to(dev)
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5); ln.bias.uniform_(-0.3, 0.3)
        dh = rnd(rows, H)
        with torch.no_grad():      # the pre-activation, torch formulation (fused.actor_l1 with the kernels off computes the same)
            zz = G.unsqueeze(1) if head is None else torch.nn.functional.linear(head.reshape(rows, HD), Wh).view(n, N, -1) + G.unsqueeze(1)
            if with_stats:
                rstd_in = torch.rsqrt(stats[..., 1] / D + 1e-5).float().unsqueeze(-1)
                zz = rstd_in * (zz - stats[..., 0].float().unsqueeze(-1) * s) + c
            else:
                zz = zz + c
            AMB["rows"] += mask_ambiguous_rows(zz.reshape(rows, H), dh)
        keep = fused.PRE_GEMM_ABOVE_HD
        fused.PRE_GEMM_ABOVE_HD = keep if rs.rand() < 0.6 else 10 ** 9       # also the in-kernel head term for many UAVs
        try:
            o_f, g_f = run_l1(True, head, G, stats, Wh, s, c, ln, 1e-5 if with_stats else None, D, dh)
        finally:
            fused.PRE_GEMM_ABOVE_HD = keep
        o_t, g_t = run_l1(False, head, G, stats, Wh, s, c, ln, 1e-5 if with_stats else None, D, dh)
        tag = ("l1", n, N, H, with_stats)
        try:
            close(o_f, o_t, "values")
            for i, (a, b) in enumerate(zip(g_f, g_t)):
                close(a, b, "grad %d" % i, 1e-3, rows)
        except AssertionError as e:
            print("MISMATCH", tag, e, flush=True); sys.exit(1)
        cases["l1"] += 1
    else:
        R = int(rs.choice([1, 2, 7, 63, 64, 65, 1000, 4099, 50000]))
        A = int(rs.choice([1, 2, 4]))
        z = (rnd(R, H) * 1.7 + 0.2).requires_grad_()
        bias = rnd(H).requires_grad_() if rs.rand() < 0.7 else None
        ln = torch.nn.LayerNorm(H).to(dev)
        lin = torch.nn.Linear(H, A).to(dev)
        dy = rnd(R, A)
        with torch.no_grad():
            AMB["rows"] += mask_ambiguous_rows(z if bias is None else z + bias, dy)
        outs = []
        for en in (True, False):
            fused.ENABLED = en
            for t in [z, bias, ln.weight, ln.bias, lin.weight, lin.bias]:
                if t is not None:
                    t.grad = None
            y = fused.relu_ln_head(z, bias, ln, lin)
            y.backward(dy)
            outs.append((y.detach().clone(), [t.grad.clone() for t in [z, bias, ln.weight, ln.bias, lin.weight, lin.bias] if t is not None]))
        fused.ENABLED = True
        try:
            close(outs[0][0], outs[1][0], "y")
            for i, (a, b) in enumerate(zip(outs[0][1], outs[1][1])):
                close(a, b, "grad %d" % i, 1e-3, R)
        except AssertionError as e:
            print("MISMATCH", ("head", R, H, A), e, flush=True); sys.exit(1)
        cases["head"] += 1
print("fuzz: %d first-block cases and %d head-tail cases in %.0f s: values within 3e-4, gradients within 1e-3 of the torch "
      "formulation (relative to the largest entry; %d rows sitting on a ReLU threshold kept out of the gradient comparison)"
      % (cases["l1"], cases["head"], time.time() - t0, AMB["rows"]))
