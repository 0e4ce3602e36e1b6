"""Library behaviour beyond 2^31 elements: the weight-gradient product dz^T x over R rows of 256 columns, as a plain GEMM and
as the split-K batched GEMM the trainer uses, against a float64 reference accumulated over 1 M-row chunks -- at the c3
batch (4.9 M rows: 1.26e9 elements per operand) and at the full c5 shard (9.8 M rows: 2.5e9 elements > 2^31).
usage: python tools/big_rows_probe.py <rows_in_units_of_150x32 envs>   e.g. 1024 (c3-equivalent rows) or 2048"""
import sys, torch
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
R, H = 150 * E * 32, 256
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(R, H, device=dev, generator=g); dz = torch.randn(R, H, device=dev, generator=g)
ref = torch.zeros(H, H, dtype=torch.float64, device=dev)
for r0 in range(0, R, 1 << 20):
    ref += dz[r0:r0 + (1 << 20)].double().t() @ x[r0:r0 + (1 << 20)].double()
scale = float(ref.abs().max())
print("rows %d  elements per operand %.3g (2^31 = 2.147e9)" % (R, R * H))
for name, fn in (("plain dz.t() @ x", lambda: dz.t() @ x),
                 ("bmm 128 chunks + sum", lambda: torch.bmm(dz.view(128, R // 128, -1).transpose(1, 2), x.view(128, R // 128, -1)).sum(0)),
                 ("bmm 16 chunks + sum", lambda: torch.bmm(dz.view(16, R // 16, -1).transpose(1, 2), x.view(16, R // 16, -1)).sum(0))):
    y = fn()
    torch.cuda.synchronize()
    print("%-22s max |err| / max |ref| = %.3e" % (name, float((y.double() - ref).abs().max()) / scale))
W = torch.randn(H, H, device=dev, generator=g) * 0.05
y = dz @ W
err = 0.0
for r0 in (0, R // 2, R - 4096):
    err = max(err, float((y[r0:r0 + 4096].double() - dz[r0:r0 + 4096].double() @ W.double()).abs().max()))
print("dgrad dz @ W           max |err| on sampled row blocks = %.3e" % err)
