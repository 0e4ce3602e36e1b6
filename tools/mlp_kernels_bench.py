"""Time the fused policy-trunk kernels (include/dcc_mlp.h) at the c3 update shapes on the GPU box:
R = 150 x 4096 x 8 agent rows, H = 256.  Prints ms per call and the algorithmic HBM rate of each."""
import os, sys, time
import torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, "dynamic-coverage-control_amd"))
import dcc_hip

dev = torch.device("cuda", 0)
n, N, M, H = 150 * 4096, 8, 64, 256
R = n * N
HD = 4 + 2 * (N - 1)


def t(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


z = torch.randn(R, H, device=dev); dh = torch.randn(R, H, device=dev)
g = torch.rand(H, device=dev) + 0.5; b = torch.randn(H, device=dev); bias = torch.randn(H, device=dev)
Wo = torch.randn(2, H, device=dev) * 0.01; bo = torch.zeros(2, device=dev); dy = torch.randn(R, 2, device=dev)
head = torch.randn(n, N, HD, device=dev); G = torch.randn(n, H, device=dev)
stats = torch.stack([torch.randn(n, N, dtype=torch.float64, device=dev) * 0.1 + 1, torch.rand(n, N, dtype=torch.float64, device=dev) * 300 + 600], -1).contiguous()
Wh = torch.randn(H, HD, device=dev) * 0.1; s = torch.randn(H, device=dev); c = torch.randn(H, device=dev)
GB = 1e-6  # bytes -> GB/s with ms
rows = [
    ("relu_ln_fwd", lambda: dcc_hip.relu_ln_fwd(z, bias, g, b, 1e-5), 2 * R * H * 4),
    ("relu_ln_bwd", lambda: dcc_hip.relu_ln_bwd(z, bias, g, dh, 1e-5), 3 * R * H * 4),
    ("relu_ln_head_fwd", lambda: dcc_hip.relu_ln_head_fwd(z, bias, g, b, 1e-5, Wo, bo), R * H * 4),
    ("relu_ln_head_bwd", lambda: dcc_hip.relu_ln_head_bwd(z, bias, g, b, 1e-5, Wo, dy), 2 * R * H * 4),
    ("actor_l1_fwd", lambda: dcc_hip.actor_l1_fwd(head, G, stats, Wh, s, c, g, b, 1e-5, 1e-5, 338), R * H * 4 + n * H * 4 + R * HD * 4),
    ("actor_l1_bwd (1 kernel: dWh in registers)", lambda: dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, g, dh, 1e-5, 1e-5, 338, two_kernel=False), R * H * 4 + 2 * n * H * 4 + R * HD * 4),
    ("actor_l1_bwd (q + GEMM)", lambda: dcc_hip.actor_l1_bwd(head, G, stats, Wh, s, c, g, dh, 1e-5, 1e-5, 338, two_kernel=True), 3 * R * H * 4 + 2 * n * H * 4 + 2 * R * HD * 4),
]
for name, fn, nbytes in rows:
    ms = t(fn)
    print("%-24s %7.2f ms  %6.0f GB/s" % (name, ms, nbytes / ms * GB))
