"""The update's three big fp32 GEMMs (c3: 4.9 M rows x 256 x 256) under the library heuristic vs torch TunableOp's pick.

Run on the GPU box: python tools/gemm_tune_probe.py [rows]
"""
import sys, time, torch, torch.nn.functional as F
dev = "cuda"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 150 * 4096 * 8
H = 256
x = torch.randn(R, H, device=dev)
dz = torch.randn(R, H, device=dev)
W = torch.randn(H, H, device=dev) * 0.05
S = 128
fl = 2 * R * H * H / 1e12


def t(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


cases = {"fwd   F.linear(x, W)": lambda: F.linear(x, W),
         "dgrad dz @ W": lambda: dz @ W,
         "wgrad bmm 128 chunks": lambda: torch.bmm(dz.view(S, R // S, H).transpose(1, 2), x.view(S, R // S, H)),
         "wgrad dz.t() @ x": lambda: dz.t() @ x}


def run(tag):
    out = {}
    for k, f in cases.items():
        ms = t(f)
        out[k] = ms
        print("%-12s %-24s %.2f ms  %.0f TF/s" % (tag, k, ms, fl / ms * 1e3), flush=True)
    return out


base = run("default")
import torch.cuda.tunable as tun
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
tun.set_max_tuning_iterations(10)
tun.set_filename("/tmp/tunable_probe.csv")
t0 = time.time()
run("tuning")
print("tuning pass took %.1f s" % (time.time() - t0))
tuned = run("tuned")
for k in cases:
    print("%-24s default %.2f ms -> tuned %.2f ms (%.2fx)" % (k, base[k], tuned[k], base[k] / tuned[k]))
for r in tun.get_results():
    print(r)
