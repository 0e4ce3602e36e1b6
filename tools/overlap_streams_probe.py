"""Do an MFMA-bound library GEMM and an HBM-bound streaming kernel overlap when issued on two HIP streams?
(the PPO update alternates them: 17.4 ms of fp32 GEMMs at 129-146 TF/s and 9.5 ms of streaming tails per c3 epoch).
Prints the time of each alone, back to back on one stream, and concurrently on two streams."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import utils.pytorch_utils as ptu
ptu.set_gpu_mode(True, 0)
ptu.use_tuned_gemms()
import dcc_hip
rows = 2457600                      # half of the c3 batch (one of two chunks)
dev = torch.device("cuda")
x = torch.randn(rows, 256, device=dev); W = torch.randn(256, 256, device=dev)
z = torch.randn(rows, 256, device=dev); g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev); bias = torch.zeros(256, device=dev)
def gemm(n):
    for _ in range(n): y = x @ W.t()
def tail(n):
    for _ in range(n): h = dcc_hip.relu_ln_fwd(z, bias, g, b, 1e-5)
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
gemm(3); tail(3)
n = 20
tg, tt = timed(lambda: gemm(n)) / n, timed(lambda: tail(n)) / n
def serial():
    for _ in range(n): gemm(1); tail(1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def conc():
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1): gemm(n)
    with torch.cuda.stream(s2): tail(n)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
def conc_alt():      # two chains, each alternating GEMM and tail (what two update chunks on two streams look like)
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for i in range(n // 2):
        with torch.cuda.stream(s1): gemm(1); tail(1)
        with torch.cuda.stream(s2): gemm(1); tail(1)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
ts, tc, ta = timed(serial) / n, timed(conc) / n, timed(conc_alt) / n
print("rows %d: GEMM alone %.3f ms (%.1f TF/s), relu_ln tail alone %.3f ms (%.2f TB/s)" % (rows, tg, 2 * rows * 65536 / tg / 1e9, tt, 2 * rows * 1024 / tt / 1e9))
print("one stream, alternating: %.3f ms per (GEMM + tail) pair;  two streams (GEMMs | tails): %.3f ms;  two streams, each alternating: %.3f ms"
      % (ts, tc, ta))
