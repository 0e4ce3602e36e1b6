"""Why does the same binary on the same box alternate between two speeds (c2: 1.20 vs 1.28 ms per 150-step launch)?
Runs the c2 rollout launch back to back for `secs` seconds, records every launch time (HIP events) and samples rocm-smi
(clocks, power, temperature) every ~2 s; prints the time series side by side.  usage: python tools/mode_probe.py [secs]"""
import os, re, subprocess, sys, threading, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "dynamic-coverage-control_amd"))
import dcc_hip
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
E, N, M, T = 4096, 8, 64, 150
poi = np.load(os.path.join(R, "dynamic-coverage-control_amd", "envs", "mpe", "pos_pois.npy"))[:M]
env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
env.reset()
out = env.alloc_out(T)
acts = torch.from_numpy(np.random.default_rng(0).uniform(-1, 1, (T, E, N, 2)).astype(np.float32)).cuda()
samples, stop = [], [False]
def smi():
    while not stop[0]:
        t = time.time()
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showuse"], capture_output=True, text=True, timeout=10).stdout
            g = lambda pat: (re.search(pat, o) or [None, "?"])[1]
            samples.append((t, g(r"sclk clock level: \S+ \((\d+)Mhz\)"), g(r"mclk clock level: \S+ \((\d+)Mhz\)"), g(r"fclk clock level: \S+ \((\d+)Mhz\)"),
                            g(r"socclk clock level: \S+ \((\d+)Mhz\)"), g(r"Power \(W\): ([\d.]+)"), g(r"Temperature \(Sensor junction\) \(C\): ([\d.]+)"),
                            g(r"Temperature \(Sensor HBM 0\) \(C\): ([\d.]+)"), g(r"GPU use \(%\): (\d+)")))
        except Exception as e:  # noqa: BLE001
            samples.append((t, "err", str(e)[:40], "", "", "", "", "", ""))
        time.sleep(1.5)
th = threading.Thread(target=smi, daemon=True); th.start()
t0 = time.time(); rows = []
while time.time() - t0 < secs:
    ev = []
    for _ in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); env.rollout(T, actions=acts, out=out); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    ms = [x.elapsed_time(y) for x, y in ev]
    rows.append((time.time() - t0, float(np.mean(ms)), float(np.min(ms)), float(np.max(ms))))
stop[0] = True; th.join(timeout=5)
print("t_s   launch_ms avg / min / max   | nearest rocm-smi sample: sclk mclk fclk socclk MHz, W, Tj, Thbm, use%")
for t, a, mn, mx in rows[::max(1, len(rows) // 60)]:
    s = min(samples, key=lambda q: abs(q[0] - (t0 + t))) if samples else None
    print("%5.1f  %.4f %.4f %.4f | %s" % (t, a, mn, mx, " ".join(str(x) for x in s[1:]) if s else ""))
av = np.array([r[1] for r in rows])
print("launch ms over %d windows of 40 launches: min %.4f median %.4f max %.4f; share of windows above 1.24 ms: %.2f" % (len(av), av.min(), np.median(av), av.max(), (av > 1.24).mean()))
