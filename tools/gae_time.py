"""GAE kernel timing at the c3 column count and 4x it.  Run on the GPU box: python tools/gae_time.py"""
import sys, torch
sys.path.insert(0, "dynamic-coverage-control_amd")
import dcc_hip
T, C = 150, 4096 * 8
g = torch.Generator(device="cuda").manual_seed(0)
rew = torch.randn(T, C, device="cuda", generator=g)
vp = torch.randn(T + 1, C, device="cuda", generator=g)
mk = (torch.rand(T + 1, C, device="cuda", generator=g) > 0.02).float()
dn = torch.tensor([0.3, 1.7], device="cuda")
for C2 in (C, C * 4):
    if C2 != C:
        rew = rew.repeat(1, 4); vp = vp.repeat(1, 4); mk = mk.repeat(1, 4)
    ret = torch.empty_like(vp); adv = torch.empty_like(rew)
    for _ in range(5):
        dcc_hip.gae_compute(rew, vp, mk, dn, 0.99, 0.95, ret, adv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dcc_hip.gae_compute(rew, vp, mk, dn, 0.99, 0.95, ret, adv)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("GAE T=%d C=%d: %.1f us, %.0f GB/s of 20 B per step and column" % (T, C2, us, 20 * T * C2 / us / 1e3))
