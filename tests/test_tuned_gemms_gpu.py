"""config/gemm_tunings_gfx950.csv: the measured-fastest library GEMM per BASELINE shape (utils/pytorch_utils.use_tuned_gemms).
Lookup only: the file must be accepted by this installation's validators, nothing may be tuned at run time, and a listed
shape must still compute x W^T."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-coverage-control_amd")
sys.path.insert(0, PKG)

pytestmark = pytest.mark.gpu


def _file_validators(path):
    out = {}
    for line in open(path):
        f = line.strip().split(",")
        if f[0] == "Validator":
            out[f[1]] = f[2]
    return out


def test_tunings_file_is_wellformed_and_loads():
    import utils.pytorch_utils as ptu
    import torch.cuda.tunable as tun
    path = ptu.TUNED_GEMMS_FILE
    assert os.path.exists(path)
    val = _file_validators(path)
    assert val["GCN_ARCH_NAME"].startswith("gfx950")
    rows = [l.strip().split(",") for l in open(path) if not l.startswith("Validator")]
    assert len(rows) >= 40 and all(len(r) == 4 and r[0].startswith("Gemm") for r in rows)
    keys = {r[1] for r in rows}
    # the update's three big GEMMs at c3 (4.9 M rows) and the rollout's 32,768-row layer
    assert {"tn_256_4915200_256_ld_256_256_256", "nn_256_4915200_256_ld_256_256_256", "tn_256_32768_256_ld_256_256_256"} <= keys
    if not torch.__version__.startswith(val["PT_VERSION"]):
        pytest.skip("tunings were recorded with torch %s" % val["PT_VERSION"])
    ptu.set_gpu_mode(True, 0)
    n = ptu.use_tuned_gemms(path)
    ptu.set_gpu_mode(False)
    if n == 0:
        tun.enable(True)
        here = {k: v for k, v in tun.get_validators()}
        tun.enable(False)
        differ = {k: (val[k], here.get(k)) for k in val if k in here and here[k] != val[k]}
        if differ:      # another library build than the one the tunings were measured with: the default GEMM path is taken
            pytest.skip("tunings recorded for other library versions: %s" % differ)
    assert n >= len(rows), "torch rejected the tunings file (validators: %s vs %s)" % (val, tun.get_validators())
    assert tun.is_enabled() and not tun.tuning_is_enabled()


def test_tuned_shape_still_computes_the_product():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    n = ptu.use_tuned_gemms(ptu.TUNED_GEMMS_FILE)
    ptu.set_gpu_mode(False)
    if n == 0:
        pytest.skip("tunings not accepted by this installation")
    g = torch.Generator(device="cuda").manual_seed(5)
    R, H = 614400, 256                      # tn_256_614400_256 / nn_256_614400_256: the critic's layer 2 at c3
    x = torch.randn(R, H, device="cuda", generator=g)
    W = torch.randn(H, H, device="cuda", generator=g) * 0.05
    y = torch.nn.functional.linear(x, W)
    dx = y @ W
    idx = torch.arange(0, R, 4801, device="cuda")
    ref = x[idx].double() @ W.double().t()
    assert torch.allclose(y[idx].double(), ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(dx[idx].double(), ref @ W.double(), rtol=1e-5, atol=1e-5)
