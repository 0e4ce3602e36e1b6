import glob
import os
import sys

import numpy as np
import pytest

os.environ.setdefault("DCC_TESTING", "1")      # the one gate in front of the package's test seams (utils/pytorch_utils.require_testing); inherited by subprocesses

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-coverage-control_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def golden_env_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "env_*.npz")))


def load_case(path):
    z = np.load(path)
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    return z, cfg


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


DELTA_LOG = []      # (label, worst relative delta error, worst tensor): printed in the terminal summary (quoted in DESIGN.md 2)


def check_param_deltas(module, before, after, tol, label, abs_tol=None):
    """Post-update parameters against a reference fixture, compared as UPDATES: for every tensor
        max |(p_got - p_before) - (p_ref_after - p_before)|  <=  tol * max |p_ref_after - p_before|
    (an absolute window on the parameter itself is 3-10 % of a 2-step Adam update and says little about the gradient).
    `before` / `after`: name -> numpy array of the fixture; abs_tol adds the usual absolute check as a second line.
    Returns the worst relative error; it is also logged per test for the summary."""
    worst, worst_name = 0.0, ""
    for k, v in module.state_dict().items():
        if k not in after:
            continue
        p0, p1 = np.asarray(before[k], np.float64), np.asarray(after[k], np.float64)
        got = v.detach().double().cpu().numpy()
        d_ref, d_got = p1 - p0, got - p0
        scale = float(np.abs(d_ref).max())
        if scale == 0.0:
            assert float(np.abs(d_got).max()) == 0.0, "%s %s: reference did not move, this did" % (label, k)
            continue
        err = float(np.abs(d_got - d_ref).max()) / scale
        if err > worst:
            worst, worst_name = err, k
        assert err <= tol, "%s %s: update differs by %.3e of max|delta| = %.3e (tolerance %.1e)" % (label, k, err, scale, tol)
        if abs_tol is not None:
            np.testing.assert_allclose(got, p1, rtol=1e-3, atol=abs_tol, err_msg="%s %s" % (label, k))
    DELTA_LOG.append((label, worst, worst_name))
    return worst


def pytest_terminal_summary(terminalreporter):
    if DELTA_LOG:
        terminalreporter.write_sep("-", "parameter-update errors vs the reference fixtures (max |d_got - d_ref| / max |d_ref| per test)")
        agg = {}
        for label, worst, name in DELTA_LOG:
            key = label.split("[")[0]
            if worst >= agg.get(key, (-1.0, ""))[0]:
                agg[key] = (worst, "%s %s" % (label, name))
        for key, (worst, where) in sorted(agg.items()):
            terminalreporter.write_line("  %-60s %.2e   (%s)" % (key, worst, where))
