"""CPU-side checks of the C-ABI: the shared library loads without a GPU and exports every symbol
declared in include/*.h; argument validation fails loudly (no compute calls here)."""
import ctypes
import glob
import os
import re

import pytest

from conftest import ROOT, PKG


def _declared():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        names += re.findall(r"DCC_API\s+[\w\s\*]+?\b(dcc_\w+)\s*\(", open(h).read())
    return sorted(set(names))


def _libpath():
    p = os.path.join(PKG, "csrc", "libdcc_hip.so")
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    return p


def test_header_declares_expected_entry_points():
    d = _declared()
    for n in ("dcc_env_create", "dcc_env_step", "dcc_env_reset", "dcc_env_rollout", "dcc_env_destroy",
              "dcc_gae_compute"):
        assert n in d, n


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_libpath())
    for n in _declared():
        assert hasattr(lib, n), "missing export %s" % n
    assert lib.dcc_abi_version() == 2


def test_library_exports_nothing_the_headers_do_not_declare():
    """Every dcc_* symbol in the dynamic symbol table of libdcc_hip.so is declared in include/*.h (no debug back doors)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _libpath()], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("dcc_")}
    assert exported, "no dcc_* exports found"
    assert exported <= set(_declared()), sorted(exported - set(_declared()))


def test_cpu_twins_cover_the_reference_path_entry_points():
    """oracle/libdcc_oracle.so carries a `_cpu` twin (same signature, host pointers) of the entry points that
    stand for reference functions: the env life cycle / step / rollout / state of include/dcc_env.h and the GAE scan of
    include/dcc_gae.h, the observation rows from a state (dcc_obs_expand) and, since round 5, their reduction to the compact policy-input
    features (dcc_obs_features / _x).  (The remaining declarations are device utilities without a reference counterpart -- kernel choice,
    write probe, the one-launch step + features, the trunk / loss / optimizer kernels -- checked against the torch formulation instead.)"""
    from oracle import oracle
    L = ctypes.CDLL(oracle.build())
    for n in ("dcc_env_create", "dcc_env_destroy", "dcc_env_obs_dim", "dcc_env_reset", "dcc_env_step", "dcc_env_rollout",
              "dcc_env_get_state", "dcc_env_set_state", "dcc_last_error", "dcc_obs_expand", "dcc_obs_features", "dcc_obs_features_x",
              "dcc_gae_compute", "dcc_returns_compute"):
        assert n in _declared() and hasattr(L, n + "_cpu"), n


def test_binding_lists_match_header():
    import dcc_hip
    assert set(dcc_hip.EXPORTS) <= set(_declared())


def test_create_validates_arguments_and_has_no_cpu_path():
    import numpy as np
    import torch
    import dcc_hip
    L = dcc_hip.load_library()
    cfg = dcc_hip.EnvCfg()
    L.dcc_env_cfg_default(ctypes.byref(cfg))
    assert cfg.dt == 0.1 and cfg.damping == 0.25 and cfg.rew_done == 1500.0 and cfg.comm_r_scale == 0.9
    h = ctypes.c_void_p()
    poi = np.zeros((4, 2))
    cfg.n_envs, cfg.n_agents, cfg.n_pois = 1, 65, 4
    cfg.poi_xy = poi.ctypes.data_as(ctypes.c_void_p)
    assert L.dcc_env_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"n_agents" in L.dcc_last_error()
    cfg.n_agents, cfg.n_pois = 4, 2000
    assert L.dcc_env_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    cfg.n_pois, cfg.poi_xy = 4, None
    assert L.dcc_env_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    if not torch.cuda.is_available():
        # no device: creation must FAIL (negative code), never fall back to a CPU implementation
        cfg.poi_xy = poi.ctypes.data_as(ctypes.c_void_p)
        assert L.dcc_env_create(ctypes.byref(cfg), ctypes.byref(h)) < 0
        with pytest.raises(dcc_hip.DccError):
            dcc_hip.HipCoverageEnv(1, 4, 4, poi)


def test_bytes_per_step_matches_survey_figures():
    import dcc_hip
    # c1 1,787 B, c2 11,851 B, c4 87,563 B, c5 676,363 B
    assert dcc_hip.bytes_per_step(4, 16) == 1787
    assert dcc_hip.bytes_per_step(8, 64) == 11851
    assert dcc_hip.bytes_per_step(16, 256) == 87563
    assert dcc_hip.bytes_per_step(32, 1024) == 676363
    assert dcc_hip.bytes_per_step(8, 64, with_actions=False) == 11851 - 64


def test_mlp_entry_points_validate_before_touching_the_device():
    """include/dcc_mlp.h: shape support query and argument validation run on the host (no GPU needed here)."""
    import dcc_hip
    L = dcc_hip.load_library()
    assert L.dcc_mlp_workspace_floats(256, 18) >= L.dcc_mlp_workspace_floats(256, 0) > 0 and L.dcc_mlp_workspace_floats(256, 66) > 0
    assert L.dcc_mlp_workspace_floats(64, 10) > 0 and L.dcc_mlp_workspace_floats(512, 0) > 0
    for H, HD in ((1000, 0), (130, 0), (0, 0), (256, 129), (512, 100), (516, 0)):     # (512, 100): Wh^T would need 200 KB of LDS
        assert L.dcc_mlp_workspace_floats(H, HD) == 0, (H, HD)
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.dcc_relu_ln_fwd(None, None, p, p, 1e-5, p, 4, 8, None) == -1            # null z
    assert L.dcc_relu_ln_fwd(p, None, p, p, 1e-5, p, 4, 1000, None) == -4            # unsupported width
    assert L.dcc_relu_ln_fwd(p, None, p, p, 1e-5, p, 0, 8, None) == 0                # empty batch: nothing to launch
    assert L.dcc_relu_ln_bwd(p, None, p, p, 1e-5, p, p, None, 4, 8, None) == -1      # no workspace
    assert L.dcc_relu_ln_head_fwd(p, None, p, p, 1e-5, p, None, p, 4, 8, 5, None) == -4    # head wider than 4
    assert L.dcc_relu_ln_head_bwd(p, None, p, p, 1e-5, p, p, p, p, None, 4, 8, 2, None) == -1
    assert L.dcc_actor_l1_fwd(p, p, None, p, p, p, p, p, 1e-5, 1e-5, 90, p, 1, 4, 130, 8, None) == -4   # head width > 128
    assert L.dcc_actor_l1_bwd(p, p, None, p, p, p, p, p, 1e-5, 1e-5, 90, p, None, None, p, p, p, p, p, 1, 4, 10, 8, None) == -1
    assert L.dcc_ppo_policy_loss(p, p, p, p, p, None, 0.2, p, p, p, 16, 5, 2, None) == -4   # more than 4 action dims
    assert L.dcc_ppo_policy_loss(p, p, p, p, p, None, 0.2, None, p, p, 16, 2, 2, None) == -1


def test_headers_are_plain_c():
    """include/*.h must be consumable from C (the boundary is a C ABI): a C99 translation unit including all of them."""
    import subprocess, tempfile
    hdrs = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert len(hdrs) >= 3
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        for h in hdrs:
            f.write('#include "%s"\n' % os.path.basename(h))
        f.write("int main(void) { return dcc_abi_version(); }\n")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I",
                        os.path.join(ROOT, "include"), f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 0, r.stderr
