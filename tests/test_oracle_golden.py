"""The CPU restatement (oracle/dcc_oracle.c) against golden vectors produced by the reference
itself (tools/gen_golden.py).  Everything is expected BIT-EXACT in float64: the restatement
follows the reference's operation order, numpy's type promotion for float32 actions, numpy's
pairwise np.sum and OpenBLAS' fused ddot in np.linalg.norm."""
import os

import numpy as np
import pytest

from conftest import golden_env_files, load_case


@pytest.mark.parametrize("path", golden_env_files(), ids=lambda p: os.path.basename(p)[4:-4])
def test_oracle_matches_reference_golden(path, oracle_mod):
    z, c = load_case(path)
    o = oracle_mod.OracleEnv(c["E"], c["N"], c["M"], z["poi"], c["r_cover"], c["r_comm"], c["comm_r_scale"],
                             c["comm_force_scale"])
    obs0 = o.reset()
    assert np.array_equal(obs0[0].astype(np.float32), z["reset_obs"])
    actions = z["actions"]
    assert actions.dtype == (np.float32 if c["act_f32"] else np.float64)
    obs_steps = list(z["obs_steps"])
    for t in range(c["T"]):
        out = o.step(actions[t])
        for k in ("pos_t", "vel_t", "reward", "coverage"):
            assert np.array_equal(out[k], z[k][t]), (k, t, np.abs(out[k] - z[k][t]).max())
        for k in ("done", "connect", "connect_s", "done_t", "assign"):
            assert np.array_equal(out[k], z[k][t]), (k, t)
        assert np.array_equal(out["energy_t"], z["energy_t"][t].astype(np.float64)), ("energy", t)
        ob32 = out["obs"].astype(np.float32)
        assert np.array_equal(ob32.astype(np.float64).reshape(c["E"], -1).sum(1), z["obs_sum"][t]), ("obs_sum", t)
        if t in obs_steps:
            assert np.array_equal(ob32, z["obs"][obs_steps.index(t)]), ("obs", t)
    o.close()


def test_known_answers_appendix_b(oracle_mod):
    """Known-answer values measured on the reference (re-measured by importing the reference; shipped N=4, M=20)."""
    z, c = load_case([p for p in golden_env_files() if p.endswith("env_n4m20_shipped.npz")][0])
    o = oracle_mod.OracleEnv(1, 4, 20, z["poi"], 0.2, 0.4, 0.9, 0.0)
    obs = o.reset()
    assert obs.shape == (1, 4, 110)
    np.testing.assert_allclose(obs[0, 0, 10:12], [-0.0663509, -0.47629586], atol=1e-8)
    out = o.step(np.zeros((1, 4, 2)))
    assert out["reward"][0] == -56.51319293982027 and out["done"][0] == 0 and out["coverage"][0] == 0.0
    o.reset()
    a = np.zeros((1, 4, 2)); a[..., 0] = 1.0
    for t in range(1, 31):
        out = o.step(a)
        assert bool(out["done"][0]) == (t == 30)
    assert out["pos_t"][0, 0, 0] == 1.5000000000000007 and out["vel_t"][0, 0, 0] == 0.5
    assert out["reward"][0] == -2510.318269572213 and out["coverage"][0] == 0.2
    assert np.array_equal(o.get_state()["pos"], np.zeros((1, 4, 2)))  # auto-reset happened


def test_rng_actions_range_and_determinism(oracle_mod):
    a = oracle_mod.rng_actions(7, 3, 64, 8)
    b = oracle_mod.rng_actions(7, 3, 64, 8)
    assert np.array_equal(a, b) and a.min() >= -1.0 and a.max() < 1.0
    assert abs(a.mean()) < 0.1 and 0.5 < a.std() < 0.65
    # sharding invariance: the stream is a function of the global env id
    lo = oracle_mod.rng_actions(7, 3, 32, 8, env0=0, env_total=64)
    hi = oracle_mod.rng_actions(7, 3, 32, 8, env0=32, env_total=64)
    assert np.array_equal(np.concatenate([lo, hi]), a)
