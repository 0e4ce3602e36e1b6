"""-m gpu: the HIP env (through the C-ABI) against (1) the reference's golden vectors and (2) the
CPU oracle on the same inputs.

Bar (BASELINE.json north_star): bit-exact connectivity / done masks, PoI energy and PoI-assignment
indices; positions / rewards / observations within 1e-5 (relative for rewards).  The kernel keeps
float64 state and the reference's operation order, so in practice positions are bit-exact whenever
the pull force is off (only + - * / sqrt fma are involved) and within a few ulp when it is on
(device exp/log1p differ from glibc in the last bits)."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_env_files, load_case

pytestmark = pytest.mark.gpu

POS_TOL = 1e-9      # absolute, float64 positions/velocities (spec: 1e-5)
OBS_TOL = 1e-5      # absolute, float32 observations
REW_RTOL = 1e-5     # relative, rewards (|R| reaches 2e4 -> float32 ulp 2e-3)


def _mk(c, z, E=None):
    import dcc_hip
    return dcc_hip.HipCoverageEnv(E or c["E"], c["N"], c["M"], z["poi"], c["r_cover"], c["r_comm"], c["comm_r_scale"],
                                  c["comm_force_scale"])


@pytest.mark.parametrize("variant", ["spec-roles", "spec-roles1", "spec-fused", "generic-roles", "generic-roles1", "generic-fused"])
@pytest.mark.parametrize("path", golden_env_files(), ids=lambda p: os.path.basename(p)[4:-4])
def test_hip_step_matches_reference_golden(path, variant, monkeypatch):
    """Every kernel family must reproduce the reference: compile-time (N, M) specialisations vs the generic
    runtime-size code (DCC_NO_SPEC=1), and the multi-wave kernels -- role-specialised physics/observation waves
    (M <= 64) or the split kernel with one physics and three observation waves per env (M > 64) -- vs the fused
    one-wave-per-env kernel (DCC_NO_ROLES=1 / DCC_NO_SPLIT=1)."""
    z, c = load_case(path)
    spec, roles = variant.split("-")
    has_spec = (c["N"], c["M"]) in ((8, 64), (4, 16), (4, 20), (16, 256))
    if spec == "generic" and not has_spec:
        pytest.skip("no specialised kernel for this size: the generic path is what 'spec' already ran")
    if roles.startswith("roles") and c["M"] > 64 and not c["act_f32"]:
        pytest.skip("the split kernel takes float32 / in-kernel actions; float64 actions use the fused kernel")
    if roles == "roles1" and c["M"] > 64:
        pytest.skip("one env per workgroup is a shape of the role-specialised kernel (<= 64 PoIs); the split kernel always is")
    # the role-specialised kernel serves two adjacent envs per workgroup (large batches) or one (small, latency-bound batches)
    monkeypatch.setenv("DCC_ROLES_ENVS", "1" if roles == "roles1" else "2")
    roles = "roles" if roles == "roles1" else roles
    monkeypatch.setenv("DCC_NO_SPEC", "1" if spec == "generic" else "0")
    monkeypatch.setenv("DCC_NO_ROLES", "1" if roles == "fused" else "0")
    monkeypatch.setenv("DCC_FORCE_ROLES", "1" if roles == "roles" else "0")   # single-step launches default to fused
    monkeypatch.setenv("DCC_NO_SPLIT", "1" if roles == "fused" else "0")
    monkeypatch.setenv("DCC_FORCE_SPLIT", "1" if roles == "roles" else "0")
    env = _mk(c, z)
    dev = env.device
    obs0 = env.reset()
    assert np.array_equal(obs0[0].cpu().numpy(), z["reset_obs"])
    actions = torch.from_numpy(z["actions"]).to(dev)
    obs_steps = list(z["obs_steps"])
    exact_pos = True
    for t in range(c["T"]):
        # terminal state is only observable before the auto-reset: run on a snapshot-able path by
        # reading the state when the env did not finish, and the outputs otherwise
        out = env.step(actions[t], env.alloc_out(reward64=True))
        st = env.get_state()
        torch.cuda.synchronize()
        done = out["done"].cpu().numpy()
        assert np.array_equal(done, z["done"][t]), ("done", t)
        assert np.array_equal(out["connect"].cpu().numpy(), z["connect"][t]), ("connect", t)
        assert np.array_equal(out["connect_s"].cpu().numpy(), z["connect_s"][t]), ("connect_s", t)
        assert np.array_equal(out["assign"].cpu().numpy(), z["assign"][t]), ("assign", t)
        np.testing.assert_allclose(out["coverage"].cpu().numpy(), z["coverage"][t].astype(np.float32), rtol=0, atol=0)
        ref_r = z["reward"][t]
        np.testing.assert_allclose(out["reward64"].cpu().numpy(), ref_r, rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(out["reward"].cpu().numpy(), ref_r, rtol=REW_RTOL, atol=1e-5)
        live = done == 0
        pos, vel = st["pos"].cpu().numpy(), st["vel"].cpu().numpy()
        assert np.abs(pos[live] - z["pos_t"][t][live]).max(initial=0) <= POS_TOL, ("pos", t)
        assert np.abs(vel[live] - z["vel_t"][t][live]).max(initial=0) <= POS_TOL, ("vel", t)
        exact_pos &= np.array_equal(pos[live], z["pos_t"][t][live]) and np.array_equal(vel[live], z["vel_t"][t][live])
        assert np.array_equal(st["energy"].cpu().numpy()[live], z["energy_t"][t][live].astype(np.float32)), ("energy", t)
        assert np.array_equal(st["done"].cpu().numpy()[live], z["done_t"][t][live]), ("done_t", t)
        # finished envs were reset to the origin (wrappers.py:104-109)
        assert not pos[~live].any() and not vel[~live].any() and not st["energy"].cpu().numpy()[~live].any()
        obs = out["obs"].cpu().numpy()
        np.testing.assert_allclose(obs.astype(np.float64).reshape(c["E"], -1).sum(1), z["obs_sum"][t], rtol=0,
                                   atol=1e-5 * obs[0].size)
        if t in obs_steps:
            np.testing.assert_allclose(obs, z["obs"][obs_steps.index(t)], rtol=0, atol=OBS_TOL)
    if c["comm_force_scale"] == 0:
        assert exact_pos, "force-off trajectories are expected bit-exact in float64"
    env.close()


@pytest.mark.parametrize("N,M,cfs,r_comm", [(8, 64, 0.0, 0.4), (8, 64, 0.5, 0.2), (5, 37, 0.5, 0.3), (16, 256, 0.5, 0.15),
                                            (32, 1024, 0.5, 0.1), (3, 130, 1.0, 0.3), (64, 70, 0.5, 0.08),
                                            (64, 1024, 0.5, 0.05), (1, 1, 0.0, 0.4), (2, 64, 1.0, 0.3), (9, 500, 0.0, 0.2),
                                            (1, 100, 0.0, 0.4), (2, 200, 1.0, 0.5), (7, 65, 0.5, 0.3)])
@pytest.mark.parametrize("multi_wave", [False, True, 2])
def test_hip_step_matches_oracle_random(N, M, cfs, r_comm, multi_wave, oracle_mod, monkeypatch):
    """Seeded random actions, E=33 envs (not a multiple of the 4 envs per workgroup), 40 steps; with the fused kernel
    and with the multi-wave kernel of the size (roles for M <= 64, split for M > 64) forced for the single-step launches."""
    if multi_wave == 2 and M > 64:
        pytest.skip("two envs per workgroup is a shape of the role-specialised kernel (<= 64 PoIs)")
    monkeypatch.setenv("DCC_FORCE_ROLES", "1" if multi_wave else "0")
    monkeypatch.setenv("DCC_FORCE_SPLIT", "1" if multi_wave else "0")
    monkeypatch.setenv("DCC_ROLES_ENVS", "2" if multi_wave == 2 else "0")     # 0: by batch size (33 envs -> one env per workgroup)
    E, T = (33, 40) if N * M < 20000 else (5, 12)
    rs = np.random.RandomState(N * 1000 + M)
    poi = rs.uniform(-1, 1, (M, 2))
    import dcc_hip
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
    o0 = env.reset().cpu().numpy()
    assert np.array_equal(o0, orc.reset().astype(np.float32))
    scale = rs.uniform(0.3, 3.0, (E, 1, 1))  # some envs drift out of bounds and reset
    bias = rs.uniform(-0.5, 0.5, (E, N, 2))
    for t in range(T):
        a = np.clip(rs.uniform(-1, 1, (E, N, 2)) * scale + bias, -1, 1).astype(np.float32)
        out = env.step(torch.from_numpy(a).to(env.device), env.alloc_out(reward64=True))
        ref = orc.step(a)
        for k in ("done", "connect", "connect_s"):
            assert np.array_equal(out[k].cpu().numpy(), ref[k]), (k, t)
        assert np.array_equal(out["assign"].cpu().numpy().astype(np.int32), ref["assign"]), ("assign", t)
        np.testing.assert_allclose(out["reward64"].cpu().numpy(), ref["reward"], rtol=1e-11, atol=1e-8)
        np.testing.assert_allclose(out["coverage"].cpu().numpy(), ref["coverage"].astype(np.float32), rtol=0, atol=0)
        np.testing.assert_allclose(out["obs"].cpu().numpy(), ref["obs"].astype(np.float32), rtol=0, atol=OBS_TOL)
        st, so = env.get_state(), orc.get_state()
        np.testing.assert_allclose(st["pos"].cpu().numpy(), so["pos"], rtol=0, atol=POS_TOL)
        np.testing.assert_allclose(st["vel"].cpu().numpy(), so["vel"], rtol=0, atol=POS_TOL)
        assert np.array_equal(st["energy"].cpu().numpy(), so["energy"].astype(np.float32))
        assert np.array_equal(st["done"].cpu().numpy(), so["done"])
    env.close()


def test_hip_f64_actions_match_oracle(oracle_mod):
    E, N, M, T = 7, 8, 64, 30
    rs = np.random.RandomState(5)
    poi = rs.uniform(-1, 1, (M, 2))
    import dcc_hip
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.2, 0.9, 0.5)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.2, 0.9, 0.5)
    env.reset(); orc.reset()
    for t in range(T):
        a = rs.uniform(-1, 1, (E, N, 2))
        out = env.step(torch.from_numpy(a).to(env.device))
        ref = orc.step(a)
        assert np.array_equal(out["done"].cpu().numpy(), ref["done"])
        assert np.array_equal(out["connect_s"].cpu().numpy(), ref["connect_s"])
        np.testing.assert_allclose(env.get_state()["pos"].cpu().numpy(), orc.get_state()["pos"], rtol=0, atol=POS_TOL)


def test_rollout_rng_matches_oracle_and_single_steps(oracle_mod):
    """K fused steps with the in-kernel generator == oracle driven by the same generator == K
    single-step launches fed the generated actions."""
    E, N, M, K = 64, 8, 64, 150
    import dcc_hip
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    env2 = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    env.reset(); env2.reset(); orc.reset()
    out = env.rollout(K, seed=1234, step0=7, env0=0, env_total=E)
    ref = orc.rollout_rng(K, 1234, 7, 0, E, want_obs_last=True)
    assert np.array_equal(out["done"].cpu().numpy(), ref["done"])
    np.testing.assert_allclose(out["reward"].cpu().numpy(), ref["reward"], rtol=REW_RTOL, atol=1e-5)
    np.testing.assert_allclose(out["coverage"].cpu().numpy(), ref["coverage"].astype(np.float32), rtol=0, atol=0)
    np.testing.assert_allclose(out["obs"][-1].cpu().numpy(), ref["obs_last"].astype(np.float32), rtol=0, atol=OBS_TOL)
    acts = np.stack([oracle_mod.rng_actions(1234, 7 + k, E, N, 0, E) for k in range(K)])
    out2 = env2.rollout(K, actions=torch.from_numpy(acts).to(env2.device))
    for k in ("obs", "reward", "done", "connect", "connect_s", "coverage", "assign"):
        assert torch.equal(out[k], out2[k]), k
    s1, s2 = env.get_state(), env2.get_state()
    for k in s1:
        assert torch.equal(s1[k], s2[k]), k


def test_sharded_envs_equal_single_device_concatenation():
    """Env shards (what each rank of an N-GPU job runs) reproduce the unsharded batch exactly."""
    E, N, M, K = 48, 8, 64, 60
    import dcc_hip
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    full = dcc_hip.HipCoverageEnv(E, N, M, poi)
    full.reset()
    ref = full.rollout(K, seed=9, env0=0, env_total=E)
    parts = []
    for r in range(3):
        sh = dcc_hip.HipCoverageEnv(E // 3, N, M, poi)
        sh.reset()
        parts.append(sh.rollout(K, seed=9, env0=r * (E // 3), env_total=E))
    for k in ("obs", "reward", "done", "coverage"):
        assert torch.equal(ref[k], torch.cat([p[k] for p in parts], dim=1)), k


def test_full_size_properties_config2():
    """BASELINE config 2 size (N=8, M=64, E=4096, T=150): size-independent invariants."""
    E, N, M, K = 4096, 8, 64, 150
    import dcc_hip
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi)
    env.reset()
    out = env.rollout(K, seed=3, out=env.alloc_out(K, obs=True, assign=True, reward64=True))
    torch.cuda.synchronize()
    obs = out["obs"]
    D = env.D
    H = 4 + 2 * (N - 1)
    feat = obs[..., H:].reshape(K, E, N, M, 5)
    energy, m_en, dn = feat[..., 2], feat[..., 3], feat[..., 4]
    assert bool((m_en == 5.0).all())
    assert bool(((dn == 1.0) == (energy >= 5.0)).all())          # done <=> energy >= m_energy
    assert bool((energy == energy.round()).all())                 # integer-valued energy
    # every agent of an env sees the same PoI energy / done
    assert bool((energy == energy[:, :, :1]).all()) and bool((dn == dn[:, :, :1]).all())
    # coverage_rate == popcount(done)/M wherever the env did not reset on that step
    live = out["done"] == 0
    cov = dn[:, :, 0].sum(-1) / M
    assert torch.allclose(cov[live], out["coverage"][live], atol=1e-6)
    # energy / done monotone between resets
    e0, e1 = energy[:-1, :, 0], energy[1:, :, 0]
    keep = (out["done"][1:] == 0)[..., None].expand_as(e0)
    assert bool((e1[keep] >= e0[keep]).all())
    # relative positions are consistent: (x_k - x_i) + (x_i - x_k) == 0 up to float32 rounding
    rel01 = obs[:, :, 0, 4:6]      # x_1 - x_0 as seen by agent 0
    rel10 = obs[:, :, 1, 4:6]      # x_0 - x_1 as seen by agent 1
    assert float((rel01 + rel10).abs().max()) < 1e-6
    # speed clamp and reward decomposition R = N*base + 75*#just  (mod 75 structure not testable
    # without base; check the bound instead): reward is finite and resets happened only on done
    vel = obs[..., 0:2]
    assert float(vel.norm(dim=-1).max()) <= 0.5 + 1e-6
    assert bool(torch.isfinite(out["reward64"]).all())
    assert 0 <= int(out["assign"].max()) < N


def test_unaligned_obs_pointer_uses_scalar_store_path(oracle_mod):
    """An obs buffer that is only 4-byte aligned cannot take the float4 path: results must not change."""
    import dcc_hip
    E, N, M = 5, 8, 64
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi)
    orc = oracle_mod.OracleEnv(E, N, M, poi)
    env.reset(); orc.reset()
    backing = torch.zeros(E * N * env.D + 1, dtype=torch.float32, device=env.device)
    obs = backing[1:].view(E, N, env.D)            # data_ptr % 16 == 4
    assert obs.data_ptr() % 16 == 4
    for t in range(6):
        a = oracle_mod.rng_actions(3, t, E, N)
        out = env.alloc_out(obs=False); out["obs"] = obs
        env.step(torch.from_numpy(a).to(env.device), out)
        ref = orc.step(a)
        np.testing.assert_allclose(obs.cpu().numpy(), ref["obs"].astype(np.float32), rtol=0, atol=OBS_TOL)
    assert float(backing[0]) == 0.0                # nothing written in front of the block


def test_odd_row_length_not_multiple_of_four(oracle_mod):
    """N*D % 4 != 0 (N=3, M=7: 3*41 = 123 floats per env): per-env blocks are not 16-byte aligned."""
    import dcc_hip
    E, N, M = 9, 3, 7
    rs = np.random.RandomState(1)
    poi = rs.uniform(-1, 1, (M, 2))
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.3, 0.25, 0.9, 1.0)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.3, 0.25, 0.9, 1.0)
    assert (N * env.D) % 4 != 0
    np.testing.assert_array_equal(env.reset().cpu().numpy(), orc.reset().astype(np.float32))
    for t in range(10):
        a = oracle_mod.rng_actions(9, t, E, N)
        out = env.step(torch.from_numpy(a).to(env.device))
        ref = orc.step(a)
        np.testing.assert_allclose(out["obs"].cpu().numpy(), ref["obs"].astype(np.float32), rtol=0, atol=OBS_TOL)
        assert np.array_equal(out["connect_s"].cpu().numpy(), ref["connect_s"])


def test_single_env_and_null_outputs(oracle_mod):
    import dcc_hip
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:20]
    env = dcc_hip.HipCoverageEnv(1, 4, 20, poi, 0.2, 0.4, 0.9, 0.0)
    orc = oracle_mod.OracleEnv(1, 4, 20, poi, 0.2, 0.4, 0.9, 0.0)
    env.reset(); orc.reset()
    a = oracle_mod.rng_actions(1, 0, 1, 4)
    out = env.step(torch.from_numpy(a).to(env.device), dict(done=torch.zeros(1, dtype=torch.uint8, device=env.device)))
    ref = orc.step(a)
    assert int(out["done"][0]) == int(ref["done"][0])          # only `done` requested: every other output skipped
    np.testing.assert_array_equal(env.get_state()["pos"].cpu().numpy(), orc.get_state()["pos"])


def test_argument_validation_raises():
    import dcc_hip
    poi = np.zeros((4, 2))
    env = dcc_hip.HipCoverageEnv(2, 3, 4, poi)
    dev = env.device
    with pytest.raises(ValueError):
        env.step(torch.zeros(2, 3, 3, device=dev))                      # wrong shape
    with pytest.raises(ValueError):
        env.step(torch.zeros(2, 3, 2, dtype=torch.float16, device=dev))  # wrong dtype
    with pytest.raises(ValueError):
        env.step(torch.zeros(2, 3, 2))                                    # host tensor
    with pytest.raises(ValueError):
        env.step(torch.zeros(2, 3, 2, device=dev), dict(reward=torch.zeros(3, device=dev)))
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.HipCoverageEnv(2, 65, 4, np.zeros((4, 2)))
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.HipCoverageEnv(0, 3, 4, poi)                                # empty batch
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.HipCoverageEnv(2, 3, 1025, np.zeros((1025, 2)))            # above DCC_MAX_POIS
    with pytest.raises(dcc_hip.DccError):
        dcc_hip.HipCoverageEnv(2, 3, 4, poi, comm_r_scale=0.0, comm_force_scale=0.5)
    L = env.lib
    assert L.dcc_env_rollout(env._h, 0, None, 0, 0, 0, 2, None, None) == -1 and b"K must" in L.dcc_last_error()


def test_connectivity_flags_match_graph_connectivity():
    """Property: connect <=> graph(A) connected; connect_ <=> connect and no A_-isolated node (N >= 3)."""
    import dcc_hip
    E, N, M = 256, 7, 8
    rs = np.random.RandomState(3)
    env = dcc_hip.HipCoverageEnv(E, N, M, rs.uniform(-1, 1, (M, 2)), 0.2, 0.25, 0.8, 0.0)
    env.reset()
    pos = rs.uniform(-1, 1, (E, N, 2)) * rs.uniform(0.2, 1.0, (E, 1, 1))
    env.set_state(pos=pos, vel=np.zeros((E, N, 2)))
    out = env.step(torch.zeros(E, N, 2, device=env.device))
    d = np.linalg.norm(pos[:, :, None] - pos[:, None], axis=-1)
    A = (d < 0.5) & ~np.eye(N, dtype=bool); As = (d < 0.8 * 0.5) & ~np.eye(N, dtype=bool)
    conn = np.zeros(E, bool)
    for e in range(E):
        seen = {0}; front = [0]
        while front:
            x = front.pop()
            for y in np.nonzero(A[e, x])[0]:
                if y not in seen:
                    seen.add(y); front.append(y)
        conn[e] = len(seen) == N
    assert np.array_equal(out["connect"].cpu().numpy().astype(bool), conn)
    assert np.array_equal(out["connect_s"].cpu().numpy().astype(bool), conn & As.any(1).all(1))
    assert 0.1 < conn.mean() < 0.9


def test_assignment_index_tie_on_rounded_distance(oracle_mod):
    """np.argmin works on the ROUNDED distances: two agents whose squared distances differ by an ulp or
    two can tie after the square root, and the earlier agent must win.  The kernel compares radicands
    and has an exact slow path for this case; build such near-ties on purpose and compare with the oracle."""
    import dcc_hip
    E, N, M = 64, 4, 3
    poi = np.array([[0.0, 0.0], [0.5, -0.25], [-0.7, 0.3]])
    rs = np.random.RandomState(0)
    pos = np.zeros((E, N, 2))
    n_tie = 0
    for e in range(E):
        dx = rs.uniform(0.3, 0.9)
        ulp = np.spacing(dx * dx)
        dy = np.sqrt(ulp) * rs.choice([0.6, 0.8, 1.0, 1.3, 1.7, 2.5])
        pos[e, 0] = (dx, dy)          # slightly FARTHER from PoI 0, but listed first
        pos[e, 1] = (dx, 0.0)
        pos[e, 2] = (2.0 * dx, 0.1)
        pos[e, 3] = (-1.2 * dx, 0.4)
        d0 = np.sqrt(np.float64(dy) * dy + dx * dx) if False else None
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.05, 0.4, 0.9, 0.0)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.05, 0.4, 0.9, 0.0)
    env.reset(); orc.reset()
    env.set_state(pos=pos, vel=np.zeros((E, N, 2)))
    orc.set_state(pos=pos, vel=np.zeros((E, N, 2)))
    a = np.zeros((E, N, 2), np.float32)
    out = env.step(torch.from_numpy(a).to(env.device))
    ref = orc.step(a)
    got = out["assign"].cpu().numpy().astype(np.int32)
    assert np.array_equal(got, ref["assign"])
    n_tie = int((ref["assign"][:, 0] == 0).sum())
    assert 0 < n_tie < E, "the construction must produce both ties (index 0 wins) and non-ties (index 1 wins)"


def test_long_rollout_stays_in_lockstep_with_oracle(oracle_mod):
    """2,000 fused steps (13 action chunks of 24 steps... and hundreds of hand-offs between the physics and the
    observation wave): done flags, rewards and the last observations still equal the oracle's."""
    import dcc_hip
    E, N, M, K = 66, 8, 64, 2000
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    env.reset(); orc.reset()
    acts = np.stack([oracle_mod.rng_actions(77, k, E, N, 0, E) for k in range(K)])
    out = env.rollout(K, actions=torch.from_numpy(acts).to(env.device))
    ref = orc.rollout_rng(K, 77, 0, 0, E, want_obs_last=True)
    assert np.array_equal(out["done"].cpu().numpy(), ref["done"])
    np.testing.assert_allclose(out["reward"].cpu().numpy(), ref["reward"], rtol=REW_RTOL, atol=1e-5)
    np.testing.assert_allclose(out["obs"][-1].cpu().numpy(), ref["obs_last"].astype(np.float32), rtol=0, atol=OBS_TOL)
    assert int(ref["done"].sum()) > 0          # auto-resets happened along the way
    st, so = env.get_state(), orc.get_state()
    assert np.array_equal(st["pos"].cpu().numpy(), so["pos"]) and np.array_equal(st["done"].cpu().numpy(), so["done"])


@pytest.mark.parametrize("N,M,cfs", [(8, 64, 0.0), (5, 37, 0.5), (16, 256, 0.5), (3, 7, 1.0)])
def test_compact_state_outputs_and_obs_expansion(N, M, cfs, oracle_mod):
    """ABI v2: the per-step compact state (post-reset pos / vel / energy / done) equals the oracle's state, and
    dcc_obs_expand(state) reproduces -- bit for bit -- the observations the same steps wrote."""
    import dcc_hip
    E, K = 21, 30
    rs = np.random.RandomState(N + M)
    poi = rs.uniform(-1, 1, (M, 2))
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.3, 0.95, cfs)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.3, 0.95, cfs)
    env.reset(); orc.reset()
    acts = np.clip(rs.normal(0, 0.8, (K, E, N, 2)), -1, 1).astype(np.float32)
    out = env.alloc_out(K)
    out.update(env.alloc_state_out(K))
    env.rollout(K, actions=torch.from_numpy(acts).to(env.device), out=out)
    for k in range(K):
        orc.step(acts[k], want_obs=False)
        so = orc.get_state()                       # post-reset state, like the compact outputs
        np.testing.assert_allclose(out["state_pos"][k].cpu().numpy(), so["pos"], rtol=0, atol=POS_TOL)
        np.testing.assert_allclose(out["state_vel"][k].cpu().numpy(), so["vel"], rtol=0, atol=POS_TOL)
        assert np.array_equal(out["state_energy"][k].cpu().numpy(), so["energy"].astype(np.float32))
        assert np.array_equal(out["state_done"][k].cpu().numpy(), so["done"])
    flat = lambda t: t.reshape((K * E,) + tuple(t.shape[2:]))
    obs2 = env.expand_obs(flat(out["state_pos"]), flat(out["state_vel"]), flat(out["state_energy"]), flat(out["state_done"]))
    assert torch.equal(obs2.view(K, E, N, env.D), out["obs"])
    # single-step path too
    o1 = env.alloc_out(); o1.update(env.alloc_state_out())
    env.step(torch.from_numpy(acts[0]).to(env.device), o1)
    assert torch.equal(env.expand_obs(o1["state_pos"], o1["state_vel"], o1["state_energy"], o1["state_done"]), o1["obs"])
    with pytest.raises(ValueError):
        env.expand_obs(o1["state_pos"].float(), o1["state_vel"], o1["state_energy"], o1["state_done"])


@pytest.mark.parametrize("N,M", [(8, 64), (4, 20), (1, 9), (5, 37), (16, 256), (33, 700)])
def test_obs_features_match_the_rows(N, M):
    """dcc_obs_features(state): head columns bit-identical to the rows dcc_obs_expand writes, PoI features equal to the
    agent-independent row columns, float64 moments equal to those of the float32 row values."""
    import dcc_hip
    from algos.algo_utils.structured import ObsLayout, features_from_obs
    E, K = 13, 12
    rs = np.random.RandomState(7 * N + M)
    poi = rs.uniform(-1, 1, (M, 2))
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.3, 0.3, 0.95, 0.0)
    env.reset()
    out = env.alloc_state_out(K)
    env.rollout(K, seed=5, out=dict(out, reward=torch.empty(K, E, device=env.device)))
    flat = lambda t: t.reshape((K * E,) + tuple(t.shape[2:]))
    st = [flat(out[k]) for k in ("state_pos", "state_vel", "state_energy", "state_done")]
    rows = env.expand_obs(*st)
    f = env.obs_features(*st)
    ref = features_from_obs(rows, ObsLayout(N, M, poi, env.m_energy))
    assert torch.equal(f["head"], ref["head"]) and torch.equal(f["poi_feat"], ref["poi_feat"])
    np.testing.assert_allclose(f["stats"][..., 0].cpu().numpy(), ref["stats"][..., 0].cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(f["stats"][..., 1].cpu().numpy(), ref["stats"][..., 1].cpu().numpy(), rtol=1e-11)
    # pooled moments of the centralised row (N rows concatenated) == moments computed on the concatenation
    np.testing.assert_allclose(f["cstats"][:, 0].cpu().numpy(), ref["cstats"][:, 0].cpu().numpy(), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(f["cstats"][:, 1].cpu().numpy(), ref["cstats"][:, 1].cpu().numpy(), rtol=1e-10)
    only_c = env.obs_features(*st, out=dict(cstats=torch.empty(K * E, 2, dtype=torch.float64, device=env.device)))
    assert torch.equal(only_c["cstats"], f["cstats"])
    # partial outputs
    only = env.obs_features(*st, out=dict(stats=torch.empty(K * E, N, 2, dtype=torch.float64, device=env.device)))
    assert torch.equal(only["stats"], f["stats"])
    with pytest.raises(ValueError):
        env.obs_features(st[0].float(), *st[1:])


def test_kernel_shape_is_measured_at_create_and_both_shapes_agree(oracle_mod, monkeypatch):
    """dcc_env_create times the role-specialised and the fused kernel shape on this device for batches that are large
    enough (which one streams faster depends on the box) and keeps the faster; DCC_AUTOTUNE=0 keeps the
    built-in default.  Whatever is chosen, the env starts from the reset state and steps like the oracle."""
    import dcc_hip
    for k in ("DCC_NO_ROLES", "DCC_FORCE_ROLES", "DCC_AUTOTUNE"):
        monkeypatch.delenv(k, raising=False)
    E, N, M = 4096, 8, 64
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    big = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    kc = big.kernel_choice()
    assert kc["choice"] in ("roles", "fused") and kc["us_per_step_roles"] > 0 and kc["us_per_step_fused"] > 0
    assert (kc["choice"] == "fused") == (kc["us_per_step_fused"] < 0.94 * kc["us_per_step_roles"])
    st = big.get_state()        # the measurement leaves the state dcc_env_create promises: every env at reset
    assert float(st["pos"].abs().sum()) == 0.0 and float(st["energy"].abs().sum()) == 0.0 and int(st["done"].sum()) == 0
    big.reset()
    out = big.rollout(6, seed=5, step0=0, env0=0, env_total=E)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    orc.reset()
    ref = orc.rollout_rng(6, 5, 0, 0, E, want_obs_last=True)
    assert np.array_equal(out["done"].cpu().numpy(), ref["done"])
    np.testing.assert_allclose(out["reward"].cpu().numpy(), ref["reward"], rtol=REW_RTOL, atol=1e-5)
    assert np.array_equal(out["obs"][-1].cpu().numpy(), ref["obs_last"].astype(np.float32))
    big.close()
    small = dcc_hip.HipCoverageEnv(64, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    assert small.kernel_choice()["choice"] == "default"          # latency-bound sizes are not measured
    small.close()
    monkeypatch.setenv("DCC_AUTOTUNE", "0")
    off = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    assert off.kernel_choice()["choice"] == "default"
    off.close()


@pytest.mark.parametrize("shape", [(8, 64, 0.0, 0.4), (4, 20, 0.0, 0.4), (5, 37, 0.5, 0.2), (16, 256, 0.5, 0.15), (32, 1024, 0.5, 0.1), (3, 130, 0.0, 0.4)],
                         ids=lambda s: "n%dm%d%s" % (s[0], s[1], "_force" if s[2] else ""))
@pytest.mark.parametrize("spec", ["spec", "generic"])
def test_step_features_equals_step_then_features(shape, spec, monkeypatch):
    """dcc_env_step_features (the step of a policy-driven rollout with structured first layers: one launch that steps the envs
    AND derives the policy-input features of the state it leaves from the stepping wave's registers) == dcc_env_step
    followed by dcc_obs_features_x on the emitted state: every per-step output, the emitted state and every feature tensor
    bit for bit, over a trajectory with auto-resets."""
    import dcc_hip
    N, M, cfs, r_comm = shape
    if spec == "generic" and (N, M) not in ((8, 64), (4, 20), (16, 256)):
        pytest.skip("no specialised kernel for this size")
    monkeypatch.setenv("DCC_NO_SPEC", "1" if spec == "generic" else "0")
    E, T = 96, 48
    from envs.hip_vec_env import load_pois
    poi = load_pois(M)
    a = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
    b = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
    a.reset(); b.reset()
    rng = np.random.RandomState(3)
    keys = ("reward", "done", "connect", "connect_s", "coverage", "assign", "state_pos", "state_vel", "state_energy", "state_done")
    mk = lambda env: {**env.alloc_out(obs=False), **env.alloc_state_out()}
    oa, ob = mk(a), mk(b)
    fb = b.alloc_features(keys=("head", "poi_feat", "stats", "cstats", "xa", "xc"))
    resets = 0
    for t in range(T):
        acts = rng.uniform(-1, 1, (E, N, 2)).astype(np.float32)
        acts[: E // 2, 0] = (1.0, 0.0)                   # UAV 0 of half the envs flies straight out of bounds -> auto-reset at step 30
        act = torch.from_numpy(acts).cuda()
        a.step(act, oa)
        fa = a.obs_features(oa["state_pos"], oa["state_vel"], oa["state_energy"], oa["state_done"])
        b.step_features(act, ob, fb)
        for k in keys:
            assert torch.equal(oa[k], ob[k]), (k, t)
        for k in fb:
            assert torch.equal(fa[k], fb[k]), (k, t)
        resets += int(oa["done"].sum())
    assert resets > 0
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    with pytest.raises(ValueError):
        b.step_features(act, {**ob, "obs": torch.empty(E, N, a.D, device="cuda")}, fb)
    a.close(); b.close()


def test_placed_observation_buffer_is_just_a_buffer(oracle_mod):
    """alloc_out(K, placed=n): the observation buffer of a rollout is the best of up to n candidate allocations, timed with
    dcc_env_obs_write_probe (where a buffer lies in HBM decides how fast the env kernels' store pattern streams into it:
    tools/placement_probe.py).  It must be nothing but a buffer: same shape / dtype, the env left in the reset state, and a
    rollout into it equals the oracle's; small buffers are not placed."""
    import dcc_hip
    E, N, M, K = 2048, 8, 64, 24          # 532 MB of rows: above the placing threshold
    poi = np.load(os.path.join(os.path.dirname(__file__), "golden", "pos_pois.npy"))[:M]
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    env.reset()
    env.rollout(3, seed=1, env0=0, env_total=E)                      # leave the reset state ...
    out = env.alloc_out(K, placed=3)
    info = env.placement_info
    assert info is not None and 1 <= info["tried"] <= 3 and len(info["probe_ms"]) == info["tried"] and min(info["probe_ms"]) > 0
    assert info["probe_ms"][info["chosen"]] == min(info["probe_ms"])
    assert tuple(out["obs"].shape) == (K, E, N, env.D) and out["obs"].dtype == torch.float32 and out["obs"].is_contiguous()
    st = env.get_state()                                             # ... the probe put it back there
    assert float(st["pos"].abs().sum()) == 0.0 and float(st["energy"].abs().sum()) == 0.0
    env.rollout(K, seed=9, step0=0, env0=0, env_total=E, out=out)
    orc = oracle_mod.OracleEnv(E, N, M, poi, 0.2, 0.4, 0.95, 0.0)
    orc.reset()
    ref = orc.rollout_rng(K, 9, 0, 0, E, want_obs_last=True)
    assert np.array_equal(out["done"].cpu().numpy(), ref["done"])
    assert np.array_equal(out["obs"][-1].cpu().numpy(), ref["obs_last"].astype(np.float32))
    small = env.alloc_out(2, placed=3)
    assert env.placement_info is None and tuple(small["obs"].shape) == (2, E, N, env.D)
    env.close()


@pytest.mark.parametrize("shape", [(8, 64, 66, 0.0, 0.4), (8, 64, 1500, 0.0, 0.4), (16, 256, 40, 0.5, 0.15), (5, 37, 33, 0.5, 0.2)],
                         ids=lambda s: "n%dm%de%d" % s[:3])
def test_store_pacing_modes_are_bit_identical(shape, monkeypatch):
    """DCC_OBS_DRAIN (KParams::obs_drain) only moves `s_waitcnt vmcnt(0)` around in the row-producing waves -- before every
    staging-window flush (2, the default), at the start of an env-step (0), nowhere (-1): every output of a fused rollout,
    observation rows included, and the state it leaves are the same bits in all three, for the role-specialised (one and two
    envs per workgroup), the split and the fused kernel shape."""
    import dcc_hip
    N, M, E, cfs, r_comm = shape
    from envs.hip_vec_env import load_pois
    poi = load_pois(M)
    monkeypatch.setenv("DCC_AUTOTUNE", "0")
    K, ref, ref_state = 12, None, None
    acts = torch.rand(K, E, N, 2, device="cuda") * 2 - 1
    for mode in ("2", "0", "-1", None):
        if mode is None:
            monkeypatch.delenv("DCC_OBS_DRAIN", raising=False)
        else:
            monkeypatch.setenv("DCC_OBS_DRAIN", mode)
        env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
        env.reset()
        out = env.rollout(K, actions=acts, seed=0, step0=0, env0=0, env_total=E)
        st = env.get_state()
        if ref is None:
            ref, ref_state = {k: v.clone() for k, v in out.items()}, st
        else:
            for k in ref:
                assert torch.equal(out[k], ref[k]), (k, mode)
            for k in st:
                assert torch.equal(st[k], ref_state[k]), (k, mode)
        env.close()


@pytest.mark.parametrize("shape", [(8, 64, 33, 0.0, 0.4), (8, 64, 513, 0.0, 0.4), (8, 64, 600, 0.0, 0.4), (5, 37, 65, 0.5, 0.2), (4, 20, 300, 0.0, 0.4)],
                         ids=lambda s: "n%dm%de%d" % s[:3])
@pytest.mark.parametrize("epw", [1, 2])
def test_two_wave_pairs_per_workgroup_are_bit_identical(shape, epw, monkeypatch):
    """KParams::roles_pairs = 2 (DCC_ROLES_PAIRS; chosen by launch() for batches that would put two or three one-pair workgroups
    on a CU) runs two independent (physics, observation) wave pairs in one 4-wave workgroup, each with its own LDS region and
    envs: same bits as the one-pair form in every output and in the state left behind, with one and two envs per pair, an odd
    number of pairs (the last workgroup's second pair has no env) and compile-time / runtime sizes; and the default policy equals both."""
    import dcc_hip
    N, M, E, cfs, r_comm = shape
    from envs.hip_vec_env import load_pois
    poi = load_pois(M)
    monkeypatch.setenv("DCC_AUTOTUNE", "0")
    monkeypatch.setenv("DCC_ROLES_ENVS", str(epw))
    K, ref, ref_state = 12, None, None
    acts = torch.rand(K, E, N, 2, device="cuda") * 2 - 1
    for pairs, slots in (("1", "2"), ("2", "2"), ("1", "8"), ("2", "4"), (None, None)):   # + the hand-off ring depth (KParams::roles_slots)
        if pairs is None:
            for k in ("DCC_ROLES_PAIRS", "DCC_ROLES_ENVS", "DCC_ROLES_SLOTS"):
                monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv("DCC_ROLES_PAIRS", pairs)
            monkeypatch.setenv("DCC_ROLES_SLOTS", slots)
        env = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.2, r_comm, 0.95, cfs)
        env.reset()
        out = env.rollout(K, actions=acts, seed=0, step0=0, env0=0, env_total=E)
        out2 = env.rollout(K, seed=7, step0=K, env0=0, env_total=E)          # in-kernel action stream, continuing
        st = env.get_state()
        got = {**{k: v.clone() for k, v in out.items()}, **{"rng_" + k: v.clone() for k, v in out2.items()}}
        if ref is None:
            ref, ref_state = got, st
        else:
            for k in ref:
                assert torch.equal(got[k], ref[k]), (k, pairs, slots)
            for k in st:
                assert torch.equal(st[k], ref_state[k]), (k, pairs, slots)
        env.close()
