"""The `_cpu` twins of include/dcc_env.h (oracle/dcc_env_cpu.c): the CPU restatement behind the SAME
C-ABI -- the product's own ctypes struct layouts (dcc_hip.EnvCfg / EnvOut), host pointers -- reproduces the reference's
golden vectors, and (-m gpu) the HIP library driven through the identical structs gives the same answers.  Test
infrastructure: the twins live in oracle/libdcc_oracle.so, never in the product library."""
import os

import numpy as np
import pytest

from conftest import golden_env_files, load_case


@pytest.mark.parametrize("path", golden_env_files(), ids=lambda p: os.path.basename(p)[4:-4])
def test_cpu_twin_reproduces_reference_golden(path, oracle_mod):
    z, c = load_case(path)
    env = oracle_mod.CpuTwinEnv(c["E"], c["N"], c["M"], z["poi"], c["r_cover"], c["r_comm"], c["comm_r_scale"], c["comm_force_scale"])
    assert np.array_equal(env.reset()[0], z["reset_obs"])
    out = env.alloc_out(state=True)
    obs_steps = list(z["obs_steps"])
    for t in range(c["T"]):
        a = z["actions"][t]
        env.step(a.astype(np.float32) if c["act_f32"] else a.astype(np.float64), out)
        assert np.array_equal(out["done"], z["done"][t]) and np.array_equal(out["connect"], z["connect"][t])
        assert np.array_equal(out["connect_s"], z["connect_s"][t]) and np.array_equal(out["assign"], z["assign"][t])
        assert np.array_equal(out["reward64"], z["reward"][t])
        assert np.array_equal(out["coverage"], z["coverage"][t].astype(np.float32))
        live = out["done"] == 0
        assert np.array_equal(out["state_pos"][live], z["pos_t"][t][live])
        assert np.array_equal(out["state_energy"][live], z["energy_t"][t][live].astype(np.float32))
        assert not out["state_pos"][~live].any()                      # finished envs restarted from the origin
        if t in obs_steps:
            assert np.array_equal(out["obs"], z["obs"][obs_steps.index(t)])
    env.close()


def test_cpu_twin_validates_like_the_hip_library(oracle_mod):
    import ctypes
    import dcc_hip
    L = oracle_mod.lib()
    oracle_mod.CpuTwinEnv(1, 2, 3, np.zeros((3, 2))).close()            # declares the prototypes
    cfg = dcc_hip.EnvCfg()
    dcc_hip.load_library().dcc_env_cfg_default(ctypes.byref(cfg))
    h = ctypes.c_void_p()
    cfg.n_envs, cfg.n_agents, cfg.n_pois = 1, 65, 4
    poi = np.zeros((4, 2))
    cfg.poi_xy = poi.ctypes.data_as(ctypes.c_void_p)
    assert L.dcc_env_create_cpu(ctypes.byref(cfg), ctypes.byref(h)) == -1 and b"bad sizes" in L.dcc_last_error_cpu()
    cfg.n_agents, cfg.bound_hard = 4, 2.0
    assert L.dcc_env_create_cpu(ctypes.byref(cfg), ctypes.byref(h)) == -4
    # in-kernel action stream through the twin == the oracle's generator
    env = oracle_mod.CpuTwinEnv(5, 3, 7, np.random.RandomState(0).uniform(-1, 1, (7, 2)), 0.3, 0.3, 0.95, 1.0)
    ref = oracle_mod.OracleEnv(5, 3, 7, env._poi, 0.3, 0.3, 0.95, 1.0)
    env.reset(); ref.reset()
    out = env.rollout(20, seed=9, step0=3, env0=2, env_total=11)
    r = ref.rollout_rng(20, 9, step0=3, env0=2, env_total=11, want_obs_last=True)
    assert np.array_equal(out["reward64"], r["reward"]) and np.array_equal(out["done"], r["done"])
    assert np.array_equal(out["obs"][-1], r["obs_last"].astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,cfs", [(8, 64, 0.0), (5, 37, 0.5)])
def test_one_binding_two_libraries(N, M, cfs, oracle_mod):
    """The same call sequence -- create, reset, K fused steps with the in-kernel action stream, get_state -- against
    libdcc_hip.so (device pointers) and against the `_cpu` twins (host pointers) through the same structs."""
    import torch
    import dcc_hip
    E, K = 33, 40
    poi = np.random.RandomState(N).uniform(-1, 1, (M, 2))
    gpu = dcc_hip.HipCoverageEnv(E, N, M, poi, 0.25, 0.3, 0.95, cfs)
    cpu = oracle_mod.CpuTwinEnv(E, N, M, poi, 0.25, 0.3, 0.95, cfs)
    assert gpu.D == cpu.D and np.array_equal(gpu.reset().cpu().numpy(), cpu.reset())
    og = gpu.alloc_out(K, reward64=True); og.update(gpu.alloc_state_out(K))
    gpu.rollout(K, seed=4, step0=0, env0=0, env_total=E, out=og)
    oc = cpu.rollout(K, seed=4, out=cpu.alloc_out(K, state=True))
    for k in ("done", "connect", "connect_s", "assign", "coverage", "state_energy", "state_done"):
        assert np.array_equal(og[k].cpu().numpy(), oc[k]), k
    np.testing.assert_allclose(og["reward64"].cpu().numpy(), oc["reward64"], rtol=1e-12 if cfs == 0 else 1e-7, atol=1e-9)
    np.testing.assert_allclose(og["state_pos"].cpu().numpy(), oc["state_pos"], rtol=0, atol=0 if cfs == 0 else 1e-9)
    np.testing.assert_allclose(og["obs"].cpu().numpy(), oc["obs"], rtol=0, atol=0 if cfs == 0 else 1e-5)
    sg, sc = gpu.get_state(), cpu.get_state()
    assert np.array_equal(sg["energy"].cpu().numpy(), sc["energy"]) and np.array_equal(sg["done"].cpu().numpy(), sc["done"])
    gpu.close(); cpu.close()


@pytest.mark.parametrize("path", [p for p in golden_env_files() if any(t in p for t in ("n8m64", "n4m20", "n5m37", "c4"))] or golden_env_files()[:3],
                         ids=lambda p: os.path.basename(p)[4:-4])
def test_obs_expand_cpu_twin_rebuilds_the_reference_rows(path, oracle_mod):
    """dcc_obs_expand_cpu (Scenario.observation from a state given as arrays, coverage.py:99-110): the rows the reference returned at the
    golden's sampled steps are rebuilt bit for bit from the state the twin emitted for those steps."""
    z, c = load_case(path)
    env = oracle_mod.CpuTwinEnv(c["E"], c["N"], c["M"], z["poi"], c["r_cover"], c["r_comm"], c["comm_r_scale"], c["comm_force_scale"])
    env.reset()
    out = env.alloc_out(state=True)
    obs_steps = list(z["obs_steps"])
    checked = 0
    for t in range(c["T"]):
        a = z["actions"][t]
        env.step(a.astype(np.float32) if c["act_f32"] else a.astype(np.float64), out)
        if t in obs_steps:
            rows = env.expand_obs(out["state_pos"], out["state_vel"], out["state_energy"], out["state_done"])
            assert np.array_equal(rows, z["obs"][obs_steps.index(t)].astype(np.float32)) and np.array_equal(rows, out["obs"])
            checked += 1
    assert checked == len(obs_steps) and checked > 0
    env.close()


@pytest.mark.gpu
def test_obs_expand_one_binding_two_libraries(oracle_mod):
    """dcc_obs_expand (device) and dcc_obs_expand_cpu (host) on the same random states: identical rows."""
    import torch
    import dcc_hip
    N, M, n = 8, 64, 300
    rs = np.random.RandomState(3)
    poi = rs.uniform(-1, 1, (M, 2))
    pos, vel = rs.uniform(-1.4, 1.4, (n, N, 2)), rs.uniform(-0.5, 0.5, (n, N, 2))
    en = rs.randint(0, 9, (n, M)).astype(np.float32); dn = (en >= 5).astype(np.uint8)
    cpu = oracle_mod.CpuTwinEnv(4, N, M, poi)
    gpu = dcc_hip.HipCoverageEnv(4, N, M, poi)
    t = lambda a: torch.from_numpy(a).to(gpu.device)
    rows_g = gpu.expand_obs(t(pos), t(vel), t(en), t(dn)).cpu().numpy()
    assert np.array_equal(rows_g, cpu.expand_obs(pos, vel, en, dn))
    gpu.close(); cpu.close()


# ---- include/dcc_gae.h: dcc_gae_compute_cpu (oracle/dcc_gae_cpu.c) -------------------------------------------------------------

def _gae_case(T, C, use_vn, seed):
    rs = np.random.RandomState(seed)
    rew = rs.normal(-40, 30, (T, C)).astype(np.float32)
    vp = rs.normal(0, 1, (T + 1, C)).astype(np.float32)
    mk = (rs.uniform(0, 1, (T + 1, C)) > 0.03).astype(np.float32)
    dn = np.array([-250.0, 97.5], np.float32) if use_vn else None
    return rew, vp, mk, dn


def test_gae_cpu_twin_reproduces_reference_golden(oracle_mod):
    """The C twin of dcc_gae_compute on the reference's own buffer (tests/golden/mappo_small.npz: SharedReplayBuffer.compute_returns
    with ValueNorm, episode ends inside the rollout): returns and raw advantages bit for bit."""
    from conftest import GOLDEN
    from oracle import mappo_oracle as mo
    Z = np.load(os.path.join(GOLDEN, "mappo_small.npz"))
    T, E, N = 16, 3, 4
    mean, std = mo.valuenorm_mean_std(Z["vn0_mean"][0], Z["vn0_mean_sq"][0], Z["vn0_debias"])
    c = lambda a, rows: np.ascontiguousarray(a, np.float32).reshape(rows, E * N)
    ret = np.zeros((T + 1, E * N), np.float32); adv = np.zeros((T, E * N), np.float32)
    oracle_mod.gae_compute_cpu(c(Z["buf_rewards"], T), c(Z["buf_value_preds_after"], T + 1), c(Z["buf_masks"], T + 1),
                               np.array([mean, std], np.float32), 0.99, 0.95, ret, adv)
    np.testing.assert_array_equal(ret.reshape(T + 1, E, N, 1)[:-1], Z["returns"][:-1])
    np.testing.assert_array_equal(adv.reshape(T, E, N, 1), Z["adv_raw"])
    assert not ret[-1].any()                                          # row T untouched, as in the reference


@pytest.mark.parametrize("T,C,use_vn", [(150, 513, True), (150, 100, False), (7, 33, True), (1, 1, True), (33, 7, False)])
def test_gae_cpu_twin_equals_the_numpy_restatement(T, C, use_vn, oracle_mod):
    from oracle import mappo_oracle as mo
    rew, vp, mk, dn = _gae_case(T, C, use_vn, T * 7 + C)
    ref, _ = mo.compute_returns_gae(rew, vp, mk, vp[-1], 0.99, 0.95, *((dn[0], dn[1]) if use_vn else (None, None)))
    ret = np.zeros((T + 1, C), np.float32); adv = np.zeros((T, C), np.float32)
    oracle_mod.gae_compute_cpu(rew, vp, mk, dn, 0.99, 0.95, ret, adv)
    np.testing.assert_array_equal(ret[:-1], ref[:-1])
    np.testing.assert_array_equal(adv, ref[:-1] - ((vp[:-1] * dn[1] + dn[0]) if use_vn else vp[:-1]))
    assert oracle_mod.lib().dcc_gae_compute_cpu(None, None, None, None, 0.99, 0.95, None, None, T, C, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("T,C,use_vn", [(150, 4096 * 8, True), (150, 1000, False), (17, 65, True), (400, 129, True)])
def test_gae_one_binding_two_libraries(T, C, use_vn, oracle_mod):
    """dcc_gae_compute (device pointers) and dcc_gae_compute_cpu (host pointers), same arguments: identical bits."""
    import torch
    import dcc_hip
    rew, vp, mk, dn = _gae_case(T, C, use_vn, T + C)
    ret_c = np.zeros((T + 1, C), np.float32); adv_c = np.zeros((T, C), np.float32)
    oracle_mod.gae_compute_cpu(rew, vp, mk, dn, 0.99, 0.95, ret_c, adv_c)
    dev = torch.device("cuda")
    ret_g = torch.zeros(T + 1, C, device=dev); adv_g = torch.zeros(T, C, device=dev)
    dcc_hip.gae_compute(torch.from_numpy(rew).to(dev), torch.from_numpy(vp).to(dev), torch.from_numpy(mk).to(dev),
                        torch.from_numpy(dn).to(dev) if use_vn else None, 0.99, 0.95, ret_g, adv_g)
    np.testing.assert_array_equal(ret_g.cpu().numpy(), ret_c)
    np.testing.assert_array_equal(adv_g.cpu().numpy(), adv_c)


# ---- include/dcc_env.h: dcc_obs_features / dcc_obs_features_x (oracle/dcc_env_cpu.c) --------------------------------------------
def _states(N, M, n, seed):
    rs = np.random.RandomState(seed)
    return (rs.uniform(-1.2, 1.2, (n, N, 2)), rs.uniform(-0.5, 0.5, (n, N, 2)), rs.randint(0, 7, (n, M)).astype(np.float32),
            (rs.uniform(size=(n, M)) < 0.3).astype(np.uint8))


@pytest.mark.parametrize("N,M", [(4, 20), (8, 64), (5, 37), (1, 9)])
def test_features_cpu_twin_is_the_reduction_of_the_rows(N, M, oracle_mod):
    """dcc_obs_features_x_cpu(state) == the same quantities derived from the rows dcc_obs_expand_cpu builds of that state (the
    package's own row-side formulation, algos/algo_utils/structured.features_from_obs / env_gemm_inputs): head and PoI columns
    bit-equal, float64 row moments, the GEMM input matrices with their ones column and zero padding."""
    import torch
    from algos.algo_utils.structured import ObsLayout, env_gemm_inputs, features_from_obs
    n = 11
    poi = np.random.RandomState(M).uniform(-1, 1, (M, 2))
    env = oracle_mod.CpuTwinEnv(3, N, M, poi, 0.25, 0.3, 0.95, 0.0)
    st = _states(N, M, n, 3 * N + M)
    f = env.obs_features(*st)
    rows = torch.from_numpy(env.expand_obs(*st))
    ref = features_from_obs(rows, ObsLayout(N, M, poi, 5.0))
    assert np.array_equal(f["head"], ref["head"].numpy()) and np.array_equal(f["poi_feat"], ref["poi_feat"].numpy())
    np.testing.assert_allclose(f["stats"], ref["stats"].numpy(), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(f["cstats"], ref["cstats"].numpy(), rtol=1e-11, atol=1e-13)
    assert np.array_equal(f["xa"], env_gemm_inputs(ref, False).numpy()) and np.array_equal(f["xc"], env_gemm_inputs(ref, True).numpy())
    only = env.obs_features(*st, out=dict(cstats=np.empty((n, 2))))              # any output may be NULL
    assert set(only) == {"cstats"} and np.array_equal(only["cstats"], f["cstats"])
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(8, 64), (5, 37), (16, 256)])
def test_features_one_binding_two_libraries(N, M, oracle_mod):
    """dcc_obs_features_x (device pointers) and dcc_obs_features_x_cpu (host pointers) on the same states: float32 outputs bit-equal,
    float64 moments to 1e-12 (the kernel sums across lanes, the twin along the row)."""
    import torch
    import dcc_hip
    n = 29
    poi = np.random.RandomState(N).uniform(-1, 1, (M, 2))
    gpu = dcc_hip.HipCoverageEnv(4, N, M, poi, 0.25, 0.3, 0.95, 0.0)
    cpu = oracle_mod.CpuTwinEnv(4, N, M, poi, 0.25, 0.3, 0.95, 0.0)
    st = _states(N, M, n, N + M)
    fg = gpu.obs_features(*[torch.from_numpy(a).to(gpu.device) for a in st])
    fc = cpu.obs_features(*st)
    for k in ("head", "poi_feat", "xa", "xc"):
        assert np.array_equal(fg[k].cpu().numpy(), fc[k]), k
    np.testing.assert_allclose(fg["stats"].cpu().numpy(), fc["stats"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(fg["cstats"].cpu().numpy(), fc["cstats"], rtol=1e-10, atol=1e-13)
    gpu.close(); cpu.close()
