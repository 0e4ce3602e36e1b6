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
