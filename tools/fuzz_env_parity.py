"""Differential fuzz of the env kernels against the oracle (test infrastructure, like tests/): random sizes, radii, force
scales, env counts, action scales and kernel forms (fused / role-specialised / split, compile-time and runtime sizes, single
steps and fused K-step rollouts) for a wall-clock budget.  Masks, assignment indices, energies and coverage must be bit-equal,
positions / rewards within the tests' tolerances; after every case the compact state is expanded to rows and reduced to the
policy-input features (dcc_obs_expand / dcc_obs_features), which must match the features computed from the rows in float64.  Run on the GPU box: python tools/fuzz_env_parity.py [seconds] [seed] [max N*M]"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-coverage-control_amd"))
sys.path.insert(0, ROOT)
import dcc_hip
from oracle import oracle
from algos.algo_utils.structured import ObsLayout, features_from_obs

POS_TOL, OBS_TOL = 1e-9, 2e-6
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
LIMIT = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
oracle.build()
t0 = time.time()
cases = steps = 0
forms = {}
while time.time() - t0 < budget:
    N = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64]))
    M = int(rs.choice([1, 2, 7, 16, 20, 37, 63, 64, 65, 100, 128, 129, 200, 256, 300, 511, 512, 700, 1024]))
    if N * M > LIMIT:
        continue
    E = int(rs.choice([1, 2, 3, 4, 5, 7, 8, 31, 33, 64, 65])) if N * M < 20000 else int(rs.choice([1, 3, 5]))
    cfs = float(rs.choice([0.0, 0.0, 0.5, 1.0, rs.uniform(0.1, 2.0)]))
    r_comm = float(rs.uniform(0.03, 0.6))
    r_cover = float(rs.uniform(0.05, 0.5))
    crs = float(rs.choice([0.95, 0.9, rs.uniform(0.5, 1.0)]))
    form = {k: str(int(rs.rand() < 0.5)) for k in ("DCC_NO_SPEC", "DCC_NO_ROLES", "DCC_FORCE_ROLES", "DCC_NO_SPLIT", "DCC_FORCE_SPLIT")}
    form["DCC_ROLES_ENVS"] = str(int(rs.choice([1, 2])))        # envs per role-specialised workgroup
    form["DCC_ROLES_SLOTS"] = str(int(rs.choice([2, 4, 8])))    # hand-off slots per env of the role-specialised kernel
    form["DCC_ROLES_PAIRS"] = str(int(rs.choice([1, 2])))       # (physics, observation) wave pairs per role-specialised workgroup
    os.environ.update(form)
    poi = rs.uniform(-1, 1, (M, 2))
    env = dcc_hip.HipCoverageEnv(E, N, M, poi, r_cover, r_comm, crs, cfs)
    orc = oracle.OracleEnv(E, N, M, poi, r_cover, r_comm, crs, cfs)
    assert np.array_equal(env.reset().cpu().numpy(), orc.reset().astype(np.float32))
    scale = rs.uniform(0.2, 3.0, (E, 1, 1))
    bias = rs.uniform(-0.6, 0.6, (E, N, 2))
    T = int(rs.choice([3, 10, 25, 60]))
    K = int(rs.choice([1, 1, 2, 5, 16]))          # K > 1: fused multi-step launches with actions from memory
    tag = (N, M, E, round(cfs, 3), round(r_comm, 3), round(r_cover, 3), round(crs, 3), K, tuple(sorted(form.items())))
    try:
        t = 0
        while t < T:
            k = min(K, T - t)
            a = np.clip(rs.uniform(-1, 1, (k, E, N, 2)) * scale + bias, -1, 1).astype(np.float32)
            if k == 1:
                outs = [env.step(torch.from_numpy(a[0]).to(env.device), env.alloc_out(reward64=True))]
            else:
                o = env.alloc_out(k, reward64=True)
                env.rollout(k, actions=torch.from_numpy(a).to(env.device), out=o)
                outs = [{kk: v[i] for kk, v in o.items()} for i in range(k)]
            for i in range(k):
                ref = orc.step(a[i])
                out = outs[i]
                for kk in ("done", "connect", "connect_s"):
                    assert np.array_equal(out[kk].cpu().numpy(), ref[kk]), (kk, t + i)
                assert np.array_equal(out["assign"].cpu().numpy().astype(np.int32), ref["assign"]), ("assign", t + i)
                np.testing.assert_allclose(out["reward64"].cpu().numpy(), ref["reward"], rtol=1e-11, atol=1e-8)
                np.testing.assert_allclose(out["coverage"].cpu().numpy(), ref["coverage"].astype(np.float32), rtol=0, atol=0)
                np.testing.assert_allclose(out["obs"].cpu().numpy(), ref["obs"].astype(np.float32), rtol=0, atol=OBS_TOL)
            st, so = env.get_state(), orc.get_state()
            np.testing.assert_allclose(st["pos"].cpu().numpy(), so["pos"], rtol=0, atol=POS_TOL)
            np.testing.assert_allclose(st["vel"].cpu().numpy(), so["vel"], rtol=0, atol=POS_TOL)
            assert np.array_equal(st["energy"].cpu().numpy(), so["energy"].astype(np.float32)), "energy"
            assert np.array_equal(st["done"].cpu().numpy(), so["done"]), "done state"
            t += k
            steps += k * E
        # compact state -> observation rows and -> policy-input features (float64 moments of the float32 row values)
        stt = [st[kk].contiguous() for kk in ("pos", "vel", "energy", "done")]
        rows = env.expand_obs(*stt)
        f = env.obs_features(*stt)
        ref = features_from_obs(rows, ObsLayout(N, M, poi, env.m_energy))
        assert torch.equal(f["head"], ref["head"]) and torch.equal(f["poi_feat"], ref["poi_feat"]), "features head / poi_feat"
        np.testing.assert_allclose(f["stats"][..., 0].cpu().numpy(), ref["stats"][..., 0].cpu().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(f["stats"][..., 1].cpu().numpy(), ref["stats"][..., 1].cpu().numpy(), rtol=1e-11)
        np.testing.assert_allclose(f["cstats"][:, 0].cpu().numpy(), ref["cstats"][:, 0].cpu().numpy(), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(f["cstats"][:, 1].cpu().numpy(), ref["cstats"][:, 1].cpu().numpy(), rtol=1e-10)
        # the rollout step of round 3: ONE launch = env step + features of the state it leaves (dcc_env_step_features) must equal
        # dcc_env_step followed by dcc_obs_features_x -- every output, the emitted state and every feature tensor bit for bit --
        # and keep tracking the oracle from the state the case ended in
        env2 = dcc_hip.HipCoverageEnv(E, N, M, poi, r_cover, r_comm, crs, cfs)
        env2.set_state(**{kk: v.clone() for kk, v in st.items()})
        oa = {**env.alloc_out(obs=False), **env.alloc_state_out()}
        ob = {**env2.alloc_out(obs=False), **env2.alloc_state_out()}
        fb = env2.alloc_features(keys=("head", "poi_feat", "stats", "cstats", "xa", "xc"))
        for t2 in range(6):
            a = np.clip(rs.uniform(-1, 1, (E, N, 2)) * scale + bias, -1, 1).astype(np.float32)
            at = torch.from_numpy(a).to(env.device)
            env.step(at, oa)
            fa = env.obs_features(oa["state_pos"], oa["state_vel"], oa["state_energy"], oa["state_done"])
            env2.step_features(at, ob, fb)
            ref = orc.step(a, want_obs=False)
            for kk in oa:
                assert torch.equal(oa[kk], ob[kk]), ("step_features output", kk, t2)
            for kk in fb:
                assert torch.equal(fa[kk], fb[kk]), ("step_features feature", kk, t2)
            for kk in ("done", "connect", "connect_s"):
                assert np.array_equal(ob[kk].cpu().numpy(), ref[kk]), ("step_features vs oracle", kk, t2)
            assert np.array_equal(ob["assign"].cpu().numpy().astype(np.int32), ref["assign"]), ("step_features assign", t2)
            steps += E
        env2.close()
    except Exception as e:  # noqa: BLE001
        print("MISMATCH in case", tag, "->", type(e).__name__, str(e)[:400], flush=True)
        sys.exit(1)
    env.close(); orc.close()
    cases += 1
    forms[tuple(sorted(form.items()))] = forms.get(tuple(sorted(form.items())), 0) + 1
print("fuzz: %d random cases (%d env-steps, %d kernel-form combinations) in %.0f s: 0 mismatches against the oracle"
      % (cases, steps, len(forms), time.time() - t0))
