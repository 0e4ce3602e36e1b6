"""Generate tests/golden/env_*.npz by running the REFERENCE env (imported read-only from
/root/reference through tools/ref_harness.py) on seeded / scripted action streams.

Container-only: needs /root/reference.  The outputs are data (inputs + expected outputs), committed
as fixtures; the reference itself never travels.  Re-run:  python tools/gen_golden.py

Per case the file holds
  cfg        : N, M, E, T, r_cover, r_comm, comm_r_scale, comm_force_scale, act_f32
  poi        : [M,2] f64 (pos_pois.npy rows, + synthetic rows for M > 1000, see EXTRA_POI_SEED)
  actions    : [T,E,N,2] f32|f64 -- exactly what was passed (a copy) to MultiAgentEnv.step
  pos_t,vel_t: [T,E,N,2] f64 state after world.step, BEFORE the vec-env auto-reset
  energy_t   : [T,E,M] u8 (integer valued), done_t [T,E,M] u8
  reward     : [T,E] f64 (the shared reward every agent receives), done [T,E] u8 (np.all(done_n))
  connect, connect_s : [T,E] u8 (world.connect / world.connect_ after the step)
  coverage   : [T,E] f64 (info["coverage_rate"])
  assign     : [T,E,M] u8 argmin_i ||x_i - p_j|| with np.argmin on np.linalg.norm (terminal pos)
  obs_steps  : [S] int, obs [S,E,N,D] f32 = the vec-env obs (post auto-reset) cast like
               SharedReplayBuffer stores it; obs_sum [T,E] f64 = sum of that f32 obs in f64
  reset_obs  : [N,D] f32
The vec-env semantics (auto-reset when np.all(done), returned obs is the reset obs) follow
envs/wrappers.py:226-232 and are applied here by hand so that the terminal state can be recorded.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import make_reference_env  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
EXTRA_POI_SEED = 2024  # synthetic PoIs appended when M > 1000 (pos_pois.npy has 1000 rows)


def policy_actions(kind, rs, world, N, t):
    """Action generators.  They read the reference world only to *choose* actions; the chosen
    actions are stored in the fixture and replayed verbatim by the tests."""
    if kind == "uniform":
        return rs.uniform(-1, 1, (N, 2))
    if kind == "zero":
        return np.zeros((N, 2))
    if kind == "east":  # constant (+1, 0): x reaches 1.5000000000000002 at step 30 -> done
        a = np.zeros((N, 2)); a[:, 0] = 1.0
        return a
    if kind == "spread":  # agents fly apart radially, then jitter: breaks connectivity, goes OOB
        ang = 2 * np.pi * np.arange(N) / N + 0.1
        a = np.stack([np.cos(ang), np.sin(ang)], 1) * (0.9 if t < 40 else 0.2)
        return a + rs.uniform(-0.3, 0.3, (N, 2))
    if kind == "seek":  # each agent heads for the nearest not-done PoI (covers PoIs -> done bonus)
        a = np.zeros((N, 2))
        undone = [l for l in world.landmarks if not l.done]
        for i, ag in enumerate(world.agents):
            if not undone:
                break
            # agents with the same target would stack: offset choice by agent index
            ds = np.array([np.linalg.norm(l.state.p_pos - ag.state.p_pos) for l in undone])
            order = np.argsort(ds)
            tgt = undone[order[min(i, len(order) - 1) if t < 5 else 0]]
            a[i] = np.clip(4.0 * (tgt.state.p_pos - ag.state.p_pos) - 1.0 * ag.state.p_vel, -1, 1)
        return a + rs.uniform(-0.05, 0.05, (N, 2))
    if kind == "wander":  # slow random walk with momentum: stays in bounds, gets partly connected graphs
        return np.clip(rs.normal(0, 0.6, (N, 2)), -1, 1)
    raise ValueError(kind)


def run_case(name, N, M, r_cover, r_comm, crs, cfs, kinds, T, act_f32, seed, n_obs_steps=6):
    extra = None
    if M > 1000:
        extra = np.random.RandomState(EXTRA_POI_SEED).uniform(-1, 1, (M - 1000, 2))
    E = len(kinds)
    envs = [make_reference_env(N, M, r_cover, r_comm, crs, cfs, extra) for _ in range(E)]
    D = 4 + 2 * (N - 1) + 5 * M
    adt = np.float32 if act_f32 else np.float64
    rss = [np.random.RandomState(seed + 17 * e) for e in range(E)]
    poi = np.array(envs[0][2].pos_pois[:M], np.float64)
    reset_obs = np.array(envs[0][0].reset(), np.float64)
    for env, _, _ in envs:
        env.reset()
    obs_steps = np.unique(np.linspace(0, T - 1, n_obs_steps).astype(int))
    rec = dict(actions=np.zeros((T, E, N, 2), adt), pos_t=np.zeros((T, E, N, 2)), vel_t=np.zeros((T, E, N, 2)),
               energy_t=np.zeros((T, E, M), np.uint8), done_t=np.zeros((T, E, M), np.uint8),
               reward=np.zeros((T, E)), done=np.zeros((T, E), np.uint8), connect=np.zeros((T, E), np.uint8),
               connect_s=np.zeros((T, E), np.uint8), coverage=np.zeros((T, E)), assign=np.zeros((T, E, M), np.uint8),
               obs=np.zeros((len(obs_steps), E, N, D), np.float32), obs_sum=np.zeros((T, E)))
    for t in range(T):
        for e, (env, world, sc) in enumerate(envs):
            a = policy_actions(kinds[e], rss[e], world, N, t).astype(adt)
            rec["actions"][t, e] = a
            ob, rew, dn, info = env.step(a.copy())  # the reference scales its argument in place (EN:190)
            assert all(r == rew[0] for r in rew)
            pos = np.array([ag.state.p_pos for ag in world.agents])
            rec["pos_t"][t, e] = pos
            rec["vel_t"][t, e] = np.array([ag.state.p_vel for ag in world.agents])
            en = np.array([l.energy for l in world.landmarks])
            assert np.all(en == np.round(en)) and en.max() < 256
            rec["energy_t"][t, e] = en.astype(np.uint8)
            rec["done_t"][t, e] = np.array([l.done for l in world.landmarks], np.uint8)
            rec["reward"][t, e] = rew[0]
            rec["done"][t, e] = np.all(dn)
            assert np.all(dn) == np.any(dn)
            rec["connect"][t, e] = world.connect
            rec["connect_s"][t, e] = world.connect_
            rec["coverage"][t, e] = world.coverage_rate
            dist = np.array([[np.linalg.norm(ag.state.p_pos - l.state.p_pos) for ag in world.agents]
                             for l in world.landmarks])
            rec["assign"][t, e] = np.argmin(dist, 1)
            if np.all(dn):
                ob = env.reset()  # WR:229-232
            ob32 = np.array(ob, np.float64).astype(np.float32)
            rec["obs_sum"][t, e] = ob32.astype(np.float64).sum()
            w = np.where(obs_steps == t)[0]
            if len(w):
                rec["obs"][w[0], e] = ob32
    cfg = dict(N=N, M=M, E=E, T=T, r_cover=r_cover, r_comm=r_comm, comm_r_scale=crs, comm_force_scale=cfs,
               act_f32=int(act_f32))
    path = os.path.join(OUT, "env_%s.npz" % name)
    np.savez_compressed(path, poi=poi, obs_steps=obs_steps, reset_obs=reset_obs.astype(np.float32),
                        kinds=np.array(kinds), **{"cfg_" + k: np.array(v) for k, v in cfg.items()}, **rec)
    n_reset = int(rec["done"].sum())
    print("%-22s N=%d M=%d E=%d T=%d cfs=%.2f f32=%d  resets=%d  connect=%.2f connect_=%.2f maxcov=%.2f  %.0f KB" % (
        name, N, M, E, T, cfs, act_f32, n_reset, rec["connect"].mean(), rec["connect_s"].mean(),
        rec["coverage"].max(), os.path.getsize(path) / 1024))


def main():
    os.makedirs(OUT, exist_ok=True)
    K5 = ["uniform", "zero", "east", "spread", "seek", "wander"]
    # shipped world: comm_r_scale 0.9 (CoverageWorld default, coverage.py never forwards 0.95), force off
    run_case("n4m20_shipped", 4, 20, 0.2, 0.4, 0.9, 0.0, K5, 150, True, 100)
    run_case("n4m20_shipped_f64", 4, 20, 0.2, 0.4, 0.9, 0.0, K5, 80, False, 101)
    # BASELINE config 1 sizes (dcc.yaml radii, comm_r_scale honoured)
    run_case("n4m16_c1", 4, 16, 0.2, 0.4, 0.95, 0.0, K5, 150, True, 102)
    # config 2/3 sizes
    run_case("n8m64_c2", 8, 64, 0.2, 0.4, 0.95, 0.0, K5, 150, True, 103)
    run_case("n8m64_force", 8, 64, 0.2, 0.2, 0.95, 0.5, K5, 120, True, 104)
    run_case("n8m64_force_f64", 8, 64, 0.2, 0.2, 0.9, 0.5, ["uniform", "spread", "wander"], 60, False, 105)
    # odd sizes: N not a power of two, M not a multiple of 64 / 4
    run_case("n5m37_force", 5, 37, 0.25, 0.3, 0.9, 0.5, K5, 100, True, 106)
    run_case("n3m7_force", 3, 7, 0.3, 0.25, 0.9, 1.0, K5, 100, True, 107)
    run_case("n2m5_force", 2, 5, 0.3, 0.25, 0.9, 1.0, ["uniform", "spread", "seek"], 60, True, 108)
    run_case("n1m9", 1, 9, 0.3, 0.25, 0.9, 0.5, ["uniform", "seek", "east"], 60, True, 109)
    # config 4 / 5 sizes (connectivity force on)
    run_case("n16m256_c4", 16, 256, 0.2, 0.15, 0.95, 0.5, ["uniform", "spread", "seek", "wander"], 48, True, 110)
    run_case("n32m1024_c5", 32, 1024, 0.2, 0.1, 0.95, 0.5, ["uniform", "wander"], 24, True, 111, n_obs_steps=2)


if __name__ == "__main__":
    main()
