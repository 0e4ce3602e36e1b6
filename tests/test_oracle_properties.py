"""Property tests (hypothesis) on the pinned CPU oracle: the invariants the HIP kernel's closed forms rely on."""
import numpy as np
from hypothesis import given, settings, strategies as st


def _bfs_connected(A):
    n = len(A)
    seen = {0}; front = [0]
    while front:
        x = front.pop()
        for y in np.nonzero(A[x])[0]:
            if y not in seen:
                seen.add(int(y)); front.append(int(y))
    return len(seen) == n


@settings(max_examples=60, deadline=None)
@given(N=st.integers(1, 9), M=st.integers(1, 40), seed=st.integers(0, 10 ** 6), spread=st.floats(0.05, 1.2),
       r_comm=st.floats(0.05, 0.6), crs=st.floats(0.5, 1.0))
def test_connectivity_flags_closed_form(oracle_mod, N, M, seed, spread, r_comm, crs):
    """The reference sums matrix powers (incl. its line-90 quirk); the kernel uses: connect <=> graph connected,
    connect_ <=> N==1, or N>=3 and connected and no A_-isolated node (never for N==2)."""
    rs = np.random.RandomState(seed)
    E = 6
    poi = rs.uniform(-1, 1, (M, 2))
    o = oracle_mod.OracleEnv(E, N, M, poi, 0.2, r_comm, crs, 0.0)
    o.reset()
    pos = rs.uniform(-1, 1, (E, N, 2)) * spread
    o.set_state(pos=pos, vel=np.zeros((E, N, 2)))
    out = o.step(np.zeros((E, N, 2), np.float32))
    for e in range(E):
        d = np.array([[np.linalg.norm(pos[e, a] - pos[e, b]) for b in range(N)] for a in range(N)])
        A = (d < r_comm + r_comm) & ~np.eye(N, dtype=bool)
        As = A & (d < crs * (r_comm + r_comm))
        conn = _bfs_connected(A)
        assert bool(out["connect"][e]) == conn
        exp_s = True if N == 1 else False if N == 2 else (conn and bool(As.any(1).all()))
        assert bool(out["connect_s"][e]) == exp_s


@settings(max_examples=25, deadline=None)
@given(N=st.integers(1, 8), M=st.integers(1, 70), seed=st.integers(0, 10 ** 6), cfs=st.sampled_from([0.0, 0.5]))
def test_energy_done_coverage_reward_invariants(oracle_mod, N, M, seed, cfs):
    rs = np.random.RandomState(seed)
    E, T = 3, 40
    poi = rs.uniform(-1, 1, (M, 2))
    o = oracle_mod.OracleEnv(E, N, M, poi, 0.25, 0.3, 0.9, cfs)
    o.reset()
    prev_e = np.zeros((E, M)); prev_d = np.zeros((E, M), np.uint8)
    for t in range(T):
        a = np.clip(rs.normal(0, 0.7, (E, N, 2)), -1, 1).astype(np.float32)
        out = o.step(a)
        en, dn = out["energy_t"], out["done_t"]
        assert np.all(en == np.round(en)) and np.all(en >= prev_e)                 # integer valued, monotone
        assert np.all(dn >= prev_d) and np.array_equal(dn == 1, en >= 5.0)          # done <=> energy >= m_energy
        np.testing.assert_allclose(out["coverage"], dn.sum(1) / M, rtol=0, atol=1e-15)
        # reward decomposition R = N*base + 75*#just with base = -sum min-dist + 1500[all done] + OOB terms
        just = (dn == 1) & (prev_d == 0)
        pos = out["pos_t"]
        for e in range(E):
            dist = np.linalg.norm(pos[e][:, None, :] - poi[None], axis=-1).min(0)
            base = -dist[dn[e] == 0].sum() + (1500.0 if dn[e].all() else 0.0)
            ap = np.abs(pos[e])
            base += -100.0 * np.clip(ap - 1, 0, None).sum() - 100.0 * (ap > 1.5).any(1).sum()
            np.testing.assert_allclose(out["reward"][e], N * base + 75.0 * just[e].sum(), rtol=1e-9, atol=1e-7)
            assert bool(out["done"][e]) == bool(dn[e].all() or (ap > 1.5).any())
        speed = np.linalg.norm(out["vel_t"], axis=-1)
        assert speed.max() <= 0.5 + 1e-12
        # after an auto-reset the stored state is the origin state
        st_ = o.get_state()
        r = out["done"] == 1
        assert not st_["pos"][r].any() and not st_["energy"][r].any()
        prev_e = np.where(r[:, None], 0.0, en); prev_d = np.where(r[:, None], 0, dn).astype(np.uint8)
