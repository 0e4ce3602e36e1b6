"""TEST-ONLY: lets the package's host logic (Learner, vec-env, rollout buffer, trainer) run on a box WITHOUT a GPU by standing the
`_cpu` twins of the C-ABI (oracle/libdcc_oracle.so: dcc_env_*_cpu, dcc_obs_expand_cpu, dcc_obs_features_x_cpu, dcc_gae_compute_cpu, dcc_returns_compute_cpu -- the C restatement of
the reference behind the same structs) in for the two device entry points the learner needs: `dcc_hip.HipCoverageEnv` and
`dcc_hip.gae_compute`.  Everything above them is the product's own code on torch CPU tensors.

This is how `-m "not gpu"` pins the ORCHESTRATOR (warmup / collect / insert / compute / rl_update / after_update, lr schedule,
ValueNorm carried over iterations, eval rollouts) against the reference's own `Learner` fixture here in the build container; the
product never takes this path (no GPU -> it fails loudly), and the `-m gpu` run of the same test goes through the HIP kernels.
"""
import contextlib

import numpy as np
import torch


class TwinAsHipEnv(object):
    """The subset of dcc_hip.HipCoverageEnv the device surface of HipCoverageVecEnv / Learner uses, on CPU tensors."""

    def __init__(self, n_envs, n_agents, n_pois, poi_xy, r_cover=0.2, r_comm=0.4, comm_r_scale=0.95, comm_force_scale=0.0,
                 device=None, **consts):
        from oracle import oracle
        assert not consts, "the twin adapter takes the default env constants"
        self._t = oracle.CpuTwinEnv(n_envs, n_agents, n_pois, poi_xy, r_cover, r_comm, comm_r_scale, comm_force_scale)
        self.device = torch.device("cpu")
        self.E, self.N, self.M, self.D = n_envs, n_agents, n_pois, self._t.D
        self.m_energy = float(self._t.cfg.m_energy)
        self.poi = np.ascontiguousarray(poi_xy, np.float64)

    def close(self):
        self._t.close()

    def reset(self, obs=None):
        o = torch.from_numpy(self._t.reset())
        if obs is None:
            return o
        obs.copy_(o)
        return obs

    def alloc_out(self, K=None, obs=True, assign=True, reward64=False, placed=0):
        out = {k: torch.from_numpy(v) for k, v in self._t.alloc_out(K, obs=obs, assign=assign).items()}
        if not reward64:
            out.pop("reward64")
        return out

    def alloc_state_out(self, K=None):
        full = self._t.alloc_out(K, obs=False, assign=False, state=True)
        return {k: torch.from_numpy(v) for k, v in full.items() if k.startswith("state_")}

    def step(self, actions, out=None):
        out = out if out is not None else self.alloc_out()
        views = {}
        for k, t in out.items():
            if t is None:
                continue
            assert t.device.type == "cpu" and t.is_contiguous(), k
            views[k] = t.detach().numpy()             # shares memory: the twin writes straight into the caller's tensors
        self._t.step(actions.detach().numpy(), views)
        return out

    def get_state(self):
        return {k: torch.from_numpy(v) for k, v in self._t.get_state().items()}

    def expand_obs(self, pos, vel, energy, done, obs=None):
        rows = torch.from_numpy(self._t.expand_obs(pos.numpy(), vel.numpy(), energy.numpy(), done.numpy()))
        if obs is None:
            return rows
        obs.copy_(rows.view_as(obs))
        return obs

    def feature_shapes(self, n):
        tt = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}
        return {k: (sh, tt[np.dtype(dt)]) for k, (sh, dt) in self._t.feature_shapes(n).items()}

    def alloc_features(self, n=None, keys=("head", "stats", "cstats", "xa", "xc")):
        return {k: torch.empty(sh, dtype=dt) for k, (sh, dt) in self.feature_shapes(n or self.E).items() if k in keys}

    def obs_features(self, pos, vel, energy, done, out=None):
        """dcc_obs_features_x_cpu; `out`: dict of destination tensors (missing keys are skipped), like dcc_hip's."""
        if out is None:
            out = {k: torch.empty(sh, dtype=dt) for k, (sh, dt) in self.feature_shapes(pos.shape[0]).items()}
        self._t.obs_features(pos.numpy(), vel.numpy(), energy.numpy(), done.numpy(),
                             {k: t.numpy() for k, t in out.items() if t is not None})
        return out


def _gae_compute(rewards, value_preds, masks, denorm, gamma, gae_lambda, returns, advantages=None):
    from oracle import oracle
    n = lambda t: None if t is None else t.detach().numpy()
    for t in (rewards, value_preds, masks, returns, advantages):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    oracle.gae_compute_cpu(n(rewards), n(value_preds), n(masks), None if denorm is None else n(denorm.contiguous()), gamma, gae_lambda,
                           n(returns), n(advantages))
    return returns


def _returns_compute(rewards, value_preds, masks, bad_masks, denorm, gamma, gae_lambda, mode, returns, advantages=None):
    from oracle import oracle
    n = lambda t: None if t is None else t.detach().numpy()
    for t in (rewards, value_preds, masks, bad_masks, returns, advantages):
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32)
    oracle.returns_compute_cpu(n(rewards), n(value_preds), n(masks), n(bad_masks), None if denorm is None else n(denorm.contiguous()),
                               gamma, gae_lambda, mode, n(returns), n(advantages))
    return returns


@contextlib.contextmanager
def cpu_twin_backend():
    import dcc_hip
    from oracle import oracle
    oracle.build()
    saved = dcc_hip.HipCoverageEnv, dcc_hip.gae_compute, dcc_hip.returns_compute
    dcc_hip.HipCoverageEnv, dcc_hip.gae_compute, dcc_hip.returns_compute = TwinAsHipEnv, _gae_compute, _returns_compute
    try:
        yield
    finally:
        dcc_hip.HipCoverageEnv, dcc_hip.gae_compute, dcc_hip.returns_compute = saved
