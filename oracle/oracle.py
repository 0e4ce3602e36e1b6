"""ctypes front-end of oracle/libdcc_oracle.so (the CPU restatement in dcc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdcc_oracle.so")
_lib = None


class _Cfg(ctypes.Structure):
    _fields_ = [("n_envs", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("n_pois", ctypes.c_int32),
                ("r_cover", ctypes.c_double), ("r_comm", ctypes.c_double),
                ("comm_r_scale", ctypes.c_double), ("comm_force_scale", ctypes.c_double)]


def build(force=False):
    """(Re)build libdcc_oracle.so when it is missing or older than its source.  Serialised with a file
    lock: the ranks of a multi-GPU bench import this module at the same time."""
    import fcntl
    src = os.path.join(_HERE, "dcc_env_cpu.c")       # includes dcc_oracle.c: one translation unit
    deps = [src, os.path.join(_HERE, "dcc_oracle.c"), os.path.join(_HERE, "dcc_gae_cpu.c"),
            os.path.join(_HERE, "..", "include", "dcc_env.h"), os.path.join(_HERE, "..", "include", "dcc_gae.h")]
    def stale():
        return force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(d) for d in deps)
    if stale():
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if stale():
                    tmp = _LIB_PATH + ".tmp%d" % os.getpid()
                    subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                                           "-fvisibility=hidden", "-std=c11", "-shared", "-o", tmp, src, "-lm"])
                    os.replace(tmp, _LIB_PATH)   # atomic: a concurrent dlopen never sees a half-written file
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp = ctypes.c_void_p
        L.dcc_oracle_create.restype = vp
        L.dcc_oracle_create.argtypes = [ctypes.POINTER(_Cfg), vp]
        L.dcc_oracle_destroy.argtypes = [vp]
        L.dcc_oracle_reset.argtypes = [vp, vp]
        L.dcc_oracle_get_state.argtypes = [vp] * 5
        L.dcc_oracle_set_state.argtypes = [vp] * 5
        L.dcc_oracle_step.argtypes = [vp, vp, ctypes.c_int] + [vp] * 12
        L.dcc_oracle_obs_dim.argtypes = [ctypes.c_int, ctypes.c_int]
        L.dcc_oracle_rng_actions.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, vp]
        L.dcc_oracle_rng_actions.restype = None
        L.dcc_oracle_rollout_rng.argtypes = [vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                             ctypes.c_int, vp, vp, vp, vp]
        L.dcc_gae_compute_cpu.argtypes = [vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, vp, vp, ctypes.c_int32, ctypes.c_int64, vp]
        L.dcc_returns_compute_cpu.argtypes = [vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_double, ctypes.c_int32, vp, vp,
                                              ctypes.c_int32, ctypes.c_int64, vp]
        _lib = L
    return _lib


def gae_compute_cpu(rewards, value_preds, masks, denorm, gamma, gae_lambda, returns, advantages=None):
    """The `_cpu` twin of include/dcc_gae.h's dcc_gae_compute on contiguous float32 numpy arrays (same argument order and
    shapes as dcc_hip.gae_compute: rewards [T,C], value_preds / masks / returns [T+1,C], denorm [2] or None, advantages [T,C])."""
    T, C = rewards.shape
    for a, shape in ((rewards, (T, C)), (value_preds, (T + 1, C)), (masks, (T + 1, C)), (returns, (T + 1, C))):
        assert a.dtype == np.float32 and a.flags.c_contiguous and a.shape == shape
    if advantages is not None:
        assert advantages.dtype == np.float32 and advantages.flags.c_contiguous and advantages.shape == (T, C)
    dn = None if denorm is None else np.ascontiguousarray(denorm, np.float32)
    rc = lib().dcc_gae_compute_cpu(_p(rewards), _p(value_preds), _p(masks), _p(dn), float(gamma), float(gae_lambda),
                                   _p(returns), _p(advantages), T, C, None)
    if rc != 0:
        raise RuntimeError("dcc_gae_compute_cpu failed: %d" % rc)
    return returns


def returns_compute_cpu(rewards, value_preds, masks, bad_masks, denorm, gamma, gae_lambda, mode, returns, advantages=None):
    """The `_cpu` twin of include/dcc_gae.h's dcc_returns_compute (mode: DCC_RETURNS_GAE = 1 | DCC_RETURNS_PROPER = 2); arrays as
    for gae_compute_cpu plus bad_masks [T+1,C] (None without DCC_RETURNS_PROPER)."""
    T, C = rewards.shape
    for a, shape in ((rewards, (T, C)), (value_preds, (T + 1, C)), (masks, (T + 1, C)), (returns, (T + 1, C)),
                     (bad_masks, (T + 1, C)), (advantages, (T, C))):
        assert a is None or (a.dtype == np.float32 and a.flags.c_contiguous and a.shape == shape)
    dn = None if denorm is None else np.ascontiguousarray(denorm, np.float32)
    rc = lib().dcc_returns_compute_cpu(_p(rewards), _p(value_preds), _p(masks), _p(bad_masks), _p(dn), float(gamma), float(gae_lambda),
                                       int(mode), _p(returns), _p(advantages), T, C, None)
    if rc != 0:
        raise RuntimeError("dcc_returns_compute_cpu failed: %d" % rc)
    return returns


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def rng_actions(seed, step, n_envs, n_agents, env0=0, env_total=None):
    out = np.empty((n_envs, n_agents, 2), np.float32)
    lib().dcc_oracle_rng_actions(seed, step, n_envs, n_agents, env0, env_total or n_envs, _p(out))
    return out


class OracleEnv:
    """Batched CPU env with the reference's semantics (float64 state).  step() returns a dict."""

    def __init__(self, n_envs, n_agents, n_pois, poi_xy, r_cover=0.2, r_comm=0.4, comm_r_scale=0.95,
                 comm_force_scale=0.0):
        self.E, self.N, self.M = n_envs, n_agents, n_pois
        self.D = lib().dcc_oracle_obs_dim(n_agents, n_pois)
        poi = np.ascontiguousarray(poi_xy, np.float64)
        assert poi.shape == (n_pois, 2)
        cfg = _Cfg(n_envs, n_agents, n_pois, r_cover, r_comm, comm_r_scale, comm_force_scale)
        self._h = lib().dcc_oracle_create(ctypes.byref(cfg), _p(poi))
        if not self._h:
            raise ValueError("dcc_oracle_create failed")

    def close(self):
        if self._h:
            lib().dcc_oracle_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        obs = np.empty((self.E, self.N, self.D), np.float64)
        lib().dcc_oracle_reset(self._h, _p(obs))
        return obs

    def get_state(self):
        pos = np.empty((self.E, self.N, 2)); vel = np.empty((self.E, self.N, 2))
        en = np.empty((self.E, self.M)); dn = np.empty((self.E, self.M), np.uint8)
        lib().dcc_oracle_get_state(self._h, _p(pos), _p(vel), _p(en), _p(dn))
        return dict(pos=pos, vel=vel, energy=en, done=dn)

    def set_state(self, pos=None, vel=None, energy=None, done=None):
        c = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
        pos, vel, energy, done = c(pos, np.float64), c(vel, np.float64), c(energy, np.float64), c(done, np.uint8)
        lib().dcc_oracle_set_state(self._h, _p(pos), _p(vel), _p(energy), _p(done))

    def step(self, actions, want_obs=True):
        a = np.ascontiguousarray(actions)
        assert a.shape == (self.E, self.N, 2) and a.dtype in (np.float32, np.float64)
        E, N, M, D = self.E, self.N, self.M, self.D
        out = dict(
            obs=np.empty((E, N, D)) if want_obs else None, reward=np.empty(E), done=np.empty(E, np.uint8),
            connect=np.empty(E, np.uint8), connect_s=np.empty(E, np.uint8), coverage=np.empty(E),
            assign=np.empty((E, M), np.int32), pos_t=np.empty((E, N, 2)), vel_t=np.empty((E, N, 2)),
            energy_t=np.empty((E, M)), done_t=np.empty((E, M), np.uint8), force_pairs=np.empty((E, N, 2), np.int32))
        rc = lib().dcc_oracle_step(self._h, _p(a), int(a.dtype == np.float32), _p(out["obs"]), _p(out["reward"]),
                                   _p(out["done"]), _p(out["connect"]), _p(out["connect_s"]), _p(out["coverage"]),
                                   _p(out["assign"]), _p(out["pos_t"]), _p(out["vel_t"]), _p(out["energy_t"]),
                                   _p(out["done_t"]), _p(out["force_pairs"]))
        if rc != 0:
            raise RuntimeError("dcc_oracle_step rc=%d" % rc)
        return out

    def rollout_rng(self, K, seed, step0=0, env0=0, env_total=None, want_obs_last=False):
        E = self.E
        rew = np.empty((K, E)); dn = np.empty((K, E), np.uint8); cov = np.empty((K, E))
        obs = np.empty((E, self.N, self.D)) if want_obs_last else None
        lib().dcc_oracle_rollout_rng(self._h, K, seed, step0, env0, env_total or E, _p(rew), _p(dn), _p(cov), _p(obs))
        return dict(reward=rew, done=dn, coverage=cov, obs_last=obs)


class CpuTwinEnv:
    """The `_cpu` twins of include/dcc_env.h (oracle/dcc_env_cpu.c) behind the product's own ctypes struct definitions:
    the same EnvCfg / EnvOut layouts as dcc_hip.py, host (numpy) arrays in place of device tensors."""

    def __init__(self, n_envs, n_agents, n_pois, poi_xy, r_cover=0.2, r_comm=0.4, comm_r_scale=0.95, comm_force_scale=0.0):
        import sys
        sys.path.insert(0, os.path.join(_HERE, "..", "dynamic-coverage-control_amd"))
        import dcc_hip                                   # struct layouts + dcc_env_cfg_default only (needs no GPU)
        self._hip = dcc_hip
        L = lib()
        vp = ctypes.c_void_p
        L.dcc_env_create_cpu.argtypes = [ctypes.POINTER(dcc_hip.EnvCfg), ctypes.POINTER(vp)]
        L.dcc_env_destroy_cpu.argtypes = [vp]
        L.dcc_env_obs_dim_cpu.argtypes = [vp]
        L.dcc_env_reset_cpu.argtypes = [vp, vp, vp]
        L.dcc_env_step_cpu.argtypes = [vp, vp, ctypes.c_int, ctypes.POINTER(dcc_hip.EnvOut), vp]
        L.dcc_env_rollout_cpu.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.POINTER(dcc_hip.EnvOut), vp]
        L.dcc_env_get_state_cpu.argtypes = [vp] * 6
        L.dcc_env_set_state_cpu.argtypes = [vp] * 6
        L.dcc_obs_expand_cpu.argtypes = [vp, ctypes.c_int64, vp, vp, vp, vp, vp, vp]
        L.dcc_obs_features_x_cpu.argtypes = [vp, ctypes.c_int64] + [vp] * 11
        L.dcc_last_error_cpu.restype = ctypes.c_char_p
        self.L = L
        cfg = dcc_hip.EnvCfg()
        dcc_hip.load_library().dcc_env_cfg_default(ctypes.byref(cfg))      # the product library's defaults
        self._poi = np.ascontiguousarray(poi_xy, np.float64)
        cfg.n_envs, cfg.n_agents, cfg.n_pois = n_envs, n_agents, n_pois
        cfg.r_cover, cfg.r_comm, cfg.comm_r_scale, cfg.comm_force_scale = r_cover, r_comm, comm_r_scale, comm_force_scale
        cfg.poi_xy = self._poi.ctypes.data_as(vp)
        self.cfg = cfg
        h = vp()
        rc = L.dcc_env_create_cpu(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            raise ValueError("dcc_env_create_cpu failed (%d): %s" % (rc, L.dcc_last_error_cpu().decode()))
        self._h = h
        self.E, self.N, self.M = n_envs, n_agents, n_pois
        self.D = L.dcc_env_obs_dim_cpu(h)

    def close(self):
        if getattr(self, "_h", None):
            self.L.dcc_env_destroy_cpu(self._h)
            self._h = None

    __del__ = close

    def alloc_out(self, K=None, obs=True, assign=True, state=False):
        lead = () if K is None else (K,)
        E, N, M, D = self.E, self.N, self.M, self.D
        out = dict(reward=np.empty(lead + (E,), np.float32), done=np.empty(lead + (E,), np.uint8),
                   connect=np.empty(lead + (E,), np.uint8), connect_s=np.empty(lead + (E,), np.uint8),
                   coverage=np.empty(lead + (E,), np.float32), reward64=np.empty(lead + (E,), np.float64))
        if obs:
            out["obs"] = np.empty(lead + (E, N, D), np.float32)
        if assign:
            out["assign"] = np.empty(lead + (E, M), np.uint8)
        if state:
            out.update(state_pos=np.empty(lead + (E, N, 2)), state_vel=np.empty(lead + (E, N, 2)),
                       state_energy=np.empty(lead + (E, M), np.float32), state_done=np.empty(lead + (E, M), np.uint8))
        return out

    def _struct(self, out):
        o = self._hip.EnvOut()
        for k, a in out.items():
            assert a.flags["C_CONTIGUOUS"]
            setattr(o, k, a.ctypes.data)
        return o

    def reset(self):
        obs = np.empty((self.E, self.N, self.D), np.float32)
        assert self.L.dcc_env_reset_cpu(self._h, _p(obs), None) == 0
        return obs

    def step(self, actions, out=None):
        a = np.ascontiguousarray(actions)
        out = out if out is not None else self.alloc_out()
        o = self._struct(out)
        rc = self.L.dcc_env_step_cpu(self._h, _p(a), 0 if a.dtype == np.float32 else 1, ctypes.byref(o), None)
        if rc != 0:
            raise RuntimeError(self.L.dcc_last_error_cpu().decode())
        return out

    def rollout(self, K, actions=None, seed=0, step0=0, env0=0, env_total=None, out=None):
        out = out if out is not None else self.alloc_out(K)
        o = self._struct(out)
        a = None if actions is None else np.ascontiguousarray(actions, np.float32)
        rc = self.L.dcc_env_rollout_cpu(self._h, K, _p(a), seed, step0, env0, env_total or self.E, ctypes.byref(o), None)
        if rc != 0:
            raise RuntimeError(self.L.dcc_last_error_cpu().decode())
        return out

    def expand_obs(self, pos, vel, energy, done):
        """dcc_obs_expand_cpu: observation rows [n,N,D] float32 of n states (pos / vel [n,N,2] f64, energy [n,M] f32, done [n,M] u8)."""
        c = lambda a, t: np.ascontiguousarray(a, t)
        pos, vel, energy, done = c(pos, np.float64), c(vel, np.float64), c(energy, np.float32), c(done, np.uint8)
        n = pos.shape[0]
        obs = np.empty((n, self.N, self.D), np.float32)
        rc = self.L.dcc_obs_expand_cpu(self._h, n, _p(pos), _p(vel), _p(energy), _p(done), _p(obs), None)
        if rc != 0:
            raise RuntimeError("dcc_obs_expand_cpu failed: %d" % rc)
        return obs

    def feature_shapes(self, n):
        HD = 4 + 2 * (self.N - 1)
        ka, kc = (2 * self.M + 1 + 7) // 8 * 8, (self.N * HD + 2 * self.M + 1 + 7) // 8 * 8
        return dict(head=((n, self.N, HD), np.float32), poi_feat=((n, 2 * self.M), np.float32), stats=((n, self.N, 2), np.float64),
                    cstats=((n, 2), np.float64), xa=((n, ka), np.float32), xc=((n, kc), np.float32))

    def obs_features(self, pos, vel, energy, done, out=None):
        """dcc_obs_features_x_cpu: the compact policy-input features of n states (same names / shapes as dcc_hip's obs_features);
        `out`: a dict naming the outputs wanted (missing keys are skipped)."""
        c = lambda a, t: np.ascontiguousarray(a, t)
        pos, vel, energy, done = c(pos, np.float64), c(vel, np.float64), c(energy, np.float32), c(done, np.uint8)
        n = pos.shape[0]
        if out is None:
            out = {k: np.empty(sh, dt) for k, (sh, dt) in self.feature_shapes(n).items()}
        for k, (sh, dt) in self.feature_shapes(n).items():
            a = out.get(k)
            assert a is None or (a.shape == sh and a.dtype == dt and a.flags["C_CONTIGUOUS"]), k
        g = lambda k: _p(out.get(k))
        rc = self.L.dcc_obs_features_x_cpu(self._h, n, _p(pos), _p(vel), _p(energy), _p(done), g("head"), g("poi_feat"), g("stats"),
                                           g("cstats"), g("xa"), g("xc"), None)
        if rc != 0:
            raise RuntimeError("dcc_obs_features_x_cpu failed: %d" % rc)
        return out

    def get_state(self):
        st = dict(pos=np.empty((self.E, self.N, 2)), vel=np.empty((self.E, self.N, 2)), energy=np.empty((self.E, self.M), np.float32),
                  done=np.empty((self.E, self.M), np.uint8))
        self.L.dcc_env_get_state_cpu(self._h, _p(st["pos"]), _p(st["vel"]), _p(st["energy"]), _p(st["done"]), None)
        return st
