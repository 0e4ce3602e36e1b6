"""ctypes binding of libdcc_hip.so (include/dcc_env.h) -- the thin C-ABI seam between the Python
host code and the hand-written HIP kernels.

There is NO CPU fallback: if the shared library is missing or no HIP device is visible, loading /
creating raises.  Tensors are torch CUDA(=HIP) tensors; only their data_ptr() crosses the ABI.
"""
import ctypes
import os

import numpy as np
import torch  # imported before the library so that both share one HIP runtime (libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCC_HIP_LIB", os.path.join(_HERE, "csrc", "libdcc_hip.so"))

ACT_F32, ACT_F64 = 0, 1
_vp = ctypes.c_void_p


class DccError(RuntimeError):
    pass


class EnvCfg(ctypes.Structure):
    _fields_ = [("n_envs", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("n_pois", ctypes.c_int32),
                ("device", ctypes.c_int32),
                ("r_cover", ctypes.c_double), ("r_comm", ctypes.c_double), ("comm_r_scale", ctypes.c_double),
                ("comm_force_scale", ctypes.c_double),
                ("dt", ctypes.c_double), ("damping", ctypes.c_double), ("max_speed", ctypes.c_double),
                ("sensitivity", ctypes.c_double), ("mass", ctypes.c_double),
                ("contact_margin", ctypes.c_double), ("m_energy", ctypes.c_double),
                ("rew_cover", ctypes.c_double), ("rew_done", ctypes.c_double), ("rew_out", ctypes.c_double),
                ("bound_soft", ctypes.c_double), ("bound_hard", ctypes.c_double),
                ("poi_xy", _vp)]


class EnvOut(ctypes.Structure):
    _fields_ = [("obs", _vp), ("reward", _vp), ("done", _vp), ("connect", _vp), ("connect_s", _vp),
                ("coverage", _vp), ("assign", _vp), ("reward64", _vp),
                ("state_pos", _vp), ("state_vel", _vp), ("state_energy", _vp), ("state_done", _vp)]


class ObsFeat(ctypes.Structure):       # include/dcc_env.h: dcc_obs_feat
    _fields_ = [("head", _vp), ("poi_feat", _vp), ("stats", _vp), ("cstats", _vp), ("xa", _vp), ("xc", _vp)]


EXPORTS = ["dcc_obs_expand", "dcc_gae_compute", "dcc_returns_compute", "dcc_abi_version", "dcc_last_error", "dcc_env_cfg_default", "dcc_env_create", "dcc_env_destroy",
           "dcc_env_obs_dim", "dcc_env_reset", "dcc_env_step", "dcc_env_step_features", "dcc_env_obs_write_probe", "dcc_env_rollout", "dcc_env_get_state",
           "dcc_env_set_state", "dcc_env_bytes_per_step", "dcc_env_kernel_choice", "dcc_obs_features", "dcc_obs_features_x",
           "dcc_relu_ln_fwd", "dcc_relu_ln_bwd", "dcc_relu_ln_head_fwd", "dcc_relu_ln_head_bwd", "dcc_mlp_workspace_floats", "dcc_actor_l1_fwd", "dcc_actor_l1_bwd", "dcc_actor_l1_pre_fwd", "dcc_actor_l1_pre_bwd",
           "dcc_ppo_policy_loss",
           "dcc_rollout_sample", "dcc_rollout_record", "dcc_rollout_record_stats", "dcc_ppo_value_loss",
           "dcc_grad_norm_workspace_floats", "dcc_grad_norm_clip", "dcc_adam_step"]

_lib = None


def load_library(path=None):
    """dlopen libdcc_hip.so and declare every prototype of include/dcc_env.h.  Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise DccError("HIP extension %s not found: build it with `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    L.dcc_abi_version.restype = ctypes.c_int
    L.dcc_last_error.restype = ctypes.c_char_p
    L.dcc_env_cfg_default.argtypes = [ctypes.POINTER(EnvCfg)]
    L.dcc_env_cfg_default.restype = None
    L.dcc_env_create.argtypes = [ctypes.POINTER(EnvCfg), ctypes.POINTER(_vp)]
    L.dcc_env_destroy.argtypes = [_vp]
    L.dcc_env_obs_dim.argtypes = [_vp]
    L.dcc_env_reset.argtypes = [_vp, _vp, _vp]
    L.dcc_env_step.argtypes = [_vp, _vp, ctypes.c_int, ctypes.POINTER(EnvOut), _vp]
    L.dcc_env_rollout.argtypes = [_vp, ctypes.c_int32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32,
                                  ctypes.c_int32, ctypes.POINTER(EnvOut), _vp]
    L.dcc_env_get_state.argtypes = [_vp] * 6
    L.dcc_env_set_state.argtypes = [_vp] * 6
    L.dcc_env_bytes_per_step.argtypes = [ctypes.c_int32] * 4
    L.dcc_env_bytes_per_step.restype = ctypes.c_int64
    L.dcc_env_kernel_choice.argtypes = [_vp, _vp, _vp]
    L.dcc_env_obs_write_probe.argtypes = [_vp, ctypes.c_int32, _vp, _vp]
    L.dcc_env_step_features.argtypes = [_vp, _vp, ctypes.c_int, ctypes.POINTER(EnvOut), ctypes.POINTER(ObsFeat), _vp]
    L.dcc_gae_compute.argtypes = [_vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp, ctypes.c_int32,
                                  ctypes.c_int64, _vp]
    L.dcc_returns_compute.argtypes = [_vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_int32, _vp, _vp,
                                      ctypes.c_int32, ctypes.c_int64, _vp]
    L.dcc_obs_expand.argtypes = [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]
    L.dcc_obs_features.argtypes = [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.dcc_obs_features_x.argtypes = [_vp, ctypes.c_int64] + [_vp] * 11
    f32, i32, i64 = ctypes.c_float, ctypes.c_int32, ctypes.c_int64
    L.dcc_relu_ln_fwd.argtypes = [_vp, _vp, _vp, _vp, f32, _vp, i64, i32, _vp]
    L.dcc_relu_ln_bwd.argtypes = [_vp, _vp, _vp, _vp, f32, _vp, _vp, _vp, i64, i32, _vp]
    L.dcc_relu_ln_head_fwd.argtypes = [_vp, _vp, _vp, _vp, f32, _vp, _vp, _vp, i64, i32, i32, _vp]
    L.dcc_relu_ln_head_bwd.argtypes = [_vp, _vp, _vp, _vp, f32, _vp, _vp, _vp, _vp, _vp, i64, i32, i32, _vp]
    L.dcc_ppo_policy_loss.argtypes = [_vp] * 6 + [f32, _vp, _vp, _vp, i64, i32, i32, _vp]
    L.dcc_ppo_value_loss.argtypes = [_vp] * 5 + [f32, f32, i32, _vp, _vp, _vp, i64, i32, _vp]
    L.dcc_rollout_sample.argtypes = [_vp] * 7 + [i64, i32, i32, i32, _vp]
    L.dcc_rollout_record.argtypes = [_vp] * 4 + [i64, i32, _vp]
    L.dcc_rollout_record_stats.argtypes = [_vp] * 7 + [i64, i32, _vp]
    L.dcc_mlp_workspace_floats.argtypes = [i32, i32]
    L.dcc_mlp_workspace_floats.restype = i64
    L.dcc_actor_l1_fwd.argtypes = [_vp] * 8 + [f32, f32, i32, _vp, i64, i32, i32, i32, _vp]
    L.dcc_actor_l1_bwd.argtypes = [_vp] * 8 + [f32, f32, i32] + [_vp] * 8 + [i64, i32, i32, i32, _vp]
    L.dcc_actor_l1_pre_fwd.argtypes = [_vp] * 7 + [f32, f32, i32, _vp, i64, i32, i32, _vp]
    L.dcc_actor_l1_pre_bwd.argtypes = [_vp] * 7 + [f32, f32, i32] + [_vp] * 7 + [i64, i32, i32, _vp]
    L.dcc_grad_norm_workspace_floats.argtypes = [i64]
    L.dcc_grad_norm_workspace_floats.restype = i64
    L.dcc_grad_norm_clip.argtypes = [_vp, i64, f32, _vp, _vp, _vp]
    L.dcc_adam_step.argtypes = [_vp, _vp, _vp, _vp, i64, f32, f32, ctypes.c_double, ctypes.c_double, f32, f32, _vp, _vp]
    if L.dcc_abi_version() != 2:
        raise DccError("libdcc_hip.so ABI version %d != 2" % L.dcc_abi_version())
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise DccError("%s failed (%d): %s" % (what, rc, load_library().dcc_last_error().decode()))


def _ptr(t):
    return None if t is None else _vp(t.data_ptr())


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def bytes_per_step(n_agents, n_pois, with_actions=True, with_obs=True):
    return int(load_library().dcc_env_bytes_per_step(n_agents, n_pois, int(with_actions), int(with_obs)))


class HipCoverageEnv:
    """Low-level handle: E batched envs resident on one GPU; all I/O are torch device tensors."""

    def __init__(self, n_envs, n_agents, n_pois, poi_xy, r_cover=0.2, r_comm=0.4, comm_r_scale=0.95,
                 comm_force_scale=0.0, device=None, **consts):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise DccError("no HIP device visible: the coverage env has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        poi = np.ascontiguousarray(poi_xy, np.float64)
        if poi.shape != (n_pois, 2):
            raise ValueError("poi_xy must be [n_pois, 2]")
        cfg = EnvCfg()
        self.lib.dcc_env_cfg_default(ctypes.byref(cfg))
        cfg.n_envs, cfg.n_agents, cfg.n_pois, cfg.device = n_envs, n_agents, n_pois, self.device.index
        cfg.r_cover, cfg.r_comm, cfg.comm_r_scale, cfg.comm_force_scale = r_cover, r_comm, comm_r_scale, comm_force_scale
        for k, v in consts.items():
            if not hasattr(cfg, k):
                raise TypeError("unknown env constant %r" % k)
            setattr(cfg, k, v)
        cfg.poi_xy = poi.ctypes.data_as(_vp)
        h = _vp()
        _check(self.lib.dcc_env_create(ctypes.byref(cfg), ctypes.byref(h)), "dcc_env_create")
        self._h = h
        self.E, self.N, self.M = n_envs, n_agents, n_pois
        self.D = self.lib.dcc_env_obs_dim(h)
        self.poi = poi
        self.m_energy = float(cfg.m_energy)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.dcc_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- allocation helpers ---------------------------------------------------------------------
    def kernel_choice(self):
        """{"choice": "roles" | "fused" | "default", "us_per_step_roles", "us_per_step_fused"}: what dcc_env_create measured on
        this device for obs-writing multi-step launches (include/dcc_env.h: dcc_env_kernel_choice)."""
        a, b = ctypes.c_float(0.0), ctypes.c_float(0.0)
        rc = self.lib.dcc_env_kernel_choice(self._h, ctypes.byref(a), ctypes.byref(b))
        if rc < 0:
            raise DccError("dcc_env_kernel_choice failed (%d): %s" % (rc, load_library().dcc_last_error().decode()))
        return {"choice": {0: "default", 1: "roles", 2: "fused"}[rc], "us_per_step_roles": a.value, "us_per_step_fused": b.value}

    PLACE_MIN_BYTES = 256 << 20     # buffers below this are not worth placing

    def alloc_placed_obs(self, K, tries=6, first=None):
        """An observation buffer [K, E, N, D] for MANY fused launches, placed well.  Where a buffer lies in HBM decides how fast
        the env kernels' store pattern streams into it: the same launch runs 6-8 % slower into some allocations than into
        others of the same process, reproducibly per buffer (tools/placement_probe.py; a sequential memset does not care, so
        this is about the scattered-chunk pattern, not about the memory).  Up to `tries` candidate allocations are timed with
        dcc_env_obs_write_probe (the observation producer alone over the WHOLE buffer, 3 launches: a buffer can be fast in one
        part and slow in another) while the earlier ones
        stay allocated, the fastest is kept and the others go back to the allocator; stops early once a candidate is clearly in
        the fast mode (>= 4 % ahead of the slowest seen).  RESETS the env state (call before the first step).
        `first`: a buffer of that shape the caller already holds (the process's first allocation): probed as candidate 0, so
        that what is kept is never a worse placement than the one the caller would have used anyway.
        Returns (tensor, info) with info = {"tried", "probe_ms", "chosen"}."""
        shape = (K, self.E, self.N, self.D)
        kp = K
        nbytes = K * self.E * self.N * self.D * 4
        free = torch.cuda.mem_get_info(self.device)[0]
        tries = max(1, min(int(tries), int(0.5 * free // max(1, nbytes))))     # the candidates are all alive at once: at most half of what is free
        cands, times = [], []
        ev = lambda: torch.cuda.Event(enable_timing=True)
        with torch.cuda.device(self.device):
            if first is not None and (tuple(first.shape) != shape or first.dtype != torch.float32 or not first.is_contiguous()):
                raise ValueError("alloc_placed_obs: `first` must be a contiguous float32 %s tensor" % (shape,))
            for i in range(max(1, tries) + (1 if first is not None else 0)):
                if i == 0 and first is not None:
                    t = first
                else:
                    try:
                        t = torch.empty(shape, dtype=torch.float32, device=self.device)
                    except torch.cuda.OutOfMemoryError:
                        break
                best = None
                for rep in range(4):        # rep 0 warms up (first touch)
                    a, b = ev(), ev()
                    a.record()
                    _check(self.lib.dcc_env_obs_write_probe(self._h, kp, _ptr(t), _stream()), "dcc_env_obs_write_probe")
                    b.record(); b.synchronize()
                    ms = a.elapsed_time(b)
                    if rep and (best is None or ms < best):
                        best = ms
                cands.append(t); times.append(best)
                if len(times) >= 2 and best <= 0.96 * max(times):
                    break
        if not cands:
            raise RuntimeError("alloc_placed_obs: out of memory")
        k = min(range(len(times)), key=times.__getitem__)
        keep = cands[k]
        del cands
        info = {"tried": len(times), "probe_ms": [round(x, 4) for x in times], "chosen": k, "probe_steps": kp}
        if first is not None:
            info["candidate_0"] = "the caller's first allocation"
        return keep, info

    def alloc_out(self, K=None, obs=True, assign=True, reward64=False, placed=0, first_obs=None):
        """Output tensors of step() (K None) / rollout(K).  placed = n > 0 (rollout buffers of >= 256 MB only): the observation
        buffer is the best-placed of up to n candidate allocations (alloc_placed_obs; resets the env state) and
        `self.placement_info` says what was tried."""
        lead = () if K is None else (K,)
        mk = lambda shape, dt: torch.empty(lead + shape, dtype=dt, device=self.device)
        out = dict(reward=mk((self.E,), torch.float32), done=mk((self.E,), torch.uint8),
                   connect=mk((self.E,), torch.uint8), connect_s=mk((self.E,), torch.uint8),
                   coverage=mk((self.E,), torch.float32))
        self.placement_info = None
        if obs and placed and K and K * self.E * self.N * self.D * 4 >= self.PLACE_MIN_BYTES:
            out["obs"], self.placement_info = self.alloc_placed_obs(K, placed, first=first_obs)
        elif obs:
            out["obs"] = mk((self.E, self.N, self.D), torch.float32)
        if assign:
            out["assign"] = mk((self.E, self.M), torch.uint8)
        if reward64:
            out["reward64"] = mk((self.E,), torch.float64)
        return out

    def alloc_state_out(self, K=None):
        """Compact post-step state outputs (ABI v2): add these to the `out` dict of step() / rollout()."""
        lead = () if K is None else (K,)
        mk = lambda shape, dt: torch.empty(lead + shape, dtype=dt, device=self.device)
        return dict(state_pos=mk((self.E, self.N, 2), torch.float64), state_vel=mk((self.E, self.N, 2), torch.float64),
                    state_energy=mk((self.E, self.M), torch.float32), state_done=mk((self.E, self.M), torch.uint8))

    def expand_obs(self, pos, vel, energy, done, obs=None):
        """obs [n,N,D] float32 from compact state (pos/vel [n,N,2] f64, energy [n,M] f32, done [n,M] u8)."""
        n = pos.shape[0]
        want = ((pos, (n, self.N, 2), torch.float64), (vel, (n, self.N, 2), torch.float64),
                (energy, (n, self.M), torch.float32), (done, (n, self.M), torch.uint8))
        for t, shape, dt in want:
            if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError("expand_obs: need contiguous %s %s on %s" % (shape, dt, self.device))
        if obs is None:
            obs = torch.empty((n, self.N, self.D), dtype=torch.float32, device=self.device)
        elif tuple(obs.shape) != (n, self.N, self.D) or obs.dtype != torch.float32 or not obs.is_contiguous():
            raise ValueError("expand_obs: obs must be contiguous float32 [n,N,D]")
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_obs_expand(self._h, n, _ptr(pos), _ptr(vel), _ptr(energy), _ptr(done), _ptr(obs), _stream()),
                   "dcc_obs_expand")
        return obs

    def obs_features(self, pos, vel, energy, done, out=None):
        """Compact policy-input features of n states (include/dcc_env.h: dcc_obs_features):
        dict(head [n,N,4+2(N-1)] f32, poi_feat [n,2M] f32, stats [n,N,2] f64 = (mean, sum sq. dev.) of each obs row,
        cstats [n,2] f64 = the same moments of the centralised row, xa / xc = the per-env GEMM inputs of
        dcc_obs_features_x).  `out`: a dict naming the outputs wanted (missing keys are skipped)."""
        n = pos.shape[0]
        want = ((pos, (n, self.N, 2), torch.float64), (vel, (n, self.N, 2), torch.float64),
                (energy, (n, self.M), torch.float32), (done, (n, self.M), torch.uint8))
        for t, shape, dt in want:
            if tuple(t.shape) != shape or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError("obs_features: need contiguous %s %s on %s" % (shape, dt, self.device))
        shapes = self.feature_shapes(n)
        if out is None:
            out = {k: torch.empty(sh, dtype=dt, device=self.device) for k, (sh, dt) in shapes.items()}
        for k, (sh, dt) in shapes.items():
            t = out.get(k)
            if t is not None and (tuple(t.shape) != sh or t.dtype != dt or not t.is_contiguous()):
                raise ValueError("obs_features: output %r must be contiguous %s %s" % (k, sh, dt))
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_obs_features_x(self._h, n, _ptr(pos), _ptr(vel), _ptr(energy), _ptr(done),
                                               _ptr(out.get("head")), _ptr(out.get("poi_feat")), _ptr(out.get("stats")),
                                               _ptr(out.get("cstats")), _ptr(out.get("xa")), _ptr(out.get("xc")), _stream()),
                   "dcc_obs_features_x")
        return out

    def _out_struct(self, out, K=None):
        # the same output tensors are passed step after step: validate and build the C struct once.  The key carries
        # everything the validation looks at (an address alone can be recycled by the caching allocator for a tensor of
        # another shape / dtype, which would then inherit a stale validation and let the kernel write out of bounds)
        key = (K,) + tuple((k, t.data_ptr(), tuple(t.shape), t.dtype, t.is_contiguous(), t.device.index)
                           for k, t in out.items() if t is not None)
        if key == getattr(self, "_out_key", None):
            return self._out_cached
        o = self._build_out_struct(out, K)
        self._out_key, self._out_cached = key, o
        return o

    def _build_out_struct(self, out, K=None):
        o = EnvOut()
        lead = () if K is None else (K,)
        shapes = dict(obs=(self.E, self.N, self.D), reward=(self.E,), done=(self.E,), connect=(self.E,),
                      connect_s=(self.E,), coverage=(self.E,), assign=(self.E, self.M), reward64=(self.E,),
                      state_pos=(self.E, self.N, 2), state_vel=(self.E, self.N, 2), state_energy=(self.E, self.M),
                      state_done=(self.E, self.M))
        dts = dict(obs=torch.float32, reward=torch.float32, done=torch.uint8, connect=torch.uint8,
                   connect_s=torch.uint8, coverage=torch.float32, assign=torch.uint8, reward64=torch.float64,
                   state_pos=torch.float64, state_vel=torch.float64, state_energy=torch.float32,
                   state_done=torch.uint8)
        for k, t in out.items():
            if t is None:
                continue
            if k not in shapes:
                raise KeyError(k)
            if tuple(t.shape) != lead + shapes[k] or t.dtype != dts[k] or not t.is_contiguous() or t.device != self.device:
                raise ValueError("output %r must be contiguous %s %s on %s" % (k, lead + shapes[k], dts[k], self.device))
            setattr(o, k, t.data_ptr())
        return o

    # ---- entry points -----------------------------------------------------------------------------
    def reset(self, obs=None):
        if obs is None:
            obs = torch.empty((self.E, self.N, self.D), dtype=torch.float32, device=self.device)
        self._out_struct(dict(obs=obs))
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_reset(self._h, _ptr(obs), _stream()), "dcc_env_reset")
        return obs

    def step(self, actions, out=None):
        if actions.device != self.device or not actions.is_contiguous() or tuple(actions.shape) != (self.E, self.N, 2):
            raise ValueError("actions must be a contiguous [E,N,2] tensor on %s" % self.device)
        if actions.dtype == torch.float32:
            dt = ACT_F32
        elif actions.dtype == torch.float64:
            dt = ACT_F64
        else:
            raise ValueError("actions must be float32 or float64")
        if out is None:
            out = self.alloc_out()
        o = self._out_struct(out)
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_step(self._h, _ptr(actions), dt, ctypes.byref(o), _stream()), "dcc_env_step")
        return out

    def feature_shapes(self, n):
        """name -> (shape, dtype) of the feature tensors of n states (dcc_obs_features_x / dcc_env_step_features)."""
        HD = 4 + 2 * (self.N - 1)
        ka, kc = (2 * self.M + 1 + 7) // 8 * 8, (self.N * HD + 2 * self.M + 1 + 7) // 8 * 8
        return dict(head=((n, self.N, HD), torch.float32), poi_feat=((n, 2 * self.M), torch.float32),
                    stats=((n, self.N, 2), torch.float64), cstats=((n, 2), torch.float64),
                    xa=((n, ka), torch.float32), xc=((n, kc), torch.float32))

    def alloc_features(self, n=None, keys=("head", "stats", "cstats", "xa", "xc")):
        return {k: torch.empty(sh, dtype=dt, device=self.device) for k, (sh, dt) in self.feature_shapes(n or self.E).items() if k in keys}

    def step_features(self, actions, out, feat):
        """One env step + the policy-input features of the state it leaves, in ONE launch (dcc_env_step_features).  `out`: as
        for step(), without "obs"; `feat`: dict of destination tensors (feature_shapes(E)); missing keys are skipped."""
        if (actions.device != self.device or not actions.is_contiguous() or tuple(actions.shape) != (self.E, self.N, 2)
                or actions.dtype != torch.float32):
            raise ValueError("actions must be a contiguous float32 [E,N,2] tensor on %s" % self.device)
        if out.get("obs") is not None:
            raise ValueError("step_features writes no observation rows")
        o = self._out_struct(out)
        key = tuple((k, t.data_ptr(), tuple(t.shape), t.dtype) for k, t in feat.items())
        cached = getattr(self, "_feat_structs", None)
        if cached is None:
            cached = self._feat_structs = {}
        fs = cached.get(key)
        if fs is None:
            shapes = self.feature_shapes(self.E)
            fs = ObsFeat()
            for k, t in feat.items():
                sh, dt = shapes[k]
                if tuple(t.shape) != sh or t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                    raise ValueError("step_features: output %r must be contiguous %s %s on %s" % (k, sh, dt, self.device))
                setattr(fs, k, t.data_ptr())
            if len(cached) > 8:
                cached.clear()
            cached[key] = fs
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_step_features(self._h, _ptr(actions), ACT_F32, ctypes.byref(o), ctypes.byref(fs), _stream()),
                   "dcc_env_step_features")
        return out

    def rollout(self, K, actions=None, seed=0, step0=0, env0=0, env_total=None, out=None):
        if actions is not None:
            if (actions.device != self.device or actions.dtype != torch.float32 or not actions.is_contiguous()
                    or tuple(actions.shape) != (K, self.E, self.N, 2)):
                raise ValueError("actions must be a contiguous float32 [K,E,N,2] tensor on %s" % self.device)
        if out is None:
            out = self.alloc_out(K)
        o = self._out_struct(out, K)
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_rollout(self._h, K, _ptr(actions), seed, step0, env0,
                                            self.E if env_total is None else env_total, ctypes.byref(o), _stream()),
                   "dcc_env_rollout")
        return out

    def get_state(self):
        d = self.device
        st = dict(pos=torch.empty((self.E, self.N, 2), dtype=torch.float64, device=d),
                  vel=torch.empty((self.E, self.N, 2), dtype=torch.float64, device=d),
                  energy=torch.empty((self.E, self.M), dtype=torch.float32, device=d),
                  done=torch.empty((self.E, self.M), dtype=torch.uint8, device=d))
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_get_state(self._h, _ptr(st["pos"]), _ptr(st["vel"]), _ptr(st["energy"]),
                                              _ptr(st["done"]), _stream()), "dcc_env_get_state")
        return st

    def set_state(self, pos=None, vel=None, energy=None, done=None):
        def c(t, dt, shape):
            if t is None:
                return None
            t = torch.as_tensor(t).to(device=self.device, dtype=dt).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError("bad state shape %s, want %s" % (tuple(t.shape), shape))
            return t
        pos = c(pos, torch.float64, (self.E, self.N, 2)); vel = c(vel, torch.float64, (self.E, self.N, 2))
        energy = c(energy, torch.float32, (self.E, self.M)); done = c(done, torch.uint8, (self.E, self.M))
        with torch.cuda.device(self.device):
            _check(self.lib.dcc_env_set_state(self._h, _ptr(pos), _ptr(vel), _ptr(energy), _ptr(done), _stream()),
                   "dcc_env_set_state")
            torch.cuda.current_stream().synchronize()  # inputs may be temporaries


def gae_compute(rewards, value_preds, masks, denorm, gamma, gae_lambda, returns, advantages=None):
    """include/dcc_gae.h: rewards [T,C], value_preds/masks/returns [T+1,C], denorm [2] or None, advantages [T,C]
    or None -- contiguous float32 tensors on one HIP device.  No CPU path."""
    L = load_library()
    T, C = rewards.shape
    ts = [("rewards", rewards, (T, C)), ("value_preds", value_preds, (T + 1, C)), ("masks", masks, (T + 1, C)),
          ("returns", returns, (T + 1, C))]
    if advantages is not None:
        ts.append(("advantages", advantages, (T, C)))
    if denorm is not None:
        ts.append(("denorm", denorm, (2,)))
    for name, t, shape in ts:
        if not t.is_cuda:
            raise DccError("gae_compute: %s is not on a HIP device (there is no CPU path)" % name)
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape or t.device != rewards.device:
            raise ValueError("gae_compute: %s must be contiguous float32 %s on %s" % (name, shape, rewards.device))
    with torch.cuda.device(rewards.device):
        rc = L.dcc_gae_compute(_ptr(rewards), _ptr(value_preds), _ptr(masks), _ptr(denorm), float(gamma),
                               float(gae_lambda), _ptr(returns), _ptr(advantages), T, C, _stream())
    _check(rc, "dcc_gae_compute")
    return returns

RETURNS_GAE, RETURNS_PROPER = 1, 2     # include/dcc_gae.h: DCC_RETURNS_GAE, DCC_RETURNS_PROPER


def returns_compute(rewards, value_preds, masks, bad_masks, denorm, gamma, gae_lambda, mode, returns, advantages=None):
    """include/dcc_gae.h: dcc_returns_compute -- every branch of the reference's compute_returns (mode = RETURNS_GAE |
    RETURNS_PROPER bits).  Tensors as for gae_compute plus bad_masks [T+1,C] (None without RETURNS_PROPER); without RETURNS_GAE
    row T of `returns` must hold the bootstrap value.  No CPU path."""
    L = load_library()
    T, C = rewards.shape
    ts = [("rewards", rewards, (T, C)), ("value_preds", value_preds, (T + 1, C)), ("masks", masks, (T + 1, C)),
          ("returns", returns, (T + 1, C))]
    if bad_masks is not None:
        ts.append(("bad_masks", bad_masks, (T + 1, C)))
    if advantages is not None:
        ts.append(("advantages", advantages, (T, C)))
    if denorm is not None:
        ts.append(("denorm", denorm, (2,)))
    for name, t, shape in ts:
        if not t.is_cuda:
            raise DccError("returns_compute: %s is not on a HIP device (there is no CPU path)" % name)
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != shape or t.device != rewards.device:
            raise ValueError("returns_compute: %s must be contiguous float32 %s on %s" % (name, shape, rewards.device))
    with torch.cuda.device(rewards.device):
        rc = L.dcc_returns_compute(_ptr(rewards), _ptr(value_preds), _ptr(masks), _ptr(bad_masks), _ptr(denorm), float(gamma),
                                   float(gae_lambda), int(mode), _ptr(returns), _ptr(advantages), T, C, _stream())
    _check(rc, "dcc_returns_compute")
    return returns


# ---- fused element-wise stages of the policy trunks (include/dcc_mlp.h) --------------------------------------------
def mlp_fused_supported(H, HD=0):
    """True when libdcc_hip.so has a compiled variant for hidden width H (and head width HD for the actor L1)."""
    return load_library().dcc_mlp_workspace_floats(int(H), int(HD)) > 0


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError("%s must be a contiguous float32 device tensor" % name)
    return t


def relu_ln_fwd(z, bias, gamma, beta, eps):
    """h = LayerNorm(ReLU(z + bias)); bias may be None."""
    R, H = z.shape
    h = torch.empty_like(_f32c(z, "z"))
    with torch.cuda.device(z.device):
        _check(load_library().dcc_relu_ln_fwd(_ptr(z), _ptr(bias), _ptr(_f32c(gamma, "gamma")), _ptr(_f32c(beta, "beta")), eps,
                                              _ptr(h), R, H, _stream()), "dcc_relu_ln_fwd")
    return h


def relu_ln_bwd(z, bias, gamma, dh, eps):
    """-> dz, dgamma, dbeta, dbias (column sums of dz)."""
    R, H = z.shape
    L = load_library()
    dz = torch.empty_like(_f32c(z, "z"))
    dp = torch.empty((3, H), dtype=torch.float32, device=z.device)
    ws = torch.empty(L.dcc_mlp_workspace_floats(H, 0), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _check(L.dcc_relu_ln_bwd(_ptr(z), _ptr(bias), _ptr(_f32c(gamma, "gamma")), _ptr(_f32c(dh, "dh")), eps, _ptr(dz),
                                 _ptr(dp), _ptr(ws), R, H, _stream()), "dcc_relu_ln_bwd")
    return dz, dp[0], dp[1], dp[2]


def relu_ln_head_fwd(z, bias, gamma, beta, eps, Wo, bo):
    """y [R,A] = LayerNorm(ReLU(z + bias)) Wo^T + bo (A <= 4)."""
    R, H = z.shape
    A = Wo.shape[0]
    y = torch.empty((R, A), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _check(load_library().dcc_relu_ln_head_fwd(_ptr(_f32c(z, "z")), _ptr(bias), _ptr(_f32c(gamma, "gamma")),
                                                   _ptr(_f32c(beta, "beta")), eps, _ptr(_f32c(Wo, "Wo")), _ptr(bo), _ptr(y),
                                                   R, H, A, _stream()), "dcc_relu_ln_head_fwd")
    return y


def relu_ln_head_bwd(z, bias, gamma, beta, eps, Wo, dy):
    """-> dz [R,H], dgamma [H], dbeta [H], dbias [H], dWo [A,H]."""
    R, H = z.shape
    A = Wo.shape[0]
    L = load_library()
    dz = torch.empty_like(z)
    dp = torch.empty((3 + A, H), dtype=torch.float32, device=z.device)
    ws = torch.empty(L.dcc_mlp_workspace_floats(H, 0), dtype=torch.float32, device=z.device)
    with torch.cuda.device(z.device):
        _check(L.dcc_relu_ln_head_bwd(_ptr(_f32c(z, "z")), _ptr(bias), _ptr(_f32c(gamma, "gamma")), _ptr(_f32c(beta, "beta")),
                                      eps, _ptr(_f32c(Wo, "Wo")), _ptr(_f32c(dy, "dy")), _ptr(dz), _ptr(dp), _ptr(ws), R, H, A,
                                      _stream()), "dcc_relu_ln_head_bwd")
    return dz, dp[0], dp[1], dp[2], dp[3:]


def ppo_policy_loss(mean, logstd, actions, old_logp, adv, active, clip):
    """-> dmean_raw [R,A], sums [8] (see include/dcc_mlp.h: dcc_ppo_policy_loss)."""
    R, A = mean.shape
    K = old_logp.shape[1]
    dev = mean.device
    dmean = torch.empty_like(_f32c(mean, "mean"))
    sums = torch.empty(8, dtype=torch.float32, device=dev)
    ws = torch.empty(8 * 2048, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(load_library().dcc_ppo_policy_loss(_ptr(mean), _ptr(_f32c(logstd, "logstd")), _ptr(_f32c(actions, "actions")),
                                                  _ptr(_f32c(old_logp, "old_logp")), _ptr(_f32c(adv, "adv")),
                                                  _ptr(None if active is None else _f32c(active, "active")), clip, _ptr(dmean),
                                                  _ptr(sums), _ptr(ws), R, A, K, _stream()), "dcc_ppo_policy_loss")
    return dmean, sums


def ppo_value_loss(values, value_preds, returns, active, norm, clip, delta, use_clipped, n_agents):
    """-> dvalues_raw [n], sums [2] (include/dcc_mlp.h: dcc_ppo_value_loss)."""
    n = values.numel()
    dev = values.device
    dv = torch.empty(n, dtype=torch.float32, device=dev)
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    ws = torch.empty(2 * 2048, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(load_library().dcc_ppo_value_loss(_ptr(_f32c(values, "values")), _ptr(_f32c(value_preds, "value_preds")),
                                                 _ptr(_f32c(returns, "returns")), _ptr(None if active is None else _f32c(active, "active")),
                                                 _ptr(None if norm is None else _f32c(norm, "norm")), clip, delta, int(use_clipped),
                                                 _ptr(dv), _ptr(sums), _ptr(ws), n, n_agents, _stream()), "dcc_ppo_value_loss")
    return dv, sums


def rollout_sample(mean, logstd, eps, value, actions_out, logp_out, value_preds_out, n_agents):
    """Sample + log-prob + insert into the buffer slots in one launch (include/dcc_mlp.h: dcc_rollout_sample)."""
    R, A = mean.shape
    K = logp_out.numel() // R
    for t, nm in ((mean, "mean"), (eps, "eps"), (actions_out, "actions"), (logp_out, "logp")):
        _f32c(t, nm)
    with torch.cuda.device(mean.device):
        _check(load_library().dcc_rollout_sample(_ptr(mean), _ptr(_f32c(logstd, "logstd")), _ptr(eps), _ptr(value),
                                                 _ptr(actions_out), _ptr(logp_out), _ptr(value_preds_out), R, n_agents, A, K,
                                                 _stream()), "dcc_rollout_sample")


def rollout_record(reward, done, rewards_out, masks_out, n_agents, coverage=None, rew_acc=None, cov_max=None):
    """rewards / masks of one env step into their buffer slots; with rew_acc (f64 [E]) / cov_max (f32 [E]) also the per-env
    logged statistics in the same launch (rew_acc += reward, cov_max = max(cov_max, coverage))."""
    R = rewards_out.numel()
    if reward.dtype != torch.float32 or done.dtype != torch.uint8 or not rewards_out.is_contiguous() or not masks_out.is_contiguous():
        raise ValueError("rollout_record: reward f32 [E], done u8 [E], contiguous float32 outputs")
    with torch.cuda.device(reward.device):
        if rew_acc is None and cov_max is None:
            _check(load_library().dcc_rollout_record(_ptr(reward), _ptr(done), _ptr(rewards_out), _ptr(masks_out), R, n_agents,
                                                     _stream()), "dcc_rollout_record")
            return
        if (rew_acc is not None and (rew_acc.dtype != torch.float64 or not rew_acc.is_contiguous())) or \
                (cov_max is not None and (cov_max.dtype != torch.float32 or not cov_max.is_contiguous() or coverage is None
                                          or coverage.dtype != torch.float32)):
            raise ValueError("rollout_record: rew_acc f64 [E], cov_max / coverage f32 [E]")
        _check(load_library().dcc_rollout_record_stats(_ptr(reward), _ptr(done), _ptr(coverage), _ptr(rewards_out), _ptr(masks_out),
                                                       _ptr(rew_acc), _ptr(cov_max), R, n_agents, _stream()), "dcc_rollout_record_stats")


def actor_l1_fwd(head, G, stats, Wh, s, c, gamma, beta, eps_in, eps_ln, D):
    """head [n,N,HD] / Wh [H,HD], or both None: one row per env without a per-row head term (N = 1, HD = 0)."""
    H = G.shape[1]
    if head is None:
        n, N, HD = G.shape[0], 1, 0
        head = Wh = G          # never dereferenced with HD = 0: any valid pointer
    else:
        n, N, HD = head.shape
    h = torch.empty((n * N, H), dtype=torch.float32, device=G.device)
    for t, nm in ((head, "head"), (G, "G"), (Wh, "Wh"), (s, "s"), (c, "c"), (gamma, "gamma"), (beta, "beta")):
        _f32c(t, nm)
    if stats is not None and (stats.dtype != torch.float64 or not stats.is_contiguous()):
        raise ValueError("stats must be contiguous float64")
    with torch.cuda.device(G.device):
        _check(load_library().dcc_actor_l1_fwd(_ptr(head), _ptr(G), _ptr(stats), _ptr(Wh), _ptr(s), _ptr(c), _ptr(gamma),
                                               _ptr(beta), eps_in, eps_ln, D, _ptr(h), n, N, HD, H, _stream()),
               "dcc_actor_l1_fwd")
    return h


def actor_l1_bwd(head, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, D, two_kernel=None):
    """-> dG, dWh, ds, dc, dgamma, dbeta.  two_kernel (default: for >= 65536 rows): the HIP kernel stores q = rstd_in * dz and
    dWh = q^T head is issued as a split-K batched GEMM (see include/dcc_mlp.h).  head = Wh = None (N = 1, HD = 0): dWh is
    None and dG [n, H] is the gradient of the per-env GEMM output."""
    H = G.shape[1]
    L = load_library()
    dev = G.device
    no_head = head is None
    if no_head:
        n, N, HD = G.shape[0], 1, 0
        head = Wh = G
        two_kernel = False
    else:
        n, N, HD = head.shape
    R = n * N
    if two_kernel is None:
        # BASELINE sizes (4 / 8 UAVs, H = 256): the one-kernel form holds Wh^T AND the dWh accumulators in registers
        # (actor_l1_bwd_env_k<NR, true>: 2.85 ms at 4.9 M rows vs 3.97 ms for q + GEMM, and no [rows, H] scratch);
        # other shapes: the generic one-kernel form is register-starved, long batches store q and run a GEMM
        base_size = H == 256 and N in (4, 8) and HD == 4 + 2 * (N - 1)
        two_kernel = (R >= 65536 and not base_size) or HD > 40 or os.environ.get("DCC_L1_TWO_KERNEL") == "1"
    dG = torch.empty_like(G)
    vecs = torch.empty((4, H), dtype=torch.float32, device=dev)     # ds, dc, dgamma, dbeta
    ws = torch.empty(L.dcc_mlp_workspace_floats(H, max(HD, 1)), dtype=torch.float32, device=dev)
    dq = torch.empty((R, H), dtype=torch.float32, device=dev) if two_kernel else None
    dWh = None if (two_kernel or no_head) else torch.empty((H, HD), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(L.dcc_actor_l1_bwd(_ptr(head), _ptr(G), _ptr(stats), _ptr(Wh), _ptr(s), _ptr(c), _ptr(gamma),
                                  _ptr(_f32c(dh, "dh")), eps_in, eps_ln, D, _ptr(dG), _ptr(dWh), _ptr(dq), _ptr(vecs[0]),
                                  _ptr(vecs[1]), _ptr(vecs[2]), _ptr(vecs[3]), _ptr(ws), n, N, HD, H, _stream()),
               "dcc_actor_l1_bwd")
    if two_kernel:
        S = 128
        while S > 1 and R % S:
            S //= 2
        hf = head.view(R, HD)
        if S > 1:
            dWh = torch.bmm(dq.view(S, R // S, H).transpose(1, 2), hf.view(S, R // S, HD)).sum(0)
        else:
            dWh = dq.t() @ hf
    return dG, dWh, vecs[0], vecs[1], vecs[2], vecs[3]


def actor_l1_pre_fwd(pre, G, stats, s, c, gamma, beta, eps_in, eps_ln, D):
    """The first block with the per-row term ready-made: pre [n*N, H] (= head . Wh^T from a library GEMM), G [n, H]."""
    n, H = G.shape
    R = pre.shape[0]
    N = R // n
    if pre.shape != (n * N, H):
        raise ValueError("pre must be [n*N, H] for G [n, H]")
    for t, nm in ((pre, "pre"), (G, "G"), (s, "s"), (c, "c"), (gamma, "gamma"), (beta, "beta")):
        _f32c(t, nm)
    if stats is not None and (stats.dtype != torch.float64 or not stats.is_contiguous()):
        raise ValueError("stats must be contiguous float64")
    h = torch.empty((R, H), dtype=torch.float32, device=G.device)
    with torch.cuda.device(G.device):
        _check(load_library().dcc_actor_l1_pre_fwd(_ptr(pre), _ptr(G), _ptr(stats), _ptr(s), _ptr(c), _ptr(gamma), _ptr(beta),
                                                   eps_in, eps_ln, D, _ptr(h), n, N, H, _stream()), "dcc_actor_l1_pre_fwd")
    return h


def actor_l1_pre_bwd(pre, G, stats, s, c, gamma, dh, eps_in, eps_ln, D):
    """-> dpre [n*N, H], dG [n, H], ds, dc, dgamma, dbeta."""
    n, H = G.shape
    R = pre.shape[0]
    N = R // n
    L = load_library()
    dev = G.device
    dG = torch.empty_like(G)
    dpre = torch.empty((R, H), dtype=torch.float32, device=dev)
    vecs = torch.empty((4, H), dtype=torch.float32, device=dev)     # ds, dc, dgamma, dbeta
    ws = torch.empty(L.dcc_mlp_workspace_floats(H, 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _check(L.dcc_actor_l1_pre_bwd(_ptr(pre), _ptr(G), _ptr(stats), _ptr(s), _ptr(c), _ptr(gamma), _ptr(_f32c(dh, "dh")),
                                      eps_in, eps_ln, D, _ptr(dG), _ptr(dpre), _ptr(vecs[0]), _ptr(vecs[1]), _ptr(vecs[2]),
                                      _ptr(vecs[3]), _ptr(ws), n, N, H, _stream()), "dcc_actor_l1_pre_bwd")
    return dpre, dG, vecs[0], vecs[1], vecs[2], vecs[3]


# ---- fused optimizer step on flat storage (include/dcc_optim.h) ------------------------------------------------------
def grad_norm_workspace_floats(n):
    return int(load_library().dcc_grad_norm_workspace_floats(int(n)))


def grad_norm_clip(flat_grad, max_norm, out, workspace):
    """out [2] <- {||flat_grad||_2, min(1, max_norm / (norm + 1e-6))} (max_norm <= 0: coefficient 1)."""
    _f32c(flat_grad, "flat_grad"); _f32c(out, "out"); _f32c(workspace, "workspace")
    with torch.cuda.device(flat_grad.device):
        _check(load_library().dcc_grad_norm_clip(_ptr(flat_grad), flat_grad.numel(), float(max_norm), _ptr(out), _ptr(workspace),
                                                 _stream()), "dcc_grad_norm_clip")


def adam_step(param, grad, exp_avg, exp_avg_sq, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay, clip=None):
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _f32c(t, nm)
        if t.numel() != param.numel():
            raise ValueError("adam_step: %s has %d elements, param %d" % (nm, t.numel(), param.numel()))
    with torch.cuda.device(param.device):
        _check(load_library().dcc_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(step_size),
                                            float(bc2_sqrt), float(beta1), float(beta2), float(eps), float(weight_decay),
                                            _ptr(clip), _stream()), "dcc_adam_step")
