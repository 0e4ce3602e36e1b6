"""The batched coverage vec-env: what `make_env(cfg)` returns.

It is the drop-in for the reference's SubprocVecEnv / DummyVecEnv (uav_dcc_control/envs/wrappers.py:
133-261) wrapping E DCEnv instances (envs/mpe/uav_dcc.py:7-58): same attributes
(`observation_space`, `share_observation_space`, `action_space`, `n_envs`, `n_agents`) and the same
`reset() / step(actions) / close() / render()` contract, including the auto-reset-on-done semantics
of wrappers.py:104-109.  Instead of one OS process per env talking over pickled pipes, the E envs are
one HIP launch (one env per wavefront) behind the C-ABI of include/dcc_env.h.

Two surfaces:
  * numpy (drop-in):  reset() -> obs [E,N,D];  step(a [E,N,2]) -> obs, rewards [E,N,1], dones [E,N] bool, infos
  * device (native):  reset_device() / step_device(a_tensor, obs_out=...) -> torch tensors on the GPU, no sync;
    the learner lets the kernel write observations straight into the rollout buffer.
"""
import os

import numpy as np
import torch

import dcc_hip
from envs.spaces import Box

_POI_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpe", "pos_pois.npy")
EXTRA_POI_SEED = 2024   # synthetic PoIs appended when num_pois > 1000 (the data file has 1000 rows)


def load_pois(num_pois):
    """First `num_pois` rows of the reference's PoI table (scenarios/coverage.py:13-16)."""
    poi = np.load(_POI_FILE)
    if num_pois > len(poi):
        extra = np.random.RandomState(EXTRA_POI_SEED).uniform(-1, 1, (num_pois - len(poi), 2))
        poi = np.concatenate([poi, extra], 0)
    return np.ascontiguousarray(poi[:num_pois], np.float64)


class _Infos:
    """Lazy sequence of per-env info dicts (`infos[i]["coverage_rate"]`, learner.py:192) backed by one
    host array: building E Python dicts per step would dominate the step at E = 4096."""

    def __init__(self, coverage):
        self._cov = coverage

    def __len__(self):
        return len(self._cov)

    def __getitem__(self, i):
        return {"coverage_rate": float(self._cov[i])}

    def __iter__(self):
        return ({"coverage_rate": float(c)} for c in self._cov)

    def coverage_rate(self):
        return self._cov


class HipCoverageVecEnv:
    def __init__(self, n_envs, num_agents=4, num_pois=20, r_cover=0.2, r_comm=0.4, comm_r_scale=0.95,
                 comm_force_scale=0.0, max_ep_len=150, device=None, poi_xy=None, obs_dtype=np.float32,
                 env0=0, env_total=None, reuse_host_buffers=False, **consts):
        self.n_envs, self.n_agents, self.n_pois = int(n_envs), int(num_agents), int(num_pois)
        self.max_ep_len = max_ep_len
        poi = load_pois(self.n_pois) if poi_xy is None else np.asarray(poi_xy, np.float64)
        self.poi_xy = poi
        self.env = dcc_hip.HipCoverageEnv(self.n_envs, self.n_agents, self.n_pois, poi, r_cover, r_comm, comm_r_scale,
                                          comm_force_scale, device=device, **consts)
        self._render_consts = dict(r_cover=r_cover, r_comm=r_comm)
        self.device = self.env.device
        self.obs_dim = self.env.D
        self.env0, self.env_total = env0, (env_total or self.n_envs)   # position of this shard in a multi-GPU job
        self.obs_dtype = np.dtype(obs_dtype)
        N, D = self.n_agents, self.obs_dim
        inf = np.inf
        # same spaces as MultiAgentEnv / DCEnv build (environment.py:52,75; uav_dcc.py:40-43)
        self.action_space = [Box(-1.0, 1.0, (2,), np.float32) for _ in range(N)]
        self.observation_space = [Box(-inf, inf, (D,), np.float32) for _ in range(N)]
        self.share_observation_space = [Box(-inf, inf, (N * D,), np.float32) for _ in range(N)]
        self._out = self._out64 = None
        self._closed = False
        # numpy surface: results leave the device through PINNED staging buffers (a pageable `.cpu()` of the 44 MB of
        # observations of a c2 step runs at 1.7 GB/s = 27 ms; pinned: 33 GB/s = 1.3 ms).  reuse_host_buffers: return views of
        # two alternating pinned sets (valid until the step after next) instead of fresh copies (+4 ms host memcpy at c2).
        self.reuse_host_buffers = bool(reuse_host_buffers)
        self._pin, self._flip = None, 0

    # ---- device surface ---------------------------------------------------------------------------
    def reset_device(self, obs_out=None):
        return self.env.reset(obs_out)

    def step_device(self, actions, obs_out=None, out=None, extra_out=None, want_obs=True, features_out=None):
        """actions: [E,N,2] float32/float64 tensor on the device.  Returns the dict of output tensors
        (obs, reward [E], done [E] u8, connect, connect_s, coverage [E], assign [E,M]).  `obs_out` lets
        the caller name the destination of the observations (e.g. a rollout-buffer slot).  The small
        per-step tensors are REUSED by the next call (no allocation per step): consume or copy them first.
        `extra_out` adds outputs to that dict (e.g. the compact state_* slots of a rollout buffer).
        `features_out` (with want_obs=False, float32 actions): dict of feature tensors (HipCoverageEnv.feature_shapes) that the
        SAME launch fills with the policy-input features of the post-step state (dcc_env_step_features)."""
        if out is None:
            if self._out is None:
                self._out = self.env.alloc_out(obs=False)
            out = dict(self._out)
            if want_obs:     # want_obs=False: the caller consumes the compact state outputs instead of rows
                out["obs"] = obs_out if obs_out is not None else torch.empty(
                    (self.n_envs, self.n_agents, self.obs_dim), dtype=torch.float32, device=self.device)
            if extra_out:
                out.update(extra_out)
        if features_out is not None and not want_obs and actions.dtype == torch.float32:
            return self.env.step_features(actions, out, features_out)
        return self.env.step(actions, out)

    # ---- numpy surface (reference contract) ----------------------------------------------------------
    def _host(self, tensors):
        """Device tensors -> numpy through the pinned staging set of this call (one stream sync for all of them)."""
        if self._pin is None:
            E, N, D = self.n_envs, self.n_agents, self.obs_dim
            mk = lambda shape, dt: torch.empty(shape, dtype=dt, pin_memory=True)
            self._pin = [dict(obs=mk((E, N, D), torch.float32), reward64=mk((E,), torch.float64), done=mk((E,), torch.uint8),
                              coverage=mk((E,), torch.float32)) for _ in range(2)]
        pin = self._pin[self._flip]
        self._flip ^= 1
        for k, t in tensors.items():
            pin[k].copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        take = (lambda a: a) if self.reuse_host_buffers else (lambda a: a.copy())
        return {k: take(pin[k].numpy()) for k in tensors}

    def reset(self):
        obs = self._host(dict(obs=self.env.reset()))["obs"]
        return obs.astype(self.obs_dtype, copy=False)

    def step(self, actions):
        a = np.ascontiguousarray(actions)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float32)
        if a.shape != (self.n_envs, self.n_agents, 2):
            raise ValueError("actions must be [n_envs, n_agents, 2], got %s" % (a.shape,))
        if self._out64 is None:        # the env's own float64 reward (dcc_env_out.reward64), what wrappers.py:161-165 hands back
            self._out64 = torch.empty(self.n_envs, dtype=torch.float64, device=self.device)
        out = self.step_device(torch.from_numpy(a).to(self.device),   # a copy: the caller's array is never mutated
                               extra_out=dict(reward64=self._out64))
        h = self._host({k: out[k] for k in ("obs", "reward64", "done", "coverage")})
        obs = h["obs"].astype(self.obs_dtype, copy=False)
        rew = h["reward64"]
        done = h["done"].astype(bool)
        E, N = self.n_envs, self.n_agents
        rewards = np.repeat(rew[:, None, None], N, axis=1)               # [E,N,1]  (wrappers.py:165)
        dones = np.repeat(done[:, None], N, axis=1)                      # [E,N]
        return obs, rewards, dones, _Infos(h["coverage"] if not self.reuse_host_buffers else h["coverage"].copy())

    def step_async(self, actions):
        self._pending = actions

    def step_wait(self):
        a, self._pending = self._pending, None
        return self.step(a)

    RENDER_MAX_ENVS = 16

    def render(self, mode="human", size=350):
        """wrappers.py:196-201,254-259.  There is no display on a GPU node: "human" is a no-op (the reference opens a pyglet
        window), "rgb_array" returns headless frames [n, 1, size, size, 3] uint8 of the first n = min(n_envs, 16) envs
        (envs/render.py), indexed like the reference's (`frame[0][0]` = the image of env 0, learner.py:199-200)."""
        if mode == "human":
            return None
        if mode != "rgb_array":
            raise NotImplementedError("render mode %r" % (mode,))
        from envs.render import rasterize
        n = min(self.n_envs, self.RENDER_MAX_ENVS)
        st = {k: v[:n].cpu().numpy() for k, v in self.env.get_state().items()}
        c = self._render_consts
        return np.stack([rasterize(st["pos"][e], self.poi_xy, st["energy"][e], st["done"][e], c["r_cover"], self.env.m_energy,
                                   c["r_comm"], size)[None] for e in range(n)])

    def get_state(self):
        return {k: v.cpu().numpy() for k, v in self.env.get_state().items()}

    def close(self):
        if not self._closed:
            self.env.close()
            self._closed = True
