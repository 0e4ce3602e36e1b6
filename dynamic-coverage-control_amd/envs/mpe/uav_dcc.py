"""`DCEnv`: the per-env Gym-style surface of the reference (uav_dcc_control/envs/mpe/uav_dcc.py:7-58)
as a one-env view of the batched HIP env, for code that drives a single environment
(`env = DCEnv("coverage", num_agents=4, num_pois=20); obs_n = env.reset(); env.step(action_n)`).
Returns per-agent lists like MultiAgentEnv.step (environment.py:86-110); NO auto-reset (that lives in
the vec-env wrapper in the reference, wrappers.py:104-109 -- here `make_env` returns the batched env).
"""
import numpy as np
import torch

import dcc_hip
from envs.hip_vec_env import load_pois
from envs.spaces import Box


class DCEnv:
    def __init__(self, scenario="coverage", num_agents=4, num_pois=20, max_ep_len=100, r_cover=0.2, r_comm=0.4,
                 comm_r_scale=0.95, comm_force_scale=0.0, **kwargs):
        if scenario != "coverage":
            raise NotImplementedError("scenario %r" % scenario)
        self.n_agents, self.max_ep_len = num_agents, max_ep_len
        self._poi = load_pois(num_pois)
        self._env = dcc_hip.HipCoverageEnv(1, num_agents, num_pois, self._poi, r_cover, r_comm, comm_r_scale,
                                           comm_force_scale, **kwargs)
        self._render_consts = dict(r_cover=r_cover, r_comm=r_comm)
        self._out = self._env.alloc_out(reward64=True)
        D = self._env.D
        self.action_space = [Box(-1.0, 1.0, (2,), np.float32) for _ in range(num_agents)]
        self.observation_space = [Box(-np.inf, np.inf, (D,), np.float32) for _ in range(num_agents)]
        self.share_observation_space = [Box(-np.inf, np.inf, (num_agents * D,), np.float32) for _ in range(num_agents)]
        self._needs_reset = False

    def reset(self):
        self._needs_reset = False
        return list(self._env.reset()[0].cpu().numpy().astype(np.float64))

    def step(self, actions):
        if self._needs_reset:
            raise RuntimeError("episode finished: call reset() (the batched kernel auto-resets like the vec-env)")
        a = np.ascontiguousarray(np.asarray(actions))[None]
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float32)
        out = self._env.step(torch.from_numpy(a).to(self._env.device), self._out)
        done = bool(out["done"][0])
        obs = out["obs"][0].cpu().numpy().astype(np.float64)
        if done:
            # the kernel returned the reset obs (vec-env semantics); a single env reports the terminal flag and
            # asks for an explicit reset() as MultiAgentEnv does
            self._needs_reset = True
        r = float(out["reward64"][0])      # the env's own float64 reward, like MultiAgentEnv.step hands it back (environment.py:106-108)
        info = {"n": [{} for _ in range(self.n_agents)], "coverage_rate": float(out["coverage"][0])}
        return list(obs), [r] * self.n_agents, [done] * self.n_agents, info

    def close(self):
        self._env.close()

    def render(self, mode="human", size=350):
        """uav_dcc.py:57-58 -> MultiAgentEnv.render.  No display on a GPU node: "human" is a no-op, "rgb_array" returns
        [frame] (one headless image [size, size, 3] uint8, envs/render.py) like the reference's one-viewer list."""
        if mode == "human":
            return None
        if mode != "rgb_array":
            raise NotImplementedError("render mode %r" % (mode,))
        from envs.render import rasterize
        st = {k: v[0].cpu().numpy() for k, v in self._env.get_state().items()}
        c = self._render_consts
        return [rasterize(st["pos"], self._poi, st["energy"], st["done"], c["r_cover"], self._env.m_energy, c["r_comm"], size)]
