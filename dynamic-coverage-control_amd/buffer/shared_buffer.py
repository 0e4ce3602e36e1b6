"""Rollout storage of MAPPO, resident on the GPU.

Reference: uav_dcc_control/buffer/shared_buffer.py:14-279 (numpy arrays [T+1, E, N, ...], `insert`,
`after_update`, `compute_returns` with GAE, `feed_forward_generator`).  Same attribute names and
shapes, but:
  * every array is a torch tensor on `ptu.device`; the env kernel writes observations straight into
    `obs[t+1]` (no host copy, no pickling);
  * `share_obs` is a VIEW: the centralised observation of an env is the concatenation of its agents'
    observations (learner.py:217-220,269-271), i.e. `obs.view(T+1, E, N*D)`.  The reference repeats
    it N times ([T+1,E,N,N*D], 53.5 GB at config 3); here it costs no memory and `share_obs[t]`
    is an expanded view with the reference's [E, N, N*D] shape;
  * `compute_returns` is one launch of the HIP GAE scan (include/dcc_gae.h) -- there is no CPU path;
  * `feed_forward_generator` yields device tensors; with one mini-batch (the shipped setting) it
    yields the whole batch without the randperm gather (the losses are means, order-free);
  * compact mode (`compact=True`): the observations are a pure function of the env state
    (pos, vel, PoI energy/done), which is 32N + 5M bytes per env-step instead of 4*N*D (576 B vs 10.8 KB at
    8 UAV x 64 PoI).  The buffer then stores the state the env kernel emits (dcc_env_out.state_*), keeps only the
    CURRENT step's observations for the policy forward, and `chunk_sample` regenerates the observations of a
    range of steps with dcc_obs_expand (bit-identical) for the chunked PPO update.
"""
import warnings

import numpy as np
import torch

import utils.pytorch_utils as ptu
from utils.util import get_shape_from_act_space, get_shape_from_obs_space


class _RowTensor(torch.Tensor):
    """`buffer.obs` where the rows are stored: a plain tensor whose item assignment also takes what the reference's
    learner assigns (numpy arrays, host tensors, learner.py:224-225).  Results of operations on it are ordinary tensors."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def __setitem__(self, idx, value):
        if isinstance(value, np.ndarray):
            value = torch.from_numpy(np.ascontiguousarray(value))
        if torch.is_tensor(value):
            value = value.to(device=self.device, dtype=self.dtype)
        torch.Tensor.__setitem__(self, idx, value)


class _Rows(object):
    """`buffer.obs` of a state-only (compact) buffer and `buffer.share_obs` of every buffer: the reference's attribute
    names (uav_dcc_control/learner.py:224-225 writes `share_obs[0]` / `obs[0]`, :233-234 reads `share_obs[step]` /
    `obs[step]`, :280 `share_obs[-1]`) where no [T+1, E, N, .] tensor backs them.
      * READ  `obs[t]` -> [E, N, D], `share_obs[t]` -> [E, N, S] (an expanded view, no memory), slices `[t0:t1]`, tuples
        `[t, e]`.  State-only buffer: the rows are regenerated from the stored env state (dcc_obs_expand, bit-identical
        to what the env kernel would have written) into a fresh tensor per access.
      * WRITE `obs[t] = rows` on a state-only buffer: rows handed in from outside have no state behind them, so the
        buffer switches to row storage on the spot (`SharedReplayBuffer.materialize_rows`, one warning) -- a learner
        written against the reference's buffer keeps working with the shipped `compact_obs: true`.  `share_obs[t] = x`
        is accepted and dropped when the centralised observation is the concatenation of the agents' rows (it is then a
        view of `obs`, which the same caller writes next); a distinct centralised observation is stored.
    Everything else (`shape`, `reshape`, `numpy`, ...) is forwarded to the full tensor / view where one exists."""

    def __init__(self, buf, shared):
        self._buf, self._shared = buf, shared

    @property
    def shape(self):
        b = self._buf
        return torch.Size((b.episode_length + 1, b.n_rollout_threads, b.num_agents, b.share_obs_dim if self._shared else b.obs_dim))

    def __len__(self):
        return self._buf.episode_length + 1

    def _full(self):
        b = self._buf
        if b.compact:
            raise RuntimeError("state-only buffer: index a step (buffer.%s[t]) or a slice of steps; the whole [T+1,E,N,.] array "
                               "is not resident (compact_obs: false keeps it)" % ("share_obs" if self._shared else "obs"))
        return b.share_obs_env.unsqueeze(2).expand(-1, -1, b.num_agents, -1) if self._shared else b.obs

    def _steps(self, t0, t1):
        """[t1-t0, E, N, D|S] rows of slots t0..t1-1"""
        b = self._buf
        if not b.compact:
            return self._full()[t0:t1]
        E, N, D = b.n_rollout_threads, b.num_agents, b.obs_dim
        rows = b.expand_rows(t0, t1, torch.empty((t1 - t0) * E, N, D, dtype=torch.float32, device=b.device)).view(t1 - t0, E, N, D)
        return rows.view(t1 - t0, E, 1, N * D).expand(-1, -1, N, -1) if self._shared else rows

    def __getitem__(self, idx):
        rest = ()
        if isinstance(idx, tuple):
            idx, rest = idx[0], idx[1:]
        T1 = self._buf.episode_length + 1
        if isinstance(idx, slice):
            t0, t1, st = idx.indices(T1)
            if st != 1:
                raise IndexError("step slices of buffer rows must be contiguous")
            out = self._steps(t0, max(t0, t1))
            return out[(slice(None),) + rest] if rest else out
        if torch.is_tensor(idx) and idx.dim() > 0 or isinstance(idx, (list, np.ndarray)):
            return self._full()[(idx,) + rest]
        t = int(idx)
        t = t + T1 if t < 0 else t
        if not 0 <= t < T1:
            raise IndexError("step %d out of range" % int(idx))
        out = self._steps(t, t + 1)[0]
        return out[rest] if rest else out

    def __setitem__(self, idx, value):
        b = self._buf
        if b.compact:
            b.materialize_rows()
        if not self._shared:
            b.obs[idx] = b._t(value)
        elif b._share_obs is not None:          # a centralised observation that is not the concatenation: [.., E, N, S] -> [.., E, S]
            v = b._t(value)
            b._share_obs[idx] = v[..., 0, :] if v.dim() >= 3 and v.shape[-2] == b.num_agents else v
        # else: share_obs is a view of obs; the rows written to `obs` carry it

    def __getattr__(self, name):
        if name.startswith("__"):        # copy / pickle / hasattr probes must see a plain AttributeError
            raise AttributeError(name)
        return getattr(self._full(), name)


class SharedReplayBuffer(object):
    def __init__(self, cfg, obs_space, cent_obs_space, act_space, device=None, compact=False, n_pois=None,
                 expander=None, featurizer=None):
        self.device = device if device is not None else ptu.device
        self.episode_length = cfg.max_ep_len
        self.n_rollout_threads = cfg.n_rollout_threads
        self.num_agents = cfg.num_agents
        self.gamma, self.gae_lambda = cfg.gamma, cfg.gae_lambda
        self._use_gae, self._use_valuenorm = bool(cfg.use_gae), cfg.use_valuenorm
        self._use_proper_time_limits = bool(cfg.use_proper_time_limits)
        if cfg.use_popart:
            raise NotImplementedError("use_popart: disabled in the reference config and not built (the reference's PopArt.update "
                                      "raises TypeError on its first call, algo_utils/popart.py:60-63)")
        T, E, N = self.episode_length, self.n_rollout_threads, self.num_agents
        D = get_shape_from_obs_space(obs_space)[0]
        S = get_shape_from_obs_space(cent_obs_space)[0]
        A = get_shape_from_act_space(act_space)
        self.obs_dim, self.share_obs_dim, self.act_dim = D, S, A
        # use_centralized_V: false -- the "shared" observation IS the agent's own row (learner.py:43-46,221-222,272-273): one
        # critic input per agent row, `share_obs` is `obs`
        self.decentralized = not bool(getattr(cfg, "use_centralized_V", True))
        if self.decentralized and (S != D or compact or featurizer is not None):
            raise ValueError("use_centralized_V: false needs the observation space as cent_obs_space and row storage "
                             "(structured_input: false, compact_obs: false)")
        self._shared_is_view = (S == N * D) and not self.decentralized
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=self.device)
        self.compact = bool(compact)
        # structured input (algos/algo_utils/structured.py): the policy consumes compact features of the state
        # (dcc_obs_features) in the rollout and in the update.  With compact=False the observation rows are still
        # stored (buffer.obs / buffer.share_obs stay available, as in the reference); with compact=True they are not.
        self._featurize = featurizer
        self.structured = featurizer is not None
        self.store_state = self.compact or self.structured
        self._feat_cache, self._feat_valid = {}, set()
        self._step_feats, self._step_feat_slot = None, -1
        if self.store_state:
            if not self._shared_is_view or n_pois is None or (self.compact and expander is None):
                raise ValueError("state-storing buffer needs share_obs == concat(obs), n_pois and (compact) an expander "
                                 "(HipCoverageEnv.expand_obs)")
            self._expand = expander
            self.state_pos = torch.zeros(T + 1, E, N, 2, dtype=torch.float64, device=self.device)
            self.state_vel = torch.zeros(T + 1, E, N, 2, dtype=torch.float64, device=self.device)
            self.state_energy = z(T + 1, E, n_pois)
            self.state_done = torch.zeros(T + 1, E, n_pois, dtype=torch.uint8, device=self.device)
        if self.compact:
            self.obs = _Rows(self, shared=False)   # lazy: rows regenerated from state on read; a row WRITE materialises storage
            self.obs_cur = z(E, N, D)         # observations of the newest slot only
            self._cur_slot = -1
            self._chunk_obs = None
        else:
            self.obs = z(T + 1, E, N, D).as_subclass(_RowTensor)
        self._share_obs = None if (self._shared_is_view or self.compact or self.decentralized) else z(T + 1, E, S)
        self.value_preds = z(T + 1, E, N, 1)
        self.returns = z(T + 1, E, N, 1)
        self.advantages_raw = z(T, E, N, 1)
        self.actions = z(T, E, N, A)
        # the reference allocates [.., act_shape] for the log-probs and broadcasts the [.,1] value into
        # it (shared_buffer.py:61-62,94); kept because the PPO surrogate sums over that axis (Q4)
        self.action_log_probs = z(T, E, N, A)
        self.rewards = z(T, E, N, 1)
        self.masks = torch.ones(T + 1, E, N, 1, dtype=torch.float32, device=self.device)
        self.bad_masks = torch.ones_like(self.masks)
        self.active_masks = torch.ones_like(self.masks)
        # GRU states of the recurrent policy variants (shared_buffer.py:42-45); feed-forward policies (the shipped
        # setting) get zero-width placeholders that keep `buffer.rnn_states[t]` indexable
        self.recurrent = bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy)
        if self.recurrent and self.store_state:
            raise ValueError("recurrent policies read observation rows: use structured_input: false, compact_obs: false")
        self.rnn_states = z(T + 1, E, N, cfg.recurrent_N, cfg.algo_hidden_size if self.recurrent else 0)
        self.rnn_states_critic = torch.zeros_like(self.rnn_states)
        self.available_actions = None
        self.minibatch_unique_pairs = False     # True: row mini-batches always work on the unique touched (step, env) pairs (tests)
        self.step = 0

    # ---- slots the env kernel writes into / the policy reads from ----------------------------------------
    def obs_slot(self, t):
        """Destination of the observations of slot t ([E,N,D]).  Compact: the single current-step scratch."""
        if self.compact:
            self._cur_slot = t
            return self.obs_cur
        return self.obs[t]

    def state_slot(self, t):
        """dcc_env_out.state_* destinations of slot t (compact / structured mode), else {}."""
        if not self.store_state:
            return {}
        return dict(state_pos=self.state_pos[t], state_vel=self.state_vel[t], state_energy=self.state_energy[t],
                    state_done=self.state_done[t])

    def set_state_slot(self, t, state):
        """Copy an env state dict (HipCoverageEnv.get_state(): pos, vel, energy, done) into slot t."""
        if self.store_state:
            self.state_pos[t].copy_(state["pos"]); self.state_vel[t].copy_(state["vel"])
            self.state_energy[t].copy_(state["energy"]); self.state_done[t].copy_(state["done"])

    def obs_at(self, t):
        """[E,N,D] observations of slot t for the policy forward (compact: only the newest slot is resident)."""
        if self.compact:
            if t != self._cur_slot:
                raise RuntimeError("compact buffer holds the observations of slot %d only, asked for %d"
                                   % (self._cur_slot, t))
            return self.obs_cur
        return self.obs[t]

    def share_obs_env_at(self, t):
        """[E, S] centralised observation of slot t."""
        if self.compact:
            return self.obs_at(t).view(self.n_rollout_threads, -1)
        return self.share_obs_env[t]

    def features_at(self, t):
        """Compact policy-input features of slot t (structured mode): dict(head [E,N,HD], poi_feat [E,2M], stats, cstats) plus
        the per-env GEMM inputs xa / xc built from them (algo_utils/structured.py: env_gemm_inputs).  When the env launch that
        produced slot t also produced its features (feature_slot / dcc_env_step_features) those are returned as they are."""
        if self._step_feats is not None and self._step_feat_slot == t:
            return self._step_feats[t & 1]
        return self._with_gemm_inputs(self._featurize(self.state_pos[t], self.state_vel[t], self.state_energy[t], self.state_done[t]))

    def feature_slot(self, t, alloc):
        """Destination for the features of slot t written by the env launch itself: two alternating sets (the policy reads the
        set of slot t while the env step fills the set of slot t + 1).  `alloc()` builds one set (HipCoverageEnv.alloc_features)."""
        if self._step_feats is None:
            self._step_feats = [alloc(), alloc()]
        self._step_feat_slot = t
        return self._step_feats[t & 1]

    @staticmethod
    def _with_gemm_inputs(f):
        """dcc_obs_features_x writes xa / xc itself; a featurizer that does not (CPU tests) gets them derived here."""
        if f.get("xa") is None or f.get("xc") is None:
            from algos.algo_utils.structured import env_gemm_inputs
            f["xa"], f["xc"] = env_gemm_inputs(f, False), env_gemm_inputs(f, True)
        return f

    def features_rows(self, t0, t1):
        """Features of slots t0..t1-1 flattened over (step, env).  They do not depend on the parameters, so all PPO
        epochs of an iteration share them; the buffers are persistent (recomputed in place after every rollout: no
        allocation per iteration)."""
        key = (t0, t1)
        f = self._feat_cache.get(key)
        if f is None or key not in self._feat_valid:
            E, N = self.n_rollout_threads, self.num_agents
            n = (t1 - t0) * E
            f = self._featurize(self.state_pos[t0:t1].reshape(n, N, 2), self.state_vel[t0:t1].reshape(n, N, 2),
                                self.state_energy[t0:t1].reshape(n, -1), self.state_done[t0:t1].reshape(n, -1), f)
            self._feat_cache[key] = self._with_gemm_inputs(f)
            self._feat_valid.add(key)
        return f

    def invalidate_features(self):
        """Mark the cached per-chunk features stale (call whenever the state slots are about to be rewritten)."""
        self._feat_valid.clear()
        self._step_feat_slot = -1

    def expand_rows(self, t0, t1, out):
        """Observation rows of slots t0..t1-1 regenerated from the stored state into `out` [(t1-t0)*E, N, D]."""
        N = self.num_agents
        n = (t1 - t0) * self.n_rollout_threads
        self._expand(self.state_pos[t0:t1].reshape(n, N, 2), self.state_vel[t0:t1].reshape(n, N, 2),
                     self.state_energy[t0:t1].reshape(n, -1), self.state_done[t0:t1].reshape(n, -1), out)
        return out

    def obs_rows(self, t0, t1):
        """[(t1-t0), E, N, D] observations of slots t0..t1-1; compact: regenerated from state into a reused chunk."""
        if not self.compact:
            return self.obs[t0:t1]
        E, N, D = self.n_rollout_threads, self.num_agents, self.obs_dim
        n = (t1 - t0) * E
        if self._chunk_obs is None or self._chunk_obs.shape[0] < n:
            self._chunk_obs = torch.empty(n, N, D, dtype=torch.float32, device=self.device)
        return self.expand_rows(t0, t1, self._chunk_obs[:n]).view(t1 - t0, E, N, D)

    def materialize_rows(self):
        """State-only -> row storage, in place: allocate `obs [T+1,E,N,D]`, fill it from the stored state, and from now on
        behave like `compact_obs: false` with dense policy inputs (the rows written from outside have no env state behind
        them, so neither the state slots nor the features derived from them can stand for those rows).  Triggered by
        the first row WRITE through the reference's buffer API (`obs[t] = rows`, `insert(share_obs, obs, ...)`)."""
        if not self.compact:
            return
        warnings.warn("SharedReplayBuffer: observation rows were written into a state-only buffer (compact_obs: true); switching "
                      "to row storage [T+1,E,N,D] -- set compact_obs: false / structured_input: false to start that way")
        T, E, N, D = self.episode_length, self.n_rollout_threads, self.num_agents, self.obs_dim
        obs = torch.zeros(T + 1, E, N, D, dtype=torch.float32, device=self.device)
        step = max(1, (1 << 28) // max(1, E * N * D))
        for t0 in range(0, T + 1, step):
            t1 = min(T + 1, t0 + step)
            self.expand_rows(t0, t1, obs[t0:t1].view((t1 - t0) * E, N, D))
        # (no copy of `obs_cur` into its slot: every slot's rows have just been regenerated from the stored state, bit-identical to
        # what the env wrote -- and `obs_cur` itself is stale whenever the steps since the last reset produced features only)
        self.obs = obs.as_subclass(_RowTensor)
        self.compact = self.structured = self.store_state = False
        self._featurize = None
        self._chunk_obs = None
        self._feat_cache, self._feat_valid = {}, set()

    def chunk_sample(self, advantages, t0, t1, dedup_critic=False):
        """The rows of steps t0..t1-1 as the reference's 12-tuple (shared_buffer.py:258-279), for the chunked
        full-batch update (MAPPOTrainer.ppo_update_chunked)."""
        E, N = self.n_rollout_threads, self.num_agents
        n = (t1 - t0) * E
        rows = lambda x: x[t0:t1].reshape(n * N, -1)
        adv = torch.as_tensor(advantages).to(self.device, torch.float32)
        if self.structured:
            f = self.features_rows(t0, t1)
            if not dedup_critic:
                raise NotImplementedError("structured input evaluates the centralised critic once per env (dedup_critic)")
            return (f, f, None, None, rows(self.actions), rows(self.value_preds), rows(self.returns),
                    rows(self.masks), rows(self.active_masks), rows(self.action_log_probs), rows(adv), None)
        obs = self.obs_rows(t0, t1)
        if self.decentralized:
            return (obs.reshape(n * N, -1), obs.reshape(n * N, -1), None, None, rows(self.actions), rows(self.value_preds),
                    rows(self.returns), rows(self.masks), rows(self.active_masks), rows(self.action_log_probs), rows(adv), None)
        so_env = obs.reshape(n, N * self.obs_dim) if (self._shared_is_view or self.compact) else self._share_obs[t0:t1].reshape(n, -1)
        so = so_env if dedup_critic else so_env.unsqueeze(1).expand(-1, N, -1).reshape(n * N, -1)
        return (so, obs.reshape(n * N, -1), None, None, rows(self.actions), rows(self.value_preds), rows(self.returns),
                rows(self.masks), rows(self.active_masks), rows(self.action_log_probs), rows(adv), None)

    # ---- centralised observation -------------------------------------------------------------------
    @property
    def share_obs_env(self):
        """[T+1, E, S]: one centralised observation per env (what the critic is fed with dedup_critic)."""
        if self.compact:
            raise RuntimeError("compact buffer: use share_obs_env_at(t) / chunk_sample()")
        if self.decentralized:
            raise RuntimeError("use_centralized_V: false -- there is no per-env centralised observation; the critic reads buffer.obs")
        if self._shared_is_view:
            T1, E, N, D = self.obs.shape
            return self.obs.view(T1, E, N * D)
        return self._share_obs

    @property
    def share_obs(self):
        """[T+1, E, N, S] with the reference's shape and no memory: reads are expanded views (state-only buffer:
        regenerated per step), writes go where `_Rows` says.  use_centralized_V: false: the observation rows themselves."""
        return self.obs if self.decentralized else _Rows(self, shared=True)

    # ---- writing -----------------------------------------------------------------------------------------
    def _t(self, x):
        return ptu.to_tensor(x) if not (torch.is_tensor(x) and x.device == self.device and x.dtype == torch.float32) else x

    def insert(self, share_obs, obs, rnn_states_actor, rnn_states_critic, actions, action_log_probs, value_preds,
               rewards, masks, bad_masks=None, active_masks=None, available_actions=None):
        """shared_buffer.py:72-105.  `obs` may be None when the env kernel already wrote obs[step+1]."""
        s = self.step
        if obs is not None and self.compact:     # rows from outside (a learner written against the reference's buffer)
            self.materialize_rows()
        if obs is not None:
            dst = self.obs_slot(s + 1)
            dst.copy_(self._t(obs).view_as(dst))
        if share_obs is not None and not self._shared_is_view and not self.decentralized:
            so = self._t(share_obs)
            self._share_obs[s + 1].copy_(so[:, 0] if so.dim() == 3 else so)
        if self.recurrent and rnn_states_actor is not None:
            self.rnn_states[s + 1].copy_(self._t(rnn_states_actor).view_as(self.rnn_states[s + 1]))
            self.rnn_states_critic[s + 1].copy_(self._t(rnn_states_critic).view_as(self.rnn_states_critic[s + 1]))
        self.actions[s].copy_(self._t(actions).view_as(self.actions[s]))
        self.action_log_probs[s].copy_(self._t(action_log_probs).view(self.n_rollout_threads, self.num_agents, -1)
                                       .expand_as(self.action_log_probs[s]))
        self.value_preds[s].copy_(self._t(value_preds).view(self.n_rollout_threads, -1, 1).expand_as(self.value_preds[s]))
        self.rewards[s].copy_(self._t(rewards).view(self.n_rollout_threads, -1, 1).expand_as(self.rewards[s]))
        self.masks[s + 1].copy_(self._t(masks).view(self.n_rollout_threads, -1, 1).expand_as(self.masks[s + 1]))
        if bad_masks is not None:
            self.bad_masks[s + 1].copy_(self._t(bad_masks))
        if active_masks is not None:
            self.active_masks[s + 1].copy_(self._t(active_masks))
        self.step = (self.step + 1) % self.episode_length

    def after_update(self):
        """shared_buffer.py:142-152: the last slot becomes slot 0."""
        if self.store_state:
            for a in (self.state_pos, self.state_vel, self.state_energy, self.state_done):
                a[0].copy_(a[-1])
        if self.compact:
            self._cur_slot = 0 if self._cur_slot == self.episode_length else self._cur_slot
        else:
            self.obs[0].copy_(self.obs[-1])
        if self._share_obs is not None:
            self._share_obs[0].copy_(self._share_obs[-1])
        if self.recurrent:
            self.rnn_states[0].copy_(self.rnn_states[-1])
            self.rnn_states_critic[0].copy_(self.rnn_states_critic[-1])
        self.masks[0].copy_(self.masks[-1])
        self.bad_masks[0].copy_(self.bad_masks[-1])
        self.active_masks[0].copy_(self.active_masks[-1])

    # ---- returns --------------------------------------------------------------------------------------------
    def compute_returns(self, next_value, value_normalizer=None):
        """shared_buffer.py:160-217 as one launch of the backward scan: GAE(gamma, lambda) on denormalised values segmented by
        `masks` (the shipped branch, :199-208 -> dcc_gae_compute), with use_proper_time_limits the `bad_masks` variants
        (:167-197) and without use_gae the plain discounted returns (:186-197, :214-217) -> dcc_returns_compute.  Also fills
        `advantages_raw` = returns - denorm(value_preds) (mappo.py:191)."""
        import dcc_hip
        T, E, N = self.episode_length, self.n_rollout_threads, self.num_agents
        nv = self._t(next_value).view(E, -1, 1)
        denorm = value_normalizer.denorm_params() if (value_normalizer is not None and self._use_valuenorm) else None
        flat = lambda x: x.view(x.shape[0], E * N)
        if self._use_gae:
            self.value_preds[-1].copy_(nv.expand_as(self.value_preds[-1]))              # :169,200
        else:
            self.returns[-1].copy_(nv.expand_as(self.returns[-1]))                      # :187,215 (value_preds[-1] is not set)
        if self._use_gae and not self._use_proper_time_limits:
            dcc_hip.gae_compute(flat(self.rewards), flat(self.value_preds), flat(self.masks), denorm, self.gamma, self.gae_lambda,
                                flat(self.returns), flat(self.advantages_raw))
            return
        mode = (dcc_hip.RETURNS_GAE if self._use_gae else 0) | (dcc_hip.RETURNS_PROPER if self._use_proper_time_limits else 0)
        dcc_hip.returns_compute(flat(self.rewards), flat(self.value_preds), flat(self.masks),
                                flat(self.bad_masks) if self._use_proper_time_limits else None, denorm, self.gamma,
                                self.gae_lambda, mode, flat(self.returns), flat(self.advantages_raw))

    # ---- sampling -----------------------------------------------------------------------------------------------
    def feed_forward_generator(self, advantages, num_mini_batch=None, mini_batch_size=None, dedup_critic=False, perm=None,
                               row_width=None):
        """shared_buffer.py:219-279.  Yields the reference's 12-tuple of [B, .] tensors (device).
        One mini-batch (the shipped setting): the whole batch in storage order, without the randperm gather (every loss is
        a mean over the batch, order-free).  More than one: the reference's mini-batches -- one permutation of the
        T*E*N agent rows (`perm` injects one), cut into num_mini_batch row sets of batch_size // num_mini_batch rows
        (`minibatch_rows`).  Works on every storage mode: row storage gathers rows, a state-only buffer gathers env states.
        dedup_critic: `share_obs_batch` carries ONE row per (step, env) pair the mini-batch touches."""
        T, E, N = self.episode_length, self.n_rollout_threads, self.num_agents
        num_mini_batch = num_mini_batch or 1
        if num_mini_batch == 1 and mini_batch_size is None:
            if self.compact:
                raise RuntimeError("compact buffer: the full batch is visited with chunk_sample() (ppo_update_chunked)")
            rows = lambda x: x.reshape(T * E * N, -1)
            adv = torch.as_tensor(advantages).to(self.device, torch.float32)
            if self.decentralized and dedup_critic:
                raise ValueError("use_centralized_V: false has one critic input per agent row: dedup_critic must be false")
            so = self.share_obs_env[:-1].reshape(T * E, -1) if dedup_critic else self.share_obs[:-1].reshape(T * E * N, -1)
            yield (so, rows(self.obs[:-1]), None, None, rows(self.actions), rows(self.value_preds[:-1]),
                   rows(self.returns[:-1]), rows(self.masks[:-1]), rows(self.active_masks[:-1]),
                   rows(self.action_log_probs), rows(adv), None)
            return
        batch_size = T * E * N
        if mini_batch_size is None:
            if batch_size < num_mini_batch:
                raise ValueError("PPO requires n_rollout_threads (%d) * max_ep_len (%d) * num_agents (%d) >= num_mini_batch (%d)"
                                 % (E, T, N, num_mini_batch))
            mini_batch_size = batch_size // num_mini_batch
        if perm is None:
            # On the CPU the permutation comes from torch's CPU generator exactly like the reference's (the same seed selects the
            # same rows: tests/test_mappo_env_golden.py).  On the GPU it is drawn on the device: a CPU randperm of the c3 batch
            # (4.9 M rows) costs ~60 ms per epoch + a 39 MB upload -- more than the epoch's kernels -- and the reference's random
            # stream cannot be followed on the device anyway (the action noise is drawn there too).
            perm = torch.randperm(batch_size, device=self.device) if self.device.type == "cuda" else torch.randperm(batch_size)
        else:
            perm = torch.as_tensor(perm).reshape(-1).long()
            # an injected sampler must not repeat rows: the structured path scatters row gradients back with index_copy_
            # (algo_utils/fused.select_rows), which keeps ONE contribution per index
            if perm.numel() and int(torch.unique(perm).numel()) != perm.numel():
                raise ValueError("feed_forward_generator: the injected row order repeats rows (sampling with replacement is not supported)")
        for i in range(num_mini_batch):
            yield self.minibatch_rows(advantages, perm[i * mini_batch_size:(i + 1) * mini_batch_size], dedup_critic, row_width)

    def features_of_pairs(self, pairs):
        """Policy-input features (dcc_obs_features) of the (step, env) states `pairs` (indices into the T*E flattening)."""
        T, E = self.episode_length, self.n_rollout_threads
        g = lambda a: a[:-1].reshape((T * E,) + tuple(a.shape[2:]))[pairs].contiguous()
        return self._with_gemm_inputs(self._featurize(g(self.state_pos), g(self.state_vel), g(self.state_energy), g(self.state_done)))

    def rows_of_pairs(self, pairs):
        """Observation rows [n, N, D] of the (step, env) states `pairs`, regenerated from the stored state (dcc_obs_expand)."""
        T, E = self.episode_length, self.n_rollout_threads
        g = lambda a: a[:-1].reshape((T * E,) + tuple(a.shape[2:]))[pairs].contiguous()
        out = torch.empty(pairs.numel(), self.num_agents, self.obs_dim, dtype=torch.float32, device=self.device)
        self._expand(g(self.state_pos), g(self.state_vel), g(self.state_energy), g(self.state_done), out)
        return out

    def minibatch_rows(self, advantages, rows, dedup_critic=False, row_width=None):
        """The reference's mini-batch for the agent rows `rows` -- UNIQUE indices into the (t, e, n) flattening of the T*E*N rows,
        what `sampler` holds at shared_buffer.py:239-240 (a slice of a permutation) -- as its 12-tuple (:258-279).
        A row (t, e, n) needs agent n's observation of state (t, e) and the centralised observation of that state.  Neither is
        gathered N times over: the (step, env) pairs the rows touch are made unique (`pairs`, sorted), and
          * row storage:      obs_batch = the rows themselves; share_obs_batch = one row per touched pair (dedup_critic) or per
                              agent row (the reference's layout);
          * state-only:       the touched states are gathered (32N + 5M bytes each) and their rows regenerated
                              (dcc_obs_expand), then as above;
          * structured input: obs_batch = share_obs_batch = the features of the touched states (dcc_obs_features).
        Which pairs are "touched": a mini-batch of at least half as many rows as there are pairs touches most of them (k mini-
        batches of N-agent envs: 1 - (1 - 1/k)^N; 99.6 % at k = 2, N = 8), so ALL T*E pairs are used -- static shapes, no
        torch.unique (a host sync + a sort per mini-batch), the per-iteration feature cache serves every mini-batch, and the
        allocator sees the same sizes every time.  Smaller mini-batches (or a batch whose [rows, row_width] activation would pass
        2^31 elements, see MAPPOTrainer.train) take the unique touched pairs.
        When a batch entry is per pair instead of per row the tuple gets a 13th element (row_sel, pair_sel): index vectors that
        pick each row's actor output out of the [pairs*N] outputs (None: obs_batch is per row already) and each row's value
        out of the [pairs] critic outputs (MAPPOTrainer._forward_losses applies them -- the row selection right after the
        actor's first block, so its 256 x 256 block and head run on the mini-batch's rows only; the gradient flows through the
        gathers)."""
        T, E, N = self.episode_length, self.n_rollout_threads, self.num_agents
        B = T * E * N
        rows = torch.as_tensor(rows).reshape(-1).to(self.device, torch.long)
        adv = torch.as_tensor(advantages).to(self.device, torch.float32)
        g = lambda x: x.reshape(B, -1)[rows]
        tail = (g(self.actions), g(self.value_preds[:-1]), g(self.returns[:-1]), g(self.masks[:-1]), g(self.active_masks[:-1]),
                g(self.action_log_probs), g(adv), None)
        pair, agent = rows // N, rows % N
        all_pairs = (rows.numel() * 2 >= T * E and B * int(row_width or 512) < 2 ** 31 and not self.minibatch_unique_pairs)
        if self.structured:
            if not dedup_critic:
                raise NotImplementedError("structured input evaluates the centralised critic once per env (dedup_critic)")
            if all_pairs:       # the whole batch's features (cached per iteration, shared by all epochs and mini-batches)
                f = self.features_rows(0, T)
                return (f, f, None, None) + tail + ((rows, pair),)
            pairs, inv = torch.unique(pair, return_inverse=True)
            f = self.features_of_pairs(pairs)
            return (f, f, None, None) + tail + ((inv * N + agent, inv),)
        if self.compact:
            pairs, inv = torch.unique(pair, return_inverse=True)
            table = self.rows_of_pairs(pairs)
            obs = table.view(pairs.numel() * N, -1)[inv * N + agent]
            so_env = table.view(pairs.numel(), -1)
            if dedup_critic:
                return (so_env, obs, None, None) + tail + ((None, inv),)
            return (so_env[inv], obs, None, None) + tail
        obs = self.obs[:-1].reshape(B, -1)[rows]
        if self.decentralized:
            if dedup_critic:
                raise ValueError("use_centralized_V: false has one critic input per agent row: dedup_critic must be false")
            return (obs, obs, None, None) + tail
        so_env = self.share_obs_env[:-1].reshape(T * E, -1)
        if dedup_critic:
            if all_pairs:       # the stored centralised rows as they are (a view: nothing gathered)
                return (so_env, obs, None, None) + tail + ((None, pair),)
            pairs, inv = torch.unique(pair, return_inverse=True)
            return (so_env[pairs], obs, None, None) + tail + ((None, inv),)
        return (so_env[pair], obs, None, None) + tail

    # ---- recurrent generators (shared_buffer.py:281-487) ----------------------------------------------------------------
    # The reference builds every mini-batch with Python loops over chunks / env columns and np.stack on host arrays.
    # Here a mini-batch is ONE advanced-indexing gather per array on the device: the reference's row order is turned
    # into (t, e, n) index vectors, and the centralised observation is gathered from the per-env view, so only the
    # mini-batch (never the whole [T,E,N,S] array) is materialised.  The permutation is drawn like the reference draws
    # it (torch.randperm on the CPU generator), so the same seed selects the same chunks.
    def _rows(self, t, e, n, advantages):
        """The 12-tuple of the rows addressed by the index vectors (t, e, n); rnn states are filled in by the callers."""
        adv = torch.as_tensor(advantages).to(self.device, torch.float32)
        so = self.obs[t, e, n] if self.decentralized else self.share_obs_env[t, e]
        return [so, self.obs[t, e, n], None, None, self.actions[t, e, n], self.value_preds[t, e, n], self.returns[t, e, n],
                self.masks[t, e, n], self.active_masks[t, e, n], self.action_log_probs[t, e, n], adv[t, e, n], None]

    def naive_recurrent_generator(self, advantages, num_mini_batch, perm=None):
        """shared_buffer.py:281-370: whole-episode sequences; a mini-batch is a set of (env, agent) columns.  Rows are
        time-major [T * cols, .]; rnn states are the columns' states at t = 0."""
        T, E, N = self.episode_length, self.n_rollout_threads, self.num_agents
        if not self.recurrent or self.compact:
            raise RuntimeError("naive_recurrent_generator needs a recurrent, row-storing buffer")
        batch_size = E * N
        if batch_size < num_mini_batch:
            raise ValueError("PPO requires n_rollout_threads (%d) * num_agents (%d) >= num_mini_batch (%d)" % (E, N, num_mini_batch))
        per = batch_size // num_mini_batch
        perm = (torch.randperm(batch_size) if perm is None else torch.as_tensor(perm)).to(self.device)
        tt = torch.arange(T, device=self.device)
        for start in range(0, batch_size, per):
            cols = perm[start:start + per]
            e, n = cols // N, cols % N
            t_i = tt.view(T, 1).expand(T, cols.numel()).reshape(-1)
            e_i = e.view(1, -1).expand(T, -1).reshape(-1)
            n_i = n.view(1, -1).expand(T, -1).reshape(-1)
            sample = self._rows(t_i, e_i, n_i, advantages)
            sample[2], sample[3] = self.rnn_states[0, e, n], self.rnn_states_critic[0, e, n]
            yield tuple(sample)

    def recurrent_generator(self, advantages, num_mini_batch, data_chunk_length, perm=None):
        """shared_buffer.py:372-487: the (env, agent) sequences are laid end to end (env-major, then agent, then time),
        cut into chunks of data_chunk_length rows; a mini-batch is a random set of chunks, rows chunk-time-major
        [L * chunks, .]; rnn states are those stored at each chunk's first row."""
        T, E, N, L = self.episode_length, self.n_rollout_threads, self.num_agents, int(data_chunk_length)
        if not self.recurrent or self.compact:
            raise RuntimeError("recurrent_generator needs a recurrent, row-storing buffer")
        data_chunks = (E * T * N) // L
        mb = data_chunks // num_mini_batch
        perm = (torch.randperm(data_chunks) if perm is None else torch.as_tensor(perm)).to(self.device)
        off = torch.arange(L, device=self.device)
        for i in range(num_mini_batch):
            idx = perm[i * mb:(i + 1) * mb]
            r = (idx.view(1, -1) * L + off.view(L, 1)).reshape(-1)        # [L * mb] rows in the (e, n, t) flattening
            e, n, t = r // (N * T), (r // T) % N, r % T
            sample = self._rows(t, e, n, advantages)
            r0 = idx * L
            e0, n0, t0 = r0 // (N * T), (r0 // T) % N, r0 % T
            sample[2], sample[3] = self.rnn_states[t0, e0, n0], self.rnn_states_critic[t0, e0, n0]
            yield tuple(sample)
