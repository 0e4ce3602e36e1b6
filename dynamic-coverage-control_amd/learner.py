"""Learner: warmup -> [collect -> env.step -> insert] x T -> compute (GAE) -> rl_update, per iteration.

Same entry point and method names as the reference orchestrator (uav_dcc_control/learner.py:21-321:
`Learner(cfg).train()`, `rollout`, `warmup`, `collect`, `insert`, `compute`, `rl_update`, `log`,
`save_model`, `load_model`), so the reference's train.py drives it unchanged.  The data path is
different: nothing crosses the host between env step, policy forward, rollout storage, GAE and the
PPO update --

  obs[t] (device) --actor/critic GEMMs--> actions --HIP env kernel--> obs[t+1] written in place
  rewards/dones/values --> buffer slots --HIP GAE scan--> returns --> 15 full-batch PPO epochs

and in a multi-GPU job each rank owns n_rollout_threads / world_size envs, with one RCCL all-reduce
of the gradients per ppo_update (algos/mappo.py).
"""
import copy
import datetime
import json
import os
import time
from argparse import Namespace

import numpy as np
import torch

import utils.pytorch_utils as ptu
from buffer.shared_buffer import SharedReplayBuffer
from envs.make_env import make_env
from utils import util as utl


def _to_namespace(cfg):
    if isinstance(cfg, Namespace):
        return copy.deepcopy(cfg)
    try:
        from omegaconf import OmegaConf
        if OmegaConf.is_config(cfg):
            return Namespace(**OmegaConf.to_container(cfg, resolve=True))
    except ImportError:
        pass
    return Namespace(**dict(cfg))


class Learner:
    def __init__(self, cfg):
        self.cfg = _to_namespace(cfg)
        for k, v in (("double_surrogate", True), ("dedup_critic", True), ("use_hip_graph", False)):
            if not hasattr(self.cfg, k):
                setattr(self.cfg, k, v)
        self.recurrent = bool(self.cfg.use_recurrent_policy or self.cfg.use_naive_recurrent_policy)
        if self.recurrent:
            # the recurrent variants (off in the shipped config) consume observation rows step by step: the rollout buffer
            # keeps rows and per-(env, agent) GRU states, and the critic runs once per agent row like in the reference
            self.cfg.structured_input = self.cfg.compact_obs = False
        self.use_centralized_V = bool(self.cfg.use_centralized_V)
        if not self.use_centralized_V:
            # use_centralized_V: false (learner.py:43-46,218-222,269-273): the critic reads each agent's OWN observation row,
            # so there is one value per agent row and nothing to share between the agents of an env -- row storage, dense
            # first layers, the critic on every row, like the reference
            self.cfg.structured_input = self.cfg.compact_obs = self.cfg.dedup_critic = False
        self.rank, self.world = ptu.init_distributed() if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or ptu.single_rank_group()) else (0, 1)
        self.dist_on = self.world > 1 or ptu.single_rank_group()
        utl.seed(self.cfg.seed + self.rank)
        # measured-fastest library GEMM per listed shape instead of the library heuristic's pick (lookup only)
        self.tuned_gemms = ptu.use_tuned_gemms() if getattr(self.cfg, "tuned_gemms", True) else 0
        if (getattr(self.cfg, "tuned_gemms", True) and self.tuned_gemms == 0 and ptu.device.type == "cuda" and self.rank == 0
                and os.environ.get("DCC_TUNED_GEMMS", "1") != "0" and "PYTORCH_TUNABLEOP_ENABLED" not in os.environ):
            import warnings
            warnings.warn("tuned_gemms: 0 entries of config/gemm_tunings_gfx950.csv match this installation (torch / ROCm / rocBLAS / "
                          "hipBLASLt versions or GPU arch differ from the file's validators): the library heuristic picks the GEMM "
                          "kernels (c3: about 6 % slower); tools/tune_gemms.sh regenerates the table")

        # 1. env (global n_rollout_threads is sharded over ranks inside make_env)
        self.train_envs = make_env(cfg=copy.deepcopy(self.cfg))
        self.n_agents = self.cfg.num_agents
        self.max_ep_len = self.cfg.max_ep_len
        self.obs_dim_n = [self.train_envs.observation_space[i].shape[0] for i in range(self.n_agents)]
        self.action_dim_n = [self.train_envs.action_space[i].shape[0] for i in range(self.n_agents)]
        self.cfg.action_dim_n, self.cfg.obs_dim_n = self.action_dim_n, self.obs_dim_n
        if self.rank == 0:
            print("initial train envs: %s, done (%d envs on this GPU, %d GPUs)"
                  % (self.cfg.save_name, self.train_envs.n_envs, self.world))

        # 2. policy / trainer
        self.share_observation_space = (self.train_envs.share_observation_space[0] if self.use_centralized_V
                                        else self.train_envs.observation_space[0])          # learner.py:43-46
        from algos.mappo import MAPPOPolicy, MAPPOTrainer
        self.policy = MAPPOPolicy(self.cfg, self.train_envs.observation_space[0], self.share_observation_space,
                                  self.train_envs.action_space[0])
        if getattr(self.cfg, "structured_input", False):
            if not self.cfg.dedup_critic:
                raise ValueError("structured_input needs dedup_critic: true")
            from algos.algo_utils.structured import ObsLayout
            env = self.train_envs
            self.policy.enable_structured_input(ObsLayout(env.n_agents, env.n_pois, env.poi_xy, env.env.m_energy))
        self.policy.broadcast_parameters(0)
        self.trainer = MAPPOTrainer(cfg=self.cfg, policy=self.policy)

        # 3. buffers (sized for the LOCAL env shard)
        self.rl_buffer = self._make_buffer(self.train_envs)
        self.test_envs = self.test_buffer = None
        if self.cfg.n_eval_rollout_threads > 0:
            tcfg = copy.deepcopy(self.cfg)
            tcfg.n_rollout_threads = max(self.world, self.cfg.n_eval_rollout_threads // self.world * self.world)
            self.test_envs = make_env(tcfg)
            self.test_buffer = self._make_buffer(self.test_envs)

        # render rollouts (reference learner.py:80-91,148-149,195-210): headless frames -> an animated GIF per render_interval.
        # Only built when GIFs are asked for (save_gifs): without a display the frames have no other consumer.
        self.render_envs = self.render_buffer = None
        self.save_gifs = bool(getattr(self.cfg, "save_gifs", False))
        if self.save_gifs and int(getattr(self.cfg, "n_render_rollout_threads", 0)) > 0:
            rcfg = copy.deepcopy(self.cfg)
            rcfg.n_rollout_threads = self.world * int(self.cfg.n_render_rollout_threads)
            self.render_envs = make_env(rcfg)
            self.render_buffer = self._make_buffer(self.render_envs)
        self.render_interval = int(getattr(self.cfg, "render_interval", 10 ** 9))

        # 4. loop parameters
        self.use_linear_lr_decay = self.cfg.use_linear_lr_decay
        self.n_iters = self.cfg.n_iters
        self.eval_interval, self.log_interval = self.cfg.eval_interval, self.cfg.log_interval
        self.is_save_model, self.save_interval = self.cfg.save_model, self.cfg.save_interval
        self.start_iter, self.cur_iter = 1, 0
        self.resume_partial = False      # a checkpoint of another job shape was loaded: replicated state only (load_checkpoint)
        self.total_env_steps = 0
        if getattr(self.cfg, "load_model", False):
            self.load_model(self.cfg.load_model_path)
        self.expt_name = datetime.datetime.now().strftime("%m%d_%H%M_") + "sd{}".format(self.cfg.seed)
        self.output_path = str(os.path.join(self.cfg.main_save_path, self.cfg.save_name, self.expt_name))
        if self.is_save_model and self.rank == 0:
            self.cfg.output_path = self.output_path
            os.makedirs(self.output_path, exist_ok=True)
            with open(os.path.join(self.output_path, "config.json"), "w") as f:
                json.dump({k: v for k, v in vars(self.cfg).items()}, f, indent=4, default=str)
        self._start_time = self._check_time = time.time()
        # cfg.use_hip_graph: after one eager pass each (buffer, envs) pair is captured into a hipGraph and
        # replayed (parameters are updated in place, so the graph always sees the current policy; the
        # sampling RNG is graph-safe).  The ~40 small kernels of a step are launch-bound when issued from
        # Python: config 3 rollout 0.13 s eager -> 0.06 s replayed, bit-identical results.
        # The captured body holds NO reduction: the logged statistics are accumulated element-wise per env inside the
        # graph (reward sum, running max of the coverage rate) and reduced over the envs after the replay, outside capture
        # (multi-block torch reductions inside a replayed graph are not reliable on this stack: tools/graph_reduce_probe.py).
        self._graphs = {}
        self._step_features = bool(getattr(self.cfg, "step_features", True)) and ptu.device.type == "cuda"
        self.use_hip_graph = bool(self.cfg.use_hip_graph) and ptu.device.type == "cuda" and not self.recurrent

    def _make_buffer(self, envs):
        bcfg = copy.deepcopy(self.cfg)
        bcfg.n_rollout_threads = envs.n_envs
        structured = bool(getattr(self.cfg, "structured_input", False))
        compact = bool(getattr(self.cfg, "compact_obs", False))
        return SharedReplayBuffer(bcfg, envs.observation_space[0],
                                  envs.share_observation_space[0] if self.use_centralized_V else envs.observation_space[0], envs.action_space[0],
                                  compact=compact, n_pois=envs.n_pois, expander=envs.env.expand_obs if compact else None,
                                  featurizer=envs.env.obs_features if structured else None)

    # ---- training loop (learner.py:132-175) ---------------------------------------------------------
    def train(self):
        self.warmup(self.rl_buffer, self.train_envs)
        for iter_ in range(self.start_iter, self.n_iters + 1):
            self.cur_iter = iter_
            if self.use_linear_lr_decay:
                self.trainer.policy.lr_decay(iter_, self.n_iters)
            rollout_info = self.rollout(self.rl_buffer, self.train_envs)
            rl_train_info = self.rl_update()
            test_rollout_info = {}
            if self.test_envs is not None and iter_ % self.eval_interval == 0:
                test_rollout_info = self.rollout(self.test_buffer, self.test_envs)
            if self.render_envs is not None and iter_ % self.render_interval == 0:
                self.rollout(self.render_buffer, self.render_envs, is_render=True, iter_=iter_)
            if iter_ % self.log_interval == 0:
                self.log(iter_=iter_, rollout_info=rollout_info, rl_train_info=rl_train_info,
                         test_rollout_info=test_rollout_info)
            if self.is_save_model and iter_ % self.save_interval == 0:      # collective: every rank contributes its RNG / env shard
                save_path = os.path.join(self.output_path, "models_%d.pt" % iter_)
                self.save_model(save_path)
                if self.rank == 0:
                    print("model saved in %s" % save_path)
        self.train_envs.close()
        if self.test_envs is not None:
            self.test_envs.close()
        if self.render_envs is not None:
            self.render_envs.close()

    # ---- rollout (learner.py:178-214) -------------------------------------------------------------------
    @torch.no_grad()
    def _rollout_body(self, r_buffer, r_envs, frames=None):
        """warmup + T x (collect -> env step -> insert) + compute, all asynchronous on the current stream.
        frames: a list that receives one headless frame of env 0 per step (render rollouts: learner.py:195-200)."""
        self.warmup(r_buffer, r_envs)
        rew_acc = torch.zeros(r_envs.n_envs, dtype=torch.float64, device=ptu.device)    # per env: element-wise only
        cov_max = torch.zeros(r_envs.n_envs, dtype=torch.float32, device=ptu.device)
        fused_glue = self._fused_glue_ok(r_buffer)
        for cur_step in range(self.max_ep_len):
            rnn_a = rnn_c = None
            if fused_glue:      # sample + log-prob + buffer insert in one launch (dcc_rollout_sample)
                actions = self.collect_into(cur_step, r_buffer)
            elif self.recurrent:
                values, actions, action_log_probs, rnn_a, rnn_c = self.collect(cur_step, r_buffer)
            else:
                values, actions, action_log_probs = self.collect(cur_step, r_buffer)
            # rows are written unless the policy reads features AND the buffer does not keep rows
            want_rows = not (r_buffer.structured and r_buffer.compact)
            # structured input without rows: the env launch also derives the features of the state it leaves (one launch
            # instead of two, the state never re-read: include/dcc_env.h dcc_env_step_features)
            feats = (r_buffer.feature_slot(cur_step + 1, r_envs.env.alloc_features)
                     if (not want_rows and self._step_features and actions.dtype == torch.float32) else None)
            out = r_envs.step_device(actions, obs_out=r_buffer.obs_slot(cur_step + 1) if want_rows else None,
                                     extra_out=r_buffer.state_slot(cur_step + 1), want_obs=want_rows, features_out=feats)
            if fused_glue:      # rewards / masks of the env step into their buffer slots (dcc_rollout_record)
                import dcc_hip
                dcc_hip.rollout_record(out["reward"], out["done"], r_buffer.rewards[cur_step], r_buffer.masks[cur_step + 1],
                                       self.n_agents, coverage=out["coverage"], rew_acc=rew_acc, cov_max=cov_max)   # + the logged statistics
                r_buffer.step = (cur_step + 1) % r_buffer.episode_length
                if frames is not None:
                    frames.append(r_envs.render("rgb_array")[0][0])
                continue
            self.insert((out, values, actions, action_log_probs, rnn_a, rnn_c), r_buffer)
            if frames is not None:
                frames.append(r_envs.render("rgb_array")[0][0])
            rew_acc += out["reward"]
            cov_max = torch.maximum(cov_max, out["coverage"])
        self.compute(r_buffer)
        return rew_acc, cov_max

    @torch.no_grad()
    def rollout(self, r_buffer, r_envs, is_render=False, iter_=0):
        key = id(r_buffer)
        r_buffer.invalidate_features()     # host-side cache: must also be dropped when the rollout is a graph replay
        if r_buffer.structured:            # parameter-derived inference tensors the (captured) rollout reads
            from algos.algo_utils.structured import refresh_folded_weights
            refresh_folded_weights(self.policy.actor, self.policy.critic)
        if is_render:      # eager, one frame per step; the GIF lands where the reference puts it (learner.py:204-210)
            frames = []
            stats = self._rollout_body(r_buffer, r_envs, frames)
            if self.save_gifs and self.is_save_model and self.rank == 0 and frames:
                from envs.render import save_gif
                save_gif(frames, os.path.join(self.output_path, "models_%d.gif" % iter_), float(getattr(self.cfg, "ifi", 0.1)))
        elif self.use_hip_graph and key in self._graphs:
            graph, stats = self._graphs[key]
            graph.replay()
        else:
            stats = self._rollout_body(r_buffer, r_envs)
            if self.use_hip_graph and key not in self._graphs:
                try:
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    # thread_local: activity of OTHER threads (the RCCL watchdog polling its events in a multi-GPU job) must
                    # not invalidate the capture; the captured body itself issues no collective
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        gstats = self._rollout_body(r_buffer, r_envs)
                    self._graphs[key] = (graph, gstats)
                except Exception as e:  # capture is an optimisation only: keep running eagerly
                    if self.rank == 0:
                        print("hipGraph capture of the rollout failed (%s); continuing eagerly" % e)
                    self.use_hip_graph = False
                    torch.cuda.synchronize()
        if r_envs is self.train_envs:      # the logged throughput counts training env steps only (not eval rollouts)
            self.total_env_steps += self.max_ep_len * r_envs.n_envs * self.world
        rew_acc, cov_max = stats           # reduced over the envs HERE, outside any captured graph
        stats = torch.stack([rew_acc.mean(), cov_max.double().mean()])
        if self.dist_on:
            ptu.all_reduce(stats)
            stats /= self.world
        r, c = stats.tolist()
        return {"reward": r, "coverage_rate": c}

    def warmup(self, r_buffer, r_envs):
        """reset every env; obs -> slot 0 (learner.py:216-225; share_obs is a view of obs here)."""
        r_envs.reset_device(r_buffer.obs_slot(0))
        if r_buffer.store_state:
            r_buffer.set_state_slot(0, r_envs.env.get_state())
        r_buffer.step = 0          # masks[0] is left as after_update set it (the last slot's), like the reference

    @torch.no_grad()
    def collect(self, cur_step, r_buffer):
        """policy forward on the E*N agent rows, critic on the E env rows (learner.py:227-252)."""
        self.trainer.prep_rollout()
        E, N = r_buffer.n_rollout_threads, self.n_agents
        if r_buffer.structured:     # compact features of the current state: no observation rows anywhere
            feats = r_buffer.features_at(cur_step)
            actions, logp, _ = self.policy.actor(feats)
            values = self.policy.critic(feats)[0].view(E, 1, 1).expand(E, N, 1)
            return values, actions.view(E, N, -1).contiguous(), logp.view(E, N, 1)
        obs = r_buffer.obs_at(cur_step).view(E * N, -1)
        if self.recurrent:          # learner.py:231-252 with the GRU states and masks of this step, one critic row per agent
            masks = r_buffer.masks[cur_step].reshape(E * N, 1)
            actions, logp, rnn_a = self.policy.actor(obs, r_buffer.rnn_states[cur_step].reshape(E * N, *r_buffer.rnn_states.shape[3:]), masks)
            so = (r_buffer.share_obs_env_at(cur_step).unsqueeze(1).expand(E, N, -1)     # a view per agent, no [T+1,E,N,S] accessor detour
                  if self.use_centralized_V else obs)
            values, rnn_c = self.policy.critic(so.reshape(E * N, -1),
                                               r_buffer.rnn_states_critic[cur_step].reshape(E * N, *r_buffer.rnn_states.shape[3:]), masks)
            return (values.view(E, N, 1), actions.view(E, N, -1).contiguous(), logp.view(E, N, 1),
                    rnn_a.reshape(E, N, *rnn_a.shape[1:]), rnn_c.reshape(E, N, *rnn_c.shape[1:]))
        actions, logp, _ = self.policy.actor(obs)
        if not self.use_centralized_V:      # one value per agent row from its own observation (learner.py:221-222)
            values = self.policy.critic(obs)[0].view(E, N, 1)
        elif self.trainer.dedup_critic:
            values = self.policy.critic(r_buffer.share_obs_env_at(cur_step))[0].view(E, 1, 1).expand(E, N, 1)
        else:
            so = r_buffer.share_obs_env_at(cur_step).unsqueeze(1).expand(E, N, -1)
            values = self.policy.critic(so.reshape(E * N, -1))[0].view(E, N, 1)
        return values, actions.view(E, N, -1).contiguous(), logp.view(E, N, 1)

    def _fused_glue_ok(self, r_buffer):
        from algos.algo_utils import fused
        return (ptu.device.type == "cuda" and fused.ENABLED and self.trainer.dedup_critic
                and r_buffer.act_dim <= fused.HEAD_MAX_OUT and not self.recurrent)

    @torch.no_grad()
    def collect_into(self, cur_step, r_buffer):
        """collect() + the policy-side half of insert() with the element-wise work in ONE HIP launch: action mean and
        value from the networks, then a = mean + std * eps, log pi(a), and the writes into the buffer's action /
        log-prob / value slots of this step (include/dcc_mlp.h: dcc_rollout_sample).  Returns the action slot."""
        import dcc_hip
        from algos.algo_utils.distributions import standard_normal
        self.trainer.prep_rollout()
        E, N = r_buffer.n_rollout_threads, self.n_agents
        if r_buffer.structured:
            x_actor = x_critic = r_buffer.features_at(cur_step)
        else:
            x_actor = r_buffer.obs_at(cur_step).view(E * N, -1)
            x_critic = r_buffer.share_obs_env_at(cur_step)
        mean = self.policy.actor._mean(x_actor)
        value = self.policy.critic(x_critic)[0]
        eps = standard_normal(mean.shape, mean.dtype, mean.device)
        logstd = self.policy.actor.act.action_out.logstd._bias.view(-1)
        dcc_hip.rollout_sample(mean, logstd, eps, value.view(E), r_buffer.actions[cur_step], r_buffer.action_log_probs[cur_step],
                               r_buffer.value_preds[cur_step], N)
        return r_buffer.actions[cur_step]

    def insert(self, data, r_buffer):
        """masks = 0 where the env finished (learner.py:254-276); obs[t+1] is already in place."""
        out, values, actions, action_log_probs, rnn_a, rnn_c = data
        E, N = r_buffer.n_rollout_threads, self.n_agents
        masks = (1.0 - out["done"].to(torch.float32)).view(E, 1, 1).expand(E, N, 1)
        rewards = out["reward"].view(E, 1, 1).expand(E, N, 1)
        if rnn_a is not None:      # GRU states of finished envs restart from zero (learner.py:258-265)
            keep = masks.reshape(E, N, 1, 1)
            rnn_a, rnn_c = rnn_a * keep, rnn_c * keep
        r_buffer.insert(None, None, rnn_a, rnn_c, actions, action_log_probs, values, rewards, masks)

    @torch.no_grad()
    def compute(self, r_buffer):
        """bootstrap value + GAE (learner.py:278-287 -> HIP scan)."""
        self.trainer.prep_rollout()
        E, N = r_buffer.n_rollout_threads, self.n_agents
        last = r_buffer.episode_length
        if self.recurrent:          # learner.py:278-287
            next_values = self.policy.critic(r_buffer.share_obs[last].reshape(E * N, -1),
                                             r_buffer.rnn_states_critic[last].reshape(E * N, *r_buffer.rnn_states.shape[3:]),
                                             r_buffer.masks[last].reshape(E * N, 1))[0].view(E, N, 1)
            r_buffer.compute_returns(next_values, self.trainer.value_normalizer)
            return
        if not self.use_centralized_V:
            next_values = self.policy.critic(r_buffer.obs_at(last).view(E * N, -1))[0].view(E, N, 1)
            r_buffer.compute_returns(next_values, self.trainer.value_normalizer)
            return
        cent = r_buffer.features_at(last) if r_buffer.structured else r_buffer.share_obs_env_at(last)
        next_values = self.policy.critic(cent)[0].view(E, 1, 1).expand(E, N, 1)
        r_buffer.compute_returns(next_values, self.trainer.value_normalizer)

    # ---- update (learner.py:292-300) ---------------------------------------------------------------------
    def rl_update(self):
        self.trainer.prep_training()
        info = self.trainer.train(buffer=self.rl_buffer, update_actor=True)
        self.rl_buffer.after_update()
        return info

    def log(self, iter_, **kwargs):
        if self.rank != 0:
            return
        now = time.time()
        print("")
        print("******** iter: %d, iter_time: %.2fs, total_time: %.2fs, agent-env-steps/s (train rollouts + updates): %.0f"
              % (iter_, now - self._check_time, now - self._start_time,
                 self.total_env_steps * self.n_agents / max(now - self._start_time, 1e-9)))
        for key, value in kwargs.items():
            print("%s" % key + "".join([", %s: %.4f" % (k, v) for k, v in value.items()]))
        if self.resume_partial:
            print("note: resumed from a checkpoint of another job shape -- per-rank RNG streams / env states were not restored, "
                  "this is not a bit-exact continuation (resume_strict: true refuses such a file)")
            self.resume_partial = False
        self._check_time = now

    def save_model(self, save_path):
        """<save_path>/agent.pkl (the reference's file name, parameters only) + <save_path>/resume.pt (full state)."""
        if self.rank == 0:
            self.trainer.save_model(save_path)
        self.save_checkpoint(os.path.join(save_path, "resume.pt"))

    # ---- faithful resume ------------------------------------------------------------
    def _rank_state(self):
        """What differs between the ranks of a multi-GPU job: the RNG streams (seeded seed + rank) and the env shard."""
        b = self.rl_buffer      # slot 0 of the rollout buffer is what after_update carried over from the last rollout (masks, GRU states:
        return {"rng_torch": torch.get_rng_state(), "rng_numpy": np.random.get_state(),       # shared_buffer.py:142-152)
                "rng_cuda": torch.cuda.get_rng_state(ptu.device) if ptu.device.type == "cuda" else None,
                "env_state": {k: v.cpu() for k, v in self.train_envs.env.get_state().items()},
                "slot0": {"masks": b.masks[0].cpu(), "rnn_states": b.rnn_states[0].cpu(), "rnn_states_critic": b.rnn_states_critic[0].cpu()}}

    def save_checkpoint(self, path):
        """Everything needed to continue a run bit-for-bit: parameters, both Adam states, ValueNorm, the
        iteration counter, and PER RANK the RNG streams, the env-shard state and slot 0 of the rollout buffer (gathered to rank 0, which writes the
        file; COLLECTIVE in a multi-GPU job: every rank must call it).  (The reference pickles the policy object only:
        ValueNorm, iteration and RNG are lost, uav_dcc_control/algos/mappo.py:237-247.)"""
        ranks = [self._rank_state()]
        if self.dist_on:
            import torch.distributed as dist
            gathered = [None] * self.world if self.rank == 0 else None
            dist.gather_object(ranks[0], gathered, dst=0)
            ranks = gathered
        if self.rank != 0:
            return
        cpu = lambda d: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in d.items()}
        ck = {"format": 2, "iter": self.cur_iter, "total_env_steps": self.total_env_steps, "world_size": self.world,
              "actor": cpu(self.policy.actor.state_dict()), "critic": cpu(self.policy.critic.state_dict()),
              "actor_optimizer": self.policy.actor_optimizer.state_dict(),
              "critic_optimizer": self.policy.critic_optimizer.state_dict(),
              "value_normalizer": cpu(self.trainer.value_normalizer.state_dict()) if self.trainer.value_normalizer is not None else None,
              "ranks": ranks}
        torch.save(ck, path)

    def load_checkpoint(self, path, strict_world=None):
        """Every rank reads the file and takes ITS OWN RNG streams and env shard (the replicated parts -- parameters,
        optimizer moments, ValueNorm, counters -- are identical on all ranks by construction).
        Job shape = (number of ranks, envs per rank).  Default rule (strict_world=None, cfg key `resume_strict` unset): a
        checkpoint that still has iterations to run (ck.iter < n_iters) is a run being CONTINUED and must come from a job of
        the same shape -- anything else raises, naming both shapes and the escape hatch; a finished one (ck.iter >= n_iters:
        evaluation, parameter hand-over) loads leniently.  `resume_strict: false` (strict_world=False) always loads leniently:
        everything that does not depend on the shape is restored (e.g. to evaluate on one GPU a model trained on eight, or to
        fine-tune with fewer envs), only the per-rank RNG streams and env-shard states are skipped -- the run then continues
        from this process's own seed and from reset envs (every rollout starts with a reset anyway, Q9) -- with a warning.
        `resume_strict: true` always insists on the bit-exact continuation."""
        ck = torch.load(path, map_location="cpu", weights_only=False)
        if ck.get("format", 1) == 1:        # single-rank layout of the first format
            ck["ranks"] = [{k: ck[k] for k in ("rng_torch", "rng_numpy", "rng_cuda", "env_state")}]
            ck["world_size"] = 1
        if strict_world is None:
            # default: a run that CONTINUES TRAINING from the file wants the bit-exact continuation and refuses a job of another
            # shape; evaluation / parameter hand-over (`resume_strict: false`, or n_iters already reached) takes the lenient path
            strict_world = getattr(self.cfg, "resume_strict", None)
            if strict_world is None:
                strict_world = int(ck["iter"]) < int(self.n_iters)
            strict_world = bool(strict_world)
        same_world = (ck["world_size"] == self.world and self.rank < len(ck["ranks"]) and
                      ck["ranks"][self.rank]["env_state"]["pos"].shape[0] == self.train_envs.n_envs)
        if not same_world and strict_world:
            raise ValueError("checkpoint %s (iteration %d of %d) was written by %d rank(s) with %d envs each; this job has %d rank(s) "
                             "with %d envs each.  The env shards and RNG streams are per rank, so a bit-exact continuation needs "
                             "the same job shape; set `resume_strict: false` to load parameters, optimizer moments, ValueNorm and "
                             "counters only (RNG streams and env states start fresh)"
                             % (path, int(ck["iter"]), int(self.n_iters), ck["world_size"],
                                ck["ranks"][0]["env_state"]["pos"].shape[0], self.world, self.train_envs.n_envs))
        self.policy.actor.load_state_dict(ck["actor"]); self.policy.critic.load_state_dict(ck["critic"])
        self.policy.actor_optimizer.load_state_dict(ck["actor_optimizer"])
        self.policy.critic_optimizer.load_state_dict(ck["critic_optimizer"])
        if ck["value_normalizer"] is not None and self.trainer.value_normalizer is not None:
            self.trainer.value_normalizer.load_state_dict(ck["value_normalizer"])
        if same_world:
            mine = ck["ranks"][self.rank]
            torch.set_rng_state(mine["rng_torch"]); np.random.set_state(mine["rng_numpy"])
            if mine["rng_cuda"] is not None and ptu.device.type == "cuda":
                torch.cuda.set_rng_state(mine["rng_cuda"], ptu.device)
            self.train_envs.env.set_state(**mine["env_state"])
            for k, v in mine.get("slot0", {}).items():         # (absent from checkpoints written before round 5)
                getattr(self.rl_buffer, k)[0].copy_(v.to(ptu.device))
        else:
            import warnings
            self.resume_partial = True        # recorded in the iteration log of every rank
            warnings.warn("checkpoint %s was written by %d rank(s) with %d envs each; this job has %d rank(s) with %d: parameters, "
                          "optimizer moments, ValueNorm and counters are restored, the per-rank RNG streams and env states are not"
                          % (path, ck["world_size"], ck["ranks"][0]["env_state"]["pos"].shape[0], self.world, self.train_envs.n_envs))
        self.cur_iter, self.start_iter = ck["iter"], ck["iter"] + 1
        self.total_env_steps = ck["total_env_steps"]
        from algos.algo_utils.structured import invalidate_folded_weights
        invalidate_folded_weights(self.policy.actor, self.policy.critic)

    # ---- headless evaluation ---------------------------------------------------------
    @torch.no_grad()
    def evaluate(self, envs=None, steps=None, deterministic=False, dump_path=None, noise_scale=1.0):
        """Roll the current policy without learning and report the metrics of the reference's README curves
        (coverage rate, steps needed to cover every PoI).  `dump_path` gets the trajectory as an .npz -- the headless
        stand-in for the pyglet viewer: per step the ACTIONS fed to the env and what they led to (post-step, post-auto-reset
        positions / velocities / PoI energies / PoI done flags, reward, env-done, coverage, connect, connect_s), plus the PoI
        table and the env constants, i.e. everything needed to replay the file through an independent implementation of the
        env (tests/test_learner_hip.py replays it through the oracle).
        Actions are SAMPLED by default, like the reference's test / render rollouts (learner.py:143-149 run the same
        `collect` as training); deterministic=True plays the distribution's mean instead.  The two differ a lot for this
        task: the policy trained for 1500 iterations on the shipped scenario covers every PoI in 44 steps when sampled
        and stalls at half coverage on its mean (profiles/r02/training_run_1500_iters_and_shard_mappo.txt): all UAVs start at
        the origin with identical observations, so identical means never separate them.  noise_scale tempers the sampling,
        a = mean + noise_scale * std * eps (1.0 = the policy's own distribution, 0.0 = its mean): how much of the
        steps-to-cover figure is exploration noise rather than policy quality (profiles/r03/training_run_*.txt)."""
        envs = envs if envs is not None else (self.test_envs if self.test_envs is not None else self.train_envs)
        T = steps or self.max_ep_len
        E, N = envs.n_envs, self.n_agents
        self.trainer.prep_rollout()
        obs = envs.reset_device()
        rec = {k: [] for k in ("actions", "pos", "vel", "energy", "poi_done", "reward", "done", "coverage", "connect", "connect_s")}
        first_done = torch.full((E,), -1, dtype=torch.int32, device=ptu.device)
        cov_max = torch.zeros(E, device=ptu.device)
        rnn = masks = None
        if self.recurrent:
            rnn = torch.zeros(E * N, self.cfg.recurrent_N, self.cfg.algo_hidden_size, device=ptu.device)
            masks = torch.ones(E * N, 1, device=ptu.device)
        for t in range(T):
            if deterministic or noise_scale == 1.0:
                actions, _, rnn = self.policy.actor(obs.view(E * N, -1), rnn, masks, deterministic=deterministic)
            else:
                mean, rnn = self.policy.actor._mean(obs.view(E * N, -1), rnn_states=rnn, masks=masks, want_states=True)
                std = self.policy.actor.act.action_out.logstd._bias.view(1, -1).exp()
                actions = mean + float(noise_scale) * std * torch.randn_like(mean)
            actions = actions.view(E, N, -1).contiguous()
            out = envs.step_device(actions)
            obs = out["obs"]
            if self.recurrent:
                masks = (1.0 - out["done"].float()).view(E, 1).expand(E, N).reshape(E * N, 1)
            cov_max = torch.maximum(cov_max, out["coverage"])
            full = (out["coverage"] >= 1.0) & (first_done < 0)
            first_done = torch.where(full, torch.full_like(first_done, t + 1), first_done)
            if dump_path is not None:
                st = envs.env.get_state()
                rec["actions"].append(actions.cpu())
                rec["pos"].append(st["pos"].cpu()); rec["vel"].append(st["vel"].cpu())
                rec["energy"].append(st["energy"].cpu()); rec["poi_done"].append(st["done"].cpu())
                for k in ("reward", "done", "coverage", "connect", "connect_s"):
                    rec[k].append(out[k].cpu())
        solved = first_done > 0
        res = {"coverage_rate": float(cov_max.mean()), "solved_fraction": float(solved.float().mean()),
               "steps_to_cover": float(first_done[solved].float().mean()) if bool(solved.any()) else float("nan")}
        if dump_path is not None:
            c = self.cfg
            np.savez_compressed(dump_path, poi=envs.env.poi, r_cover=c.r_cover, r_comm=c.r_comm, comm_r_scale=c.comm_r_scale,
                                comm_force_scale=c.comm_force_scale, **{k: torch.stack(v).numpy() for k, v in rec.items()})
        return res

    def load_model(self, load_path):
        """cfg.load_model / cfg.load_model_path (expt.yaml, like the reference): a directory written by save_model.  When
        it holds resume.pt the run CONTINUES from it (iteration counter, lr schedule, optimizer moments, ValueNorm, RNG,
        env state); with agent.pkl alone -- also one written by the reference -- only the parameters are taken."""
        resume = os.path.join(load_path, "resume.pt")
        if os.path.exists(resume):
            self.load_checkpoint(resume)
        else:
            self.trainer.load_model(load_path)
