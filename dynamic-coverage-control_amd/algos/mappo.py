"""MAPPO policy and trainer (reference: uav_dcc_control/algos/mappo.py:15-247), device-resident.

What is kept from the reference (same entry points, same numerics, its quirks behind flags that
default to the reference behaviour):
  * MAPPOPolicy.get_actions / get_values / evaluate_actions / act / lr_decay        (mappo.py:39-65)
  * MAPPOTrainer.cal_value_loss / ppo_update / train / prep_* / save_model / load_model
  * PPO-clip surrogate summed over the action-log-prob columns: the reference stores the [B,1]
    log-prob broadcast into an [.,2] buffer, so the surrogate is DOUBLED  ->
    `double_surrogate=True`;  one-sided Huber (Q5);  ValueNorm.update inside every ppo_update (Q11).
What is MI355X-first:
  * every tensor of the update already lives on the GPU (no per-epoch host gather of B x (D+S));
  * the centralised critic input is identical for the N agents of an env (learner.py:269-271), so
    with `dedup_critic` it is evaluated once per env and broadcast (N x fewer critic FLOPs);
  * parameters, gradients and Adam moments of each network are flat arrays (algo_utils/optim.py): clipping + Adam is
    three HIP launches per network (include/dcc_optim.h), zero_grad one memset;
  * one process per GPU: the flat gradient arrays are all-reduced over RCCL in place, BEFORE clipping (so the clip
    sees the global gradient); the critic's all-reduce is in flight while the actor's backward runs; advantage
    statistics and ValueNorm moments are all-reduced too, which makes G ranks on E/G envs each equivalent to one
    rank on E.
"""
import os
import pickle

import torch

import utils.pytorch_utils as ptu
from algos.algo_utils import fused
from algos.algo_utils.optim import FlatAdam
from algos.algo_utils.structured import invalidate_folded_weights
from algos.r_actor_critic import R_Actor, R_Critic
from utils.util import huber_loss, mse_loss, update_linear_schedule
from utils.valuenorm import ValueNorm


def _dist():
    import torch.distributed as dist
    return dist if ptu.dist_active() else None


class MAPPOPolicy:
    def __init__(self, cfg, obs_space, cent_obs_space, act_space):
        self.device = ptu.device
        self.actor_lr, self.critic_lr = cfg.actor_lr, cfg.critic_lr
        self.opti_eps, self.weight_decay = cfg.opti_eps, cfg.weight_decay
        self.obs_space, self.share_obs_space, self.act_space = obs_space, cent_obs_space, act_space
        self.actor = R_Actor(cfg, obs_space, act_space, ptu.device)
        self.critic = R_Critic(cfg, cent_obs_space, ptu.device)
        self.actor_optimizer = FlatAdam(self.actor.parameters(), lr=self.actor_lr, eps=self.opti_eps,
                                        weight_decay=self.weight_decay)
        self.critic_optimizer = FlatAdam(self.critic.parameters(), lr=self.critic_lr, eps=self.opti_eps,
                                         weight_decay=self.weight_decay)

    def enable_structured_input(self, layout):
        """Let actor and critic accept compact features (algo_utils/structured.py) in place of observation rows."""
        self.actor.obs_layout = self.critic.obs_layout = layout

    def broadcast_parameters(self, src=0):
        """Replicas start identical: rank `src`'s parameters are broadcast once (no-op single process)."""
        dist = _dist()
        if dist is None:
            return
        for p in list(self.actor.parameters()) + list(self.critic.parameters()):
            ptu.broadcast(p.data, src)

    def lr_decay(self, episode, episodes):
        update_linear_schedule(self.actor_optimizer, episode, episodes, self.actor_lr)
        update_linear_schedule(self.critic_optimizer, episode, episodes, self.critic_lr)

    def get_actions(self, cent_obs, obs, rnn_states_actor=None, rnn_states_critic=None, masks=None,
                    available_actions=None, deterministic=False):
        actions, logp, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        values, rnn_states_critic = self.critic(cent_obs, rnn_states_critic, masks)
        return values, actions, logp, rnn_states_actor, rnn_states_critic

    def get_values(self, cent_obs, rnn_states_critic=None, masks=None):
        return self.critic(cent_obs, rnn_states_critic, masks)[0]

    def evaluate_actions(self, cent_obs, obs, rnn_states_actor, rnn_states_critic, action, masks,
                         available_actions=None, active_masks=None):
        logp, ent = self.actor.evaluate_actions(obs, rnn_states_actor, action, masks, available_actions, active_masks)
        values, _ = self.critic(cent_obs, rnn_states_critic, masks)
        return values, logp, ent

    def act(self, obs, rnn_states_actor=None, masks=None, available_actions=None, deterministic=False):
        actions, _, rnn_states_actor = self.actor(obs, rnn_states_actor, masks, available_actions, deterministic)
        return actions, rnn_states_actor

    def state_dict(self):
        return {"actor": self.actor.state_dict(), "critic": self.critic.state_dict(),
                "actor_optimizer": self.actor_optimizer.state_dict(),
                "critic_optimizer": self.critic_optimizer.state_dict()}

    def load_state_dict(self, sd):
        self.actor.load_state_dict(sd["actor"]); self.critic.load_state_dict(sd["critic"])
        if "actor_optimizer" in sd:
            self.actor_optimizer.load_state_dict(sd["actor_optimizer"])
            self.critic_optimizer.load_state_dict(sd["critic_optimizer"])


def _allreduce_start(opt):
    """Start the sum of `opt`'s flat gradient array over the ranks (asynchronous; None on a single process)."""
    dist = _dist()
    if dist is None:
        return None
    return ptu.all_reduce(opt.flat_grad, async_op=True)


def _allreduce_finish(opt, work):
    """Wait for the all-reduce and turn the sum into the mean (each rank's loss is a mean over its own equally sized
    batch shard, so the mean of the per-rank gradients is the gradient of the global mean)."""
    if work is None:
        return
    work.wait()
    opt.flat_grad.div_(_dist().get_world_size())


class MAPPOTrainer:
    def __init__(self, cfg, policy, agent_id=0):
        self.tpdv = dict(dtype=torch.float32, device=ptu.device)
        self.policy = policy
        self.clip_param, self.ppo_epoch = cfg.clip_param, cfg.ppo_epoch
        self.num_mini_batch = cfg.num_mini_batch
        self.value_loss_coef, self.entropy_coef = cfg.value_loss_coef, cfg.entropy_coef
        self.max_grad_norm, self.huber_delta = cfg.max_grad_norm, cfg.huber_delta
        self._use_max_grad_norm = cfg.use_max_grad_norm
        self._use_clipped_value_loss = cfg.use_clipped_value_loss
        self._use_huber_loss = cfg.use_huber_loss
        self._use_valuenorm = cfg.use_valuenorm
        self._use_value_active_masks = cfg.use_value_active_masks
        self._use_policy_active_masks = cfg.use_policy_active_masks
        if cfg.use_popart:
            raise NotImplementedError("use_popart: disabled in the reference config and not built (the reference's PopArt.update "
                                      "raises TypeError on its first call, algo_utils/popart.py:60-63)")
        self._use_recurrent_policy = bool(cfg.use_recurrent_policy)
        self._use_naive_recurrent = bool(cfg.use_naive_recurrent_policy)
        self.data_chunk_length = int(getattr(cfg, "data_chunk_length", 10))
        # reference quirk Q4 (doubled surrogate) on by default; critic de-duplication is exact
        self.double_surrogate = bool(getattr(cfg, "double_surrogate", True))
        self.dedup_critic = bool(getattr(cfg, "dedup_critic", True)) and bool(getattr(cfg, "use_centralized_V", True))
        # compute the parameter-free part of the input LayerNorm once per train() instead of once per epoch
        self.cache_normalized_inputs = bool(getattr(cfg, "cache_normalized_inputs", True))
        # > 0: visit the batch in chunks of this many rollout steps with gradient accumulation (exact);
        # required by (and defaulted for) the compact-state rollout buffer
        self.update_chunk_steps = int(getattr(cfg, "update_chunk_steps", 0))
        self.value_normalizer = ValueNorm(1, device=ptu.device) if self._use_valuenorm else None

    # ---- losses ---------------------------------------------------------------------------------
    def cal_value_loss(self, values, value_preds_batch, return_batch, active_masks_batch, update_norm=True):
        """mappo.py:103-131 (ValueNorm.update happens here, once per ppo_update -- Q11; the chunked update calls
        it once on the whole return batch and passes update_norm=False for its chunks)."""
        value_pred_clipped = value_preds_batch + (values - value_preds_batch).clamp(-self.clip_param, self.clip_param)
        if self._use_valuenorm:
            if update_norm:
                self.value_normalizer.update(return_batch)
            target = self.value_normalizer.normalize(return_batch)
        else:
            target = return_batch
        error_clipped = target - value_pred_clipped
        error_original = target - values
        if self._use_huber_loss:
            loss_clipped = huber_loss(error_clipped, self.huber_delta)
            loss_original = huber_loss(error_original, self.huber_delta)
        else:
            loss_clipped, loss_original = mse_loss(error_clipped), mse_loss(error_original)
        value_loss = torch.max(loss_original, loss_clipped) if self._use_clipped_value_loss else loss_original
        if self._use_value_active_masks:
            return (value_loss * active_masks_batch).sum() / active_masks_batch.sum()
        return value_loss.mean()

    def _forward_losses(self, sample, prenormalized=False, update_norm=True):
        """Forward pass + the three loss terms of mappo.py:139-164 on one batch (or one chunk of it)."""
        (share_obs_batch, obs_batch, rnn_states_batch, rnn_states_critic_batch, actions_batch, value_preds_batch,
         return_batch, masks_batch, active_masks_batch, old_action_log_probs_batch, adv_targ,
         available_actions_batch) = sample[:12]
        # row mini-batches whose inputs are held per (step, env) pair (SharedReplayBuffer.minibatch_rows): which actor output /
        # which critic output belongs to each row of the mini-batch
        row_sel, pair_sel = sample[12] if len(sample) > 12 else (None, None)
        t = lambda x: ptu.to_tensor(x) if x is not None else None
        old_logp, adv_targ = t(old_action_log_probs_batch), t(adv_targ)
        value_preds_batch, return_batch, active_masks_batch = t(value_preds_batch), t(return_batch), t(active_masks_batch)
        actions_batch = t(actions_batch)
        structured_in = isinstance(obs_batch, dict)     # compact features of (step, env) states instead of rows
        if not structured_in:
            obs_batch, share_obs_batch = t(obs_batch), t(share_obs_batch)
        n_rows = actions_batch.shape[0]

        if not self.double_surrogate:      # single log-prob column: the PPO surrogate without the reference's doubling (Q4)
            old_logp = old_logp[:, :1]
        actor = self.policy.actor
        on_gpu = ptu.device.type == "cuda"
        fused_loss = on_gpu and available_actions_batch is None and fused.policy_loss_usable(actions_batch, old_logp)
        mean = actor._mean(obs_batch, prenormalized, rnn_states_batch, masks_batch, row_sel=row_sel)
        if not fused_loss:  # (fused: surrogate, entropy and their gradients in one HIP pass over [B, A], dcc_ppo_policy_loss)
            action_log_probs, dist_entropy = actor.act.evaluate_actions(
                None, actions_batch, available_actions_batch,
                active_masks=active_masks_batch if actor._use_policy_active_masks else None, mean=mean)
        values = self.policy.critic(share_obs_batch, rnn_states_critic_batch, masks_batch, prenormalized=prenormalized)[0]
        if pair_sel is not None:
            values = values.index_select(0, pair_sel)
        n_rep = n_rows // values.shape[0]
        fused_vloss = on_gpu and fused.value_loss_usable(values) and values.shape[0] * n_rep == n_rows
        if values.shape[0] != n_rows and not fused_vloss:
            values = values.unsqueeze(1).expand(-1, n_rep, -1).reshape(-1, 1)

        if fused_loss:
            policy_loss, dist_entropy, imp_weights = fused.policy_loss(
                mean, actor.act.action_out.logstd._bias.view(-1), actions_batch, old_logp, adv_targ, active_masks_batch,
                self.clip_param, self._use_policy_active_masks)
        else:
            imp_weights = torch.exp(action_log_probs - old_logp)   # [B, A] when old_logp keeps the reference's [.,2] layout
            surr1 = imp_weights * adv_targ
            surr2 = torch.clamp(imp_weights, 1.0 - self.clip_param, 1.0 + self.clip_param) * adv_targ
            surr = torch.sum(torch.min(surr1, surr2), dim=-1, keepdim=True)
            if self._use_policy_active_masks:
                policy_loss = (-surr * active_masks_batch).sum() / active_masks_batch.sum()
            else:
                policy_loss = -surr.mean()
        if fused_vloss:     # clipped Huber loss and its gradient from the per-env values in one HIP pass (dcc_ppo_value_loss)
            norm = None
            if self._use_valuenorm:
                if update_norm:
                    self.value_normalizer.update(return_batch)
                norm = self.value_normalizer.denorm_params()          # {mean, std}: normalize(x) = (x - mean) / std
            value_loss = fused.value_loss(values, value_preds_batch, return_batch, active_masks_batch, norm, self.clip_param,
                                          self.huber_delta if self._use_huber_loss else None, self._use_clipped_value_loss,
                                          self._use_value_active_masks, n_rep)
        else:
            value_loss = self.cal_value_loss(values, value_preds_batch, return_batch, active_masks_batch, update_norm)
        return policy_loss, dist_entropy, value_loss, imp_weights

    def _backward(self, policy_loss, dist_entropy, value_loss, update_actor, weight=1.0, last=True):
        """total_loss.backward() of mappo.py:166-174.  Actor and critic share no parameter, so the total is the sum of two
        independent graphs; in a multi-GPU job the critic's is run first and -- on the last (or only) chunk of the batch --
        its flat gradient starts its RCCL all-reduce while the actor's backward is still running.  Returns the pending
        all-reduces [(optimizer, work)]."""
        critic_loss = value_loss * self.value_loss_coef
        actor_loss = (policy_loss - dist_entropy * self.entropy_coef) if update_actor else None
        if weight != 1.0:
            critic_loss = critic_loss * weight
            actor_loss = actor_loss * weight if actor_loss is not None else None
        if _dist() is None:
            (critic_loss if actor_loss is None else actor_loss + critic_loss).backward()
            return []
        pending = []
        critic_loss.backward()
        if last:
            pending.append((self.policy.critic_optimizer, _allreduce_start(self.policy.critic_optimizer)))
        if actor_loss is not None:
            actor_loss.backward()
        if last:
            pending.append((self.policy.actor_optimizer, _allreduce_start(self.policy.actor_optimizer)))
        return pending

    def _optimizer_step(self, pending=()):
        """finish the all-reduces (multi-GPU) -> clip -> Adam, for both networks (mappo.py:176-185): three launches per
        network on the flat arrays (include/dcc_optim.h).  Returns the two pre-clip gradient norms as 0-d tensors."""
        for opt, work in pending:
            _allreduce_finish(opt, work)
        max_norm = self.max_grad_norm if self._use_max_grad_norm else None     # None: norm only (get_gard_norm)
        actor_grad_norm = self.policy.actor_optimizer.clip_and_step(max_norm)
        critic_grad_norm = self.policy.critic_optimizer.clip_and_step(max_norm)
        invalidate_folded_weights(self.policy.actor, self.policy.critic)
        return actor_grad_norm, critic_grad_norm

    def ppo_update(self, sample, update_actor=True, prenormalized=False):
        """One full-batch PPO step (mappo.py:133-187).  `sample` is the 12-tuple of the reference's
        feed_forward_generator; tensors may be numpy (drop-in) or device tensors (native path).
        If `share_obs_batch` has fewer rows than `obs_batch` it holds ONE row per (step, env) and the
        values are broadcast over the agents (dedup_critic)."""
        policy_loss, dist_entropy, value_loss, imp_weights = self._forward_losses(sample, prenormalized)
        self.policy.actor_optimizer.zero_grad()
        self.policy.critic_optimizer.zero_grad()
        pending = self._backward(policy_loss, dist_entropy, value_loss, update_actor)
        actor_grad_norm, critic_grad_norm = self._optimizer_step(pending)
        return value_loss, critic_grad_norm, policy_loss, dist_entropy, actor_grad_norm, imp_weights

    def ppo_update_chunked(self, buffer, advantages, update_actor=True):
        """The same full-batch PPO step with the batch visited in chunks of `update_chunk_steps` rollout steps
        and the gradients accumulated: every loss term is a mean over the batch, so the chunk losses weighted by
        their share of the rows add up to the full-batch loss (same gradient up to summation order).  Peak
        activation memory is that of one chunk, and a compact-state buffer only ever materialises one chunk of
        observations (dcc_obs_expand)."""
        T = buffer.episode_length
        step = max(1, int(self.update_chunk_steps))
        if self._use_valuenorm:   # once per update on the whole return batch, like cal_value_loss does (Q11)
            self.value_normalizer.update(buffer.returns[:-1].reshape(-1, 1))
        # every row is active on this path (the env never deactivates an agent), so a chunk's weight is its row share
        total_rows = float(T * buffer.n_rollout_threads * buffer.num_agents)
        self.policy.actor_optimizer.zero_grad()
        self.policy.critic_optimizer.zero_grad()
        acc, pending = None, []
        for t0 in range(0, T, step):
            t1 = min(T, t0 + step)
            sample = buffer.chunk_sample(advantages, t0, t1, dedup_critic=self.dedup_critic)
            w = sample[4].shape[0] / total_rows
            policy_loss, dist_entropy, value_loss, imp_weights = self._forward_losses(sample, False, update_norm=False)
            pending = self._backward(policy_loss, dist_entropy, value_loss, update_actor, weight=w, last=t1 == T)
            part = torch.stack([value_loss.detach(), policy_loss.detach(), dist_entropy.detach(),
                                imp_weights.detach().mean()]).double() * w
            acc = part if acc is None else acc + part
        actor_grad_norm, critic_grad_norm = self._optimizer_step(pending)
        return acc[0], critic_grad_norm, acc[1], acc[2], actor_grad_norm, acc[3]

    def _epoch(self, buffer, advantages, update_actor):
        """One chunked full-batch PPO epoch -> metrics [6] (float64, on the device)."""
        vl, cgn, pl, ent, agn, ratio = self.ppo_update_chunked(buffer, advantages, update_actor)
        return torch.stack([vl.double(), pl.double(), ent.double(), agn.double(), cgn.double(), ratio.double()])

    # ---- one training phase -------------------------------------------------------------------------
    def normalized_advantages(self, buffer):
        """mappo.py:190-198: adv = returns - denorm(V); (adv - mean) / (std + 1e-5) over active entries
        (population std, like np.nanstd).  Moments are all-reduced over ranks."""
        if self._use_valuenorm:
            adv = buffer.returns[:-1] - self.value_normalizer.denormalize(buffer.value_preds[:-1])
        else:
            adv = buffer.returns[:-1] - buffer.value_preds[:-1]
        act = buffer.active_masks[:-1]
        w = (act != 0).to(adv.dtype)
        stats = torch.stack([(adv * w).sum().double(), (adv.double() ** 2 * w).sum(), w.sum().double()])
        dist = _dist()
        if dist is not None:
            ptu.all_reduce(stats)
        mean = stats[0] / stats[2]
        var = (stats[1] / stats[2] - mean ** 2).clamp(min=0.0)
        return ((adv - mean.float()) / (torch.sqrt(var).float() + 1e-5))

    def train(self, buffer, update_actor=True):
        """mappo.py:189-227: ppo_epoch passes, each over num_mini_batch mini-batches (1 = whole batch)."""
        advantages = self.normalized_advantages(buffer)
        info = {"value_loss": 0.0, "policy_loss": 0.0, "dist_entropy": 0.0, "actor_grad_norm": 0.0,
                "critic_grad_norm": 0.0, "ratio": 0.0}
        acc = torch.zeros(6, dtype=torch.float64, device=ptu.device)
        recurrent = self._use_recurrent_policy or self._use_naive_recurrent
        if self.num_mini_batch > 1 and not recurrent:
            return self._train_mini_batches(buffer, advantages, update_actor, info)
        chunked = self.update_chunk_steps > 0 or getattr(buffer, "compact", False) or getattr(buffer, "structured", False)
        if chunked and recurrent:
            # chunk_sample hands out feed-forward rows (no GRU states): a recurrent policy is updated through the recurrent generators below
            raise ValueError("update_chunk_steps / state-only storage apply to feed-forward policies: a recurrent policy needs "
                             "update_chunk_steps: 0, compact_obs: false, structured_input: false")
        if chunked:
            if self.update_chunk_steps <= 0:   # rows are regenerated per chunk: keep them small; features are tiny
                self.update_chunk_steps = buffer.episode_length if getattr(buffer, "structured", False) else 10
            # No [rows, hidden] activation of a chunk may reach 2^31 elements: beyond that a torch kernel of the backward pass
            # faults on this stack (hipErrorIllegalAddress at 9.8 M rows x 256 = the full c5 shard, 32 UAV x 2048 envs x 150
            # steps; the library GEMMs themselves stay correct there, tools/big_rows_probe.py).  The chunked step is exact
            # (gradient accumulation of a mean loss), so the batch is simply visited in more pieces.
            rows_per_step = buffer.n_rollout_threads * buffer.num_agents
            width = self.policy.actor.hidden_size
            if not getattr(buffer, "structured", False):
                # dense first layers: the regenerated observation chunk [rows, D] (and the critic's input, [rows / N, N D] =
                # the same element count, or N x that when the critic is not de-duplicated) is the widest tensor of a chunk
                # (c5 shard: D = 5186, 65,536 rows per step -> the default 10 steps would be 3.4e9 elements, 13.6 GB)
                width = max(width, buffer.obs_dim, buffer.share_obs_dim if not self.dedup_critic else 0)
            max_steps = max(1, (2 ** 31 - 1) // max(1, rows_per_step * width))
            if self.update_chunk_steps > max_steps:
                self.update_chunk_steps = max_steps
            if getattr(buffer, "structured", False):
                # per-chunk state features: parameter-free, shared by all epochs, recomputed IN PLACE after a rollout
                T, step = buffer.episode_length, max(1, int(self.update_chunk_steps))
                for t0 in range(0, T, step):
                    buffer.features_rows(t0, min(T, t0 + step))
            for _ in range(self.ppo_epoch):
                acc += self._epoch(buffer, advantages, update_actor)
            acc /= self.ppo_epoch
            for k, v in zip(("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"),
                            self._global_metrics(acc).tolist()):
                info[k] = v
            return info
        cached = None
        if self.cache_normalized_inputs and self.num_mini_batch == 1 and not recurrent:
            with torch.no_grad():   # parameter-free: (x - mean) / sqrt(var + eps), once for all epochs
                full = next(buffer.feed_forward_generator(advantages, 1, dedup_critic=self.dedup_critic))
                cached = (self.policy.critic.base.normalize_input(ptu.to_tensor(full[0])),
                          self.policy.actor.base.normalize_input(ptu.to_tensor(full[1])))
        for _ in range(self.ppo_epoch):
            if self._use_recurrent_policy:        # mappo.py:204-209
                gen = buffer.recurrent_generator(advantages, self.num_mini_batch, self.data_chunk_length)
            elif self._use_naive_recurrent:
                gen = buffer.naive_recurrent_generator(advantages, self.num_mini_batch)
            else:
                gen = buffer.feed_forward_generator(advantages, self.num_mini_batch, dedup_critic=self.dedup_critic)
            for sample in gen:
                if cached is not None:
                    sample = cached + tuple(sample[2:])
                vl, cgn, pl, ent, agn, imp = self.ppo_update(sample, update_actor, prenormalized=cached is not None)
                acc += torch.stack([vl.detach().double(), pl.detach().double(), ent.detach().double(), agn.double(), cgn.double(),
                                    imp.detach().mean().double()])
        acc /= (self.ppo_epoch * self.num_mini_batch)
        vals = self._global_metrics(acc).tolist()   # the only host sync of the update
        for k, v in zip(("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"), vals):
            info[k] = v
        return info

    def _train_mini_batches(self, buffer, advantages, update_actor, info):
        """num_mini_batch > 1 with a feed-forward policy (mappo.py:203-213 over shared_buffer.py:239-279): every epoch one
        permutation of the T*E*N agent rows, cut into num_mini_batch row sets; one ppo_update -- ValueNorm update on that
        set's returns (Q11), losses as means over the set, clip, Adam -- per set.  Same code for every storage mode: the
        buffer hands out rows, regenerated rows or state features (SharedReplayBuffer.minibatch_rows).
        `self.minibatch_perms` (a list, consumed front to back) injects the permutations; otherwise the buffer draws them
        (torch.randperm: on the CPU generator -- what the reference draws from -- when running on the CPU, on the device else)."""
        acc = torch.zeros(6, dtype=torch.float64, device=ptu.device)
        n_updates = 0
        # No activation of a mini-batch may reach 2^31 elements (see train(): a torch kernel of the backward pass faults beyond
        # that on this stack).  The full-batch path splits the batch into chunks for that; a mini-batch IS the unit the user
        # asked for, so an oversized one is refused with the way out instead of being split behind the user's back.
        T, E, N = buffer.episode_length, buffer.n_rollout_threads, buffer.num_agents
        mb_rows = T * E * N // self.num_mini_batch
        H = self.policy.actor.hidden_size
        structured = getattr(buffer, "structured", False)
        D, S = buffer.obs_dim, buffer.share_obs_dim
        pairs = min(T * E, mb_rows)          # (step, env) states a mini-batch can touch
        if structured:       # the first block is evaluated for every agent of every touched state (up to all of them)
            widest = pairs * N * H
        else:                # per storage mode, the largest tensor SharedReplayBuffer.minibatch_rows / the two trunks materialise
            widest = mb_rows * max(H, D)                                          # the actor's rows
            if getattr(buffer, "decentralized", False) or not self.dedup_critic:
                widest = max(widest, mb_rows * S)                                 # one critic input per agent row
            elif getattr(buffer, "compact", False):
                widest = max(widest, pairs * max(S, N * D))                       # the regenerated rows of the touched states
            else:                                                                 # row storage, critic once per pair: ALL pairs when
                widest = max(widest, (T * E if mb_rows * 2 >= T * E else pairs) * S)   # a mini-batch touches most of them (static shapes)
        if widest >= 2 ** 31:
            need = -(-widest // (2 ** 31 - 1))
            raise ValueError("num_mini_batch = %d leaves activations of %.2e elements per mini-batch (limit 2^31 on this stack): use "
                             "num_mini_batch >= %d%s" % (self.num_mini_batch, widest, self.num_mini_batch * need,
                                                        " (more than %d, so that a mini-batch touches fewer env states)" % (2 * N) if structured else ""))
        if structured:       # the whole batch's state features: parameter-free, computed once per iteration
            buffer.features_rows(0, buffer.episode_length)
        for _ in range(self.ppo_epoch):
            perm = None
            if getattr(self, "minibatch_perms", None):        # TEST SEAM (reference fixtures): honoured only with DCC_TESTING=1
                ptu.require_testing("MAPPOTrainer.minibatch_perms")
                perm = self.minibatch_perms.pop(0)
            for sample in buffer.feed_forward_generator(advantages, self.num_mini_batch, dedup_critic=self.dedup_critic, perm=perm,
                                                        row_width=self.policy.actor.hidden_size):
                vl, cgn, pl, ent, agn, imp = self.ppo_update(sample, update_actor)
                acc += torch.stack([vl.detach().double(), pl.detach().double(), ent.detach().double(), agn.double(), cgn.double(),
                                    imp.detach().mean().double()])
                n_updates += 1
        acc /= n_updates
        for k, v in zip(("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"),
                        self._global_metrics(acc).tolist()):
            info[k] = v
        return info

    @staticmethod
    def _global_metrics(acc):
        """Losses / entropy / ratio are means over the local env shard: averaged over the (equally sized) shards they are the
        job-wide means every rank logs (the gradient norms are global already).  One 6-element all-reduce per train()."""
        dist = _dist()
        if dist is not None:
            acc = acc.clone()
            ptu.all_reduce(acc)
            acc /= dist.get_world_size()
        return acc

    def prep_training(self):
        self.policy.actor.train(); self.policy.critic.train()

    def prep_rollout(self):
        self.policy.actor.eval(); self.policy.critic.eval()

    # ---- checkpoints ----------------------------------------------------------------------------------
    def save_model(self, save_path):
        """Same location/name as the reference (mappo.py:237-240: <save_path>/agent.pkl).  The file is a
        pickled dict of state_dicts (+ ValueNorm, which the reference's pickle of the policy object omits)."""
        os.makedirs(save_path, exist_ok=True)
        sd = self.policy.state_dict()
        sd = {k: ({kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} if k in ("actor", "critic") else v)
              for k, v in sd.items()}
        if self.value_normalizer is not None:
            sd["value_normalizer"] = {k: v.cpu() for k, v in self.value_normalizer.state_dict().items()}
        with open(os.path.join(save_path, "agent.pkl"), "wb") as f:
            pickle.dump(sd, f)

    def load_model(self, load_path):
        """Reads this package's agent.pkl (a dict of state_dicts) and also a checkpoint written by the REFERENCE
        (mappo.py:236-247 pickles the whole MAPPOPolicy object): its parameters are taken over, the never-trained
        `mlp.fc_h` template is dropped, optimizer moments and ValueNorm -- which the reference's pickle does
        not carry in a usable form / at all -- start fresh."""
        with open(os.path.join(load_path, "agent.pkl"), "rb") as f:
            sd = _CheckpointUnpickler(f).load()
        if not isinstance(sd, dict):
            live = lambda m: {k: v for k, v in m.state_dict().items() if ".fc_h." not in k}
            sd = {"actor": live(sd.actor), "critic": live(sd.critic)}
        self.policy.load_state_dict(sd)
        if self.value_normalizer is not None and "value_normalizer" in sd:
            self.value_normalizer.load_state_dict(sd["value_normalizer"])
        invalidate_folded_weights(self.policy.actor, self.policy.critic)


class _Opaque(object):
    """Stand-in for classes a pickled reference policy mentions that do not exist here (gym spaces, ...): holds state,
    does nothing."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__["_state"] = state


class _CheckpointUnpickler(pickle.Unpickler):
    """Unpickler for `agent.pkl`.  The reference pickles its whole MAPPOPolicy object (mappo.py:237-240), which drags in
    classes that do not exist here and carry no parameters -- gym spaces, argparse / omegaconf containers, whatever the
    writing script defined in its __main__: those become `_Opaque`.  A class of the reference's OWN packages or of
    torch (algos.* / buffer.* / utils.* / envs.* / torch.*) that cannot be resolved is a network component this package
    does not build (the CNN base, PopArt, ...) or a genuine import bug: that error is re-raised with the class named
    instead of surfacing later as an unrelated AttributeError inside state_dict().
    Like the reference's `pickle.load`, this executes what the file says: load checkpoints you trust only."""
    _MUST_RESOLVE = ("algos", "buffer", "utils", "envs", "torch", "learner")

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError) as e:
            if not any(module == m or module.startswith(m + ".") for m in self._MUST_RESOLVE):
                return _Opaque
            raise pickle.UnpicklingError("agent.pkl refers to %s.%s, which this package does not provide (%s); only the MLP / GRU "
                                         "policies of the reference's shipped configuration can be loaded" % (module, name, e)) from e
