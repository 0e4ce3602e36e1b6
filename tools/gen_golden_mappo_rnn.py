"""Generate tests/golden/mappo_rnn_small.npz by importing the REFERENCE's recurrent MAPPO path (algos.mappo with
use_recurrent_policy / use_naive_recurrent_policy, algos.algo_utils.rnn.RNNLayer, buffer.shared_buffer's
recurrent_generator / naive_recurrent_generator) on CPU.  Container-only; outputs are data.
Re-run: python tools/gen_golden_mappo_rnn.py

Sizes: hidden 16, N=3 agents, E=4 envs, T=12 steps, D=10, S=30, A=2, recurrent_N=1, data_chunk_length=4, 2 mini-batches.
For each mode m in {chunk, naive}:
  m/actor/..., m/critic/...     reference state_dicts after construction (seed 5)
  m/ev_*                        evaluate_actions on a [T*B] sequence batch with episode ends inside (RNNLayer sequence path)
  m/step_*                      one-step get_actions outputs (RNNLayer single-step path; deterministic action = mean)
  m/gen_perm, m/gen<i>_<field>  the permutation the generator drew (torch.manual_seed(21)) and the mini-batches it yielded
  m/info_*, m/actor2/..., m/critic2/...   MAPPOTrainer.train (ppo_epoch=2, torch.manual_seed(3)) results
buf_*: the synthetic rollout buffer shared by both modes.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

REF = "/root/reference/uav_dcc_control"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mappo_rnn_small.npz")
FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks",
          "active_masks", "old_logp", "adv")


class Box:
    def __init__(self, n):
        self.shape = (n,)


def main():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer

    base = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml"):
        base.update(yaml.safe_load(open(os.path.join(REF, f))))
    for k in ("actor_lr", "critic_lr", "opti_eps"):
        base[k] = float(base[k])
    N, E, T, D, A, H, L, MB = 3, 4, 12, 10, 2, 16, 4, 2
    S = N * D
    base.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2, num_mini_batch=MB,
                data_chunk_length=L, recurrent_N=1)
    rs = np.random.RandomState(19)
    obs = rs.normal(0, 1, (T + 1, E, N, D)).astype(np.float32)
    actions = rs.uniform(-1, 1, (T, E, N, A)).astype(np.float32)
    logp = rs.normal(-2.5, 0.3, (T, E, N, 1)).astype(np.float32)
    rew = np.repeat(rs.normal(-50, 30, (T, E, 1, 1)).astype(np.float32), N, axis=2)
    vp = np.repeat(rs.normal(0, 1, (T + 1, E, 1, 1)).astype(np.float32), N, axis=2)
    masks = np.ones((T + 1, E, N, 1), np.float32)
    masks[3, 0] = 0; masks[7, 2] = 0; masks[8, 2] = 0; masks[T, 1] = 0
    rnn_a = rs.normal(0, 0.5, (T + 1, E, N, 1, H)).astype(np.float32)
    rnn_c = rs.normal(0, 0.5, (T + 1, E, N, 1, H)).astype(np.float32)
    rnn_a[masks[..., 0] == 0] = 0; rnn_c[masks[..., 0] == 0] = 0      # what Learner.insert stores for finished envs
    next_value = np.repeat(rs.normal(0, 1, (E, 1, 1)).astype(np.float32), N, axis=1)
    out = dict(buf_obs=obs, buf_actions=actions, buf_logp=logp, buf_rewards=rew, buf_value_preds=vp, buf_masks=masks,
               buf_rnn_states=rnn_a, buf_rnn_states_critic=rnn_c, next_value=next_value)

    for mode in ("chunk", "naive"):
        cfg = Namespace(**dict(base, use_recurrent_policy=(mode == "chunk"), use_naive_recurrent_policy=(mode == "naive")))
        torch.manual_seed(5); np.random.seed(5)
        policy = MAPPOPolicy(cfg, Box(D), Box(S), Box(A))
        trainer = MAPPOTrainer(cfg, policy)
        pre = mode + "/"
        for k, v in policy.actor.state_dict().items():
            out[pre + "actor/" + k] = v.numpy().copy()
        for k, v in policy.critic.state_dict().items():
            out[pre + "critic/" + k] = v.numpy().copy()

        buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A))
        buf.obs[:] = obs
        buf.share_obs[:] = np.repeat(obs.reshape(T + 1, E, 1, S), N, axis=2)
        buf.actions[:] = actions
        buf.action_log_probs[:] = logp
        buf.rewards[:] = rew
        buf.value_preds[:] = vp
        buf.masks[:] = masks
        buf.rnn_states[:] = rnn_a
        buf.rnn_states_critic[:] = rnn_c

        # sequence evaluate: B = E*N columns, T steps, time-major rows; states at t = 0
        B = E * N
        trainer.prep_rollout()
        ev_obs = obs[:-1].reshape(T, B, D).reshape(T * B, D)
        ev_sobs = buf.share_obs[:-1].reshape(T, B, S).reshape(T * B, S)
        ev_act = actions.reshape(T * B, A)
        ev_masks = masks[:-1].reshape(T * B, 1)
        with torch.no_grad():
            v, lp, ent = policy.evaluate_actions(ev_sobs, ev_obs, rnn_a[0].reshape(B, 1, H), rnn_c[0].reshape(B, 1, H), ev_act,
                                                 ev_masks, None, np.ones((T * B, 1), np.float32))
            sv, sa, slp, sra, src = policy.get_actions(buf.share_obs[2].reshape(B, S), obs[2].reshape(B, D),
                                                       rnn_a[2].reshape(B, 1, H), rnn_c[2].reshape(B, 1, H),
                                                       masks[2].reshape(B, 1), deterministic=True)
        out.update({pre + "ev_values": v.numpy(), pre + "ev_logp": lp.numpy(), pre + "ev_entropy": np.array(ent.item()),
                    pre + "step_values": sv.numpy(), pre + "step_actions": sa.numpy(), pre + "step_rnn_actor": sra.numpy(),
                    pre + "step_rnn_critic": src.numpy()})

        vn = trainer.value_normalizer
        vn.update(np.random.RandomState(23).normal(-300, 120, (200, 1)).astype(np.float32))
        out.update({pre + "vn0_mean": vn.running_mean.numpy().copy(), pre + "vn0_mean_sq": vn.running_mean_sq.numpy().copy(),
                    pre + "vn0_debias": vn.debiasing_term.numpy().copy()})
        buf.compute_returns(next_value, vn)
        out[pre + "returns"] = buf.returns.copy()
        adv = buf.returns[:-1] - vn.denormalize(buf.value_preds[:-1])
        adv = (adv - np.nanmean(adv)) / (np.nanstd(adv) + 1e-5)
        out[pre + "adv_norm"] = adv.copy()

        torch.manual_seed(21)
        n_perm = (E * T * N) // L if mode == "chunk" else E * N
        out[pre + "gen_perm"] = torch.randperm(n_perm).numpy()
        torch.manual_seed(21)
        gen = (buf.recurrent_generator(adv, MB, L) if mode == "chunk" else buf.naive_recurrent_generator(adv, MB))
        for i, sample in enumerate(gen):
            for name, arr in zip(FIELDS, sample[:11]):
                out["%sgen%d_%s" % (pre, i, name)] = np.asarray(arr).copy()

        trainer.prep_training()
        torch.manual_seed(3)
        info = trainer.train(buf, update_actor=True)
        for k, v in info.items():
            out[pre + "info_" + k] = np.array(float(v))
        for k, v in policy.actor.state_dict().items():
            out[pre + "actor2/" + k] = v.numpy().copy()
        for k, v in policy.critic.state_dict().items():
            out[pre + "critic2/" + k] = v.numpy().copy()
        print(mode, {k: round(float(v), 5) for k, v in info.items()})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
