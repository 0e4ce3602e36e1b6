"""Generate tests/golden/mappo_small.npz by importing the REFERENCE MAPPO code (algos.mappo,
buffer.shared_buffer, utils.valuenorm, utils.util from /root/reference/uav_dcc_control) on CPU.
Container-only; the outputs are data.  Re-run: python tools/gen_golden_mappo.py

Contents (hidden size 32, N=4 agents, E=3 envs, T=16 steps, D=20, S=N*D=80, A=2):
  actor/..., critic/...     reference state_dicts right after construction (seed 7)
  ev_*                      a 64-row evaluate_actions batch and its outputs (values, logp, entropy, mean action)
  buf_*                     a synthetic rollout buffer (obs, actions, rewards, value_preds, masks with episode ends)
  vn0_*                     ValueNorm state before compute_returns (after two updates on synthetic returns)
  returns                   SharedReplayBuffer.compute_returns(next_value, value_normalizer)  (GAE, live branch)
  adv_norm                  the normalised advantages MAPPOTrainer.train builds
  info_*                    train_info of MAPPOTrainer.train (ppo_epoch=2, one mini-batch)
  actor2/..., critic2/...   parameters after that train() call; vn1_* ValueNorm state after it
  huber_*                   utils.util.huber_loss on a vector that includes e < -delta (one-sided quirk)
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

REF = "/root/reference/uav_dcc_control"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "mappo_small.npz")


class Box:  # container only (the reference checks __class__.__name__ == "Box" and .shape)
    def __init__(self, n):
        self.shape = (n,)


def main():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer
    from utils.util import huber_loss

    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(REF, f))))
    for k in ("actor_lr", "critic_lr", "opti_eps"):
        cfg[k] = float(cfg[k])  # PyYAML reads 5e-4 as a string
    N, E, T, D, A, H = 4, 3, 16, 20, 2, 32
    cfg.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2)
    cfg = Namespace(**cfg)
    S = N * D
    torch.manual_seed(7); np.random.seed(7)
    policy = MAPPOPolicy(cfg, Box(D), Box(S), Box(A))
    trainer = MAPPOTrainer(cfg, policy)
    out = {}
    for k, v in policy.actor.state_dict().items():
        out["actor/" + k] = v.numpy().copy()
    for k, v in policy.critic.state_dict().items():
        out["critic/" + k] = v.numpy().copy()

    rs = np.random.RandomState(11)
    ev_obs = rs.normal(0, 1, (64, D)).astype(np.float32)
    ev_sobs = rs.normal(0, 1, (64, S)).astype(np.float32)
    ev_act = rs.uniform(-1, 1, (64, A)).astype(np.float32)
    ones = np.ones((64, 1), np.float32)
    trainer.prep_rollout()
    with torch.no_grad():
        values, logp, ent = policy.evaluate_actions(ev_sobs, ev_obs, np.zeros((64, 1, H), np.float32),
                                                    np.zeros((64, 1, H), np.float32), ev_act, ones,
                                                    None, ones)
        mean_act, _ = policy.act(ev_obs, np.zeros((64, 1, H), np.float32), ones, deterministic=True)
    out.update(ev_obs=ev_obs, ev_sobs=ev_sobs, ev_act=ev_act, ev_values=values.numpy(), ev_logp=logp.numpy(),
               ev_entropy=np.array(ent.item()), ev_mean_act=mean_act.numpy())

    buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A))
    obs = rs.normal(0, 1, (T + 1, E, N, D)).astype(np.float32)
    buf.obs[:] = obs
    buf.share_obs[:] = np.repeat(obs.reshape(T + 1, E, 1, S), N, axis=2)
    buf.actions[:] = rs.uniform(-1, 1, (T, E, N, A))
    buf.action_log_probs[:] = rs.normal(-2.5, 0.3, (T, E, N, 1))          # broadcast into [.,2] like insert() does
    rew = rs.normal(-50, 30, (T, E, 1, 1)).astype(np.float32)
    buf.rewards[:] = np.repeat(rew, N, axis=2)
    vp = rs.normal(0, 1, (T + 1, E, 1, 1)).astype(np.float32)            # critic output is identical per agent
    buf.value_preds[:] = np.repeat(vp, N, axis=2)
    masks = np.ones((T + 1, E, N, 1), np.float32)
    masks[5, 0] = 0; masks[9, 1] = 0; masks[10, 1] = 0; masks[T, 2] = 0    # episode ends
    buf.masks[:] = masks
    vn = trainer.value_normalizer
    vn.update(rs.normal(-300, 120, (200, 1)).astype(np.float32))
    vn.update(rs.normal(-280, 100, (200, 1)).astype(np.float32))
    out.update(vn0_mean=vn.running_mean.numpy().copy(), vn0_mean_sq=vn.running_mean_sq.numpy().copy(),
               vn0_debias=vn.debiasing_term.numpy().copy())
    next_value = np.repeat(rs.normal(0, 1, (E, 1, 1)).astype(np.float32), N, axis=1)
    out.update(buf_obs=obs, buf_actions=buf.actions.copy(), buf_logp=buf.action_log_probs.copy(),
               buf_rewards=buf.rewards.copy(), buf_value_preds=buf.value_preds.copy(), buf_masks=masks,
               next_value=next_value)
    buf.compute_returns(next_value, vn)
    out["returns"] = buf.returns.copy()
    out["buf_value_preds_after"] = buf.value_preds.copy()
    adv = buf.returns[:-1] - vn.denormalize(buf.value_preds[:-1])
    out["adv_raw"] = adv.copy()
    out["adv_norm"] = (adv - np.nanmean(adv)) / (np.nanstd(adv) + 1e-5)

    trainer.prep_training()
    torch.manual_seed(3)
    info = trainer.train(buf, update_actor=True)
    for k, v in info.items():
        out["info_" + k] = np.array(float(v))
    for k, v in policy.actor.state_dict().items():
        out["actor2/" + k] = v.numpy().copy()
    for k, v in policy.critic.state_dict().items():
        out["critic2/" + k] = v.numpy().copy()
    out.update(vn1_mean=vn.running_mean.numpy().copy(), vn1_mean_sq=vn.running_mean_sq.numpy().copy(),
               vn1_debias=vn.debiasing_term.numpy().copy())
    e = torch.tensor([-25.0, -10.0, -3.0, 0.0, 2.0, 10.0, 10.5, 40.0])
    out.update(huber_e=e.numpy(), huber_out=huber_loss(e, 10.0).numpy())
    out["cfg_json"] = np.array(str({k: v for k, v in vars(cfg).items()}))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB; info:", {k: round(float(v), 5) for k, v in info.items()})


if __name__ == "__main__":
    main()
