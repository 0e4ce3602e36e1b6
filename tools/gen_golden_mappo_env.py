"""Generate tests/golden/mappo_env_small.npz: the REFERENCE MAPPO update (algos.mappo / buffer.shared_buffer / utils.valuenorm
imported from /root/reference/uav_dcc_control) on a rollout of the REFERENCE env (tools/ref_harness.py), i.e. with real
observation rows -- the fixture the structured-input path (first layers from env-state features) is checked against directly.
Container-only; the outputs are data.  Re-run: python tools/gen_golden_mappo_env.py [small|n8m64|n12m24|small_mb2|n8m64_mb3|n8m64_h256]
(`*_mb<k>`: the same rollouts with num_mini_batch = k -- the reference's row mini-batches; the permutations it drew are stored)
Second case `n8m64` (tests/golden/mappo_env_n8m64.npz): the BASELINE c2/c3 shape, 8 UAV x 64 PoI, E=2, T=31, hidden 32.
`n8m64_h256`: the n8m64 rollout with the SHIPPED width, algo_hidden_size 256 (config/algo_config/mappo.yaml) -- the kernel
instantiations and tuned GEMM picks BASELINE c3 runs.  The ~1 M initial parameters are stored in full, the post-update ones as
sampled snapshots (tests/_sampling.py: 4,096 elements + max|delta| + ||delta||_2 per tensor); `fwd_*` holds the reference
networks' own forward on the stored rollout (values, log-probs / entropy of the stored actions) before the update.
Third case `n12m24`: 12 UAVs (more than 8: the learner's first block forms head . Wh^T with a library GEMM), 24 PoI, E=2, T=20.

4 UAV x 20 PoI (shipped world constants), E=3 envs, T=34 steps (env 1 flies east and finishes at step 30: one episode end + auto-reset), hidden 32, ppo_epoch 2.  Contents:
  poi [M,2]; state_pos/state_vel [T+1,E,N,2] f64, state_energy [T+1,E,M] f32, state_done [T+1,E,M] u8: the env state each
  stored observation was built from (slot 0 = reset; after a finished episode the reset state, wrappers.py:229-232)
  obs [T+1,E,N,D] f32 as SharedReplayBuffer stores it; actions (uniform, fed to the env), action_log_probs, value_preds
  (synthetic), rewards / masks from the env; vn0_* ValueNorm state; next_value; returns; adv_norm; info_*;
  actor/ critic/ parameters before, actor2/ critic2/ after MAPPOTrainer.train; vn1_*.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from _sampling import snapshot  # noqa: E402
from ref_harness import make_reference_env  # noqa: E402  (installs the gym stub, puts the reference on sys.path)

REF = "/root/reference/uav_dcc_control"
CASES = {"small": (4, 20, 3, 34, 32), "n8m64": (8, 64, 2, 31, 32), "n12m24": (12, 24, 2, 20, 32),     # N, M, E, T, H
         # [, num_mini_batch]: the reference's feed_forward_generator with more than one mini-batch (shared_buffer.py:239-279):
         # per epoch one torch.randperm over the T*E*N agent rows, cut into num_mini_batch row sets, one ppo_update (and one
         # ValueNorm update, Q11) per set.  The permutations the run drew are stored as perm<epoch>.
         "small_mb2": (4, 20, 3, 34, 32, 2), "n8m64_mb3": (8, 64, 2, 31, 32, 3),
         "n8m64_h256": (8, 64, 2, 32, 256)}


class Box:
    def __init__(self, n):
        self.shape = (n,)


def state_of(world):
    pos = np.array([a.state.p_pos for a in world.agents], np.float64)
    vel = np.array([a.state.p_vel for a in world.agents], np.float64)
    en = np.array([l.energy for l in world.landmarks], np.float32)
    dn = np.array([l.done for l in world.landmarks], np.uint8)
    return pos, vel, en, dn


def main(case="small"):
    OUT = os.path.join(HERE, "..", "tests", "golden", "mappo_env_%s.npz" % case)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer

    N, M, E, T, H = CASES[case][:5]
    MB = CASES[case][5] if len(CASES[case]) > 5 else 1
    A = 2
    D = 4 + 2 * (N - 1) + 5 * M
    S = N * D
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(REF, f))))
    for k in ("actor_lr", "critic_lr", "opti_eps"):
        cfg[k] = float(cfg[k])
    cfg.update(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2, num_mini_batch=MB)
    cfg = Namespace(**cfg)
    torch.manual_seed(17); np.random.seed(17)
    policy = MAPPOPolicy(cfg, Box(D), Box(S), Box(A))
    trainer = MAPPOTrainer(cfg, policy)
    out = {}
    for k, v in policy.actor.state_dict().items():
        out["actor/" + k] = v.numpy().copy()
    for k, v in policy.critic.state_dict().items():
        out["critic/" + k] = v.numpy().copy()

    rs = np.random.RandomState(23)
    envs = [make_reference_env(N, M, 0.2, 0.4, 0.9, 0.0, None) for _ in range(E)]
    out["poi"] = np.array(envs[0][2].pos_pois[:M], np.float64)
    obs = np.zeros((T + 1, E, N, D), np.float32)
    st = dict(state_pos=np.zeros((T + 1, E, N, 2)), state_vel=np.zeros((T + 1, E, N, 2)),
              state_energy=np.zeros((T + 1, E, M), np.float32), state_done=np.zeros((T + 1, E, M), np.uint8))
    rewards = np.zeros((T, E, N, 1), np.float32)
    masks = np.ones((T + 1, E, N, 1), np.float32)
    actions = np.zeros((T, E, N, A), np.float32)

    def record(t, e, ob, world):
        obs[t, e] = np.array(ob, np.float64).astype(np.float32)
        for k, v in zip(("state_pos", "state_vel", "state_energy", "state_done"), state_of(world)):
            st[k][t, e] = v

    for e, (env, world, sc) in enumerate(envs):
        record(0, e, env.reset(), world)
    # env 1 flies east at full speed from step 3 on: leaves the arena (|x| > 1.5) within the rollout -> done, auto-reset
    for t in range(T):
        for e, (env, world, sc) in enumerate(envs):
            a = rs.uniform(-1, 1, (N, A))
            if e == 1 and t >= 0:
                a[:, 0] = 1.0; a[:, 1] = 0.0
            a = a.astype(np.float32)
            actions[t, e] = a
            ob, rew, dn, info = env.step(a.copy())
            rewards[t, e, :, 0] = rew[0]
            if np.all(dn):
                ob = env.reset()
                masks[t + 1, e] = 0.0
            record(t + 1, e, ob, world)
    assert masks.min() == 0.0 or T < 30, "no episode end in the rollout"

    buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A))
    buf.obs[:] = obs
    buf.share_obs[:] = np.repeat(obs.reshape(T + 1, E, 1, S), N, axis=2)
    buf.actions[:] = actions
    buf.action_log_probs[:] = rs.normal(-2.6, 0.25, (T, E, N, 1))
    buf.rewards[:] = rewards
    vp = rs.normal(0, 1, (T + 1, E, 1, 1)).astype(np.float32)
    buf.value_preds[:] = np.repeat(vp, N, axis=2)
    buf.masks[:] = masks
    vn = trainer.value_normalizer
    vn.update(rs.normal(-300, 120, (200, 1)).astype(np.float32))
    out.update(vn0_mean=vn.running_mean.numpy().copy(), vn0_mean_sq=vn.running_mean_sq.numpy().copy(),
               vn0_debias=vn.debiasing_term.numpy().copy())
    next_value = np.repeat(rs.normal(0, 1, (E, 1, 1)).astype(np.float32), N, axis=1)
    out.update(obs=obs, actions=actions, action_log_probs=buf.action_log_probs.copy(), rewards=rewards,
               value_preds=buf.value_preds.copy(), masks=masks, next_value=next_value, **st)
    buf.compute_returns(next_value, vn)
    out["returns"] = buf.returns.copy()
    out["value_preds_after"] = buf.value_preds.copy()
    adv = buf.returns[:-1] - vn.denormalize(buf.value_preds[:-1])
    out["adv_norm"] = (adv - np.nanmean(adv)) / (np.nanstd(adv) + 1e-5)
    if H >= 256:      # the reference networks' forward on the stored rollout, before the update (policy.evaluate_actions,
        trainer.prep_rollout()           # mappo.py:84-99 -> r_actor_critic.py:60-75,110-121): values [T*E*N,1], log-probs, entropy
        with torch.no_grad():
            B = T * E * N
            v, lp, ent = policy.evaluate_actions(buf.share_obs[:-1].reshape(B, S), buf.obs[:-1].reshape(B, D),
                                                 buf.rnn_states[:-1].reshape(B, *buf.rnn_states.shape[3:]),
                                                 buf.rnn_states_critic[:-1].reshape(B, *buf.rnn_states_critic.shape[3:]),
                                                 buf.actions.reshape(B, A), buf.masks[:-1].reshape(B, 1))
        out.update(fwd_values=v.numpy().copy(), fwd_log_probs=lp.numpy().copy(), fwd_entropy=np.array(float(ent)))
    trainer.prep_training()
    torch.manual_seed(3)
    perms, randperm = [], torch.randperm

    def recording_randperm(*a, **k):       # torch's function, not the reference's: note every permutation the generator draws
        p = randperm(*a, **k)
        perms.append(p.numpy().copy())
        return p

    torch.randperm = recording_randperm
    try:
        info = trainer.train(buf, update_actor=True)
    finally:
        torch.randperm = randperm
    assert len(perms) == cfg.ppo_epoch and all(len(p) == T * E * N for p in perms)
    if MB > 1:
        for i, p in enumerate(perms):
            out["perm%d" % i] = p.astype(np.int64)
    for k, v in info.items():
        out["info_" + k] = np.array(float(v))
    for pre, mod in (("actor", policy.actor), ("critic", policy.critic)):
        after = {k: v.numpy().copy() for k, v in mod.state_dict().items()}
        if H >= 256:
            snapshot(out, pre + "2s/", after, {k: out[pre + "/" + k] for k in after})
        else:
            out.update({pre + "2/" + k: v for k, v in after.items()})
    out.update(vn1_mean=vn.running_mean.numpy().copy(), vn1_mean_sq=vn.running_mean_sq.numpy().copy(),
               vn1_debias=vn.debiasing_term.numpy().copy())
    out["dims"] = np.array([N, M, E, T, A, H, D] + ([MB] if MB > 1 else []))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB; episode ends:", int((masks[:, :, 0, 0] == 0).sum()),
          "info:", {k: round(float(v), 5) for k, v in info.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "small")
