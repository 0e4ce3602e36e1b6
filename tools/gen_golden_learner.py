"""Generate tests/golden/learner_ref_e<E>.npz: a multi-iteration trajectory of the REFERENCE's own orchestrator,
`Learner(cfg).train()` imported from /root/reference/uav_dcc_control/learner.py (learner.py:132-300: train -> lr_decay ->
rollout = warmup + [collect -> step -> insert] x T + compute -> rl_update -> after_update, eval rollouts every
eval_interval).  Container-only (needs /root/reference); the outputs are data.  Re-run:

    python tools/gen_golden_learner.py 1      # n_rollout_threads 1 -> DummyVecEnv        (envs/wrappers.py:204-236)
    python tools/gen_golden_learner.py 2      # n_rollout_threads 2 -> SubprocVecEnv      (envs/wrappers.py:133-202)
    python tools/gen_golden_learner.py 2 8 64 # the BASELINE c2 / c3 task size (8 UAV x 64 PoI) -> learner_ref_e2_n8m64.npz
    python tools/gen_golden_learner.py 2 rnn  # use_recurrent_policy: true (GRU actor / critic)  -> learner_ref_e2_rnn.npz
    python tools/gen_golden_learner.py 2 mb2  # num_mini_batch: 2 (row mini-batches in every epoch) -> learner_ref_e2_mb2.npz
    python tools/gen_golden_learner.py 2 decv # use_centralized_V: false (critic on each agent's own row)  -> learner_ref_e2_decv.npz
    python tools/gen_golden_learner.py 2 nogae # use_gae: false -> plain discounted returns (shared_buffer.py:214-217) -> learner_ref_e2_nogae.npz
    python tools/gen_golden_learner.py 2 8 64 h256   # 8 x 64 at the SHIPPED width, algo_hidden_size 256, ppo_epoch 2
                                                      # -> learner_ref_e2_n8m64_h256.npz (after gen_golden_mappo_env.py n8m64_h256)

h256: the ~1 M initial parameters are not stored a second time: the run starts from the `actor/` / `critic/` parameters of
tests/golden/mappo_env_n8m64_h256.npz (loaded into the reference's networks before training, like the action-mean bias below),
and the parameters after each iteration are kept as sampled snapshots (tests/_sampling.py: 4,096 elements + max|delta| +
||delta||_2 per tensor, against the previous iteration and against the start).  ppo_epoch is 2 there (as in the
mappo_env_* fixtures), not 15: at width 256 on a 640-row batch the reference's own longer updates are chaotic at the 1-ulp
level.  tools/h256_fixture_sensitivity.sh (-> profiles/r06/h256_fixture_sensitivity.txt) runs the reference with its initial
parameters scaled by 1 + 1e-7 N(0,1), six noise seeds: with 15 epochs every seed moves single elements of the iteration-2
update by 9e-2 ... 2e-1 of max|delta| and ||delta||_2 by 9e-3; with 5 (3) epochs one seed in six bifurcates (elements 0.59
(0.22), ||delta||_2 1.8e-2 (7.6e-4)); with 2 epochs all six stay within 2.0e-3 / 8.4e-6 over the whole run -- the only
trajectory other arithmetic orders can be held to a tight bar on.  (`ep<k>`, `pert=<x>`, `seed=<n>`, `out=<dir>`: probe flags.)

The third form needs the size-generalised scenario: the shipped one hard-codes 4 x 20 in make_world (coverage.py:40-41), so
`scenarios.load` is pointed at tools/ref_harness.sized_scenario_class (ONLY make_world replaced) before the envs are built;
DCEnv, MultiAgentEnv, CoverageWorld, the vec-env wrappers, make_env and the Learner run unmodified.  To keep the files small the
observation rows are stored for the first training rollout (and, at 4 x 20, the first eval rollout) only.

Nothing of the reference is modified: the class is driven through its public `train()`; the only instrumentation is
  * a wrapper around the bound `learner.rollout` / `learner.rl_update` that snapshots the buffers / parameters after each call,
  * a forward hook on the actor's DiagGaussian head that records (mean, std) of every `collect` so that the Gaussian noise
    eps = (action - mean) / std the reference drew can be stored (the replaying test injects it in place of its own RNG).
Third-party container modules absent from this image (gym, omegaconf, wandb, imageio) are in-memory / test stand-ins with no
arithmetic (SURVEY.md 8c iii).

Configuration = the reference's three YAMLs merged like train.py:12-19 does, with only: n_rollout_threads E, max_ep_len 40,
algo_hidden_size 32, n_iters 4, eval_interval 2, save_model / log_wandb off.  The shipped scenario (4 UAV x 20 PoI) and every
other key (ppo_epoch 15, num_mini_batch 1, lr 5e-4 with linear decay -> the 4th iteration runs with lr 0) are untouched.
Before training, the action-mean bias is set to (+1.2, 0.15): the swarm drifts east and leaves the arena around step 33-36, so
each rollout holds an episode end + auto-reset (masks = 0, the value bootstrap cut, the reset observation in the next slot).

Contents (k = rollout call in order of execution, i = iteration 1..4):
  cfg_json                         the merged configuration
  init/actor/*, init/critic/*      parameters the run starts from
  r<k>/kind (0 train, 1 eval), r<k>/iter, r<k>/eps [T,E,N,2] f64, r<k>/mean, r<k>/std, r<k>/actions, r<k>/action_log_probs,
  r<k>/rewards, r<k>/masks, r<k>/value_preds, r<k>/returns, r<k>/obs [T+1,E,N,D] f32 (k = 0, 2 only), r<k>/info_reward, r<k>/info_coverage_rate,
  r<k>/vn_* ValueNorm state the returns were computed with
  i<i>/lr_actor, lr_critic (after lr_decay), i<i>/info_* (rl_update's dict), i<i>/vn_* (after the update),
  i<i>/actor/*, i<i>/critic/* (parameters after the update), i<i>/masks0 (slot 0 after after_update)
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/uav_dcc_control"
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests", "_standin"))        # omegaconf stand-in (PyYAML-backed container)
sys.path.insert(1, os.path.join(HERE, "..", "tests"))
from _sampling import snapshot  # noqa: E402
from ref_harness import _install_gym_stub  # noqa: E402


def _stub_modules():
    _install_gym_stub()
    for name in ("wandb", "imageio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)


def sd_np(module, prefix, out):
    for k, v in module.state_dict().items():
        out[prefix + k] = v.detach().cpu().numpy().copy()


def vn_np(vn, prefix, out):
    out[prefix + "vn_mean"] = vn.running_mean.numpy().copy()
    out[prefix + "vn_mean_sq"] = vn.running_mean_sq.numpy().copy()
    out[prefix + "vn_debias"] = vn.debiasing_term.numpy().copy()


def main(E, N=None, M=None, rnn=False, mb=1, h256=False, decv=False, nogae=False, epochs=2, pert=0.0, outdir=None, pert_seed=1):
    _stub_modules()
    init_file = os.path.join(HERE, "..", "tests", "golden", "mappo_env_n8m64_h256.npz")
    sized = N is not None
    os.chdir(REF)                         # the reference resolves ./config/... and ./envs/... relative to its own directory
    sys.path.insert(0, REF)
    from omegaconf import OmegaConf
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from learner import Learner

    cfg = OmegaConf.merge(OmegaConf.load("./config/env_config/dcc.yaml"), OmegaConf.load("./config/algo_config/mappo.yaml"),
                          OmegaConf.load("./config/expt.yaml"))
    cfg.log_wandb = False
    cfg.save_model = False
    cfg.n_rollout_threads = E
    cfg.max_ep_len = 40
    cfg.algo_hidden_size = 32
    cfg.n_iters = 4
    cfg.eval_interval = 2
    if h256:
        assert (N, M) == (8, 64), "h256 starts from the parameters of mappo_env_n8m64_h256.npz"
        cfg.algo_hidden_size = 256
        cfg.ppo_epoch = epochs
    if decv:       # learner.py:43-46,218-222,269-273: the critic's input is each agent's own observation row
        cfg.use_centralized_V = False
    if nogae:      # compute_returns' last branch: returns[-1] = next_value, discounted sum segmented by masks (shared_buffer.py:214-217)
        cfg.use_gae = False
    if mb > 1:     # the reference's mini-batch loop (mappo.py:203-213 over shared_buffer.py:239-279); the permutations it draws with
        cfg.num_mini_batch = mb              # torch.randperm are recorded (i<i>/perms [ppo_epoch, T*E*N])
    if rnn:        # the reference's recurrent branch of the orchestrator: GRU states in collect / insert (zeroed on episode ends,
        cfg.use_recurrent_policy = True      # learner.py:258-265), carried over by after_update, recurrent_generator in the update
    if sized:
        cfg.num_agents, cfg.num_pois = N, M
        from argparse import Namespace as _NS
        from ref_harness import sized_scenario_class
        import envs.mpe.multiagent.scenarios as scenarios
        Sized = sized_scenario_class()
        scenarios.load = lambda name: _NS(Scenario=Sized)      # what DCEnv.__init__ calls (uav_dcc.py:21); forked workers inherit it
    torch.set_num_threads(1)
    learner = Learner(cfg)
    out = {"cfg_json": np.array(json.dumps(OmegaConf.to_container(cfg, resolve=True), default=str))}

    if h256:
        Zi = np.load(init_file)
        for mod, pre in ((learner.policy.actor, "actor/"), (learner.policy.critic, "critic/")):
            mod.load_state_dict({k[len(pre):]: torch.from_numpy(Zi[k]) for k in Zi.files if k.startswith(pre)}, strict=True)
    with torch.no_grad():
        learner.policy.actor.act.action_out.fc_mean.bias.copy_(torch.tensor([1.2, 0.15]))
    if pert:      # sensitivity probe only (never for a committed fixture): relative 1-ulp-sized noise on the initial parameters
        g = torch.Generator().manual_seed(pert_seed)
        with torch.no_grad():
            for p in list(learner.policy.actor.parameters()) + list(learner.policy.critic.parameters()):
                p.mul_(1.0 + pert * torch.randn(p.shape, generator=g))
    init_np = {}
    sd_np(learner.policy.actor, "init/actor/", init_np)
    sd_np(learner.policy.critic, "init/critic/", init_np)
    if h256:      # only what differs from the file the parameters came from
        out["init/actor/act.action_out.fc_mean.bias"] = init_np["init/actor/act.action_out.fc_mean.bias"]
    else:
        out.update(init_np)
    prev_np = dict(init_np)

    rec = {"on": False, "mean": [], "std": []}

    def hook(mod, inp, dist):
        if rec["on"]:
            rec["mean"].append(dist.mean.detach().numpy().copy())
            rec["std"].append(dist.stddev.detach().numpy().copy())

    learner.policy.actor.act.action_out.register_forward_hook(hook)
    T, N = learner.max_ep_len, learner.n_agents
    state = {"k": 0, "iter": 0}
    orig_rollout, orig_update = learner.rollout, learner.rl_update
    orig_decay = learner.trainer.policy.lr_decay

    def lr_decay(episode, episodes):
        orig_decay(episode, episodes)
        state["iter"] = episode
        out["i%d/lr_actor" % episode] = np.array(learner.policy.actor_optimizer.param_groups[0]["lr"], np.float64)
        out["i%d/lr_critic" % episode] = np.array(learner.policy.critic_optimizer.param_groups[0]["lr"], np.float64)

    def rollout(r_buffer, r_envs, is_render=False, iter_=0):
        rec["on"], rec["mean"], rec["std"] = True, [], []
        pre = "r%d/" % state["k"]
        vn_np(learner.trainer.value_normalizer, pre, out)         # the state compute() denormalises with
        info = orig_rollout(r_buffer, r_envs, is_render, iter_)
        rec["on"] = False
        Eb = r_buffer.n_rollout_threads
        mean = np.stack(rec["mean"]).reshape(T, Eb, N, -1).astype(np.float64)
        std = np.stack(rec["std"]).reshape(T, Eb, N, -1).astype(np.float64)
        actions = r_buffer.actions.copy()
        out.update({pre + "kind": np.array(0 if r_buffer is learner.rl_buffer else 1), pre + "iter": np.array(state["iter"]),
                    pre + "eps": (actions.astype(np.float64) - mean) / std, pre + "mean": mean.astype(np.float32),
                    pre + "std": std.astype(np.float32), pre + "actions": actions,
                    pre + "action_log_probs": r_buffer.action_log_probs.copy(), pre + "rewards": r_buffer.rewards.copy(),
                    pre + "masks": r_buffer.masks.copy(), pre + "value_preds": r_buffer.value_preds.copy(),
                    pre + "returns": r_buffer.returns.copy(),
                    **({pre + "rnn_states": r_buffer.rnn_states.copy(), pre + "rnn_states_critic": r_buffer.rnn_states_critic.copy()}
                       if rnn else {}),
                    pre + "info_reward": np.array(float(info["reward"])),
                    pre + "info_coverage_rate": np.array(float(info["coverage_rate"]))})
        if state["k"] in ((0,) if sized else (0, 2)):     # rows of the first training rollout (+ the first eval rollout): the env itself
            out[pre + "obs"] = r_buffer.obs.copy()          # is pinned elsewhere, later rollouts are covered by rewards / returns
        assert np.array_equal(r_buffer.share_obs, r_buffer.obs) if decv else \
            np.array_equal(r_buffer.share_obs[:, :, 0], r_buffer.obs.reshape(T + 1, Eb, -1))
        state["k"] += 1
        return info

    def rl_update():
        perms, randperm = [], torch.randperm

        def recording_randperm(*a, **k):     # torch's function, not the reference's
            p = randperm(*a, **k)
            perms.append(p.numpy().copy())
            return p

        torch.randperm = recording_randperm
        try:
            info = orig_update()
        finally:
            torch.randperm = randperm
        pre = "i%d/" % state["iter"]
        if mb > 1:
            out[pre + "perms"] = np.stack(perms).astype(np.int64)
        for k, v in info.items():
            out[pre + "info_" + k] = np.array(float(v))
        vn_np(learner.trainer.value_normalizer, pre, out)
        if h256:
            for tag, mod in (("actor/", learner.policy.actor), ("critic/", learner.policy.critic)):
                now = {k: v.detach().numpy().copy() for k, v in mod.state_dict().items()}
                snapshot(out, pre + tag, now, {k: prev_np["init/" + tag + k] for k in now}, {k: init_np["init/" + tag + k] for k in now})
                prev_np.update({"init/" + tag + k: v for k, v in now.items()})
        else:
            sd_np(learner.policy.actor, pre + "actor/", out)
            sd_np(learner.policy.critic, pre + "critic/", out)
        out[pre + "masks0"] = learner.rl_buffer.masks[0].copy()
        return info

    learner.rollout, learner.rl_update = rollout, rl_update
    learner.trainer.policy.lr_decay = lr_decay
    learner.train()

    out["dims"] = np.array([E, N, learner.cfg.num_pois, T, learner.cfg.algo_hidden_size, learner.cfg.n_iters, state["k"]])
    assert outdir or (not pert and epochs == 2 and pert_seed == 1), "probe runs must not overwrite the committed fixture: pass out=<dir>"
    path = os.path.join(outdir or os.path.join(HERE, "..", "tests", "golden"), "learner_ref_e%d%s%s%s.npz" % (E, "_n%dm%d" % (N, M) if sized else "", "_rnn" if rnn else "",
                                                                                     ("_mb%d" % mb if mb > 1 else "") + ("_h256" if h256 else "") + ("_decv" if decv else "") + ("_nogae" if nogae else "")))
    np.savez_compressed(path, **out)
    ends = [int((out["r%d/masks" % k][1:, :, 0, 0] == 0).sum()) for k in range(state["k"])]
    print("wrote", os.path.normpath(path), os.path.getsize(path) // 1024, "KB; rollouts:", state["k"], "kinds:",
          [int(out["r%d/kind" % k]) for k in range(state["k"])], "episode ends per rollout:", ends)
    print("rollout infos:", [(round(float(out["r%d/info_reward" % k]), 3), round(float(out["r%d/info_coverage_rate" % k]), 3))
                             for k in range(state["k"])])


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a in ("rnn", "h256", "decv", "nogae") or a.startswith(("mb", "ep", "pert=", "out=", "seed="))]
    argv = [a for a in sys.argv[1:] if a not in flags]
    opt = lambda pre, conv, default: ([conv(a[len(pre):]) for a in flags if a.startswith(pre)] or [default])[0]
    main(*([int(v) for v in argv[:3]] or [2]), rnn="rnn" in flags, mb=opt("mb", int, 1), h256="h256" in flags, decv="decv" in flags, nogae="nogae" in flags,
         epochs=opt("ep", int, 2), pert=opt("pert=", float, 0.0), outdir=opt("out=", str, None),
         pert_seed=opt("seed=", int, 1))
