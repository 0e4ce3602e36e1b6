"""-m gpu: this package's `Learner.train()` replays a multi-iteration trajectory of the REFERENCE's own orchestrator.

tests/golden/learner_ref_{e1,e2,e2_n8m64}.npz were written by tools/gen_golden_learner.py, which imports
/root/reference/uav_dcc_control/learner.py and runs `Learner(cfg).train()` unmodified for 4 iterations (shipped 4 UAV x 20 PoI
scenario, T = 40, hidden 32, ppo_epoch 15, eval rollout every 2nd iteration; E = 1 -> DummyVecEnv, E = 2 -> SubprocVecEnv; and
the 8 UAV x 64 PoI task of BASELINE c2 / c3 through the size-generalised scenario), recording the Gaussian noise of every `collect`.  Here the same configuration drives this package's Learner with that noise
injected (algos/algo_utils/distributions.set_noise_source) and every quantity the reference's loop produced is compared:

  per rollout   actions, log-probs, rewards, masks (bit-exact), value predictions, GAE returns, observations in the buffer,
                the `rollout()` metrics dict (reward, coverage_rate)                                   learner.py:178-214,278-287
  per iteration lr after lr_decay, `rl_update()`'s dict, ValueNorm state (15 EMA updates per iteration, Q11), per-tensor
                parameter DELTAS of the iteration, masks[0] after after_update                          learner.py:132-175,292-300

The run is free: parameters are set once (iteration 0) and never re-synchronised with the fixture, so errors compound over the
4 iterations the way they would in a real run.  Two storage configurations: the shipped default (compact state + structured
first layers, no observation rows anywhere) and the reference-like row storage.  Tolerances are stated next to each check;
the achieved maxima are printed (pytest -s) and quoted in DESIGN.md section 2.
"""
import json
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import yaml

from conftest import GOLDEN, PKG
from _sampling import sample_indices

# keys this package adds to the reference's configuration (config/algo_config/mappo.yaml, last block) or gives another default
OWN_KEYS = {"double_surrogate", "dedup_critic", "cache_normalized_inputs", "use_hip_graph", "structured_input", "compact_obs",
            "tuned_gemms", "update_chunk_steps"}


def _shipped_cfg(ref_cfg, **over):
    """This package's three YAMLs merged like train.py does, then the generator's overrides; every key of the reference's
    merged configuration must come out with the reference's value (save_gifs is the one documented difference: no viewer)."""
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    for k in ("n_rollout_threads", "max_ep_len", "algo_hidden_size", "n_iters", "eval_interval", "save_model", "log_wandb",
              "num_agents", "num_pois", "use_recurrent_policy", "num_mini_batch", "use_centralized_V", "use_gae") + (("ppo_epoch",) if ref_cfg["algo_hidden_size"] == 256 else ()):
        cfg[k] = ref_cfg[k]      # (ppo_epoch: 2 in the hidden-256 fixture, see tools/gen_golden_learner.py on why)
    for k, v in ref_cfg.items():
        if k in ("save_gifs",):
            continue
        assert k in cfg, "reference key %s missing from the shipped configuration" % k
        assert cfg[k] == v or str(cfg[k]) == str(v), "key %s: %r here, %r in the reference run" % (k, cfg[k], v)
    cfg.update(over)
    return Namespace(**cfg)


class _Track(object):
    """max of |got - want| / scale per named quantity over the whole run (printed at the end)."""

    def __init__(self):
        self.worst, self.failures = {}, []

    def close(self, name, got, want, tol, scale=None):
        """Violations are collected, not raised: the run goes on, so one test run shows every quantity that is off and by how
        much (the test fails at the end with the full list)."""
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
        s = float(np.abs(want).max()) if scale is None else float(scale)
        err = float(np.abs(got - want).max()) / max(s, 1e-30)
        key = name.split("@")[0]
        self.worst[key] = max(self.worst.get(key, 0.0), err)
        if not err <= tol:
            self.failures.append("%s: max error %.3e of scale %.3e exceeds %.1e" % (name, err, s, tol))


def _set_params(module, init):
    sd = module.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(torch.from_numpy(init[k]).to(v.device))
    return sd


def _initial_parameters(Z, tag, module):
    """name -> ndarray the reference run started from.  The hidden-256 fixture does not store its ~1 M initial parameters a
    second time: they are the `actor/` / `critic/` entries of mappo_env_n8m64_h256.npz, with the overrides stored under init/."""
    pre = "init/%s/" % tag
    if any(k.startswith(pre + "base.") for k in Z.files):
        return {k: Z[pre + k] for k in module.state_dict()}
    Zi = np.load(os.path.join(GOLDEN, "mappo_env_n8m64_h256.npz"))
    return {k: (Z[pre + k] if pre + k in Z.files else Zi["%s/%s" % (tag, k)]) for k in module.state_dict()}


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["shipped", "rows"])
@pytest.mark.parametrize("fixture", ["e1", "e2", "e2_n8m64", "e2_rnn", "e2_mb2", "e2_n8m64_h256", "e2_decv", "e2_nogae"])
def test_learner_replays_the_reference_learner(fixture, storage, capsys):
    """e1 / e2: the shipped 4 UAV x 20 PoI task on 1 env (DummyVecEnv in the reference) / 2 envs (SubprocVecEnv); e2_n8m64: the
    BASELINE c2 / c3 task size, 8 UAV x 64 PoI, through the size-generalised scenario (tools/gen_golden_learner.py); e2_rnn:
    `use_recurrent_policy: true` -- the orchestrator's GRU branch (states through collect / insert, zeroed on episode ends,
    carried over by after_update: learner.py:231-265) and the recurrent generator in the update; on the shipped YAMLs the Learner
    falls back to row storage by itself for it; e2_mb2: `num_mini_batch: 2` -- 15 epochs x 2 row mini-batches per iteration, the
    reference's permutations injected (on the shipped state-only storage this is SURVEY.md 8f row 4 end to end);
    e2_n8m64_h256: the 8 x 64 task at the SHIPPED width algo_hidden_size 256, 3 iterations -- the rollout-step and update kernels
    in the instantiations BASELINE c3 runs (parameter snapshots sampled: tests/_sampling.py); e2_decv: `use_centralized_V: false`
    -- the critic on each agent's own observation row (learner.py:43-46,218-222,269-273), one value per agent; like the GRU
    variants the Learner falls back to row storage for it by itself; e2_nogae: `use_gae: false` -- compute_returns' plain
    discounted returns (shared_buffer.py:214-217 -> dcc_returns_compute mode 0) through the orchestrator."""
    _replay(fixture, storage, True, capsys)


@pytest.mark.parametrize("fixture,storage", [(f, s) for f in ("e1", "e2", "e2_n8m64", "e2_mb2") for s in ("rows", "state-only", "shipped")]
                         + [("e2_rnn", "rows"), ("e2_n8m64_h256", "shipped"), ("e2_decv", "rows"), ("e2_decv", "shipped"), ("e2_nogae", "shipped")])
def test_orchestrator_replays_the_reference_learner_on_the_cpu(fixture, storage, capsys, oracle_mod):
    """The same replay without a GPU: the package's Learner / vec-env / buffer / trainer on torch CPU tensors, with the `_cpu` twins
    of the C-ABI standing in for the two device entry points (tests/_cpu_twin_backend.py).  Pins the host-side orchestration --
    what learner.py:132-300 does between the kernels -- in the build container; the kernels themselves are the -m gpu run's job.
    Row storage, the state-only buffer with regenerated rows (dcc_obs_expand_cpu), and the SHIPPED configuration -- state-only
    storage + structured first layers fed by dcc_obs_features_x_cpu (the fused trunk kernels in their torch formulation)."""
    from _cpu_twin_backend import cpu_twin_backend
    with cpu_twin_backend():
        _replay(fixture, storage, False, capsys)


def _replay(fixture, storage, gpu, capsys):
    import utils.pytorch_utils as ptu
    from algos.algo_utils import distributions
    from algos.algo_utils.structured import invalidate_folded_weights
    Z = np.load(os.path.join(GOLDEN, "learner_ref_%s.npz" % fixture))
    ref_cfg = json.loads(str(Z["cfg_json"]))
    E, N, M, T, H, n_iters, n_roll = [int(x) for x in Z["dims"]]
    assert (N, M) == ((8, 64) if "n8m64" in fixture else (4, 20)) and ref_cfg["num_agents"] == N
    assert H == (256 if fixture.endswith("h256") else 32) == ref_cfg["algo_hidden_size"]
    sampled = fixture.endswith("h256")
    # Tolerances sit <= 10x above what the runs achieve (profiles/r05/learner_replay_errors.txt).  One env
    # means a 160-row batch: its Adam updates amplify fp32 noise (achieved 6.0e-3 of max|delta| per iteration, 1.1e-5 in the later
    # rollouts); two envs achieve 1.6e-4 and 3.7e-6; the GRU policies on the GPU 2.2e-3 (CPU 5.6e-5: the recurrent update's
    # chunked BPTT sums in another order there) and 4.1e-7.
    DELTA, LATER = (1e-2, 5.0) if (E == 1 or fixture.endswith("rnn")) else (1.5e-3, 3.0)
    # hidden 256 (sampled snapshots): ||delta||_2 of every tensor is held to the same 1.5e-3, single sampled elements to ELEM =
    # 1e-2 of max|delta|: with ~1 M parameters some always sit where Adam's m / sqrt(v) is ill-conditioned -- the REFERENCE's own
    # run moves single elements by up to 2.0e-3 (||delta||_2 by 8.4e-6) when its initial parameters are perturbed by 1e-7 relative
    # (tools/h256_fixture_sensitivity.sh -> profiles/r06/h256_fixture_sensitivity.txt, ppo_epoch 2, six noise seeds)
    ELEM = 1e-2
    over = dict(use_hip_graph=False)          # the injected noise replaces the in-graph philox stream
    if storage == "rows":
        over.update(structured_input=False, compact_obs=False)
    elif storage == "state-only":
        over.update(structured_input=False, compact_obs=True)
    ptu.set_gpu_mode(gpu, 0)
    from learner import Learner
    lr = Learner(_shipped_cfg(ref_cfg, **over))
    rnn = bool(ref_cfg["use_recurrent_policy"])
    decv = not ref_cfg["use_centralized_V"]
    if decv:      # one critic input per agent row: row storage, no de-duplication, whatever the YAMLs ask for
        b = lr.rl_buffer
        assert b.decentralized and not b.compact and not b.structured and not lr.trainer.dedup_critic and b.share_obs is b.obs
        assert lr.policy.critic.base.mlp.fc1[0].in_features == b.obs_dim
    elif rnn:     # recurrent policies read rows step by step: the Learner switches the shipped state-only storage off on its own
        assert lr.recurrent and not lr.rl_buffer.compact and not lr.rl_buffer.structured
    else:
        assert lr.rl_buffer.compact == (storage != "rows") and lr.rl_buffer.structured == (storage == "shipped")
    init = {"a": _initial_parameters(Z, "actor", lr.policy.actor), "c": _initial_parameters(Z, "critic", lr.policy.critic)}
    _set_params(lr.policy.actor, init["a"])
    _set_params(lr.policy.critic, init["c"])
    invalidate_folded_weights(lr.policy.actor, lr.policy.critic)

    trk = _Track()
    st = {"k": 0, "iter": 0, "draws": 0, "cur": None}
    prev = {"a": {k: v.clone() for k, v in lr.policy.actor.state_dict().items()},
            "c": {k: v.clone() for k, v in lr.policy.critic.state_dict().items()},
            "ra": dict(init["a"]), "rc": dict(init["c"])}

    def noise(shape, dtype, device):         # one draw per collect: the reference's eps of rollout k, step t
        eps = st["cur"][st["draws"]]
        st["draws"] += 1
        return torch.from_numpy(eps.reshape(shape)).to(device=device, dtype=dtype)

    orig_rollout, orig_update, orig_decay = lr.rollout, lr.rl_update, lr.trainer.policy.lr_decay

    def lr_decay(episode, episodes):
        orig_decay(episode, episodes)
        st["iter"] = episode
        for name, opt in (("lr_actor", lr.policy.actor_optimizer), ("lr_critic", lr.policy.critic_optimizer)):
            trk.close("lr@%d" % episode, opt.param_groups[0]["lr"], Z["i%d/%s" % (episode, name)], 1e-12, scale=5e-4)

    def rollout(r_buffer, r_envs, is_render=False, iter_=0):
        k = st["k"]
        pre = "r%d/" % k
        assert k < n_roll and int(Z[pre + "kind"]) == (0 if r_buffer is lr.rl_buffer else 1), "rollout order differs"
        assert int(Z[pre + "iter"]) == st["iter"]
        Eb = r_buffer.n_rollout_threads
        st["cur"], st["draws"] = Z[pre + "eps"].reshape(T, Eb * N, 2), 0
        info = orig_rollout(r_buffer, r_envs, is_render, iter_)
        assert st["draws"] == T, "one noise draw per collect"
        g = lambda name: getattr(r_buffer, name).cpu().numpy()
        np.testing.assert_array_equal(g("masks"), Z[pre + "masks"], err_msg=pre + "masks")               # bit-exact
        assert float(Z[pre + "masks"][1:].min()) == 0.0                                                  # an episode ended
        # Iteration 1 runs on the fixture's own parameters: pure forward / env / GAE parity, <= 1e-5 relative (2e-5 for what comes
        # straight out of the fp32 networks).  From iteration 2 on the parameters are this run's own (updates within 1 % of the
        # reference's, see rl_update below), and that drift feeds back through actions -> positions -> rewards: 5x the window.
        w = 1.0 if st["iter"] == 1 else LATER
        tag = "%s@%d" % ("" if st["iter"] == 1 else "_later", k)
        trk.close("actions" + tag, g("actions"), Z[pre + "actions"], 2e-5 * w)
        trk.close("action_log_probs" + tag, g("action_log_probs"), Z[pre + "action_log_probs"], 2e-5 * w)
        trk.close("rewards" + tag, g("rewards"), Z[pre + "rewards"], 1e-5 * w)
        trk.close("value_preds" + tag, g("value_preds"), Z[pre + "value_preds"], 2e-5 * w)
        trk.close("returns" + tag, g("returns"), Z[pre + "returns"], 1e-5 * w)
        if rnn:     # GRU states of every slot, zeros where an episode ended, slot 0 = what after_update carried over
            trk.close("rnn_states" + tag, g("rnn_states"), Z[pre + "rnn_states"], 2e-5 * w, scale=1.0)
            trk.close("rnn_states_critic" + tag, g("rnn_states_critic"), Z[pre + "rnn_states_critic"], 2e-5 * w, scale=1.0)
            assert not g("rnn_states")[1:][Z[pre + "masks"][1:, :, :, 0] == 0].any()
        if pre + "obs" in Z.files:      # (the 8 x 64 fixture stores the rows of its first rollout only)
            obs = torch.stack([torch.as_tensor(r_buffer.obs[t]) for t in range(T + 1)]).cpu().numpy()   # rows / regenerated from state
            trk.close("obs" + tag, obs, Z[pre + "obs"], 2e-5 * w, scale=1.0)
        trk.close("info_reward" + tag, info["reward"], Z[pre + "info_reward"], 1e-5 * w)
        trk.close("info_coverage_rate" + tag, info["coverage_rate"], Z[pre + "info_coverage_rate"], 1e-6, scale=1.0)
        st["k"] += 1
        return info

    def rl_update():
        i = st["iter"] + 0
        if "i%d/perms" % (st["iter"]) in Z.files:      # the row permutations the reference's generator drew in this iteration
            lr.trainer.minibatch_perms = [torch.from_numpy(p) for p in Z["i%d/perms" % st["iter"]]]
        info = orig_update()
        assert not getattr(lr.trainer, "minibatch_perms", None)          # one permutation per epoch, all consumed
        i = st["iter"]
        pre = "i%d/" % i
        for key in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"):
            # policy_loss is a difference of O(1) terms that nearly cancel (|policy_loss| ~ 1e-2 and 0 on the lr = 0 iteration)
            trk.close("info_%s@%d" % (key, i), info[key], Z[pre + "info_" + key], 1e-4,
                      scale=max(abs(float(Z[pre + "info_" + key])), 0.1 if key == "policy_loss" else 0.0))
        vn = lr.trainer.value_normalizer
        w = 1.0 if i == 1 else LATER        # moments of the returns: from iteration 2 on those of this run's own rollouts (see rollout)
        trk.close("vn_mean@%d" % i, vn.running_mean.cpu().numpy(), Z[pre + "vn_mean"], 1e-5 * w)
        trk.close("vn_mean_sq@%d" % i, vn.running_mean_sq.cpu().numpy(), Z[pre + "vn_mean_sq"], 1e-5 * w)
        trk.close("vn_debias@%d" % i, vn.debiasing_term.cpu().numpy(), Z[pre + "vn_debias"], 1e-6)
        np.testing.assert_array_equal(lr.rl_buffer.masks[0].cpu().numpy(), Z[pre + "masks0"])
        for tag, mod, rtag in (("a", lr.policy.actor, "actor/"), ("c", lr.policy.critic, "critic/")):
            for name, v in mod.state_dict().items():
                if sampled:
                    # sampled snapshot (tests/_sampling.py): the reference's values at sample_indices(numel), max|delta| and
                    # ||delta||_2 of the whole tensor against the previous iteration (#d*) and against the start (#c*)
                    key = pre + rtag + name
                    idx = sample_indices(v.numel())
                    ref_now = Z[key + "#val"].astype(np.float64)
                    got = v.double().cpu().numpy().reshape(-1)
                    ref_prev = np.asarray(prev["r" + tag][name], np.float64).reshape(-1)
                    if i == 1:
                        ref_prev = ref_prev[idx]               # iteration 1: the full initial tensor; later: the sampled values
                    ref_init = np.asarray(init[tag][name], np.float64).reshape(-1)
                    for kind, c, d_ref, d_got in (("delta", "d", ref_now - ref_prev, got - prev[tag][name].double().cpu().numpy().reshape(-1)),
                                                  ("drift", "c", ref_now - ref_init[idx], got - ref_init)):
                        dmax, dl2 = float(Z[key + "#%smax" % c]), float(Z[key + "#%sl2" % c])
                        if dmax == 0.0:        # lr = 0: nothing may move
                            assert float(np.abs(d_got).max()) == 0.0, "%s%s moved on the lr = 0 iteration" % (rtag, name)
                            continue
                        trk.close("%s_%s%s@%d" % (kind, rtag, name, i), d_got[idx], d_ref, ELEM, scale=dmax)
                        trk.close("%s_max_%s%s@%d" % (kind, rtag, name, i), np.abs(d_got).max(), dmax, ELEM)
                        trk.close("l2%s_%s%s@%d" % (kind, rtag, name, i), np.sqrt((d_got * d_got).sum()), dl2, DELTA)
                    prev[tag][name] = v.clone()
                    prev["r" + tag][name] = ref_now            # (sampled values from now on)
                    continue
                d_ref = Z[pre + rtag + name].astype(np.float64) - prev["r" + tag][name].astype(np.float64)
                d_got = (v.double() - prev[tag][name].double()).cpu().numpy()
                scale = float(np.abs(d_ref).max())
                if scale == 0.0:            # lr = 0 (last iteration of the linear schedule): nothing may move
                    assert float(np.abs(d_got).max()) == 0.0, "%s%s moved on the lr = 0 iteration" % (rtag, name)
                else:
                    trk.close("delta_%s%s@%d" % (rtag, name, i), d_got, d_ref, DELTA, scale=scale)
                # ... and the drift of the parameter itself since iteration 0, against the distance the reference has moved
                c_ref = Z[pre + rtag + name].astype(np.float64) - init[tag][name].astype(np.float64)
                trk.close("drift_%s%s@%d" % (rtag, name, i), v.double().cpu().numpy() - init[tag][name].astype(np.float64), c_ref,
                          DELTA, scale=float(np.abs(c_ref).max()))
                prev[tag][name] = v.clone()
                prev["r" + tag][name] = Z[pre + rtag + name]
        return info

    lr.rollout, lr.rl_update, lr.trainer.policy.lr_decay = rollout, rl_update, lr_decay
    old = distributions.set_noise_source(noise)
    try:
        lr.train()
    finally:
        distributions.set_noise_source(old)
        ptu.set_gpu_mode(False)
    with capsys.disabled():
        big = lambda pre: max(((v, k) for k, v in trk.worst.items() if k.startswith(pre)), default=(0.0, ""))
        print("\n[learner replay %s %s %s] worst relative errors: " % (fixture, storage, "gpu" if gpu else "cpu")
              + ", ".join("%s %.1e" % (k, v) for k, v in sorted(trk.worst.items()) if not k.startswith(("delta_", "drift_", "l2d")))
              + "; per-iteration parameter updates %.1e (%s), drift since iteration 0 %.1e (%s)" % (big("delta_") + big("drift_"))
              + ("; ||update||_2 %.1e (%s), ||drift||_2 %.1e (%s)" % (big("l2delta_") + big("l2drift_")) if sampled else ""))
    assert st["k"] == n_roll and st["iter"] == n_iters
    assert not trk.failures, "\n".join(trk.failures)
