"""-m gpu: the device-resident MAPPO loop end to end (env kernel -> policy GEMMs -> buffer -> GAE
kernel -> PPO update), the numpy drop-in surface of the vec-env, and the single-env DCEnv view."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch
import yaml

from conftest import PKG, GOLDEN, load_case

pytestmark = pytest.mark.gpu


def _cfg(**over):
    cfg = {}
    for f in ("config/env_config/dcc.yaml", "config/algo_config/mappo.yaml", "config/expt.yaml"):
        cfg.update(yaml.safe_load(open(os.path.join(PKG, f))))
    cfg.update(over)
    return Namespace(**cfg)


def test_vec_env_numpy_surface_matches_reference_contract():
    """make_env(cfg) -> reset()/step() with the shapes/dtypes of wrappers.py:161-165 and the golden values."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from envs.make_env import make_env
    z, c = load_case(os.path.join(GOLDEN, "env_n4m20_shipped.npz"))
    cfg = _cfg(n_rollout_threads=c["E"], num_agents=4, num_pois=20, comm_r_scale=0.9)
    env = make_env(cfg)
    assert env.observation_space[0].shape == (110,) and env.share_observation_space[0].shape == (440,)
    assert env.action_space[0].__class__.__name__ == "Box" and env.action_space[0].shape == (2,)
    obs = env.reset()
    assert obs.shape == (c["E"], 4, 110)
    np.testing.assert_array_equal(obs[0], z["reset_obs"])
    for t in range(40):
        a = z["actions"][t].copy()
        a0 = a.copy()
        obs, rew, done, infos = env.step(a)
        assert np.array_equal(a, a0)                        # the caller's actions are not mutated (Q2)
        assert rew.shape == (c["E"], 4, 1) and done.shape == (c["E"], 4) and done.dtype == bool
        assert rew.dtype == np.float64                      # the env's own float64 reward (wrappers.py:161-165), not a rounded float32
        np.testing.assert_allclose(rew[:, 0, 0], z["reward"][t], rtol=1e-12, atol=1e-9)
        assert np.array_equal(done[:, 0], z["done"][t].astype(bool))
        assert len(infos) == c["E"]
        np.testing.assert_allclose([i["coverage_rate"] for i in infos], z["coverage"][t], atol=1e-6)
    env.close()


def test_vec_env_numpy_surface_pinned_staging_buffers():
    """The numpy surface leaves the device through pinned staging buffers: fresh arrays by default (the reference's
    contract), or -- reuse_host_buffers -- views of two alternating pinned sets that stay valid for one more step."""
    from envs.hip_vec_env import HipCoverageVecEnv
    a = np.random.RandomState(0).uniform(-1, 1, (6, 4, 2)).astype(np.float32)
    fresh, reuse = HipCoverageVecEnv(6, 4, 20), HipCoverageVecEnv(6, 4, 20, reuse_host_buffers=True)
    assert np.array_equal(fresh.reset(), reuse.reset())
    o1, r1, d1, i1 = fresh.step(a); o2, r2, d2, i2 = fresh.step(a)
    p1, q1, e1, j1 = reuse.step(a); keep = p1.copy(); p2, q2, e2, j2 = reuse.step(a)
    assert np.array_equal(o1, keep) and np.array_equal(o2, p2) and np.array_equal(r2, q2) and np.array_equal(d2, e2)
    assert not np.shares_memory(o1, o2) and not np.shares_memory(p1, p2)
    assert np.array_equal(p1, keep)                       # the previous step's view is still intact after one more step
    p3 = reuse.step(a)[0]
    assert np.shares_memory(p3, p1)                       # ... and is recycled by the step after that
    assert [x["coverage_rate"] for x in i2] == [x["coverage_rate"] for x in j2]
    fresh.close(); reuse.close()


def test_single_env_dcenv_view():
    from envs.mpe.uav_dcc import DCEnv
    z, c = load_case(os.path.join(GOLDEN, "env_n4m20_shipped.npz"))
    env = DCEnv("coverage", num_agents=4, num_pois=20, comm_r_scale=0.9)
    obs_n = env.reset()
    assert len(obs_n) == 4 and obs_n[0].shape == (110,)
    o, r, d, info = env.step(np.zeros((4, 2)))
    assert abs(r[0] - (-56.51319293982027)) < 1e-9 and d == [False] * 4 and info["coverage_rate"] == 0.0   # SURVEY Appendix B, float64
    assert env.render("human") is None
    frame = env.render("rgb_array")
    assert len(frame) == 1 and frame[0].shape == (350, 350, 3) and frame[0].dtype == np.uint8
    env.close()


def test_learner_runs_and_improves_nothing_breaks(tmp_path):
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    cfg = _cfg(n_rollout_threads=64, n_eval_rollout_threads=8, num_agents=4, num_pois=16, max_ep_len=20, n_iters=2,
               ppo_epoch=3, algo_hidden_size=64, save_model=True, save_interval=2, eval_interval=2,
               main_save_path=str(tmp_path))
    lr = Learner(cfg)
    lr.train()
    assert os.path.exists(os.path.join(lr.output_path, "models_2.pt", "agent.pkl"))
    # rollout invariants on the device buffer
    b = lr.rl_buffer
    assert b.compact and b.structured and not torch.is_tensor(b.obs)       # shipped default: compact env state, no observation rows
    assert torch.isfinite(b.returns).all() and torch.isfinite(b.state_pos).all() and torch.isfinite(b.state_energy).all()
    assert bool(((b.masks == 0) | (b.masks == 1)).all())
    # values identical across the agents of an env (centralised critic evaluated once per env)
    assert float((b.value_preds - b.value_preds[:, :, :1]).abs().max()) == 0.0
    # cfg.load_model / cfg.load_model_path (expt.yaml keys, like the reference): a fresh Learner CONTINUES the run from the
    # directory save_model wrote -- iteration counter, parameters, optimizer moments, ValueNorm
    lr2 = Learner(_cfg(**dict(vars(cfg), save_model=False, load_model=True, seed=123,
                              load_model_path=os.path.join(lr.output_path, "models_2.pt"))))
    assert lr2.start_iter == 3 and lr2.total_env_steps == lr.total_env_steps
    for (k, a), (_, b2) in zip(lr.policy.actor.state_dict().items(), lr2.policy.actor.state_dict().items()):
        assert torch.equal(a, b2), k
    assert torch.equal(lr.policy.critic_optimizer.exp_avg_sq, lr2.policy.critic_optimizer.exp_avg_sq)
    assert lr2.policy.actor_optimizer.step_count == lr.policy.actor_optimizer.step_count == 2 * 3
    # agent.pkl alone (what the reference writes): parameters only
    os.remove(os.path.join(lr.output_path, "models_2.pt", "resume.pt"))
    lr2.policy.actor.state_dict()["act.action_out.fc_mean.weight"].zero_()
    lr2.load_model(os.path.join(lr.output_path, "models_2.pt"))
    assert torch.equal(lr2.policy.actor.state_dict()["act.action_out.fc_mean.weight"],
                       lr.policy.actor.state_dict()["act.action_out.fc_mean.weight"])
    info = lr.trainer.train(lr.rl_buffer)          # the buffer of the finished run is still trainable
    for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"):
        assert np.isfinite(info[k]), k
    ptu.set_gpu_mode(False)


def test_rollout_obs_in_buffer_equal_oracle_replay(oracle_mod):
    """The observations the kernel wrote into the rollout buffer are those the oracle produces when it is
    fed the actions stored in the buffer (policy-driven actions, auto-resets included)."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    from envs.hip_vec_env import load_pois
    cfg = _cfg(n_rollout_threads=16, n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=30, n_iters=1,
               ppo_epoch=1, algo_hidden_size=32, save_model=False, compact_obs=False)
    lr = Learner(cfg)
    torch.manual_seed(0)
    lr.rollout(lr.rl_buffer, lr.train_envs)
    b = lr.rl_buffer
    orc = oracle_mod.OracleEnv(16, 4, 16, load_pois(16), 0.2, 0.4, 0.95, 0.0)
    o = orc.reset()
    np.testing.assert_array_equal(b.obs[0].cpu().numpy(), o.astype(np.float32))
    for t in range(30):
        ref = orc.step(b.actions[t].cpu().numpy())
        np.testing.assert_allclose(b.obs[t + 1].cpu().numpy(), ref["obs"].astype(np.float32), rtol=0, atol=1e-5)
        np.testing.assert_allclose(b.rewards[t, :, 0, 0].cpu().numpy(), ref["reward"], rtol=1e-5, atol=1e-5)
        assert np.array_equal(1 - b.masks[t + 1, :, 0, 0].cpu().numpy(), ref["done"])
    ptu.set_gpu_mode(False)


def test_checkpoint_resume_is_bit_faithful(tmp_path):
    """Run 2 iterations; checkpoint after the first; a fresh Learner restored from the checkpoint must
    reproduce the second iteration exactly (parameters, ValueNorm, rollout statistics)."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_rollout_threads=32, n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=12, n_iters=2,
              ppo_epoch=2, algo_hidden_size=32, save_model=False, seed=3)
    a = Learner(_cfg(**kw))
    a.cur_iter = 1
    a.policy.lr_decay(1, 2); r1 = a.rollout(a.rl_buffer, a.train_envs); a.rl_update()
    ck = str(tmp_path / "resume.pt")
    a.save_checkpoint(ck)
    a.policy.lr_decay(2, 2); r2 = a.rollout(a.rl_buffer, a.train_envs); i2 = a.rl_update()
    b = Learner(_cfg(**dict(kw, seed=99)))          # different seed: everything must come from the checkpoint
    b.load_checkpoint(ck)
    assert b.start_iter == 2
    b.policy.lr_decay(2, 2); q2 = b.rollout(b.rl_buffer, b.train_envs); j2 = b.rl_update()
    assert r2 == q2
    for k in i2:
        assert i2[k] == j2[k], k
    for (ka, va), (kb, vb) in zip(a.policy.actor.state_dict().items(), b.policy.actor.state_dict().items()):
        assert torch.equal(va, vb), ka
    assert torch.equal(a.trainer.value_normalizer.running_mean, b.trainer.value_normalizer.running_mean)
    assert torch.equal(a.rl_buffer.masks, b.rl_buffer.masks)        # incl. slot 0 = the last slot of the rollout before the checkpoint
    ptu.set_gpu_mode(False)


def test_checkpoint_of_another_job_size_loads_the_replicated_state(tmp_path):
    """Advisor finding (r02): a checkpoint written by a job of another size (rank count / envs per rank) still carries
    everything that does not depend on it.  It loads with a warning -- parameters, Adam moments, ValueNorm, counters -- and
    the run continues -- when asked for (`resume_strict: false`) or when the file's iteration count says there is no training
    left to continue (evaluation / hand-over).  A run that would CONTINUE TRAINING refuses by default (advisor finding r03: a
    silent partial resume is not the bit-exact continuation a user expects), and `resume_strict: true` always refuses."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=10, n_iters=3, ppo_epoch=2, algo_hidden_size=32,
              save_model=False)
    a = Learner(_cfg(n_rollout_threads=16, **kw))
    a.cur_iter = 1
    a.rollout(a.rl_buffer, a.train_envs); a.rl_update()
    ck = str(tmp_path / "resume.pt")
    a.save_checkpoint(ck)
    d = Learner(_cfg(n_rollout_threads=8, seed=5, **kw))          # default: iteration 1 of 3 -> training would continue -> refuse
    with pytest.raises(ValueError, match=r"1 rank\(s\) with 16 envs each; this job has 1 rank\(s\) with 8 envs each.*resume_strict: false"):
        d.load_checkpoint(ck)                                      # same rank count, other shard size: both named, with the way out
    e = Learner(_cfg(n_rollout_threads=8, seed=5, **dict(kw, n_iters=1)))   # nothing left to train: the lenient hand-over
    with pytest.warns(UserWarning, match="per-rank RNG streams and env states are not"):
        e.load_checkpoint(ck)
    b = Learner(_cfg(n_rollout_threads=8, seed=5, resume_strict=False, **kw))
    with pytest.warns(UserWarning, match="per-rank RNG streams and env states are not"):
        b.load_checkpoint(ck)
    assert b.resume_partial
    assert b.start_iter == 2 and b.total_env_steps == a.total_env_steps
    for (ka, va), (kb, vb) in zip(a.policy.actor.state_dict().items(), b.policy.actor.state_dict().items()):
        assert torch.equal(va, vb), ka
    assert torch.equal(a.policy.critic_optimizer.exp_avg_sq, b.policy.critic_optimizer.exp_avg_sq)
    assert torch.equal(a.trainer.value_normalizer.running_mean, b.trainer.value_normalizer.running_mean)
    res = b.evaluate()                                   # e.g. evaluating on one GPU what eight trained
    assert 0.0 <= res["coverage_rate"] <= 1.0
    b.rollout(b.rl_buffer, b.train_envs); info = b.rl_update()
    assert all(np.isfinite(v) for v in info.values())
    c = Learner(_cfg(n_rollout_threads=8, resume_strict=True, **kw))
    with pytest.raises(ValueError):
        c.load_checkpoint(ck)
    ptu.set_gpu_mode(False)


@pytest.mark.parametrize("force", [0.0, 0.5])
def test_headless_evaluation_dump_replays_through_the_oracle(tmp_path, force, oracle_mod):
    """Learner.evaluate(dump_path) (reference learner.py:143-149 test rollout, :196-210 viewer): the
    dumped trajectory is replayed through the oracle -- the ACTIONS of the file fed to an OracleEnv built from the file's own
    constants must reproduce every dumped array: env-done / connect / connect_s / PoI-done masks and PoI energies bit for
    bit, positions and velocities to 1e-9 (bit-identical with the pull force off), rewards and coverage to 1e-5.  The
    metrics evaluate() returns are recomputed from the file."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    T, E, N, M = 40, 8, 4, 16
    lr = Learner(_cfg(n_rollout_threads=8, n_eval_rollout_threads=E, num_agents=N, num_pois=M, max_ep_len=T, n_iters=1,
                      ppo_epoch=1, algo_hidden_size=32, save_model=False, comm_force_scale=force, r_comm=0.4 if force == 0 else 0.15))
    with torch.no_grad():       # a larger action noise, so that UAVs separate, cover PoIs and (force on) lose connectivity
        lr.policy.actor.act.action_out.logstd._bias.fill_(0.7)
    path = str(tmp_path / "traj.npz")
    res = lr.evaluate(dump_path=path)
    z = np.load(path)
    assert z["pos"].shape == (T, E, N, 2) and z["vel"].shape == (T, E, N, 2) and z["actions"].shape == (T, E, N, 2)
    assert z["energy"].shape == (T, E, M) and z["poi_done"].shape == (T, E, M) and z["poi"].shape == (M, 2)
    assert z["actions"].dtype == np.float32
    orc = oracle_mod.OracleEnv(E, N, M, z["poi"], float(z["r_cover"]), float(z["r_comm"]), float(z["comm_r_scale"]),
                               float(z["comm_force_scale"]))
    orc.reset()
    for t in range(T):
        ref = orc.step(z["actions"][t], want_obs=False)
        st = orc.get_state()                                    # post-auto-reset, like the dump
        for k in ("done", "connect", "connect_s"):
            assert np.array_equal(z[k][t], ref[k]), (k, t)
        assert np.array_equal(z["poi_done"][t], st["done"]), t
        assert np.array_equal(z["energy"][t], st["energy"].astype(np.float32)), t
        if force == 0:
            assert np.array_equal(z["pos"][t], st["pos"]) and np.array_equal(z["vel"][t], st["vel"]), t
        else:
            np.testing.assert_allclose(z["pos"][t], st["pos"], rtol=0, atol=1e-9)
            np.testing.assert_allclose(z["vel"][t], st["vel"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(z["reward"][t], ref["reward"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(z["coverage"][t], ref["coverage"], rtol=0, atol=1e-6)
    assert z["poi_done"].any() and (z["energy"] > 0).any()       # the trajectory does something
    if force > 0:
        assert (z["connect_s"] == 0).any()                         # ... including the branchy force path
    # the returned metrics are functions of the dumped arrays
    cov = z["coverage"]
    np.testing.assert_allclose(res["coverage_rate"], cov.max(0).mean(), rtol=1e-6)
    full = cov >= 1.0
    solved = full.any(0)
    assert abs(res["solved_fraction"] - solved.mean()) < 1e-6
    if solved.any():
        np.testing.assert_allclose(res["steps_to_cover"], (full.argmax(0)[solved] + 1).mean(), rtol=1e-6)
    ptu.set_gpu_mode(False)


def test_reference_buffer_attribute_names_in_the_shipped_default_config(oracle_mod):
    """SURVEY.md 8b B3 with the shipped YAMLs (compact_obs: true, structured_input: true).
    (1) After a native rollout `buffer.obs[t]` / `buffer.share_obs[t]` -- the names the reference's learner reads
        (learner.py:233-234,280) -- return the rows of every slot although none is stored: checked against the ORACLE
        replaying the buffer's actions.
    (2) A learner written against the reference's buffer (warmup writes share_obs[0] / obs[0], learner.py:224-225; collect
        reads share_obs[step] / obs[step]; insert(share_obs, obs, ...) with host rows, :254-276; compute reads share_obs[-1],
        :280; trainer.train(buffer)) runs on the same default config: the first row write switches the buffer to row
        storage."""
    import warnings
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    T, E, N, M = 8, 8, 4, 16
    lr = Learner(_cfg(n_rollout_threads=E, n_eval_rollout_threads=0, num_agents=N, num_pois=M, max_ep_len=T, n_iters=1,
                      ppo_epoch=2, algo_hidden_size=32, save_model=False))
    buf, envs, tr = lr.rl_buffer, lr.train_envs, lr.trainer
    assert buf.compact and buf.structured and not torch.is_tensor(buf.obs)
    lr.rollout(buf, envs)
    D = envs.obs_dim
    assert tuple(buf.obs.shape) == (T + 1, E, N, D) and tuple(buf.share_obs.shape) == (T + 1, E, N, N * D)
    c = lr.cfg
    orc = oracle_mod.OracleEnv(E, N, M, envs.poi_xy, c.r_cover, c.r_comm, c.comm_r_scale, c.comm_force_scale)
    rows = orc.reset().astype(np.float32)
    for t in range(T + 1):
        got = buf.obs[t]
        assert got.is_cuda and tuple(got.shape) == (E, N, D)
        assert np.array_equal(got.cpu().numpy(), rows), t                    # force off: bit-identical to the reference's rows
        so = buf.share_obs[t]
        assert tuple(so.shape) == (E, N, N * D) and np.array_equal(so[:, 2].cpu().numpy(), rows.reshape(E, N * D))
        if t < T:
            rows = orc.step(buf.actions[t].cpu().numpy())["obs"].astype(np.float32)
    assert np.array_equal(buf.share_obs[-1][:, 0].cpu().numpy(), rows.reshape(E, N * D))
    assert buf.compact                                                       # reads do not change the storage

    # (2) the reference learner's call sequence, host rows in
    obs = envs.reset()
    share = np.expand_dims(obs.reshape(E, -1), 1).repeat(N, axis=1)
    with pytest.warns(UserWarning, match="switching to row storage"):
        buf.share_obs[0] = share.copy()
        buf.obs[0] = obs.copy()
    assert not buf.compact and torch.is_tensor(buf.obs) and np.array_equal(buf.obs[0].cpu().numpy(), obs)
    buf.step = 0
    for step in range(T):
        tr.prep_rollout()
        with torch.no_grad():
            value, action, logp, _, _ = tr.policy.get_actions(buf.share_obs[step].reshape(E * N, -1), buf.obs[step].reshape(E * N, -1),
                                                              None, None, buf.masks[step].reshape(E * N, 1))
        actions = action.view(E, N, 2).cpu().numpy()
        obs, rewards, dones, infos = envs.step(actions.copy())
        masks = np.ones((E, N, 1), np.float32); masks[dones] = 0.0
        share = np.expand_dims(obs.reshape(E, -1), 1).repeat(N, axis=1)
        buf.insert(share, obs, None, None, actions, logp.view(E, N, 1).cpu().numpy(), value.view(E, N, 1).cpu().numpy(), rewards, masks)
        assert np.array_equal(buf.obs[step + 1].cpu().numpy(), obs)
    with torch.no_grad():
        nv = tr.policy.get_values(buf.share_obs[-1].reshape(E * N, -1), None, buf.masks[-1].reshape(E * N, 1))
    buf.compute_returns(nv.view(E, N, 1), tr.value_normalizer)
    tr.prep_training()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        info = tr.train(buf, update_actor=True)
    buf.after_update()
    assert all(np.isfinite(v) for v in info.values()) and info["ratio"] > 0
    ptu.set_gpu_mode(False)


def test_train_py_launcher_end_to_end(tmp_path):
    """`cd dynamic-coverage-control_amd && python train.py 0 key=value...` (the reference's entry-point shape)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "train.py", "0", "n_iters=2", "n_rollout_threads=32", "n_eval_rollout_threads=0",
                        "max_ep_len=10", "ppo_epoch=2", "algo_hidden_size=32", "num_agents=4", "num_pois=16",
                        "save_interval=2", "main_save_path=%s/" % tmp_path], cwd=PKG, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "iter: 2" in r.stdout and "value_loss" in r.stdout and "model saved" in r.stdout


def test_graph_replay_does_not_reuse_stale_features():
    """structured_input + use_hip_graph: the per-chunk feature cache is host state, so a graph replay (which runs no
    Python of the rollout body) must still drop it -- the update has to see the features of the NEW rollout."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    lr = Learner(_cfg(n_rollout_threads=32, n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=12, n_iters=1,
                      ppo_epoch=1, algo_hidden_size=32, save_model=False, structured_input=True, use_hip_graph=True))
    b = lr.rl_buffer
    for it in range(3):        # eager (+capture), then two replays
        lr.rollout(b, lr.train_envs)
        assert b._feat_valid == set()
        lr.trainer.prep_training()
        lr.trainer.train(b)        # (not rl_update: its after_update() rewrites slot 0)
        T = b.episode_length
        f = b.features_rows(0, T)
        n = T * b.n_rollout_threads
        fresh = lr.train_envs.env.obs_features(b.state_pos[:T].reshape(n, 4, 2), b.state_vel[:T].reshape(n, 4, 2),
                                                b.state_energy[:T].reshape(n, -1), b.state_done[:T].reshape(n, -1))
        assert torch.equal(f["head"], fresh["head"]) and torch.equal(f["stats"], fresh["stats"]), it
    # the replayed rollout used the CURRENT parameters (the inference cache of the folded first-layer weights is
    # refreshed in place before every replay): values stored for slot 3 == an autograd-path forward now
    lr.rollout(b, lr.train_envs)
    v_now = lr.policy.critic(b.features_at(3))[0].detach()
    np.testing.assert_allclose(b.value_preds[3, :, 0, 0].cpu().numpy(), v_now.view(-1).cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert lr.use_hip_graph and len(lr._graphs) == 1
    ptu.set_gpu_mode(False)


@pytest.mark.parametrize("branch", [{}, {"use_centralized_V": False}, {"use_gae": False}, {"use_proper_time_limits": True},
                                    {"use_gae": False, "use_proper_time_limits": True}],
                         ids=["shipped", "decentralized_V", "no_gae", "proper_time_limits", "no_gae_proper_time_limits"])
def test_hip_graph_rollout_equals_the_eager_rollout(branch):
    """use_hip_graph (default on): the captured rollout holds no reduction -- per-env reward sums / coverage maxima are
    accumulated element-wise inside the graph and reduced after the replay -- and a replay fills the buffer and returns
    the statistics exactly like the same rollout issued eagerly from the same RNG state.  Also with the configuration branches
    served since round 6 (critic per agent row; dcc_returns_compute inside the captured body)."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_rollout_threads=32, n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=12, n_iters=1,
              ppo_epoch=2, algo_hidden_size=32, save_model=False, seed=17, **branch)
    g, e = Learner(_cfg(**dict(kw, use_hip_graph=True))), Learner(_cfg(**dict(kw, use_hip_graph=False)))
    torch.manual_seed(5); g.rollout(g.rl_buffer, g.train_envs)            # eager pass + capture
    assert g.use_hip_graph and len(g._graphs) == 1, "capture must have succeeded on the GPU box"
    for it in range(3):
        st = torch.cuda.get_rng_state()
        r_replay = g.rollout(g.rl_buffer, g.train_envs)                   # graph replay
        torch.cuda.set_rng_state(st)
        r_eager = e.rollout(e.rl_buffer, e.train_envs)
        assert r_replay == r_eager, it
        for name in ("actions", "rewards", "masks", "value_preds", "returns") + (("state_pos",) if g.rl_buffer.store_state else ("obs",)):
            assert torch.equal(torch.as_tensor(getattr(g.rl_buffer, name)), torch.as_tensor(getattr(e.rl_buffer, name))), (name, it)
        if branch.get("use_centralized_V") is False:     # one value per AGENT row (the agents of an env see different rows)
            assert float((g.rl_buffer.value_preds - g.rl_buffer.value_preds[:, :, :1]).abs().max()) > 0.0
        ig, ie = g.rl_update(), e.rl_update()
        assert ig == ie and all(np.isfinite(v) for v in ig.values())
    ptu.set_gpu_mode(False)


def test_rollout_with_step_features_equals_the_two_launch_rollout():
    """step_features (default on): the env launch of a rollout step also writes the policy-input features of the new state
    (dcc_env_step_features) and the logged statistics ride in the record launch -- three launches fewer per step than
    env step + dcc_obs_features + two element-wise torch ops.  Same RNG state -> the same rollout, bit for bit, eagerly and
    as a replayed graph, and the same update."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    for size in (dict(num_agents=4, num_pois=16), dict(num_agents=8, num_pois=64, comm_force_scale=0.5, r_comm=0.2)):
        kw = dict(n_rollout_threads=32, n_eval_rollout_threads=0, max_ep_len=12, n_iters=1, ppo_epoch=2, algo_hidden_size=32,
                  save_model=False, seed=23, **size)
        a, b = Learner(_cfg(**dict(kw, step_features=True))), Learner(_cfg(**dict(kw, step_features=False)))
        assert a._step_features and not b._step_features and a.rl_buffer.structured and a.rl_buffer.compact
        for it in range(3):                                   # eager + capture, then replays
            st = torch.cuda.get_rng_state()
            ra = a.rollout(a.rl_buffer, a.train_envs)
            torch.cuda.set_rng_state(st)
            rb = b.rollout(b.rl_buffer, b.train_envs)
            assert ra == rb, it
            for name in ("actions", "action_log_probs", "rewards", "masks", "value_preds", "returns", "state_pos", "state_energy"):
                assert torch.equal(getattr(a.rl_buffer, name), getattr(b.rl_buffer, name)), (name, it)
            ia, ib = a.rl_update(), b.rl_update()
            assert ia == ib and all(np.isfinite(v) for v in ia.values())
        assert a.rl_buffer._step_feats is not None and b.rl_buffer._step_feats is None
    ptu.set_gpu_mode(False)


def test_compact_state_buffer_trains_like_the_full_buffer():
    """compact_obs: the rollout buffer keeps env state instead of observations and the update regenerates them
    chunk by chunk with dcc_obs_expand.  Same seed -> the same rollout (bit-identical actions/rewards/returns),
    regenerated observations bit-identical to the stored ones, and the same PPO step up to summation order."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_rollout_threads=48, n_eval_rollout_threads=0, num_agents=8, num_pois=64, max_ep_len=25, n_iters=1,
              ppo_epoch=3, algo_hidden_size=64, save_model=False, seed=5, cache_normalized_inputs=False)
    kw["structured_input"] = False          # dense first layers on rows: stored (full) vs regenerated per chunk (comp)
    full = Learner(_cfg(**dict(kw, compact_obs=False)))
    comp = Learner(_cfg(**dict(kw, compact_obs=True, update_chunk_steps=7)))
    for (k, a), (_, b) in zip(full.policy.actor.state_dict().items(), comp.policy.actor.state_dict().items()):
        assert torch.equal(a, b), k
    torch.manual_seed(11); r_full = full.rollout(full.rl_buffer, full.train_envs)
    torch.manual_seed(11); r_comp = comp.rollout(comp.rl_buffer, comp.train_envs)
    fb, cb = full.rl_buffer, comp.rl_buffer
    assert cb.compact and not torch.is_tensor(cb.obs)
    assert r_full == r_comp
    for name in ("actions", "rewards", "masks", "value_preds", "returns", "action_log_probs"):
        assert torch.equal(getattr(fb, name), getattr(cb, name)), name
    T = fb.episode_length
    for t0, t1 in ((0, 7), (7, 20), (20, T + 1)):
        assert torch.equal(cb.obs_rows(t0, t1), fb.obs[t0:t1]), (t0, t1)
    i_full, i_comp = full.rl_update(), comp.rl_update()
    for k in i_full:
        np.testing.assert_allclose(i_comp[k], i_full[k], rtol=2e-4, atol=1e-6, err_msg=k)
    for (k, a), (_, b) in zip(full.policy.state_dict()["actor"].items(), comp.policy.state_dict()["actor"].items()):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
    # a second iteration keeps working (after_update + warmup on the compact slots)
    r2 = comp.rollout(comp.rl_buffer, comp.train_envs)
    assert np.isfinite(r2["reward"]) and all(np.isfinite(v) for v in comp.rl_update().values())
    ptu.set_gpu_mode(False)


def test_structured_input_trains_like_the_full_buffer():
    """structured_input: actor and critic consume dcc_obs_features of the state; no observation row is ever built.
    The rollout it collects is replayed into a full-row learner (rows regenerated with dcc_obs_expand): both must
    compute the same values / log-probs for the stored actions and make the same PPO step up to fp32 re-association."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_rollout_threads=40, n_eval_rollout_threads=0, num_agents=8, num_pois=64, max_ep_len=20, n_iters=1,
              ppo_epoch=3, algo_hidden_size=64, save_model=False, seed=9, cache_normalized_inputs=False)
    st = Learner(_cfg(**dict(kw, structured_input=True, compact_obs=True, update_chunk_steps=8)))
    full = Learner(_cfg(**dict(kw, structured_input=False, compact_obs=False)))
    torch.manual_seed(4)
    r = st.rollout(st.rl_buffer, st.train_envs)
    sb, fb = st.rl_buffer, full.rl_buffer
    assert sb.structured and sb.compact and not torch.is_tensor(sb.obs) and np.isfinite(r["reward"])
    T = sb.episode_length
    fb.obs.copy_(sb.obs_rows(0, T + 1))
    for name in ("actions", "rewards", "masks", "value_preds", "returns", "action_log_probs", "advantages_raw"):
        getattr(fb, name).copy_(getattr(sb, name))
    # the stored values / log-probs came from the structured forward: the dense forward on the rows agrees
    E, N = sb.n_rollout_threads, 8
    with torch.no_grad():
        full.trainer.prep_rollout()
        v = full.policy.critic(fb.share_obs_env[3])[0]
        logp, _ = full.policy.actor.evaluate_actions(fb.obs[3].view(E * N, -1), None, fb.actions[3].view(E * N, -1), None)
    np.testing.assert_allclose(v.view(E).cpu().numpy(), sb.value_preds[3, :, 0, 0].cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(logp.view(E, N).cpu().numpy(), sb.action_log_probs[3, :, :, 0].cpu().numpy(), rtol=1e-4, atol=1e-4)
    i_st, i_full = st.rl_update(), full.rl_update()
    for k in i_full:
        np.testing.assert_allclose(i_st[k], i_full[k], rtol=5e-4, atol=1e-6, err_msg=k)
    for (k, a), (_, b) in zip(full.policy.state_dict()["actor"].items(), st.policy.state_dict()["actor"].items()):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=3e-5, err_msg=k)
    for (k, a), (_, b) in zip(full.policy.state_dict()["critic"].items(), st.policy.state_dict()["critic"].items()):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=3e-5, err_msg=k)
    r2 = st.rollout(st.rl_buffer, st.train_envs)
    assert np.isfinite(r2["reward"]) and all(np.isfinite(v) for v in st.rl_update().values())
    ptu.set_gpu_mode(False)


def test_structured_input_with_shapes_outside_the_fused_variants():
    """64 UAVs -> 130 head columns (> the 128 the fused actor-L1 kernel reaches with two head registers) and hidden width 72 (no float4
    variant): the structured path must still run, on the unfused torch formulation of the same algebra."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    import dcc_hip
    from learner import Learner
    assert not dcc_hip.mlp_fused_supported(64, 130) and dcc_hip.mlp_fused_supported(64, 66) and dcc_hip.mlp_fused_supported(64, 0)
    for hidden in (64, 72):
        lr = Learner(_cfg(n_rollout_threads=8, n_eval_rollout_threads=0, num_agents=64, num_pois=30, max_ep_len=6, n_iters=1,
                          ppo_epoch=2, algo_hidden_size=hidden, save_model=False, structured_input=True))
        r = lr.rollout(lr.rl_buffer, lr.train_envs)
        info = lr.rl_update()
        assert np.isfinite(r["reward"]) and all(np.isfinite(v) for v in info.values())
        # and it is the same function as the dense rows
        b = lr.rl_buffer
        rows = b.obs_rows(2, 3)[0]
        feats = b.features_at(2)
        with torch.no_grad():
            v_s = lr.policy.critic(feats)[0]
            v_d = lr.policy.critic(rows.reshape(8, -1))[0]
            a_s, _, _ = lr.policy.actor(feats, deterministic=True)
            a_d, _, _ = lr.policy.actor(rows.reshape(8 * 64, -1), deterministic=True)
        np.testing.assert_allclose(v_s.cpu().numpy(), v_d.cpu().numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(a_s.cpu().numpy(), a_d.cpu().numpy(), rtol=1e-4, atol=1e-5)
    ptu.set_gpu_mode(False)


def test_structured_learner_full_train_loop_with_eval_envs(tmp_path):
    """Learner.train() end to end on the fast path: structured input, hipGraph rollouts for the train AND the eval
    buffer, LR decay, periodic eval rollout, checkpoint written and reloadable."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    cfg = _cfg(n_rollout_threads=64, n_eval_rollout_threads=16, num_agents=4, num_pois=20, max_ep_len=15, n_iters=3,
               ppo_epoch=2, algo_hidden_size=64, save_model=True, save_interval=3, eval_interval=1, log_interval=1,
               main_save_path=str(tmp_path), structured_input=True, compact_obs=False)
    lr = Learner(cfg)
    lr.train()
    assert lr.rl_buffer.structured and lr.test_buffer.structured and len(lr._graphs) == 2
    assert torch.is_tensor(lr.rl_buffer.obs) and not lr.rl_buffer.compact       # compact_obs: false -> features AND rows
    assert os.path.exists(os.path.join(lr.output_path, "models_3.pt", "agent.pkl"))
    lr2 = Learner(_cfg(**dict(vars(cfg), save_model=False, seed=7)))
    lr2.load_checkpoint(os.path.join(lr.output_path, "models_3.pt", "resume.pt"))
    for (k, a), (_, b) in zip(lr.policy.actor.state_dict().items(), lr2.policy.actor.state_dict().items()):
        assert torch.equal(a, b), k
    res = lr2.evaluate(steps=10)
    assert 0.0 <= res["coverage_rate"] <= 1.0
    ptu.set_gpu_mode(False)


def test_flat_adam_matches_torch_adam_and_clip_grad_norm():
    """algo_utils/optim.py on the GPU (dcc_grad_norm_clip + dcc_adam_step, include/dcc_optim.h) against
    nn.utils.clip_grad_norm_ + torch.optim.Adam on the same gradients: norms, parameters and moments over 5 steps, with
    the clip active and inactive, odd parameter sizes (tails that are not a multiple of 4), weight decay, and run-to-run
    bit equality."""
    from algos.algo_utils.optim import FlatAdam
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    shapes = [(64, 37), (64,), (3, 5), (1,), (130, 64), (7,)]

    def run(max_norm, wd, gscale):
        ps = [torch.nn.Parameter(torch.randn(*sh, device=dev)) for sh in shapes]
        qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
        flat = FlatAdam(ps, lr=5e-4, eps=1e-5, weight_decay=wd)
        ref = torch.optim.Adam(qs, lr=5e-4, eps=1e-5, weight_decay=wd)
        norms = []
        for step in range(5):
            flat.zero_grad()
            gs = [torch.randn(*sh, device=dev) * gscale for sh in shapes]
            for p, q, g in zip(ps, qs, gs):
                p.grad.add_(g)                    # what autograd does with a pre-existing .grad
                q.grad = g.clone()
            rn = torch.nn.utils.clip_grad_norm_(qs, max_norm) if max_norm else torch.linalg.vector_norm(torch.cat([g.reshape(-1) for g in gs]))
            ref.step()
            n = flat.clip_and_step(max_norm)
            norms.append(float(n))
            np.testing.assert_allclose(float(n), float(rn), rtol=2e-6)
            for p, q in zip(ps, qs):
                np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-5, atol=1e-7)
        for i, (p, off) in enumerate(zip(ps, flat._offsets)):
            st = ref.state[qs[i]]
            np.testing.assert_allclose(flat.exp_avg[off:off + p.numel()].cpu().numpy(), st["exp_avg"].reshape(-1).cpu().numpy(), rtol=1e-5, atol=1e-7)
            np.testing.assert_allclose(flat.exp_avg_sq[off:off + p.numel()].cpu().numpy(), st["exp_avg_sq"].reshape(-1).cpu().numpy(), rtol=1e-5, atol=1e-10)
        assert all(p.data_ptr() % 256 == 0 for p in ps) and float(flat.flat_param[flat._offsets[2] + 15:flat._offsets[3]].abs().sum()) == 0.0
        return norms, flat.flat_param.clone()

    for max_norm, wd, gscale in ((10.0, 0.0, 1.0), (10.0, 0.0, 0.01), (None, 0.0, 1.0), (0.5, 0.01, 3.0)):
        torch.manual_seed(9); n1, p1 = run(max_norm, wd, gscale)
        torch.manual_seed(9); n2, p2 = run(max_norm, wd, gscale)
        assert n1 == n2 and torch.equal(p1, p2)               # fixed-order reduction: bit-reproducible
    # checkpoint round trip, and taking over a torch.optim.Adam state_dict
    ps = [torch.nn.Parameter(torch.randn(5, 3, device=dev)), torch.nn.Parameter(torch.randn(9, device=dev))]
    a = FlatAdam(ps, lr=1e-3)
    for p in ps:
        p.grad.add_(torch.randn_like(p))
    a.clip_and_step(1.0)
    b = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=7.0)
    b.load_state_dict(a.state_dict())
    assert b.step_count == 1 and torch.equal(a.exp_avg, b.exp_avg) and b.param_groups[0]["lr"] == 1e-3
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    t = torch.optim.Adam(qs, lr=2e-3)
    for q in qs:
        q.grad = torch.ones_like(q)
    t.step()
    b.load_state_dict(t.state_dict())
    assert b.step_count == 1 and b.param_groups[0]["lr"] == 2e-3 and float(b.exp_avg[:15].min()) > 0


def test_entry_points_report_through_dcc_last_error():
    """dcc_gae_compute / dcc_mlp.h / dcc_optim.h failures carry a message (round-1 verdict: they returned a bare -1)."""
    import dcc_hip
    L = dcc_hip.load_library()
    assert L.dcc_gae_compute(None, None, None, None, 0.99, 0.95, None, None, 4, 4, None) != 0
    assert b"dcc_gae_compute" in L.dcc_last_error()
    assert L.dcc_relu_ln_fwd(None, None, None, None, 1e-5, None, 4, 256, None) != 0
    assert b"dcc_relu_ln_fwd" in L.dcc_last_error()
    x = torch.zeros(8, 7, device="cuda")
    assert L.dcc_relu_ln_fwd(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), 1e-5, x.data_ptr(), 8, 1000, None) == -4
    assert b"compiled variants" in L.dcc_last_error()
    assert L.dcc_adam_step(None, None, None, None, 4, 0.1, 1.0, 0.9, 0.999, 1e-8, 0.0, None, None) != 0
    assert b"dcc_adam_step" in L.dcc_last_error()
    with pytest.raises(dcc_hip.DccError, match="dcc_gae_compute"):
        dcc_hip.gae_compute(torch.zeros(0, 4, device="cuda"), torch.zeros(1, 4, device="cuda"), torch.zeros(1, 4, device="cuda"),
                            None, 0.99, 0.95, torch.zeros(1, 4, device="cuda"))


def test_fused_rollout_glue_fills_the_buffer_like_collect_and_insert():
    """dcc_rollout_sample / dcc_rollout_record (one launch each per step) leave the rollout buffer as the unfused
    collect() + insert() do: same actions bit for bit (same RNG stream), same values / rewards / masks, log-probs to 1e-6."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    kw = dict(n_rollout_threads=40, n_eval_rollout_threads=0, num_agents=8, num_pois=64, max_ep_len=30, n_iters=1,
              ppo_epoch=1, algo_hidden_size=64, save_model=False, seed=21, use_hip_graph=False)
    a, b = Learner(_cfg(**kw)), Learner(_cfg(**kw))
    assert a._fused_glue_ok(a.rl_buffer)
    b._fused_glue_ok = lambda r_buffer: False
    torch.manual_seed(77); ra = a.rollout(a.rl_buffer, a.train_envs)
    torch.manual_seed(77); rb = b.rollout(b.rl_buffer, b.train_envs)
    A, B = a.rl_buffer, b.rl_buffer
    assert torch.equal(A.actions, B.actions)
    for name in ("value_preds", "rewards", "masks", "returns"):
        assert torch.equal(getattr(A, name), getattr(B, name)), name
    np.testing.assert_allclose(A.action_log_probs.cpu().numpy(), B.action_log_probs.cpu().numpy(), rtol=0, atol=2e-6)
    assert int((A.masks == 0).sum()) == int((B.masks == 0).sum()) and A.step == B.step
    assert ra == rb
    ptu.set_gpu_mode(False)


@pytest.mark.parametrize("mode", ["use_recurrent_policy", "use_naive_recurrent_policy"])
def test_recurrent_policy_variants_train_on_the_device(mode):
    """GRU policies (off in the shipped config) through the whole loop on the GPU -- per-(env, agent)
    GRU states in the rollout buffer, zeroed when an env finishes, chunked / whole-episode generators, eval rollout."""
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(True, 0)
    from learner import Learner
    lr = Learner(_cfg(n_rollout_threads=12, n_eval_rollout_threads=0, num_agents=4, num_pois=16, max_ep_len=40, n_iters=1,
                      ppo_epoch=2, algo_hidden_size=32, save_model=False, num_mini_batch=2, data_chunk_length=5,
                      use_hip_graph=False, **{mode: True}))
    b = lr.rl_buffer
    assert lr.recurrent and b.recurrent and not b.structured and not b.compact and b.rnn_states.shape == (41, 12, 4, 1, 32)
    for _ in range(2):
        r = lr.rollout(b, lr.train_envs)
        info = lr.rl_update()
        assert np.isfinite(r["reward"]) and all(np.isfinite(v) for v in info.values())
    assert float(b.rnn_states[1:].abs().sum()) > 0
    done_next = b.masks[1:, :, :, 0] == 0                      # state stored after a finishing step restarts from zero
    assert float(b.rnn_states[1:][done_next].abs().sum()) == 0.0
    res = lr.evaluate(steps=10)
    assert 0.0 <= res["coverage_rate"] <= 1.0
    ptu.set_gpu_mode(False)
