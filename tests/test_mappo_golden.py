"""MAPPO numerics against fixtures produced by the reference's own algos.mappo / buffer.shared_buffer /
utils.valuenorm (tools/gen_golden_mappo.py).  CPU torch: this is host-side PyTorch logic, the HIP
GAE kernel is checked in test_gae_hip.py (-m gpu)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_param_deltas

DELTA_TOL = 3e-4      # post-update parameters as UPDATES: of max|delta| per tensor (achieved on the CPU: 3.0e-5, see the run summary)

Z = np.load(os.path.join(GOLDEN, "mappo_small.npz"))
N, E, T, D, A, H = 4, 3, 16, 20, 2, 32
S = N * D


def make_cfg(**over):
    c = dict(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2, layer_N=1,
             use_ReLU=True, use_popart=False, use_valuenorm=True, use_feature_normalization=True, use_orthogonal=True,
             gain=0.01, use_recurrent_policy=False, use_naive_recurrent_policy=False, recurrent_N=1,
             actor_lr=5e-4, critic_lr=5e-4, opti_eps=1e-5, weight_decay=0, use_clipped_value_loss=True,
             clip_param=0.2, num_mini_batch=1, entropy_coef=0.01, value_loss_coef=1, use_max_grad_norm=True,
             max_grad_norm=10.0, use_gae=True, gamma=0.99, gae_lambda=0.95, use_proper_time_limits=False,
             use_huber_loss=True, use_value_active_masks=True, use_policy_active_masks=True, huber_delta=10.0,
             data_chunk_length=10, stacked_frames=1)
    c.update(over)
    return Namespace(**c)


class Box:
    def __init__(self, n):
        self.shape = (n,)


def _sd(prefix):
    return {k[len(prefix):]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(prefix)}


def _policy(cfg, dev="cpu"):
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(dev == "cuda", 0)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    pol = MAPPOPolicy(cfg, Box(D), Box(S), Box(A))
    # the reference's state_dict also holds its dead `mlp.fc_h` template: ignored
    missing, unexpected = pol.actor.load_state_dict(_sd("actor/"), strict=False)
    assert not missing and all(".fc_h." in k for k in unexpected)
    missing, unexpected = pol.critic.load_state_dict(_sd("critic/"), strict=False)
    assert not missing and all(".fc_h." in k for k in unexpected)
    return pol, MAPPOTrainer(cfg, pol)


def test_parameter_counts_match_reference_live_parameters():
    pol, _ = _policy(make_cfg())
    live = lambda pre: sum(Z[k].size for k in Z.files if k.startswith(pre) and ".fc_h." not in k)
    assert sum(p.numel() for p in pol.actor.parameters()) == live("actor/")
    assert sum(p.numel() for p in pol.critic.parameters()) == live("critic/")
    # SURVEY.md 8e closed forms (H=256): actor 258*D + 67,588 ; critic 258*S + 67,329
    h = 32
    assert live("actor/") == (2 * D) + (D * h + h) + 2 * h + (h * h + h) + 2 * h + (h * A + A) + A


def test_fresh_policy_entropy_is_2p8379():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from algos.mappo import MAPPOPolicy
    pol = MAPPOPolicy(make_cfg(), Box(D), Box(S), Box(A))
    _, _, ent = pol.evaluate_actions(torch.zeros(5, S), torch.zeros(5, D), None, None, torch.zeros(5, A), None, None,
                                     torch.ones(5, 1))
    assert abs(float(ent) - 2.8379) < 1e-3     # 2 * (0.5 + 0.5 ln 2 pi), sigma = 1


def test_evaluate_actions_matches_reference():
    pol, tr = _policy(make_cfg())
    tr.prep_rollout()
    with torch.no_grad():
        v, logp, ent = pol.evaluate_actions(Z["ev_sobs"], Z["ev_obs"], None, None, Z["ev_act"], None, None,
                                            torch.ones(64, 1))
        mean_act, _ = pol.act(Z["ev_obs"], deterministic=True)
    np.testing.assert_allclose(v.numpy(), Z["ev_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.numpy(), Z["ev_logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(ent), float(Z["ev_entropy"]), rtol=1e-6)
    np.testing.assert_allclose(mean_act.numpy(), Z["ev_mean_act"], rtol=1e-5, atol=1e-7)


def test_huber_is_one_sided_like_the_reference():
    from utils.util import huber_loss
    out = huber_loss(torch.from_numpy(Z["huber_e"]), 10.0).numpy()
    np.testing.assert_array_equal(out, Z["huber_out"])
    assert out[0] == 0.0          # e = -25 < -delta contributes nothing (Q5)


def _set_vn(vn, pre):
    vn.running_mean.copy_(torch.from_numpy(Z[pre + "_mean"]))
    vn.running_mean_sq.copy_(torch.from_numpy(Z[pre + "_mean_sq"]))
    vn.debiasing_term.copy_(torch.from_numpy(Z[pre + "_debias"]))


def test_oracle_gae_matches_reference_bit_exact():
    from oracle import mappo_oracle as mo
    mean, std = mo.valuenorm_mean_std(Z["vn0_mean"][0], Z["vn0_mean_sq"][0], Z["vn0_debias"])
    ret, vp = mo.compute_returns_gae(Z["buf_rewards"], Z["buf_value_preds"], Z["buf_masks"], Z["next_value"], 0.99,
                                     0.95, mean, std)
    np.testing.assert_array_equal(vp, Z["buf_value_preds_after"])
    np.testing.assert_array_equal(ret, Z["returns"])


def test_valuenorm_matches_reference():
    from utils.valuenorm import ValueNorm
    from oracle import mappo_oracle as mo
    vn = ValueNorm(1)
    rs = np.random.RandomState(11)   # replay the fixture's draws up to the ValueNorm updates
    rs.normal(0, 1, (64, D)); rs.normal(0, 1, (64, S)); rs.uniform(-1, 1, (64, A))
    rs.normal(0, 1, (T + 1, E, N, D)); rs.uniform(-1, 1, (T, E, N, A)); rs.normal(-2.5, 0.3, (T, E, N, 1))
    rs.normal(-50, 30, (T, E, 1, 1)); rs.normal(0, 1, (T + 1, E, 1, 1))
    vn.update(rs.normal(-300, 120, (200, 1)).astype(np.float32))
    vn.update(rs.normal(-280, 100, (200, 1)).astype(np.float32))
    np.testing.assert_allclose(vn.running_mean.numpy(), Z["vn0_mean"], rtol=1e-6)
    np.testing.assert_allclose(vn.running_mean_sq.numpy(), Z["vn0_mean_sq"], rtol=1e-6)
    np.testing.assert_allclose(vn.debiasing_term.numpy(), Z["vn0_debias"], rtol=1e-6)
    _set_vn(vn, "vn0")
    mean, std = mo.valuenorm_mean_std(Z["vn0_mean"][0], Z["vn0_mean_sq"][0], Z["vn0_debias"])
    p = vn.denorm_params().numpy()
    assert p[0] == mean and p[1] == std
    x = torch.from_numpy(Z["buf_value_preds"])
    np.testing.assert_array_equal(vn.denormalize(x).numpy(), Z["buf_value_preds"] * std + mean)


def _filled_buffer(cfg):
    from buffer.shared_buffer import SharedReplayBuffer
    buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A))
    buf.obs.copy_(torch.from_numpy(Z["buf_obs"]))
    buf.actions.copy_(torch.from_numpy(Z["buf_actions"]))
    buf.action_log_probs.copy_(torch.from_numpy(Z["buf_logp"]))
    buf.rewards.copy_(torch.from_numpy(Z["buf_rewards"]))
    buf.value_preds.copy_(torch.from_numpy(Z["buf_value_preds_after"]))
    buf.masks.copy_(torch.from_numpy(Z["buf_masks"]))
    buf.returns.copy_(torch.from_numpy(Z["returns"]))   # the HIP GAE kernel is tested on the GPU
    return buf


def test_share_obs_is_a_view_with_reference_shape():
    buf = _filled_buffer(make_cfg())
    so = buf.share_obs
    assert tuple(so.shape) == (T + 1, E, N, S)
    ref = np.repeat(Z["buf_obs"].reshape(T + 1, E, 1, S), N, axis=2)
    np.testing.assert_array_equal(so.numpy(), ref)
    assert buf.share_obs_env.data_ptr() == buf.obs.data_ptr()      # no copy


@pytest.mark.parametrize("cache", [False, True])
@pytest.mark.parametrize("dedup", [False, True])
def test_train_matches_reference(dedup, cache):
    cfg = make_cfg(dedup_critic=dedup, cache_normalized_inputs=cache)
    pol, tr = _policy(cfg)
    _set_vn(tr.value_normalizer, "vn0")
    buf = _filled_buffer(cfg)
    adv = tr.normalized_advantages(buf)
    np.testing.assert_allclose(adv.numpy(), Z["adv_norm"], rtol=2e-5, atol=2e-6)
    tr.prep_training()
    info = tr.train(buf, update_actor=True)
    for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"):
        np.testing.assert_allclose(info[k], float(Z["info_" + k]), rtol=2e-4, atol=1e-6, err_msg=k)
    _check_updates(pol, "train[dedup=%s,cache=%s]" % (dedup, cache))
    np.testing.assert_allclose(tr.value_normalizer.running_mean.numpy(), Z["vn1_mean"], rtol=1e-5)
    np.testing.assert_allclose(tr.value_normalizer.debiasing_term.numpy(), Z["vn1_debias"], rtol=1e-6)


def _check_updates(pol, label, tol=DELTA_TOL):
    """Post-update parameters as UPDATES (conftest.check_param_deltas) + the absolute window as a second line."""
    for pre0, pre1, mod in (("actor/", "actor2/", pol.actor), ("critic/", "critic2/", pol.critic)):
        before = {k[len(pre0):]: Z[k] for k in Z.files if k.startswith(pre0)}
        after = {k[len(pre1):]: Z[k] for k in Z.files if k.startswith(pre1)}
        check_param_deltas(mod, before, after, tol, "mappo_small " + label + " " + pre1[:-2], abs_tol=2e-5)


def test_surrogate_doubling_flag():
    """Q4: with the reference's [.,2] log-prob layout the policy loss is exactly twice the [.,1] one."""
    cfg = make_cfg()
    pol, tr = _policy(cfg)
    _set_vn(tr.value_normalizer, "vn0")
    buf = _filled_buffer(cfg)
    adv = tr.normalized_advantages(buf)
    s = next(buf.feed_forward_generator(adv, 1))
    tr.prep_training()
    torch.manual_seed(0)
    _, _, pl2, _, _, imp = tr.ppo_update(s)
    assert imp.shape[-1] == 2
    pol1, tr1 = _policy(cfg)
    _set_vn(tr1.value_normalizer, "vn0")
    s1 = list(s); s1[9] = s[9][:, :1].contiguous()
    _, _, pl1, _, _, imp1 = tr1.ppo_update(tuple(s1))
    assert imp1.shape[-1] == 1
    np.testing.assert_allclose(float(pl2), 2 * float(pl1), rtol=1e-5)


def test_buffer_gae_has_no_cpu_path():
    import dcc_hip
    cfg = make_cfg()
    buf = _filled_buffer(cfg)
    from utils.valuenorm import ValueNorm
    with pytest.raises(dcc_hip.DccError):
        buf.compute_returns(torch.zeros(E, N, 1), ValueNorm(1))


@pytest.mark.parametrize("k", [3, 12])
@pytest.mark.parametrize("dedup", [False, True])
def test_mini_batch_generator_partitions_the_rollout(dedup, k):
    """num_mini_batch > 1 follows the reference's feed_forward_generator (shared_buffer.py:239-279): one
    permutation of the T*E*N agent rows drawn with torch.randperm on the CPU generator, cut into num_mini_batch row sets;
    every field is the gather of those rows, the critic rows stay aligned with the agent rows."""
    cfg = make_cfg(num_mini_batch=k, dedup_critic=dedup)
    buf = _filled_buffer(cfg)
    B = T * E * N
    # k = 3: 64-row mini-batches over 48 (step, env) pairs touch most pairs -> every pair is handed out (static shapes);
    # k = 12: 16 rows -> the unique touched pairs
    adv = torch.arange(B, dtype=torch.float32).view(T, E, N, 1)     # row id as the advantage
    seen = []
    torch.manual_seed(0)
    want = torch.randperm(B)                                          # what the reference's generator would draw after this seed
    torch.manual_seed(0)
    for i, s in enumerate(buf.feed_forward_generator(adv, k, dedup_critic=dedup)):
        share, obs, acts, ids = s[0], s[1], s[4], s[10].view(-1).long()
        assert torch.equal(ids, want[i * (B // k):(i + 1) * (B // k)])                 # the reference's row sets, in its order
        np.testing.assert_array_equal(obs.numpy(), Z["buf_obs"][:-1].reshape(B, D)[ids.numpy()])
        np.testing.assert_array_equal(acts.numpy(), Z["buf_actions"].reshape(B, A)[ids.numpy()])
        np.testing.assert_array_equal(s[5].numpy(), buf.value_preds[:-1].reshape(B, 1)[ids].numpy())
        np.testing.assert_array_equal(s[9].numpy(), buf.action_log_probs.reshape(B, -1)[ids].numpy())
        so_rows = Z["buf_obs"][:-1].reshape(T * E, S)[(ids // N).numpy()]            # the centralised row of each agent row
        if dedup:                                                                      # one row per touched (step, env) pair + selectors
            row_sel, pair_sel = s[12]
            assert row_sel is None and share.shape[0] == (T * E if k == 3 else ids.div(N, rounding_mode="floor").unique().numel())
            np.testing.assert_array_equal(share[pair_sel].numpy(), so_rows)
        else:                                                                          # the reference's 12-tuple exactly
            assert len(s) == 12
            np.testing.assert_array_equal(share.numpy(), so_rows)
        seen.append(ids)
    allids = torch.cat(seen)
    assert allids.numel() == (B // k) * k and allids.unique().numel() == allids.numel()


def test_train_with_two_mini_batches_runs():
    cfg = make_cfg(num_mini_batch=2)
    pol, tr = _policy(cfg)
    _set_vn(tr.value_normalizer, "vn0")
    info = tr.train(_filled_buffer(cfg))
    assert all(np.isfinite(v) for v in info.values())


# ---- chunked full-batch update / compact-state buffer ----------------------------------
@pytest.mark.parametrize("chunk", [1, 3, 1000])
@pytest.mark.parametrize("dedup", [False, True])
def test_chunked_update_matches_reference(dedup, chunk):
    """Gradient accumulation over chunks of rollout steps is the same full-batch PPO step: the reference's
    post-update parameters, losses and ValueNorm state are reproduced to the same tolerances."""
    cfg = make_cfg(dedup_critic=dedup, update_chunk_steps=chunk)
    pol, tr = _policy(cfg)
    _set_vn(tr.value_normalizer, "vn0")
    buf = _filled_buffer(cfg)
    tr.prep_training()
    info = tr.train(buf, update_actor=True)
    for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"):
        np.testing.assert_allclose(info[k], float(Z["info_" + k]), rtol=2e-4, atol=1e-6, err_msg=k)
    _check_updates(pol, "chunked[dedup=%s,chunk=%d]" % (dedup, chunk))
    np.testing.assert_allclose(tr.value_normalizer.running_mean.numpy(), Z["vn1_mean"], rtol=1e-5)
    np.testing.assert_allclose(tr.value_normalizer.debiasing_term.numpy(), Z["vn1_debias"], rtol=1e-6)


def _compact_buffer(cfg, n_pois=3):
    """Compact buffer whose 'expander' is a table lookup keyed by state_energy[:, 0] (the real one is the HIP
    kernel dcc_obs_expand, tested on the GPU): exercises the storage / chunking logic on the CPU."""
    from buffer.shared_buffer import SharedReplayBuffer
    table = torch.from_numpy(Z["buf_obs"]).reshape((T + 1) * E, N, D)
    calls = []

    def expander(pos, vel, energy, done, out):
        assert pos.dtype == torch.float64 and done.dtype == torch.uint8 and pos.shape[0] == out.shape[0]
        calls.append(out.shape[0])
        out.copy_(table[energy[:, 0].long()])
        return out

    buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A), compact=True, n_pois=n_pois, expander=expander)
    buf.state_energy[:, :, 0] = torch.arange((T + 1) * E, dtype=torch.float32).view(T + 1, E)
    for name, key in (("actions", "buf_actions"), ("action_log_probs", "buf_logp"), ("rewards", "buf_rewards"),
                      ("value_preds", "buf_value_preds_after"), ("masks", "buf_masks"), ("returns", "returns")):
        getattr(buf, name).copy_(torch.from_numpy(Z[key]))
    return buf, calls


def test_compact_buffer_update_matches_reference():
    cfg = make_cfg(update_chunk_steps=4)
    pol, tr = _policy(cfg)
    _set_vn(tr.value_normalizer, "vn0")
    buf, calls = _compact_buffer(cfg)
    assert not torch.is_tensor(buf.obs) and buf.obs_cur.shape == (E, N, D)       # no [T+1,E,N,D] array is resident
    full = (T + 1) * E * N * D * 4
    compact = sum(a.numel() * a.element_size() for a in (buf.state_pos, buf.state_vel, buf.state_energy, buf.state_done))
    assert compact == (T + 1) * E * (32 * N + 5 * 3) and compact < full
    tr.prep_training()
    info = tr.train(buf, update_actor=True)
    n_chunks = -(-T // 4)
    assert len(calls) == n_chunks * cfg.ppo_epoch and max(calls) == 4 * E     # one chunk of observations at a time
    for k in ("value_loss", "policy_loss", "dist_entropy", "actor_grad_norm", "critic_grad_norm", "ratio"):
        np.testing.assert_allclose(info[k], float(Z["info_" + k]), rtol=2e-4, atol=1e-6, err_msg=k)
    _check_updates(pol, "compact-buffer")


def test_compact_buffer_answers_the_reference_attribute_names():
    """SURVEY.md 8b B3 in the shipped default (compact_obs: true): `buffer.obs[t]` / `buffer.share_obs[t]` as the
    reference's learner reads them (learner.py:233-234, :280) are regenerated from the stored state; rows WRITTEN through
    the reference's API (learner.py:224-225 `share_obs[0] = ..; obs[0] = ..`, insert(share_obs, obs, ...)) switch the
    buffer to row storage, after which it behaves like the reference's buffer."""
    cfg = make_cfg()
    buf, calls = _compact_buffer(cfg)
    ref = Z["buf_obs"]
    assert tuple(buf.obs.shape) == (T + 1, E, N, D) and tuple(buf.share_obs.shape) == (T + 1, E, N, S) and len(buf.obs) == T + 1
    np.testing.assert_array_equal(buf.obs[3].numpy(), ref[3])
    np.testing.assert_array_equal(buf.obs[-1].numpy(), ref[-1])
    np.testing.assert_array_equal(buf.obs[2:5].numpy(), ref[2:5])
    np.testing.assert_array_equal(buf.obs[4, 1].numpy(), ref[4, 1])
    so = buf.share_obs[5]
    assert tuple(so.shape) == (E, N, S)
    np.testing.assert_array_equal(so.numpy(), np.repeat(ref[5].reshape(E, 1, S), N, axis=1))
    np.testing.assert_array_equal(np.concatenate(buf.share_obs[-1].numpy()), np.repeat(ref[-1].reshape(E, 1, S), N, axis=1).reshape(E * N, S))
    a, b = buf.obs[1], buf.obs[2]
    assert a.data_ptr() != b.data_ptr()                     # every read is its own tensor
    with pytest.raises(RuntimeError):
        buf.obs.numpy()                                     # the whole array is not resident
    assert buf.compact and calls
    # --- the reference's warmup: rows from outside -> row storage
    new0 = np.random.RandomState(0).randn(E, N, D).astype(np.float32)
    with pytest.warns(UserWarning, match="switching to row storage"):
        buf.share_obs[0] = np.repeat(new0.reshape(E, 1, S), N, axis=1).copy()
        buf.obs[0] = new0.copy()
    assert not buf.compact and not buf.structured and torch.is_tensor(buf.obs) and tuple(buf.obs.shape) == (T + 1, E, N, D)
    np.testing.assert_array_equal(buf.obs[0].numpy(), new0)
    np.testing.assert_array_equal(buf.obs[1:].numpy(), ref[1:])                 # the other slots were filled from the state
    np.testing.assert_array_equal(buf.share_obs[0].numpy(), np.repeat(new0.reshape(E, 1, S), N, axis=1))
    assert buf.share_obs_env.data_ptr() == buf.obs.data_ptr() and tuple(buf.share_obs.numpy().shape) == (T + 1, E, N, S)
    # --- the reference's insert(share_obs, obs, ...) on a fresh state-only buffer
    buf2, _ = _compact_buffer(cfg)
    rows = np.random.RandomState(1).randn(E, N, D).astype(np.float32)
    z1 = np.zeros((E, N, 1), np.float32)
    with pytest.warns(UserWarning):
        buf2.insert(np.repeat(rows.reshape(E, 1, S), N, axis=1), rows, None, None, np.zeros((E, N, A), np.float32),
                    np.zeros((E, N, 1), np.float32), z1, z1, np.ones((E, N, 1), np.float32))
    np.testing.assert_array_equal(buf2.obs[1].numpy(), rows)
    sample = next(buf2.feed_forward_generator(torch.zeros(T, E, N, 1), 1, dedup_critic=True))     # the reference's generator works now
    assert tuple(sample[1].shape) == (T * E * N, D)


def test_compact_buffer_slots_and_guards():
    cfg = make_cfg()
    buf, _ = _compact_buffer(cfg)
    o1 = buf.obs_slot(1)
    assert o1.data_ptr() == buf.obs_cur.data_ptr() and buf.obs_at(1) is buf.obs_cur
    with pytest.raises(RuntimeError):
        buf.obs_at(0)                       # only the newest slot's observations are resident
    with pytest.raises(RuntimeError):
        buf.share_obs_env
    with pytest.raises(RuntimeError):
        next(buf.feed_forward_generator(torch.zeros(T, E, N, 1)))
    st = buf.state_slot(2)
    assert st["state_pos"].data_ptr() == buf.state_pos[2].data_ptr() and st["state_done"].dtype == torch.uint8
    buf.set_state_slot(0, dict(pos=torch.ones(E, N, 2, dtype=torch.float64), vel=torch.zeros(E, N, 2, dtype=torch.float64),
                               energy=torch.full((E, 3), 7.0), done=torch.ones(E, 3, dtype=torch.uint8)))
    assert float(buf.state_pos[0].sum()) == E * N * 2 and float(buf.state_energy[0, 0, 0]) == 7.0
    buf.obs_slot(T)
    buf.state_pos[-1].fill_(5.0)
    buf.after_update()
    assert float(buf.state_pos[0, 0, 0, 0]) == 5.0 and buf.obs_at(0) is buf.obs_cur
    # the non-compact buffer answers the same slot API
    plain = _filled_buffer(cfg)
    assert plain.obs_slot(3).data_ptr() == plain.obs[3].data_ptr() and plain.state_slot(3) == {}
    assert plain.share_obs_env_at(2).data_ptr() == plain.obs[2].data_ptr()


@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_load_model_accepts_a_checkpoint_written_by_the_reference(dev):
    """tests/golden/ref_agent_small/agent.pkl was written by the reference's MAPPOTrainer.save_model (a pickle of its
    MAPPOPolicy object, tools/gen_golden_ref_checkpoint.py; same seed as mappo_small.npz).  load_model takes over the
    parameters (dropping the dead mlp.fc_h template) although classes the pickle mentions do not exist here.  On the device
    (`-m gpu`) the loaded policy then reproduces the reference's own evaluate_actions / act outputs."""
    import utils.pytorch_utils as ptu
    pol, tr = _policy(make_cfg(), dev)
    for p in list(pol.actor.parameters()) + list(pol.critic.parameters()):
        assert p.device.type == dev
        p.data.add_(1.0)                                        # make sure values really come from the file
    tr.load_model(os.path.join(GOLDEN, "ref_agent_small"))
    for name, net in (("actor/", pol.actor), ("critic/", pol.critic)):
        for k, v in net.state_dict().items():
            assert v.device.type == dev
            np.testing.assert_array_equal(v.cpu().numpy(), Z[name + k], err_msg=k)
    # parameters are still the views of the flat optimizer storage (the load copies in place)
    opt = pol.actor_optimizer
    assert all(p.data_ptr() == opt.flat_param.data_ptr() + 4 * off for p, off in zip(opt._params, opt._offsets))
    # the loaded networks compute what the reference computed with these parameters (reference fixtures ev_*)
    tr.prep_rollout()
    t = lambda a: torch.from_numpy(a).to(ptu.device)
    with torch.no_grad():
        v, logp, ent = pol.evaluate_actions(t(Z["ev_sobs"]), t(Z["ev_obs"]), None, None, t(Z["ev_act"]), None, None,
                                            torch.ones(64, 1, device=ptu.device))
        mean_act, _ = pol.act(t(Z["ev_obs"]), deterministic=True)
    np.testing.assert_allclose(v.cpu().numpy(), Z["ev_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(logp.cpu().numpy(), Z["ev_logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(ent), float(Z["ev_entropy"]), rtol=1e-6)
    np.testing.assert_allclose(mean_act.cpu().numpy(), Z["ev_mean_act"], rtol=1e-5, atol=1e-7)
    # round trip through this package's own format
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr.save_model(d)
        pol2, tr2 = _policy(make_cfg(), dev)
        pol2.actor.act.action_out.logstd._bias.data.fill_(3.0)
        tr2.load_model(d)
        assert float(pol2.actor.act.action_out.logstd._bias.abs().sum()) == 0.0
    ptu.set_gpu_mode(False)


def test_double_surrogate_false_uses_one_log_prob_column():
    """cfg.double_surrogate (advisor finding: the flag used to be dead).  With one column the policy surrogate -- and its
    gradient -- is half the reference's doubled one; value loss and entropy are untouched."""
    out = {}
    for flag in (True, False):
        pol, tr = _policy(make_cfg(double_surrogate=flag, ppo_epoch=1))
        buf = _filled_buffer(make_cfg())
        _set_vn(tr.value_normalizer, "vn0")
        adv = torch.from_numpy(Z["adv_norm"])
        sample = next(buf.feed_forward_generator(adv, 1, dedup_critic=True))
        pl, ent, vl, imp = tr._forward_losses(sample)
        out[flag] = (float(pl), float(ent), float(vl), tuple(imp.shape))
    assert out[True][3][1] == 2 and out[False][3][1] == 1
    np.testing.assert_allclose(out[False][0], 0.5 * out[True][0], rtol=1e-6)
    assert out[False][1] == out[True][1] and out[False][2] == out[True][2]


def test_flat_adam_reattaches_parameters_that_left_its_storage():
    """Advisor finding (r02): kernels write parameters through raw pointers into flat_param, so a parameter whose storage was
    moved behind the optimizer's back (module._apply / .to(), GRU.flatten_parameters, p.data = ...) must not silently keep
    computing with a detached tensor.  zero_grad / clip_and_step compare pointers (host side), re-attach a moved parameter
    with its current values, and raise when dtype / size changed."""
    from algos.algo_utils.optim import FlatAdam
    torch.manual_seed(0)
    lin = torch.nn.Linear(5, 3)
    opt = FlatAdam(lin.parameters(), lr=1e-2)
    attached = lambda: all(p.data_ptr() == opt.flat_param.data_ptr() + 4 * off for p, off in zip(opt._params, opt._offsets))
    assert attached()
    lin.weight.data = lin.weight.data.clone() + 1.0          # moved, with values the network now computes with
    moved = lin.weight.data.clone()
    assert not attached()
    opt.zero_grad()
    assert attached() and torch.equal(lin.weight.data, moved)
    lin(torch.randn(4, 5)).sum().backward()
    before = lin.weight.data.clone()
    lin.bias.data = lin.bias.data.clone()                    # moved between backward and the step
    opt.clip_and_step(10.0)
    assert attached() and not torch.equal(lin.weight.data, before)       # the step reached the tensors the module uses
    lin.weight.data = lin.weight.data.double()
    with pytest.raises(RuntimeError, match="left the optimizer's flat storage"):
        opt.zero_grad()


def test_checkpoint_unpickler_only_hides_known_parameter_free_classes(tmp_path):
    """Advisor finding (r02): a reference checkpoint mentioning gym spaces loads (those objects carry no parameters), one
    mentioning a network class this package does not build names that class instead of failing later in state_dict()."""
    import pickle, sys, types
    from algos.mappo import _CheckpointUnpickler, _Opaque
    def dump_with_fake(modname, clsname, path):
        mod = types.ModuleType(modname)
        cls = type(clsname, (object,), {"__module__": modname})
        setattr(mod, clsname, cls)
        parents = modname.split(".")
        added = []
        for i in range(1, len(parents) + 1):
            name = ".".join(parents[:i])
            if name not in sys.modules:
                sys.modules[name] = mod if i == len(parents) else types.ModuleType(name)
                added.append(name)
        try:
            if modname in sys.modules and sys.modules[modname] is not mod:
                setattr(sys.modules[modname], clsname, cls)
            with open(path, "wb") as f:
                pickle.dump({"x": cls()}, f)
        finally:
            for name in added:
                sys.modules.pop(name, None)
            if modname in sys.modules and hasattr(sys.modules[modname], clsname):
                delattr(sys.modules[modname], clsname)
    ok, bad = str(tmp_path / "ok.pkl"), str(tmp_path / "bad.pkl")
    dump_with_fake("gym.spaces.box", "Box", ok)
    dump_with_fake("algos.algo_utils.cnn", "CNNBase", bad)
    with open(ok, "rb") as f:
        assert isinstance(_CheckpointUnpickler(f).load()["x"], _Opaque)
    with open(bad, "rb") as f, pytest.raises(pickle.UnpicklingError, match="algos.algo_utils.cnn.CNNBase"):
        _CheckpointUnpickler(f).load()


def test_chunk_clamp_counts_the_widest_tensor_of_a_chunk():
    """Advisor finding (r02): the 2^31-element guard of the chunked update must use the observation width on the dense
    first-layer path (c5 shard: D = 5186 >> hidden), not only [rows, hidden]."""
    cfg = make_cfg(update_chunk_steps=1000, ppo_epoch=1)
    pol, tr = _policy(cfg)

    class Stop(Exception):
        pass

    class FakeBuf:            # only what train() reads before the first epoch
        compact, structured = True, False
        episode_length, n_rollout_threads, num_agents = 150, 2048, 32
        obs_dim, share_obs_dim = 5186, 32 * 5186
        returns = value_preds = torch.zeros(151, 1, 1, 1)
        active_masks = torch.ones(151, 1, 1, 1)
    tr._epoch = lambda *a: (_ for _ in ()).throw(Stop())
    with pytest.raises(Stop):
        tr.train(FakeBuf())
    assert tr.update_chunk_steps == (2 ** 31 - 1) // (2048 * 32 * 5186) == 6
    FakeBuf.structured = True
    FakeBuf.features_rows = lambda self, a, b: None
    tr.update_chunk_steps = 1000
    with pytest.raises(Stop):
        tr.train(FakeBuf())
    assert tr.update_chunk_steps == 1000          # features are narrow: [rows, hidden] = 65,536 x 32 per step, no clamp needed
