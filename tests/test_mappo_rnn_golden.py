"""Recurrent MAPPO variants (reference buffer/shared_buffer.py:281-487, algos/algo_utils/rnn.py,
r_actor_critic.py with use_recurrent_policy / use_naive_recurrent_policy) against fixtures produced by the reference
itself (tools/gen_golden_mappo_rnn.py).  Every test runs twice: on CPU torch (host-side logic, `-m "not gpu"`) and on the
device (`-m gpu`: MIOpen GRU, device gathers of both generators, the fused loss / optimizer kernels in the update)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, check_param_deltas
from test_mappo_golden import make_cfg, Box

Z = np.load(os.path.join(GOLDEN, "mappo_rnn_small.npz"))
N, E, T, D, A, H, L, MB = 3, 4, 12, 10, 2, 16, 4, 2
S, B = N * D, E * N
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
FIELDS = ("share_obs", "obs", "rnn_states", "rnn_states_critic", "actions", "value_preds", "returns", "masks",
          "active_masks", "old_logp", "adv")


def _cfg(mode, **over):
    return make_cfg(num_agents=N, n_rollout_threads=E, max_ep_len=T, algo_hidden_size=H, ppo_epoch=2, num_mini_batch=MB,
                    data_chunk_length=L, recurrent_N=1, use_recurrent_policy=(mode == "chunk"),
                    use_naive_recurrent_policy=(mode == "naive"), dedup_critic=False, **over)


def _sd(prefix):
    return {k[len(prefix):]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith(prefix)}


def _dev():
    import utils.pytorch_utils as ptu
    return ptu.device


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _np(t):
    return t.detach().cpu().numpy()


def _setup(mode, dev="cpu"):
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(dev == "cuda", 0)
    from algos.mappo import MAPPOPolicy, MAPPOTrainer
    from buffer.shared_buffer import SharedReplayBuffer
    cfg = _cfg(mode)
    pol = MAPPOPolicy(cfg, Box(D), Box(S), Box(A))
    for net, name in ((pol.actor, "actor"), (pol.critic, "critic")):
        missing, unexpected = net.load_state_dict(_sd("%s/%s/" % (mode, name)), strict=False)
        assert not missing and all(".fc_h." in k for k in unexpected), (missing, unexpected)
    tr = MAPPOTrainer(cfg, pol)
    buf = SharedReplayBuffer(cfg, Box(D), Box(S), Box(A))
    assert buf.obs.device.type == dev and next(pol.actor.parameters()).device.type == dev
    buf.obs.copy_(_t(Z["buf_obs"]))
    buf.actions.copy_(_t(Z["buf_actions"]))
    buf.action_log_probs.copy_(_t(Z["buf_logp"]).expand_as(buf.action_log_probs))
    buf.rewards.copy_(_t(Z["buf_rewards"]))
    buf.value_preds.copy_(_t(Z["buf_value_preds"]))
    buf.masks.copy_(_t(Z["buf_masks"]))
    buf.rnn_states.copy_(_t(Z["buf_rnn_states"]))
    buf.rnn_states_critic.copy_(_t(Z["buf_rnn_states_critic"]))
    return cfg, pol, tr, buf


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("mode", ["chunk", "naive"])
def test_rnn_layer_sequence_and_single_step_match_reference(mode, dev):
    """RNNLayer: the mask-every-step formulation (no host round trip) equals the reference's segment-wise GRU calls on
    a sequence with episode ends inside; the single-step (rollout) path returns the reference's new states."""
    cfg, pol, tr, buf = _setup(mode, dev)
    assert buf.rnn_states.shape == (T + 1, E, N, 1, H)
    tr.prep_rollout()
    pre = mode + "/"
    obs, masks = Z["buf_obs"], Z["buf_masks"]
    ev_obs = _t(obs[:-1].reshape(T * B, D))
    ev_sobs = buf.share_obs[:-1].reshape(T * B, S)
    with torch.no_grad():
        v, lp, ent = pol.evaluate_actions(ev_sobs, ev_obs, buf.rnn_states[0].reshape(B, 1, H), buf.rnn_states_critic[0].reshape(B, 1, H),
                                          _t(Z["buf_actions"].reshape(T * B, A)),
                                          _t(masks[:-1].reshape(T * B, 1)), None, torch.ones(T * B, 1, device=_dev()))
        sv, sa, _, sra, src = pol.get_actions(buf.share_obs[2].reshape(B, S), buf.obs[2].reshape(B, D),
                                              buf.rnn_states[2].reshape(B, 1, H), buf.rnn_states_critic[2].reshape(B, 1, H),
                                              buf.masks[2].reshape(B, 1), deterministic=True)
    np.testing.assert_allclose(_np(v), Z[pre + "ev_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_np(lp), Z[pre + "ev_logp"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(ent), float(Z[pre + "ev_entropy"]), rtol=1e-6)
    np.testing.assert_allclose(_np(sv), Z[pre + "step_values"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_np(sa), Z[pre + "step_actions"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(_np(sra), Z[pre + "step_rnn_actor"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_np(src), Z[pre + "step_rnn_critic"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("mode", ["chunk", "naive"])
def test_recurrent_generators_yield_the_reference_minibatches(mode, dev):
    """Same seed -> same permutation -> the same mini-batches, field by field, bit for bit (they are gathers)."""
    cfg, pol, tr, buf = _setup(mode, dev)
    pre = mode + "/"
    buf.returns.copy_(_t(Z[pre + "returns"]))
    adv = _t(Z[pre + "adv_norm"])
    torch.manual_seed(21)
    gen = buf.recurrent_generator(adv, MB, L) if mode == "chunk" else buf.naive_recurrent_generator(adv, MB)
    n = 0
    for i, sample in enumerate(gen):
        for name, arr in zip(FIELDS, sample[:11]):
            ref = Z["%sgen%d_%s" % (pre, i, name)]
            assert arr.device.type == dev
            got = _np(arr)
            assert got.shape == ref.shape, (name, got.shape, ref.shape)
            np.testing.assert_array_equal(got, ref, err_msg="%s minibatch %d" % (name, i))
        assert sample[11] is None
        n += 1
    assert n == MB
    # an explicit permutation selects the same rows as the seeded draw
    gen2 = (buf.recurrent_generator(adv, MB, L, perm=Z[pre + "gen_perm"]) if mode == "chunk"
            else buf.naive_recurrent_generator(adv, MB, perm=Z[pre + "gen_perm"]))
    first = next(iter(gen2))
    np.testing.assert_array_equal(_np(first[1]), Z[pre + "gen0_obs"])


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("mode", ["chunk", "naive"])
def test_recurrent_train_reproduces_the_reference_update(mode, dev):
    """MAPPOTrainer.train with the recurrent generators (2 epochs x 2 mini-batches): losses, gradient norms and the
    post-update parameters of actor / critic (GRU weights included) equal the reference's."""
    cfg, pol, tr, buf = _setup(mode, dev)
    pre = mode + "/"
    vn = tr.value_normalizer
    vn.running_mean.copy_(_t(Z[pre + "vn0_mean"]))
    vn.running_mean_sq.copy_(_t(Z[pre + "vn0_mean_sq"]))
    vn.debiasing_term.copy_(_t(Z[pre + "vn0_debias"]).reshape(vn.debiasing_term.shape))
    buf.returns.copy_(_t(Z[pre + "returns"]))
    np.testing.assert_allclose(_np(tr.normalized_advantages(buf)), Z[pre + "adv_norm"], rtol=2e-5, atol=2e-6)
    tr.prep_training()
    torch.manual_seed(3)
    info = tr.train(buf, update_actor=True)
    for k, v in info.items():
        np.testing.assert_allclose(v, float(Z[pre + "info_" + k]), rtol=2e-4, atol=1e-6, err_msg=k)
    for name, net in (("actor", pol.actor), ("critic", pol.critic)):
        sd = net.state_dict()
        for k, v in sd.items():
            np.testing.assert_allclose(_np(v), Z["%s%s2/%s" % (pre, name, k)], rtol=2e-4, atol=2e-6, err_msg=k)
        # ... and as UPDATES: max |d_got - d_ref| <= tol * max |d_ref| per tensor (achieved: run summary)
        check_param_deltas(net, {k: Z["%s%s/%s" % (pre, name, k)] for k in sd}, {k: Z["%s%s2/%s" % (pre, name, k)] for k in sd},
                           1e-3 if dev == "cuda" else 3e-4, "mappo_rnn %s %s %s" % (mode, dev, name))      # achieved: 1.12e-4 / 4.5e-5
    assert "rnn.rnn.weight_hh_l0" in pol.actor.state_dict() and "rnn.norm.weight" in pol.critic.state_dict()


def test_recurrent_buffer_refuses_state_only_storage():
    import utils.pytorch_utils as ptu
    ptu.set_gpu_mode(False)
    from buffer.shared_buffer import SharedReplayBuffer
    with pytest.raises(ValueError):
        SharedReplayBuffer(_cfg("chunk"), Box(D), Box(S), Box(A), compact=True, n_pois=1, expander=lambda *a: None)
    buf = SharedReplayBuffer(make_cfg(), Box(20), Box(80), Box(2))
    with pytest.raises(RuntimeError):
        next(buf.recurrent_generator(torch.zeros(16, 3, 4, 1), 1, 4))
