"""Actor and centralised critic of MAPPO (reference: uav_dcc_control/algos/r_actor_critic.py:19-121).

Both are MLPBase trunks (LayerNorm -> Linear/ReLU/LayerNorm x2) on PyTorch-ROCm: the three GEMMs
per network are the only dense contractions of the whole hot path and run on MFMA through
hipBLASLt.  Inputs are expected to live on the device already (the rollout never leaves the GPU);
numpy inputs are still accepted for drop-in use.  The recurrent variants (`use_recurrent_policy`,
`use_naive_recurrent_policy`: a GRU + LayerNorm between trunk and head, r_actor_critic.py:36-37,52-53,100-101,117-118)
are built (algo_utils/rnn.py) although the shipped config keeps them off; the CNN / PopArt variants are not
(disabled by mappo.yaml, cfg keys accepted and must be off).
"""
import torch
import torch.nn as nn

from algos.algo_utils.act import ACTLayer
from algos.algo_utils.mlp import MLPBase
from algos.algo_utils.rnn import RNNLayer
from algos.algo_utils import structured
from algos.algo_utils.util import init
from utils.util import check
from utils.util import get_shape_from_obs_space


def _require_mlp(cfg, shape):
    if len(shape) != 1:
        raise NotImplementedError("only flat observations are on the coverage hot path")


def _recurrent(cfg):
    return bool(cfg.use_recurrent_policy or cfg.use_naive_recurrent_policy)


class R_Actor(nn.Module):
    def __init__(self, cfg, obs_space, action_space, device=torch.device("cpu")):
        super().__init__()
        self.hidden_size = cfg.algo_hidden_size
        self._use_policy_active_masks = cfg.use_policy_active_masks
        self.tpdv = dict(dtype=torch.float32, device=device)
        obs_shape = get_shape_from_obs_space(obs_space)
        _require_mlp(cfg, obs_shape)
        self.base = MLPBase(cfg, obs_shape)
        self.rnn = RNNLayer(self.hidden_size, self.hidden_size, cfg.recurrent_N, cfg.use_orthogonal) if _recurrent(cfg) else None
        self.act = ACTLayer(action_space, self.hidden_size, cfg.use_orthogonal, cfg.gain)
        self.obs_layout = None    # set by MAPPOPolicy.enable_structured_input
        self.to(device)

    def _mean(self, obs, prenormalized=False, rnn_states=None, masks=None, want_states=False, row_sel=None):
        """Gaussian mean [B, A] = fc_mean(trunk(obs)) (the head is fused with the trunk's last block on the GPU).
        obs: rows [B, D], or the dict of compact features of B/N env states (algo_utils/structured.py).
        Recurrent variants: fc_mean(rnn(trunk(obs), rnn_states, masks)); want_states also returns the new states.
        row_sel: keep only these of the rows `obs` describes (row mini-batches over per-env features: the selection is applied
        right after the first block, so the 256 x 256 block and the head run on the selected rows only)."""
        head = self.act.action_out.fc_mean
        trunk_head = head if self.rnn is None else None
        if isinstance(obs, dict):
            if self.obs_layout is None:
                raise RuntimeError("compact features passed to an actor without an observation layout")
            out = structured.actor_trunk(self.base, self.obs_layout, obs, trunk_head, row_sel)
        else:
            obs = check(obs).to(**self.tpdv)
            if row_sel is not None:
                obs = obs.index_select(0, row_sel)
            out = self.base.forward_prenormalized(obs, trunk_head) if prenormalized else self.base(obs, trunk_head)
        if self.rnn is not None:
            feats, rnn_states = self.rnn(out, check(rnn_states).to(**self.tpdv), check(masks).to(**self.tpdv))
            out = head(feats)
        return (out, rnn_states) if want_states else out

    def forward(self, obs, rnn_states=None, masks=None, available_actions=None, deterministic=False):
        mean, rnn_states = self._mean(obs, rnn_states=rnn_states, masks=masks, want_states=True)
        actions, logp = self.act(None, available_actions, deterministic, mean=mean)
        return actions, logp, rnn_states

    def evaluate_actions(self, obs, rnn_states, action, masks, available_actions=None, active_masks=None,
                         prenormalized=False):
        """prenormalized: `obs` already went through MLPBase.normalize_input (see mlp.py)."""
        action = check(action).to(**self.tpdv)
        if active_masks is not None:
            active_masks = check(active_masks).to(**self.tpdv)
        return self.act.evaluate_actions(None, action, available_actions,
                                         active_masks=active_masks if self._use_policy_active_masks else None,
                                         mean=self._mean(obs, prenormalized, rnn_states, masks))


class R_Critic(nn.Module):
    def __init__(self, cfg, cent_obs_space, device=torch.device("cpu")):
        super().__init__()
        self.hidden_size = cfg.algo_hidden_size
        if cfg.use_popart:
            raise NotImplementedError("PopArt is disabled in the reference config and not built")
        self.tpdv = dict(dtype=torch.float32, device=device)
        shape = get_shape_from_obs_space(cent_obs_space)
        _require_mlp(cfg, shape)
        self.base = MLPBase(cfg, shape)
        self.rnn = RNNLayer(self.hidden_size, self.hidden_size, cfg.recurrent_N, cfg.use_orthogonal) if _recurrent(cfg) else None
        init_method = nn.init.orthogonal_ if cfg.use_orthogonal else nn.init.xavier_uniform_
        self.v_out = init(nn.Linear(self.hidden_size, 1), init_method, lambda b: nn.init.constant_(b, 0))
        self.obs_layout = None
        self.to(device)

    def forward(self, cent_obs, rnn_states=None, masks=None, prenormalized=False):
        """cent_obs: rows [B, N*D], or the dict of compact features of B env states (one value per env)."""
        head = self.v_out if self.rnn is None else None
        if isinstance(cent_obs, dict):
            if self.obs_layout is None:
                raise RuntimeError("compact features passed to a critic without an observation layout")
            v = structured.critic_trunk(self.base, self.obs_layout, cent_obs, head)
        else:
            cent_obs = check(cent_obs).to(**self.tpdv)
            v = self.base.forward_prenormalized(cent_obs, head) if prenormalized else self.base(cent_obs, head)
        if self.rnn is not None:
            feats, rnn_states = self.rnn(v, check(rnn_states).to(**self.tpdv), check(masks).to(**self.tpdv))
            v = self.v_out(feats)
        return v, rnn_states
