"""MLP trunk of actor and critic: LayerNorm(in) -> [Linear -> ReLU -> LayerNorm] x (1 + layer_N).

Reference: uav_dcc_control/algos/algo_utils/mlp.py:7-58.  Parameter names (`feature_norm`, `mlp.fc1`,
`mlp.fc2.<i>`) match the reference so that its state_dicts load; the reference's registered-but-
never-run `fc_h` template (mlp.py:21-23, 66,304 dead parameters per network) is not
created -- it would have `None` gradients and only complicate the gradient all-reduce.
"""
import torch.nn as nn
import torch.nn.functional as F

from . import fused
from .util import init


def _block(in_dim, hidden, use_orthogonal, use_relu):
    act = nn.ReLU() if use_relu else nn.Tanh()
    init_method = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
    gain = nn.init.calculate_gain("relu" if use_relu else "tanh")
    lin = init(nn.Linear(in_dim, hidden), init_method, lambda b: nn.init.constant_(b, 0), gain=gain)
    return nn.Sequential(lin, act, nn.LayerNorm(hidden))


class MLPLayer(nn.Module):
    def __init__(self, input_dim, hidden_size, layer_N, use_orthogonal, use_ReLU):
        super().__init__()
        self._layer_N = layer_N
        self.fc1 = _block(input_dim, hidden_size, use_orthogonal, use_ReLU)
        self.fc2 = nn.ModuleList([_block(hidden_size, hidden_size, use_orthogonal, use_ReLU) for _ in range(layer_N)])

    @staticmethod
    def run_block(blk, x):
        """Linear -> act -> LayerNorm; the ReLU + LayerNorm tail is one fused HIP pass on the GPU (dcc_mlp.h)."""
        if isinstance(blk[1], nn.ReLU):
            return fused.relu_ln(fused.linear_nobias(x, blk[0]), blk[0].bias, blk[2])
        return blk(x)

    def forward(self, x, head=None):
        """head: optional narrow nn.Linear applied to the features; fused with the last block's tail on the GPU."""
        blocks = [self.fc1] + list(self.fc2)
        for i, blk in enumerate(blocks):
            if head is not None and i == len(blocks) - 1 and isinstance(blk[1], nn.ReLU):
                return fused.relu_ln_head(fused.linear_nobias(x, blk[0]), blk[0].bias, blk[2], head)
            x = self.run_block(blk, x)
        return x if head is None else head(x)


class MLPBase(nn.Module):
    def __init__(self, cfg, obs_shape):
        super().__init__()
        self._use_feature_normalization = cfg.use_feature_normalization
        self.hidden_size = cfg.algo_hidden_size
        obs_dim = obs_shape[0]
        if self._use_feature_normalization:
            self.feature_norm = nn.LayerNorm(obs_dim)
        self.mlp = MLPLayer(obs_dim, self.hidden_size, cfg.layer_N, cfg.use_orthogonal, cfg.use_ReLU)

    def forward(self, x, head=None):
        if self._use_feature_normalization:
            x = self.feature_norm(x)
        return self.mlp(x, head)

    # ---- input-normalisation cache for the PPO epochs ---------------------------------------------------
    # LayerNorm(x) = xhat * gamma + beta with xhat = (x - mean) / sqrt(var + eps) independent of the
    # parameters.  The observations of a rollout do not change over the ppo_epoch passes, so xhat is computed
    # once per iteration and the affine part is folded into the first Linear:
    #     W (xhat * gamma + beta) + b  =  (W * gamma) xhat + (W beta + b)
    # which removes one LayerNorm forward + backward over the [B, obs_dim] input per epoch and network
    # (the largest memory-bound kernels of the update).  Same function, gradients for gamma / beta flow
    # through the folded weight and bias.
    def normalize_input(self, x):
        if not self._use_feature_normalization:
            return x
        ln = self.feature_norm
        return F.layer_norm(x, ln.normalized_shape, None, None, ln.eps)

    def forward_prenormalized(self, xhat, head=None):
        lin = self.mlp.fc1[0]
        if self._use_feature_normalization:
            ln = self.feature_norm
            w = lin.weight * ln.weight.unsqueeze(0)
            b = lin.bias + lin.weight @ ln.bias
        else:
            w, b = lin.weight, lin.bias
        relu1 = isinstance(self.mlp.fc1[1], nn.ReLU)
        blocks = list(self.mlp.fc2)
        if relu1:
            z = fused.linear_w(xhat, w)
            if head is not None and not blocks:
                return fused.relu_ln_head(z, b, self.mlp.fc1[2], head)
            h = fused.relu_ln(z, b, self.mlp.fc1[2])
        else:
            h = self.mlp.fc1[2](self.mlp.fc1[1](F.linear(xhat, w, b)))
        for i, blk in enumerate(blocks):
            if head is not None and i == len(blocks) - 1 and isinstance(blk[1], nn.ReLU):
                return fused.relu_ln_head(fused.linear_nobias(h, blk[0]), blk[0].bias, blk[2], head)
            h = self.mlp.run_block(blk, h)
        return h if head is None else head(h)
