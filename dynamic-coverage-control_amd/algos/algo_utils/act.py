"""Action layer (reference: algos/algo_utils/act.py:5-184).  Only the branch the coverage task
uses is built: a continuous `Box` action space -> DiagGaussian (act.py:22-25,79-84,165-184)."""
import torch.nn as nn

from .distributions import DiagGaussian


class ACTLayer(nn.Module):
    def __init__(self, action_space, inputs_dim, use_orthogonal, gain):
        super().__init__()
        if action_space.__class__.__name__ != "Box":
            raise NotImplementedError("only continuous Box action spaces are on the coverage hot path, got %s"
                                      % action_space.__class__.__name__)
        self.continuous_action = True
        self.action_out = DiagGaussian(inputs_dim, action_space.shape[0], use_orthogonal, gain)

    def forward(self, x, available_actions=None, deterministic=False, mean=None):
        """mean: the Gaussian mean fc_mean(x) when the caller already has it (fused with the trunk), else from x."""
        dist = self.action_out(x, mean=mean)
        actions = dist.mode() if deterministic else dist.sample()
        return actions, dist.log_probs(actions)

    def evaluate_actions(self, x, action, available_actions=None, active_masks=None, mean=None):
        """log pi(a|s) summed over action dims [B,1]; entropy summed over dims and averaged over the
        (active) batch -- act.py:173-179 multiplies the per-dim entropy [B,A] by the mask [B,1]."""
        dist = self.action_out(x, mean=mean)
        logp = dist.log_probs(action)
        ent = dist.entropy()
        if active_masks is not None:
            ent = (ent * active_masks).sum() / active_masks.sum()
        else:
            ent = ent.mean()
        return logp, ent
