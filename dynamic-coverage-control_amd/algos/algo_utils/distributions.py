"""Diagonal Gaussian action head (reference: algos/algo_utils/distributions.py:31-41,72-92,108-119):
mean = Linear(H, A) (orthogonal init, gain cfg.gain), state-independent log-std parameter (zeros)."""
import math

import torch
import torch.nn as nn

from .util import init


class AddBias(nn.Module):
    """Holds the log-std as `_bias` of shape [A, 1] (the reference's parameter name and shape)."""

    def __init__(self, bias):
        super().__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        return x + self._bias.t().view(1, -1)


_noise_source = None


def set_noise_source(fn):
    """Replace the standard-normal draws of action sampling: fn(shape, dtype, device) -> tensor, None restores torch.randn.
    The reference draws from torch's global generator (torch.distributions.Normal.sample, act.py:79-84); this seam lets a
    recorded noise sequence of a reference run be replayed through this package's rollout (tests/test_learner_reference_replay.py).
    Returns the previous source.  A TEST SEAM: installing a source needs DCC_TESTING=1 (utils/pytorch_utils.require_testing)."""
    global _noise_source
    if fn is not None:
        import utils.pytorch_utils as ptu
        ptu.require_testing("distributions.set_noise_source")
    prev, _noise_source = _noise_source, fn
    return prev


def standard_normal(shape, dtype, device):
    """eps ~ N(0, 1) of action sampling (a = mean + std * eps), from the active noise source."""
    if _noise_source is not None:
        return _noise_source(tuple(shape), dtype, device)
    return torch.randn(shape, dtype=dtype, device=device)


class FixedNormal(torch.distributions.Normal):
    def sample(self, sample_shape=torch.Size()):
        """mean + std * eps.  torch.normal(mean, std) -- what Normal.sample calls -- checks `std.min() >= 0` on the
        host: a device synchronisation per rollout step and illegal inside a hipGraph capture."""
        shape = self._extended_shape(sample_shape)
        with torch.no_grad():
            return self.loc.expand(shape) + self.scale.expand(shape) * standard_normal(shape, self.loc.dtype, self.loc.device)

    def log_probs(self, actions):
        return super().log_prob(actions).sum(-1, keepdim=True)

    def mode(self):
        return self.mean


class DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs, use_orthogonal=True, gain=0.01):
        super().__init__()
        init_method = nn.init.orthogonal_ if use_orthogonal else nn.init.xavier_uniform_
        self.fc_mean = init(nn.Linear(num_inputs, num_outputs), init_method, lambda b: nn.init.constant_(b, 0), gain)
        self.logstd = AddBias(torch.zeros(num_outputs))

    def forward(self, x, mean=None):
        if mean is None:
            mean = self.fc_mean(x)
        logstd = self.logstd(torch.zeros_like(mean))
        # validate_args=False: the default argument validation does `(scale > 0).all()` on the host -- a device
        # synchronisation per call (and illegal inside a hipGraph capture); exp() is positive by construction
        return FixedNormal(mean, logstd.exp(), validate_args=False)


LOG_SQRT_2PI = 0.5 * math.log(2 * math.pi)
