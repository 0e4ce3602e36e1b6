"""ValueNorm: debiased exponential moving mean / mean-square of the return targets.

Reference: uav_dcc_control/utils/valuenorm.py:8-79 (beta 0.99999, epsilon 1e-5, var clamp 1e-2).
Device-resident: statistics never leave the GPU (`denormalize` returns a tensor, not numpy, and
`denorm_params()` hands {mean, sqrt(var)} to the HIP GAE kernel as a 2-float device tensor).
With torch.distributed initialised the batch moments are averaged over ranks (equal batch sizes),
which equals the single-process statistics of the concatenated batch.
"""
import torch
import torch.nn as nn


class ValueNorm(nn.Module):
    def __init__(self, input_shape=1, norm_axes=1, beta=0.99999, per_element_update=False, epsilon=1e-5,
                 device=torch.device("cpu")):
        super().__init__()
        self.input_shape, self.norm_axes = input_shape, norm_axes
        self.beta, self.epsilon, self.per_element_update = beta, epsilon, per_element_update
        self.register_buffer("running_mean", torch.zeros(input_shape, dtype=torch.float32, device=device))
        self.register_buffer("running_mean_sq", torch.zeros(input_shape, dtype=torch.float32, device=device))
        self.register_buffer("debiasing_term", torch.zeros((), dtype=torch.float32, device=device))

    def running_mean_var(self):
        """valuenorm.py:32-36."""
        db = self.debiasing_term.clamp(min=self.epsilon)
        mean = self.running_mean / db
        mean_sq = self.running_mean_sq / db
        var = (mean_sq - mean ** 2).clamp(min=1e-2)
        return mean, var

    @torch.no_grad()
    def update(self, x):
        """valuenorm.py:38-55; `x` is [B, 1] (norm_axes leading dims are reduced)."""
        x = torch.as_tensor(x).to(self.running_mean.device, torch.float32)
        dims = tuple(range(self.norm_axes))
        batch_mean = x.mean(dim=dims)
        batch_sq_mean = (x ** 2).mean(dim=dims)
        import torch.distributed as dist
        import utils.pytorch_utils as ptu
        if ptu.dist_active():
            both = torch.stack([batch_mean.reshape(-1), batch_sq_mean.reshape(-1)])
            ptu.all_reduce(both)
            both /= dist.get_world_size()
            batch_mean, batch_sq_mean = both[0].reshape(batch_mean.shape), both[1].reshape(batch_sq_mean.shape)
        if self.per_element_update:
            n = 1
            for d in x.shape[:self.norm_axes]:
                n *= d
            weight = self.beta ** n
        else:
            weight = self.beta
        self.running_mean.mul_(weight).add_(batch_mean * (1.0 - weight))
        self.running_mean_sq.mul_(weight).add_(batch_sq_mean * (1.0 - weight))
        self.debiasing_term.mul_(weight).add_(1.0 * (1.0 - weight))

    def normalize(self, x):
        x = torch.as_tensor(x).to(self.running_mean.device, torch.float32)
        mean, var = self.running_mean_var()
        return (x - mean[(None,) * self.norm_axes]) / torch.sqrt(var)[(None,) * self.norm_axes]

    def denormalize(self, x):
        """valuenorm.py:68-79 without the .cpu().numpy() round trip."""
        x = torch.as_tensor(x).to(self.running_mean.device, torch.float32)
        mean, var = self.running_mean_var()
        return x * torch.sqrt(var)[(None,) * self.norm_axes] + mean[(None,) * self.norm_axes]

    def denorm_params(self):
        """[mean, sqrt(var)] as a contiguous float32 device tensor (input of dcc_gae_compute)."""
        mean, var = self.running_mean_var()
        return torch.stack([mean.reshape(-1)[0], torch.sqrt(var).reshape(-1)[0]]).contiguous()
