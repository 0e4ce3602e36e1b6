"""Small numeric helpers of the MAPPO path (reference: uav_dcc_control/utils/util.py)."""
import math
import random

import numpy as np
import torch


def seed(_seed):
    """util.py:7-12."""
    random.seed(_seed)
    torch.manual_seed(_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(_seed)
    np.random.seed(_seed)


def check(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x


def get_gard_norm(params):
    """L2 norm over all gradients (util.py:20-26; the reference's spelling is kept)."""
    tot = 0.0
    for p in params:
        if p.grad is not None:
            tot += float(p.grad.norm()) ** 2
    return math.sqrt(tot)


def update_linear_schedule(optimizer, epoch, total_num_epochs, initial_lr):
    """lr = lr0 * (1 - epoch/total), floored at 0 (util.py:29-33)."""
    lr = max(initial_lr - (initial_lr * (epoch / float(total_num_epochs))), 0.0)
    for g in optimizer.param_groups:
        g["lr"] = lr


def huber_loss(e, d):
    """ONE-SIDED Huber exactly as the reference writes it (util.py:36-39): the linear
    branch is selected by `e > d`, not `|e| > d`, so errors below -d contribute 0."""
    a = (e.abs() <= d).to(e.dtype)
    b = (e > d).to(e.dtype)
    return a * e ** 2 / 2 + b * d * (e.abs() - d / 2)


def mse_loss(e):
    return e ** 2 / 2


def get_shape_from_obs_space(obs_space):
    if obs_space.__class__.__name__ == "Box":
        return tuple(obs_space.shape)
    if isinstance(obs_space, (list, tuple)):
        return tuple(obs_space)
    raise NotImplementedError(type(obs_space))


def get_shape_from_act_space(act_space):
    name = act_space.__class__.__name__
    if name == "Discrete":
        return 1
    if name == "Box":
        return act_space.shape[0]
    raise NotImplementedError(name)
