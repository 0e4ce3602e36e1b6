import numpy as np
import torch


def init(module, weight_init, bias_init, gain=1):
    """Initialise a Linear in place (reference: algos/algo_utils/util.py:7-10)."""
    weight_init(module.weight.data, gain=gain)
    bias_init(module.bias.data)
    return module


def check(x):
    return torch.from_numpy(x) if isinstance(x, np.ndarray) else x
