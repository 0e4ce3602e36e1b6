"""Layer initialisation helper of the MLP / action heads."""


def init(module, weight_init, bias_init, gain=1):
    """Apply `weight_init(weight, gain=gain)` and `bias_init(bias)` to a Linear in place and return it
    (same call shape as the reference's algos/algo_utils/util.py:7-10, so layers are built identically)."""
    weight_init(module.weight.data, gain=gain)
    bias_init(module.bias.data)
    return module
