"""numpy restatement of the reference's returns / advantage arithmetic (TEST INFRASTRUCTURE ONLY).

Pinned against tests/golden/mappo_small.npz, which tools/gen_golden_mappo.py produced by running the
reference's own SharedReplayBuffer.compute_returns / ValueNorm (tests/test_mappo_golden.py).
Reference paths relative to uav_dcc_control/.
"""
import numpy as np


def valuenorm_mean_std(running_mean, running_mean_sq, debias, epsilon=1e-5):
    """utils/valuenorm.py:32-36 in float32 (torch CPU semantics): returns (mean, sqrt(var))."""
    f = np.float32
    db = max(f(debias), f(epsilon))
    mean = f(running_mean) / db
    mean_sq = f(running_mean_sq) / db
    var = max(f(mean_sq - mean * mean), f(1e-2))
    return f(mean), f(np.sqrt(f(var)))


def compute_returns_gae(rewards, value_preds, masks, next_value, gamma, gae_lambda, mean=None, std=None):
    """buffer/shared_buffer.py:199-208 (use_gae, use_valuenorm, no proper time limits) in numpy
    float32, statement by statement.  rewards [T,...], value_preds/masks [T+1,...].
    Returns (returns [T+1,...], value_preds with the bootstrap row set)."""
    rewards = np.asarray(rewards, np.float32)
    vp = np.array(value_preds, np.float32, copy=True)
    masks = np.asarray(masks, np.float32)
    vp[-1] = next_value
    T = rewards.shape[0]
    returns = np.zeros_like(vp)

    def denorm(v):  # valuenorm.py:75: v * sqrt(var) + mean (two float32 ops)
        if mean is None:
            return v
        return v * np.float32(std) + np.float32(mean)

    gae = 0
    for step in reversed(range(T)):
        delta = rewards[step] + gamma * denorm(vp[step + 1]) * masks[step + 1] - denorm(vp[step])
        gae = delta + gamma * gae_lambda * masks[step + 1] * gae
        returns[step] = gae + denorm(vp[step])
    return returns, vp


def compute_returns(rewards, value_preds, masks, bad_masks, next_value, gamma, gae_lambda, use_gae, use_proper_time_limits,
                    mean=None, std=None):
    """Every branch of buffer/shared_buffer.py:160-217 in numpy float32, statement by statement (mean / std None: the
    `else` arms without ValueNorm).  Returns (returns [T+1,...], value_preds as the method leaves them).  Pinned against
    tests/golden/returns_modes.npz (tools/gen_golden_returns.py ran the reference's own method for the 8 flag combinations)."""
    rewards = np.asarray(rewards, np.float32)
    vp = np.array(value_preds, np.float32, copy=True)
    masks, bad_masks = np.asarray(masks, np.float32), np.asarray(bad_masks, np.float32)
    T = rewards.shape[0]
    returns = np.zeros_like(vp)
    vn = mean is not None

    def denorm(v):
        return v * np.float32(std) + np.float32(mean)

    if use_proper_time_limits:
        if use_gae:
            vp[-1] = next_value
            gae = 0
            for step in reversed(range(T)):
                if vn:                                                                                       # :171-178
                    delta = rewards[step] + gamma * denorm(vp[step + 1]) * masks[step + 1] - denorm(vp[step])
                    gae = delta + gamma * gae_lambda * gae * masks[step + 1]
                    gae = gae * bad_masks[step + 1]
                    returns[step] = gae + denorm(vp[step])
                else:                                                                                        # :180-185
                    delta = rewards[step] + gamma * vp[step + 1] * masks[step + 1] - vp[step]
                    gae = delta + gamma * gae_lambda * masks[step + 1] * gae
                    gae = gae * bad_masks[step + 1]
                    returns[step] = gae + vp[step]
        else:
            returns[-1] = next_value
            for step in reversed(range(T)):                                                                  # :188-197
                v = denorm(vp[step]) if vn else vp[step]
                returns[step] = (returns[step + 1] * gamma * masks[step + 1] + rewards[step]) * bad_masks[step + 1] \
                    + (1 - bad_masks[step + 1]) * v
    else:
        if use_gae:
            return compute_returns_gae(rewards, vp, masks, next_value, gamma, gae_lambda, mean, std)        # :199-213
        returns[-1] = next_value
        for step in reversed(range(T)):                                                                      # :215-217
            returns[step] = returns[step + 1] * gamma * masks[step + 1] + rewards[step]
    return returns, vp
