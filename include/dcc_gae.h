/*
 * dcc_gae.h -- C-ABI of the device-side GAE / returns scan in libdcc_hip.so.
 *
 * Replaces (reference paths relative to uav_dcc_control/):
 *   SharedReplayBuffer.compute_returns, live branch  buffer/shared_buffer.py:199-208      (dcc_gae_compute)
 *   ... and its three other branches                 buffer/shared_buffer.py:167-197,209-217 (dcc_returns_compute)
 *   ValueNorm.denormalize                            utils/valuenorm.py:68-79
 *   the advantage line of MAPPOTrainer.train         algos/mappo.py:190-191
 * The reference runs a T-long Python loop over numpy float32 arrays and round-trips through torch
 * inside denormalize() at every step; here one launch walks every (env, agent) column backwards
 * in time: a first-order linear recurrence segmented by `masks` (0 at episode ends).
 *
 * Arithmetic is float32 in exactly the reference's operation order (numpy promotes the Python
 * float gamma / gamma*lambda to float32 scalars), so results are bit-identical to the reference.
 * All pointers are DEVICE pointers; the call is asynchronous on `stream`.
 */
#ifndef DCC_GAE_H
#define DCC_GAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef DCC_API
#if defined(__GNUC__)
#define DCC_API __attribute__((visibility("default")))
#else
#define DCC_API
#endif
#endif

/*
 * rewards      [T,   C] float32            (C = E*N columns; the trailing 1 of the reference's
 * value_preds  [T+1, C] float32             [T,E,N,1] arrays is dropped)
 *              row T must already hold the bootstrap value V(s_T) (shared_buffer.py:200)
 * masks        [T+1, C] float32            0 where the episode ended before that step
 * denorm       [2] float32 or NULL         {mean, sqrt(var)} of ValueNorm.running_mean_var()
 *                                          (valuenorm.py:32-36); NULL = values are not normalised
 * returns      [T+1, C] float32 out        rows 0..T-1 written (row T untouched, as in the reference)
 * advantages   [T,   C] float32 out or NULL  returns[t] - denorm(value_preds[t]) (mappo.py:191)
 */
DCC_API int dcc_gae_compute(const float* rewards, const float* value_preds, const float* masks,
                            const float* denorm, double gamma, double gae_lambda,
                            float* returns, float* advantages, int32_t T, int64_t C, void* stream);

/*
 * Every branch of SharedReplayBuffer.compute_returns (buffer/shared_buffer.py:160-217) as one backward scan per column.
 * `mode` selects the recurrence (cfg.use_gae, cfg.use_proper_time_limits of config/algo_config/mappo.yaml):
 *   DCC_RETURNS_GAE                         :199-213   = dcc_gae_compute (the shipped configuration)
 *   DCC_RETURNS_GAE | DCC_RETURNS_PROPER    :167-185   the same with gae *= bad_masks[t+1] after every step
 *   DCC_RETURNS_PROPER                      :186-197   returns[t] = (returns[t+1] gamma masks[t+1] + rewards[t]) bad[t+1]
 *                                                                    + (1 - bad[t+1]) denorm(value_preds[t])
 *   0                                       :214-217   returns[t] = returns[t+1] gamma masks[t+1] + rewards[t]
 * Arguments as for dcc_gae_compute, plus
 *   bad_masks   [T+1, C] float32 or NULL    0 where an episode was cut by the time limit (shared_buffer.py:67,98-99);
 *                                           required with DCC_RETURNS_PROPER
 *   returns     without DCC_RETURNS_GAE row T is READ: the caller stores the bootstrap value there first
 *                                           (`self.returns[-1] = next_value`, :187,215 -- the critic's raw output even
 *                                           when the values are normalised, as in the reference); with it row T of
 *                                           value_preds is read as in dcc_gae_compute
 * float32 in the reference's operation order per branch (bit-identical to its numpy loop).
 */
#define DCC_RETURNS_GAE 1
#define DCC_RETURNS_PROPER 2
DCC_API int dcc_returns_compute(const float* rewards, const float* value_preds, const float* masks, const float* bad_masks,
                                const float* denorm, double gamma, double gae_lambda, int32_t mode,
                                float* returns, float* advantages, int32_t T, int64_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCC_GAE_H */
