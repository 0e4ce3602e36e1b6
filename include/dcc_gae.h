/*
 * dcc_gae.h -- C-ABI of the device-side GAE / returns scan in libdcc_hip.so.
 *
 * Replaces (reference paths relative to uav_dcc_control/):
 *   SharedReplayBuffer.compute_returns, live branch  buffer/shared_buffer.py:199-208
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

#ifdef __cplusplus
}
#endif
#endif /* DCC_GAE_H */
