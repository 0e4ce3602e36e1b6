/*
 * dcc_optim.h -- C-ABI of the fused optimizer step of the MAPPO update in libdcc_hip.so.
 *
 * The reference ends every PPO step with, per network, `nn.utils.clip_grad_norm_(params, 10)` and `Adam.step()`
 * (uav_dcc_control/algos/mappo.py:176-185; optimizers built at mappo.py:27-37 with lr, eps = opti_eps, weight_decay).
 * Issued through PyTorch that is ~25 small launches per network and step (per-tensor norms, stack, norm, clamp, scale,
 * then the foreach Adam chain).  Here the parameters, gradients and both Adam moments of a network live in FLAT float32
 * arrays (the host side keeps the nn.Parameters and their .grad as views into them, 16-byte aligned), and a step is
 * three launches:
 *
 *   dcc_grad_norm_clip   out[0] = ||grad||_2 over the flat array, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))
 *                        (torch.nn.utils.clip_grad_norm_ semantics; max_norm <= 0: out[1] = 1, norm only = get_gard_norm,
 *                        utils/util.py:20-26).  Two stages with a fixed summation order: bit-reproducible run to run.
 *   dcc_adam_step        g = clip * grad (+ weight_decay * p);  m = lerp(m, g, 1-beta1);  v = beta2 v + (1-beta2) g^2;
 *                        p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)      -- torch.optim.Adam's update rule, with the
 *                        bias corrections step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t) evaluated by the
 *                        caller in double precision like torch does.  `clip` is read from DEVICE memory (out + 1 of the call
 *                        above), so no host synchronisation sits between the two.
 *
 * All pointers are device pointers to contiguous float32; calls are asynchronous on `stream`; return 0 or a negative
 * DCC_E* code with the message in dcc_last_error().  The same flat gradient array is what the multi-GPU path hands to
 * RCCL (one all-reduce per network, no packing copies).
 */
#ifndef DCC_OPTIM_H
#define DCC_OPTIM_H

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

/* Floats of `workspace` dcc_grad_norm_clip needs for n elements. */
DCC_API int64_t dcc_grad_norm_workspace_floats(int64_t n);

/* out [2] = {||grad||_2, clip coefficient}; workspace: dcc_grad_norm_workspace_floats(n) floats. */
DCC_API int dcc_grad_norm_clip(const float* grad, int64_t n, float max_norm, float* out, float* workspace, void* stream);

/* One Adam step on n elements in place (param, exp_avg, exp_avg_sq); clip: device pointer to the scale applied to the
 * gradient (NULL = 1).  beta1 / beta2 are doubles: 1 - beta is formed in double precision like torch forms it. */
DCC_API int dcc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float step_size,
                          float bc2_sqrt, double beta1, double beta2, float eps, float weight_decay, const float* clip,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCC_OPTIM_H */
