/*
 * dcc_gae_cpu.c -- the `_cpu` twin of include/dcc_gae.h's entry point.  TEST INFRASTRUCTURE: part of
 * oracle/libdcc_oracle.so, never linked into or loaded by the product.  Same signature as dcc_gae_compute with HOST
 * pointers and `stream` ignored.
 *
 * Restates, column by column in float32 and in the reference's operation order (paths relative to uav_dcc_control/):
 *   SharedReplayBuffer.compute_returns, live branch   buffer/shared_buffer.py:199-208
 *       gae = 0;  for step in reversed(range(T)):
 *           delta = rewards[step] + gamma * denorm(value_preds[step+1]) * masks[step+1] - denorm(value_preds[step])
 *           gae = delta + gamma * gae_lambda * masks[step+1] * gae
 *           returns[step] = gae + denorm(value_preds[step])
 *   ValueNorm.denormalize                              utils/valuenorm.py:68-79     v * sqrt(var) + mean (two float32 ops)
 *   the advantage line of MAPPOTrainer.train           algos/mappo.py:190-191       returns[:-1] - denorm(value_preds[:-1])
 * numpy turns the Python floats into float32 scalars when they meet float32 arrays: gamma -> (float)gamma, and
 * `gamma * gae_lambda` is formed in double first (left to right) and then rounded.  Built with -ffp-contract=off, so no
 * multiply-add is fused.  Pinned bit-exact to tests/golden/mappo_small.npz (tests/test_cpu_twin_abi.py).
 */
#include <stddef.h>
#include <stdint.h>

#include "../include/dcc_gae.h"

DCC_API int dcc_gae_compute_cpu(const float *rewards, const float *value_preds, const float *masks, const float *denorm,
                                double gamma, double gae_lambda, float *returns, float *advantages, int32_t T, int64_t C,
                                void *stream)
{
    (void)stream;
    if (!rewards || !value_preds || !masks || !returns) return -1;
    if (T < 1 || C < 1) return -1;
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    const int dn = denorm != NULL;
    const float mean = dn ? denorm[0] : 0.f, sd = dn ? denorm[1] : 1.f;
    for (int64_t c = 0; c < C; ++c) {
        float v_next = value_preds[(int64_t)T * C + c];           /* shared_buffer.py:200 put the bootstrap value there */
        if (dn) v_next = v_next * sd + mean;
        float gae = 0.f;
        for (int32_t t = T - 1; t >= 0; --t) {
            const float r = rewards[(int64_t)t * C + c], m = masks[(int64_t)(t + 1) * C + c];
            float v_cur = value_preds[(int64_t)t * C + c];
            if (dn) v_cur = v_cur * sd + mean;
            const float delta = (r + (g * v_next) * m) - v_cur;      /* :203-205 */
            gae = delta + (gl * m) * gae;                            /* :206 */
            const float ret = gae + v_cur;                           /* :207 */
            returns[(int64_t)t * C + c] = ret;
            if (advantages) advantages[(int64_t)t * C + c] = ret - v_cur;
            v_next = v_cur;
        }
    }
    return 0;
}


/*
 * dcc_returns_compute_cpu -- twin of dcc_returns_compute: every branch of SharedReplayBuffer.compute_returns
 * (buffer/shared_buffer.py:160-217), one column at a time, float32 in the reference's operation order:
 *   use_proper_time_limits, use_gae   :167-185   delta as in the live branch;
 *                                                with ValueNorm  gae = delta + gamma * gae_lambda * gae * masks[step+1]   (:176)
 *                                                without         gae = delta + gamma * gae_lambda * masks[step+1] * gae   (:183)
 *                                                gae = gae * bad_masks[step+1];  returns[step] = gae + denorm(value_preds[step])
 *   use_proper_time_limits, no gae    :186-197   returns[-1] = next_value;
 *                                                returns[step] = (returns[step+1] * gamma * masks[step+1] + rewards[step]) * bad_masks[step+1]
 *                                                                + (1 - bad_masks[step+1]) * denorm(value_preds[step])
 *   neither                           :214-217   returns[-1] = next_value;  returns[step] = returns[step+1] * gamma * masks[step+1] + rewards[step]
 *   use_gae only                      :199-213   dcc_gae_compute_cpu above
 * Pinned bit-exact to tests/golden/returns_modes.npz (the reference's own loops, tools/gen_golden_returns.py).
 */
DCC_API int dcc_returns_compute_cpu(const float *rewards, const float *value_preds, const float *masks, const float *bad_masks,
                                    const float *denorm, double gamma, double gae_lambda, int32_t mode, float *returns,
                                    float *advantages, int32_t T, int64_t C, void *stream)
{
    if (mode == DCC_RETURNS_GAE)
        return dcc_gae_compute_cpu(rewards, value_preds, masks, denorm, gamma, gae_lambda, returns, advantages, T, C, stream);
    if (mode < 0 || mode > (DCC_RETURNS_GAE | DCC_RETURNS_PROPER)) return -1;
    if (!rewards || !value_preds || !masks || !returns) return -1;
    if ((mode & DCC_RETURNS_PROPER) && !bad_masks) return -1;
    if (T < 1 || C < 1) return -1;
    const int use_gae = (mode & DCC_RETURNS_GAE) != 0, ptl = (mode & DCC_RETURNS_PROPER) != 0;
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    const int dn = denorm != NULL;
    const float mean = dn ? denorm[0] : 0.f, sd = dn ? denorm[1] : 1.f;
    for (int64_t c = 0; c < C; ++c) {
        float carry = use_gae ? value_preds[(int64_t)T * C + c] : returns[(int64_t)T * C + c];
        if (use_gae && dn) carry = carry * sd + mean;
        float gae = 0.f;
        for (int32_t t = T - 1; t >= 0; --t) {
            const float r = rewards[(int64_t)t * C + c], m = masks[(int64_t)(t + 1) * C + c];
            const float b = ptl ? bad_masks[(int64_t)(t + 1) * C + c] : 1.f;
            float v_cur = value_preds[(int64_t)t * C + c];
            if (dn) v_cur = v_cur * sd + mean;
            float ret;
            if (use_gae) {
                const float delta = (r + (g * carry) * m) - v_cur;
                if (dn) gae = delta + (gl * gae) * m;                /* :176 */
                else gae = delta + (gl * m) * gae;                   /* :183 */
                gae = gae * b;                                       /* :177,184 */
                ret = gae + v_cur;
                carry = v_cur;
            } else {
                const float disc = (carry * g) * m + r;
                ret = ptl ? disc * b + (1.f - b) * v_cur : disc;     /* :190-197 | :217 */
                carry = ret;
            }
            returns[(int64_t)t * C + c] = ret;
            if (advantages) advantages[(int64_t)t * C + c] = ret - v_cur;
        }
    }
    return 0;
}
