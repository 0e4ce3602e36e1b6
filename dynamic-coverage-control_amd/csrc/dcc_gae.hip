// dcc_gae.hip -- GAE(gamma, lambda) returns (and the reference's three other return branches) as a backwards, mask-segmented
// scan on the device.
//
// One lane per (env, agent) column; columns are contiguous in memory ([T, E, N] row-major), so
// every load/store of a wavefront is one coalesced 256-byte transaction.  The recurrence runs in
// float32 in the reference's exact operation order (buffer/shared_buffer.py:199-208 with the
// torch float32 `v * sqrt(var) + mean` of utils/valuenorm.py:75 folded in), compiled with
// -ffp-contract=off, so the result is bit-identical to the reference's numpy loop.  The loads do not
// depend on the recurrence, so 16 steps of them are kept in flight ahead of it (software prefetch).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "dcc_gae.h"
#include "dcc_internal.h"

namespace {

// U steps of loads are in flight per lane while the previous U steps' recurrence runs: the loads do not depend on the
// recurrence, so the kernel is bound by HBM bandwidth (20 B per step and column), not by one memory latency per step.
constexpr int kGaeAhead = 16;
constexpr int kGaeBlock = 64;   // one wavefront per workgroup: C/64 workgroups spread over all CUs at c3 (32,768 columns)

struct GaeChunk {
    float r[kGaeAhead], v[kGaeAhead], m[kGaeAhead];
};

// steps t0, t0-1, .., t0-U+1 of column c (steps below 0 are not read)
__device__ __forceinline__ void gae_fetch(GaeChunk& q, const float* __restrict__ rewards, const float* __restrict__ vpred,
                                          const float* __restrict__ masks, int t0, long long C, long long c) {
#pragma unroll
    for (int u = 0; u < kGaeAhead; ++u) {
        const int t = t0 - u;
        if (t >= 0) {
            q.r[u] = rewards[(long long)t * C + c];
            q.v[u] = vpred[(long long)t * C + c];
            q.m[u] = masks[(long long)(t + 1) * C + c];
        }
    }
}

__global__ __launch_bounds__(kGaeBlock) void dcc_gae_kernel(const float* __restrict__ rewards,
                                                            const float* __restrict__ vpred,
                                                            const float* __restrict__ masks,
                                                            const float* __restrict__ denorm, float gamma, float gl,
                                                            float* __restrict__ returns, float* __restrict__ adv, int T,
                                                            long long C) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean = 0.f, sd = 1.f;
    const bool dn = denorm != nullptr;
    if (dn) { mean = denorm[0]; sd = denorm[1]; }
    // denormalised V(s_{t+1}) carried across iterations
    float v_next = vpred[(long long)T * C + c];
    if (dn) v_next = v_next * sd + mean;
    float gae = 0.f;
    GaeChunk cur, nxt;
    gae_fetch(cur, rewards, vpred, masks, T - 1, C, c);
    for (int t0 = T - 1; t0 >= 0; t0 -= kGaeAhead) {
        gae_fetch(nxt, rewards, vpred, masks, t0 - kGaeAhead, C, c);
#pragma unroll
        for (int u = 0; u < kGaeAhead; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const float r = cur.r[u], m = cur.m[u];
                const float v_cur = dn ? (cur.v[u] * sd + mean) : cur.v[u];
                // delta = r[t] + gamma * V[t+1] * mask[t+1] - V[t]          (shared_buffer.py:203-205)
                const float delta = (r + (gamma * v_next) * m) - v_cur;
                // gae = delta + gamma*lambda * mask[t+1] * gae               (shared_buffer.py:206)
                gae = delta + (gl * m) * gae;
                const float ret = gae + v_cur;                              // shared_buffer.py:207
                returns[(long long)t * C + c] = ret;
                if (adv) adv[(long long)t * C + c] = ret - v_cur;           // mappo.py:191
                v_next = v_cur;
            }
        }
        cur = nxt;
    }
}

// ---- the other branches of compute_returns (buffer/shared_buffer.py:167-197,209-217) -------------------------------------------
// Same layout and look-ahead as above, the recurrence chosen at compile time.  Off the shipped configuration (use_gae: true,
// use_proper_time_limits: false runs dcc_gae_kernel), so one generic body instead of four tuned ones.
template <bool PTL>
struct RetChunk {
    float r[kGaeAhead], v[kGaeAhead], m[kGaeAhead], b[PTL ? kGaeAhead : 1];
};

template <bool PTL>
__device__ __forceinline__ void ret_fetch(RetChunk<PTL>& q, const float* __restrict__ rewards, const float* __restrict__ vpred,
                                          const float* __restrict__ masks, const float* __restrict__ bad, int t0, long long C,
                                          long long c) {
#pragma unroll
    for (int u = 0; u < kGaeAhead; ++u) {
        const int t = t0 - u;
        if (t >= 0) {
            q.r[u] = rewards[(long long)t * C + c];
            q.v[u] = vpred[(long long)t * C + c];
            q.m[u] = masks[(long long)(t + 1) * C + c];
            if constexpr (PTL) q.b[u] = bad[(long long)(t + 1) * C + c];
        }
    }
}

template <bool GAE, bool PTL>
__global__ __launch_bounds__(kGaeBlock) void dcc_returns_kernel(const float* __restrict__ rewards, const float* __restrict__ vpred,
                                                                const float* __restrict__ masks, const float* __restrict__ bad,
                                                                const float* __restrict__ denorm, float gamma, float gl,
                                                                float* __restrict__ returns, float* __restrict__ adv, int T,
                                                                long long C) {
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean = 0.f, sd = 1.f;
    const bool dn = denorm != nullptr;
    if (dn) { mean = denorm[0]; sd = denorm[1]; }
    // GAE: the denormalised V(s_{t+1}); otherwise returns[t+1], seeded with the bootstrap the caller stored in row T (:187,215)
    float carry = GAE ? vpred[(long long)T * C + c] : returns[(long long)T * C + c];
    if (GAE && dn) carry = carry * sd + mean;
    float gae = 0.f;
    RetChunk<PTL> cur, nxt;
    ret_fetch<PTL>(cur, rewards, vpred, masks, bad, T - 1, C, c);
    for (int t0 = T - 1; t0 >= 0; t0 -= kGaeAhead) {
        ret_fetch<PTL>(nxt, rewards, vpred, masks, bad, t0 - kGaeAhead, C, c);
#pragma unroll
        for (int u = 0; u < kGaeAhead; ++u) {
            const int t = t0 - u;
            if (t >= 0) {
                const float r = cur.r[u], m = cur.m[u];
                const float v_cur = dn ? (cur.v[u] * sd + mean) : cur.v[u];
                float ret;
                if constexpr (GAE) {
                    const float delta = (r + (gamma * carry) * m) - v_cur;           // :173-175,181-182,203-205,210-211
                    if (PTL && dn) gae = delta + (gl * gae) * m;                     // :176  gamma * gae_lambda * gae * masks
                    else gae = delta + (gl * m) * gae;                               // :183,206,212
                    if constexpr (PTL) gae = gae * cur.b[u];                         // :177,184
                    ret = gae + v_cur;                                               // :178,185,207,213
                    carry = v_cur;
                } else {
                    const float disc = (carry * gamma) * m + r;                      // returns[t+1] * gamma * masks[t+1] + rewards[t]
                    if constexpr (PTL) ret = disc * cur.b[u] + (1.f - cur.b[u]) * v_cur;   // :190-197
                    else ret = disc;                                                 // :217
                    carry = ret;
                }
                returns[(long long)t * C + c] = ret;
                if (adv) adv[(long long)t * C + c] = ret - v_cur;                    // mappo.py:190-191
            }
        }
        cur = nxt;
    }
}

}  // namespace

extern "C" {

DCC_API int dcc_gae_compute(const float* rewards, const float* value_preds, const float* masks, const float* denorm,
                            double gamma, double gae_lambda, float* returns, float* advantages, int32_t T, int64_t C,
                            void* stream) {
    if (!rewards || !value_preds || !masks || !returns) return dcc_fail(-1, "dcc_gae_compute: rewards / value_preds / masks / returns must not be NULL");
    if (T < 1 || C < 1) return dcc_fail(-1, "dcc_gae_compute: T and C must be >= 1");
    const int block = kGaeBlock;
    const long long grid = (C + block - 1) / block;
    if (grid > 0x7fffffffLL) return dcc_fail(-1, "dcc_gae_compute: too many columns for one launch");
    // numpy turns the Python floats into float32 scalars: gamma -> f32(gamma), gamma*lambda (computed
    // in float64 first, shared_buffer.py:206 evaluates left to right) -> f32
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    hipLaunchKernelGGL(dcc_gae_kernel, dim3((unsigned)grid), dim3(block), 0, reinterpret_cast<hipStream_t>(stream),
                       rewards, value_preds, masks, denorm, g, gl, returns, advantages, (int)T, (long long)C);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : dcc_fail(-2, std::string("dcc_gae_compute: ") + hipGetErrorString(e));
}

DCC_API int dcc_returns_compute(const float* rewards, const float* value_preds, const float* masks, const float* bad_masks,
                                const float* denorm, double gamma, double gae_lambda, int32_t mode, float* returns,
                                float* advantages, int32_t T, int64_t C, void* stream) {
    if (mode == DCC_RETURNS_GAE)      // the shipped branch keeps its own kernel
        return dcc_gae_compute(rewards, value_preds, masks, denorm, gamma, gae_lambda, returns, advantages, T, C, stream);
    if (mode < 0 || mode > (DCC_RETURNS_GAE | DCC_RETURNS_PROPER)) return dcc_fail(-1, "dcc_returns_compute: unknown mode");
    if (!rewards || !value_preds || !masks || !returns) return dcc_fail(-1, "dcc_returns_compute: rewards / value_preds / masks / returns must not be NULL");
    if ((mode & DCC_RETURNS_PROPER) && !bad_masks) return dcc_fail(-1, "dcc_returns_compute: DCC_RETURNS_PROPER needs bad_masks");
    if (T < 1 || C < 1) return dcc_fail(-1, "dcc_returns_compute: T and C must be >= 1");
    const long long grid = (C + kGaeBlock - 1) / kGaeBlock;
    if (grid > 0x7fffffffLL) return dcc_fail(-1, "dcc_returns_compute: too many columns for one launch");
    const float g = (float)gamma, gl = (float)(gamma * gae_lambda);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 gr((unsigned)grid), bl(kGaeBlock);
    if (mode == (DCC_RETURNS_GAE | DCC_RETURNS_PROPER))
        hipLaunchKernelGGL((dcc_returns_kernel<true, true>), gr, bl, 0, st, rewards, value_preds, masks, bad_masks, denorm, g, gl,
                           returns, advantages, (int)T, (long long)C);
    else if (mode == DCC_RETURNS_PROPER)
        hipLaunchKernelGGL((dcc_returns_kernel<false, true>), gr, bl, 0, st, rewards, value_preds, masks, bad_masks, denorm, g, gl,
                           returns, advantages, (int)T, (long long)C);
    else
        hipLaunchKernelGGL((dcc_returns_kernel<false, false>), gr, bl, 0, st, rewards, value_preds, masks, bad_masks, denorm, g, gl,
                           returns, advantages, (int)T, (long long)C);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : dcc_fail(-2, std::string("dcc_returns_compute: ") + hipGetErrorString(e));
}

}  // extern "C"
