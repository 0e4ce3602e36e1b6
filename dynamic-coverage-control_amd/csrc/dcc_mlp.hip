// dcc_mlp.hip -- fused element-wise stages of the MAPPO policy trunks (include/dcc_mlp.h).
//
// Layout: one wavefront owns a row of H activations at a time; lane l holds the VEC consecutive columns starting at
// (v*64 + l)*VEC for v < VPL, so every global access of a wave is a run of 64*VEC consecutive floats (float4 per
// lane when H % 4 == 0).  LayerNorm reductions are wave reductions; nothing but the parameter-gradient partial
// sums leaves registers.  Waves walk rows (or envs) with a grid stride; each keeps private partial sums of the
// parameter gradients in registers and writes them once to `workspace`; a second tiny kernel adds the partials in a
// fixed order (no float atomics -> bit-reproducible updates, which the checkpoint/resume test relies on).
#include <hip/hip_runtime.h>

#include <cstdint>

#include <string>

#include "dcc_internal.h"
#include "dcc_mlp.h"

// The Makefile builds the library with -ffp-contract=off for the env kernel's float64 parity; nothing here is
// compared bit-for-bit with a CPU (the tests use tolerances), and these kernels are VALU-bound without fused
// multiply-adds, so contraction is switched back on for this file.
#pragma clang fp contract(fast)

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = 4;
constexpr int kReluLnBlocks = 1024;   // 4096 waves: 16 per CU
constexpr int kL1Blocks = 512;        // 2048 waves (the L1 backward keeps ~100 accumulators per lane)

// Wave-wide sum, result in every lane (a scalar register).  Four DPP adds (VALU, no LDS round trip) leave each 16-lane
// row with its row sum, two row-broadcast adds carry them into lane 63, one readlane makes it wave-uniform.
// (__shfl_xor compiles to ds_bpermute: six dependent LDS round trips per reduction.)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    // row_bcast:15 into rows 1 and 3 (r0+r1, r2+r3), row_bcast:31 into rows 2 and 3 (lane 63 = r0+r1+r2+r3); rows outside
    // row_mask keep their value.  Written as instructions because the compiler's DPP combiner only fuses a masked
    // float add when all rows are enabled (it emits mov 0 / mov_dpp / add otherwise); the s_nop are the two wait states a
    // DPP (and the readlane that follows) needs after a VALU write of its source.
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return lane_bcast(v, 63);
}

template <int VEC>
__device__ __forceinline__ void ld(const float* __restrict__ p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void st(float* __restrict__ p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

// 1 / sqrt(x) as ONE v_rsq_f32 (1 ulp; x = variance + eps is O(1e-5 .. 1e3) here, far from the denormal range).  The
// IEEE-exact `1.0f / sqrtf(x)` is ~30 instructions under -fno-fast-math, once or twice per row of every kernel here.  Every
// forward and its backward use this same function, so a backward's recomputed activations are the forward's.
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }

// ReLU + LayerNorm statistics of one row held in a[VPL][VEC] (already ReLU'd, zeros in masked slots).
template <int VEC, int VPL>
__device__ __forceinline__ void row_stats(const float (&a)[VPL][VEC], const bool (&ok)[VPL], float invH, float eps,
                                          float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += a[v][j];
    mean = wave_sum(s) * invH;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
        if (ok[v]) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = a[v][j] - mean; q += d * d; }
        }
    rstd = fast_rsqrt(wave_sum(q) * invH + eps);
}

// ---- h = LayerNorm(ReLU(z)) -------------------------------------------------------------------------------------
template <int VEC, int VPL>
__global__ __launch_bounds__(kBlock) void relu_ln_fwd_k(const float* __restrict__ z, const float* __restrict__ bias,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        float* __restrict__ h, long long R, int H) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], b[VPL][VEC], zb[VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[v][j] = 0.f; b[v][j] = 0.f; zb[v][j] = 0.f; }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); ld<VEC>(beta + cb[v], b[v]); if (bias) ld<VEC>(bias + cb[v], zb[v]); }
    }
    const float invH = 1.0f / (float)H;
    for (long long r = gw; r < R; r += nw) {
        float a[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[v][j] = 0.f;
            if (ok[v]) {
                ld<VEC>(z + r * H + cb[v], a[v]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) a[v][j] = fmaxf(a[v][j] + zb[v][j], 0.f);
            }
        }
        float mean, rstd;
        row_stats<VEC, VPL>(a, ok, invH, eps, mean, rstd);
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (ok[v]) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = (a[v][j] - mean) * rstd * g[v][j] + b[v][j];
                st<VEC>(h + r * H + cb[v], o);
            }
    }
}

// LayerNorm + ReLU backward for one row: a = ReLU(z) values, dh the incoming gradient (overwritten by dz);
// accumulates dgamma / dbeta partials.
template <int VEC, int VPL>
__device__ __forceinline__ void row_bwd(const float (&zr)[VPL][VEC], const float (&a)[VPL][VEC], float (&dh)[VPL][VEC],
                                        const float (&g)[VPL][VEC], const bool (&ok)[VPL], float invH, float mean,
                                        float rstd, float (&acc_g)[VPL][VEC], float (&acc_b)[VPL][VEC]) {
    float s1 = 0.f, s2 = 0.f;
    float xh[VPL][VEC];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            xh[v][j] = ok[v] ? (a[v][j] - mean) * rstd : 0.f;
            acc_g[v][j] += dh[v][j] * xh[v][j];
            acc_b[v][j] += dh[v][j];
            const float dx = dh[v][j] * g[v][j];
            dh[v][j] = dx;
            s1 += dx;
            s2 += dx * xh[v][j];
        }
    const float m1 = wave_sum(s1) * invH, m2 = wave_sum(s2) * invH;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            dh[v][j] = (zr[v][j] > 0.f) ? rstd * (dh[v][j] - m1 - xh[v][j] * m2) : 0.f;
}

template <int VEC, int VPL>
__global__ __launch_bounds__(kBlock) void relu_ln_bwd_k(const float* __restrict__ z, const float* __restrict__ bias,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ dh, float eps,
                                                        float* __restrict__ dz, float* __restrict__ ws, long long R,
                                                        int H) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], zb[VPL][VEC], acc_g[VPL][VEC], acc_b[VPL][VEC], acc_z[VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[v][j] = 0.f; zb[v][j] = 0.f; acc_g[v][j] = 0.f; acc_b[v][j] = 0.f; acc_z[v][j] = 0.f; }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); if (bias) ld<VEC>(bias + cb[v], zb[v]); }
    }
    const float invH = 1.0f / (float)H;
    for (long long r = gw; r < R; r += nw) {
        float zr[VPL][VEC], a[VPL][VEC], d[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { zr[v][j] = 0.f; d[v][j] = 0.f; }
            if (ok[v]) {
                ld<VEC>(z + r * H + cb[v], zr[v]); ld<VEC>(dh + r * H + cb[v], d[v]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) zr[v][j] += zb[v][j];
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[v][j] = fmaxf(zr[v][j], 0.f);
        }
        float mean, rstd;
        row_stats<VEC, VPL>(a, ok, invH, eps, mean, rstd);
        row_bwd<VEC, VPL>(zr, a, d, g, ok, invH, mean, rstd, acc_g, acc_b);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc_z[v][j] += d[v][j];
            if (ok[v]) st<VEC>(dz + r * H + cb[v], d[v]);
        }
    }
    float* w = ws + gw * 3 * H;   // per-wave partials: [dgamma | dbeta | dbias]
#pragma unroll
    for (int v = 0; v < VPL; ++v)
        if (ok[v]) { st<VEC>(w + cb[v], acc_g[v]); st<VEC>(w + H + cb[v], acc_b[v]); st<VEC>(w + 2 * H + cb[v], acc_z[v]); }
}

// ---- y = Linear_{Wo,bo}(LayerNorm(ReLU(z))) with a narrow output (A <= 4 columns: action mean / value) -------------
// The normalised activations h never reach memory: the forward writes only y [R,A]; the backward recomputes h from z,
// forms dh = dy Wo on the fly and accumulates dWo += dy^T h next to dgamma / dbeta.
constexpr int kAMax = 4;

template <int VEC, int VPL>
__global__ __launch_bounds__(kBlock) void relu_ln_head_fwd_k(const float* __restrict__ z, const float* __restrict__ zbias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             const float* __restrict__ Wo, const float* __restrict__ bo,
                                                             float* __restrict__ y, long long R, int H, int A) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], b[VPL][VEC], zb[VPL][VEC], wo[kAMax][VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[v][j] = 0.f; b[v][j] = 0.f; zb[v][j] = 0.f; }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); ld<VEC>(beta + cb[v], b[v]); if (zbias) ld<VEC>(zbias + cb[v], zb[v]); }
#pragma unroll
        for (int o = 0; o < kAMax; ++o) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) wo[o][v][j] = 0.f;
            if (ok[v] && o < A) ld<VEC>(Wo + o * H + cb[v], wo[o][v]);
        }
    }
    float bias = 0.f;
    if (lane < A) bias = bo ? bo[lane] : 0.f;
    const float invH = 1.0f / (float)H;
    // the only traffic is the z row (1 KB per wave and row) and the chain of reductions below is long: with the load
    // issued where it is needed the kernel holds ~4 MB in flight, short of what HBM latency x bandwidth asks for, so the
    // next row of the wave is fetched before this one is reduced
    float nx[VPL][VEC];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) nx[v][j] = 0.f;
        if (ok[v] && gw < R) ld<VEC>(z + gw * H + cb[v], nx[v]);
    }
    for (long long r = gw; r < R; r += nw) {
        float a[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[v][j] = ok[v] ? fmaxf(nx[v][j] + zb[v][j], 0.f) : 0.f;
            if (ok[v] && r + nw < R) ld<VEC>(z + (r + nw) * H + cb[v], nx[v]);
        }
        float mean, rstd;
        row_stats<VEC, VPL>(a, ok, invH, eps, mean, rstd);
        float part[kAMax] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (ok[v]) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float hh = (a[v][j] - mean) * rstd * g[v][j] + b[v][j];
#pragma unroll
                    for (int o = 0; o < kAMax; ++o) part[o] += hh * wo[o][v][j];
                }
            }
        float out = 0.f;
#pragma unroll
        for (int o = 0; o < kAMax; ++o)
            if (o < A) {
                const float t = wave_sum(part[o]);
                if (lane == o) out = t + bias;
            }
        if (lane < A) y[r * A + lane] = out;
    }
}

template <int VEC, int VPL>
__global__ __launch_bounds__(kBlock) void relu_ln_head_bwd_k(const float* __restrict__ z, const float* __restrict__ zbias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             const float* __restrict__ Wo, const float* __restrict__ dy,
                                                             float* __restrict__ dz, float* __restrict__ ws, long long R,
                                                             int H, int A) {
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], b[VPL][VEC], zb[VPL][VEC], wo[kAMax][VPL][VEC];
    float acc_g[VPL][VEC], acc_b[VPL][VEC], acc_z[VPL][VEC], acc_w[kAMax][VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[v][j] = 0.f; b[v][j] = 0.f; zb[v][j] = 0.f; acc_g[v][j] = 0.f; acc_b[v][j] = 0.f; acc_z[v][j] = 0.f; }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); ld<VEC>(beta + cb[v], b[v]); if (zbias) ld<VEC>(zbias + cb[v], zb[v]); }
#pragma unroll
        for (int o = 0; o < kAMax; ++o) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { wo[o][v][j] = 0.f; acc_w[o][v][j] = 0.f; }
            if (ok[v] && o < A) ld<VEC>(Wo + o * H + cb[v], wo[o][v]);
        }
    }
    const float invH = 1.0f / (float)H;
    // the wave's next row (z and its dy) is fetched before this one is processed, as in the forward
    float nx[VPL][VEC], ndy[kAMax];
#pragma unroll
    for (int o = 0; o < kAMax; ++o) ndy[o] = (o < A && gw < R) ? dy[gw * A + o] : 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) nx[v][j] = 0.f;
        if (ok[v] && gw < R) ld<VEC>(z + gw * H + cb[v], nx[v]);
    }
    for (long long r = gw; r < R; r += nw) {
        float zr[VPL][VEC], a[VPL][VEC], d[VPL][VEC];
        float dyr[kAMax];
        const bool more = r + nw < R;
#pragma unroll
        for (int o = 0; o < kAMax; ++o) { dyr[o] = ndy[o]; if (o < A && more) ndy[o] = dy[(r + nw) * A + o]; }
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) zr[v][j] = ok[v] ? nx[v][j] + zb[v][j] : 0.f;
            if (ok[v] && more) ld<VEC>(z + (r + nw) * H + cb[v], nx[v]);
#pragma unroll
            for (int j = 0; j < VEC; ++j) a[v][j] = fmaxf(zr[v][j], 0.f);
        }
        float mean, rstd;
        row_stats<VEC, VPL>(a, ok, invH, eps, mean, rstd);
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float hh = ok[v] ? (a[v][j] - mean) * rstd * g[v][j] + b[v][j] : 0.f;
                float t = 0.f;
#pragma unroll
                for (int o = 0; o < kAMax; ++o) { t += dyr[o] * wo[o][v][j]; acc_w[o][v][j] += dyr[o] * hh; }
                d[v][j] = t;
            }
        row_bwd<VEC, VPL>(zr, a, d, g, ok, invH, mean, rstd, acc_g, acc_b);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc_z[v][j] += d[v][j];
            if (ok[v]) st<VEC>(dz + r * H + cb[v], d[v]);
        }
    }
    float* w = ws + gw * (3 + kAMax) * H;   // per-wave partials: [dgamma | dbeta | dbias | dWo rows]
#pragma unroll
    for (int v = 0; v < VPL; ++v)
        if (ok[v]) {
            st<VEC>(w + cb[v], acc_g[v]);
            st<VEC>(w + H + cb[v], acc_b[v]);
            st<VEC>(w + 2 * H + cb[v], acc_z[v]);
#pragma unroll
            for (int o = 0; o < kAMax; ++o) st<VEC>(w + (3 + o) * H + cb[v], acc_w[o][v]);
        }
}

// Stage 1 of the partial-sum reduction: the nw per-wave vectors are cut into gridDim.y segments of `seg` waves; every
// segment is summed (fixed order) into its own first slot, in place.  Stage 2 (reduce_partials_k / l1_reduce_k with
// nw = number of segments and stride = seg * stride) adds the segment sums.  One thread per column p, so no two
// threads touch the same address.
__global__ __launch_bounds__(kBlock) void reduce_segments_k(float* __restrict__ ws, long long nw, int seg, int P, int stride) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const long long w0 = (long long)blockIdx.y * seg;
    long long w1 = w0 + seg;
    if (w1 > nw) w1 = nw;
    if (w0 >= w1) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long w = w0;
    for (; w + 3 < w1; w += 4) {
        s0 += ws[w * stride + p]; s1 += ws[(w + 1) * stride + p]; s2 += ws[(w + 2) * stride + p]; s3 += ws[(w + 3) * stride + p];
    }
    for (; w < w1; ++w) s0 += ws[w * stride + p];
    ws[w0 * stride + p] = (s0 + s1) + (s2 + s3);
}

// out[p] = sum over the nw per-wave partial vectors (fixed order).
__global__ __launch_bounds__(kBlock) void reduce_partials_k(const float* __restrict__ ws, long long nw, int P, int stride,
                                                            float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long w = 0;
    for (; w + 3 < nw; w += 4) {
        s0 += ws[w * stride + p]; s1 += ws[(w + 1) * stride + p]; s2 += ws[(w + 2) * stride + p]; s3 += ws[(w + 3) * stride + p];
    }
    for (; w < nw; ++w) s0 += ws[w * stride + p];
    out[p] = (s0 + s1) + (s2 + s3);
}

// ---- PPO-clip policy surrogate of a diagonal Gaussian, forward and gradient in one pass ----------------------------
// One thread per row.  logp = sum_d [-(a_d-mu_d)^2 / (2 sigma_d^2) - log sigma_d - log sqrt(2 pi)];  for each of the K
// stored old-log-prob columns (the reference keeps K = A identical columns) ratio = exp(logp - old),
// surr = min(ratio adv, clamp(ratio, 1-eps, 1+eps) adv) summed over the columns.  Gradients follow PyTorch's rules
// (minimum splits a tie evenly, clamp passes the gradient inside the closed range).  Outputs: dmean_raw =
// d(-sum_r act_r surr_r)/d mean, per-block partial sums [S_surr, S_act, S_ratio, 0, dlogstd_raw[0..3]].
constexpr int kPpoP = 8;
__global__ __launch_bounds__(kBlock) void ppo_policy_loss_k(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                            const float* __restrict__ actions,
                                                            const float* __restrict__ old_logp,
                                                            const float* __restrict__ adv, const float* __restrict__ active,
                                                            float clip, float* __restrict__ dmean, float* __restrict__ ws,
                                                            long long R, int A, int K) {
    __shared__ float red[kWavesPerBlock][kPpoP];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ls[kAMax], inv_var[kAMax];
    float logc = 0.f;
#pragma unroll
    for (int d = 0; d < kAMax; ++d) {
        ls[d] = d < A ? logstd[d] : 0.f;
        inv_var[d] = __expf(-2.f * ls[d]);
        if (d < A) logc += ls[d] + 0.91893853320467274178f;   // log sigma + log sqrt(2 pi)
    }
    const float lo = 1.f - clip, hi = 1.f + clip;
    float acc[kPpoP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long r = (long long)blockIdx.x * kBlock + threadIdx.x; r < R; r += (long long)gridDim.x * kBlock) {
        float dev[kAMax], q = 0.f;
#pragma unroll
        for (int d = 0; d < kAMax; ++d) {
            dev[d] = 0.f;
            if (d < A) { dev[d] = actions[r * A + d] - mean[r * A + d]; q += dev[d] * dev[d] * inv_var[d]; }
        }
        const float logp = -0.5f * q - logc;
        const float a = adv[r];
        const float act = active ? active[r] : 1.f;
        float surr = 0.f, coef = 0.f;
#pragma unroll
        for (int k = 0; k < kAMax; ++k)
            if (k < K) {
                const float ratio = expf(logp - old_logp[r * K + k]);
                const float s1 = ratio * a, s2 = fminf(fmaxf(ratio, lo), hi) * a;
                const float g1 = ratio * a;                                   // d s1 / d logp
                const float g2 = (ratio >= lo && ratio <= hi) ? ratio * a : 0.f;   // d s2 / d logp
                surr += fminf(s1, s2);
                coef += s1 < s2 ? g1 : (s1 > s2 ? g2 : 0.5f * (g1 + g2));
                acc[2] += ratio;
            }
        acc[0] += surr * act;
        acc[1] += act;
        const float w = -coef * act;          // d(-act surr)/d logp
#pragma unroll
        for (int d = 0; d < kAMax; ++d)
            if (d < A) {
                dmean[r * A + d] = w * dev[d] * inv_var[d];
                acc[4 + d] += w * (dev[d] * dev[d] * inv_var[d] - 1.f);
            }
    }
#pragma unroll
    for (int i = 0; i < kPpoP; ++i) {
        const float t = wave_sum(acc[i]);
        if (lane == 0) red[wid][i] = t;
    }
    __syncthreads();
    if (threadIdx.x < kPpoP) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kWavesPerBlock; ++w2) t += red[w2][threadIdx.x];
        ws[(long long)blockIdx.x * kPpoP + threadIdx.x] = t;
    }
}

// ---- clipped value loss (Huber / MSE) of the centralised critic, forward and gradient in one pass ----------------------
// One thread per critic row e (one value for the N agents of an env-step); per agent row r = e*N + i:
//   target = (ret - mean) / std (ValueNorm) or ret;  vpc = vp + clamp(v - vp, -clip, clip)
//   loss = max(h(target - v), h(target - vpc))   with the reference's ONE-SIDED Huber h (utils/util.py:36-39) or e^2/2
// PyTorch gradient conventions (maximum splits ties evenly, clamp passes the gradient on the closed interval).
__device__ __forceinline__ float vl_h(float e, float d, bool huber) {
    if (!huber) return e * e * 0.5f;
    const float a = fabsf(e) <= d ? 1.f : 0.f, b = e > d ? 1.f : 0.f;
    return a * e * e * 0.5f + b * d * (fabsf(e) - d * 0.5f);
}
__device__ __forceinline__ float vl_dh(float e, float d, bool huber) {
    if (!huber) return e;
    return (fabsf(e) <= d ? e : 0.f) + (e > d ? d : 0.f);
}
__global__ __launch_bounds__(kBlock) void ppo_value_loss_k(const float* __restrict__ values, const float* __restrict__ vpred,
                                                           const float* __restrict__ returns, const float* __restrict__ active,
                                                           const float* __restrict__ norm, float clip, float delta,
                                                           int use_clipped, float* __restrict__ dvalues,
                                                           float* __restrict__ ws, long long n, int N) {
    __shared__ float red[kWavesPerBlock][2];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool huber = delta > 0.f;
    float mean = 0.f, sd = 1.f;
    if (norm) { mean = norm[0]; sd = norm[1]; }
    float acc_l = 0.f, acc_a = 0.f;
    for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < n; e += (long long)gridDim.x * kBlock) {
        const float v = values[e];
        float dv = 0.f;
        for (int i = 0; i < N; ++i) {
            const long long r = e * N + i;
            const float vp = vpred[r];
            const float target = norm ? (returns[r] - mean) / sd : returns[r];
            const float act = active ? active[r] : 1.f;
            const float dlt = v - vp;
            const float vpc = vp + fminf(fmaxf(dlt, -clip), clip);
            const float eo = target - v, ec = target - vpc;
            const float lo = vl_h(eo, delta, huber), lc = vl_h(ec, delta, huber);
            const float go = -vl_dh(eo, delta, huber);                                        // d lo / d v
            const float gc = (dlt >= -clip && dlt <= clip) ? -vl_dh(ec, delta, huber) : 0.f;  // d lc / d v
            float l = lo, g = go;
            if (use_clipped) {
                l = fmaxf(lo, lc);
                g = lo > lc ? go : (lo < lc ? gc : 0.5f * (go + gc));
            }
            acc_l += l * act;
            acc_a += act;
            dv += g * act;
        }
        dvalues[e] = dv;
    }
    const float tl = wave_sum(acc_l), ta = wave_sum(acc_a);
    if (lane == 0) { red[wid][0] = tl; red[wid][1] = ta; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kWavesPerBlock; ++w2) t += red[w2][threadIdx.x];
        ws[(long long)blockIdx.x * 2 + threadIdx.x] = t;
    }
}

// ---- rollout glue: sample + log-prob + buffer insert in one launch, reward / mask record in another -------------------
// (Learner.collect / insert, learner.py:227-276: ~18 element-wise launches per env step otherwise.)
__global__ __launch_bounds__(kBlock) void rollout_sample_k(const float* __restrict__ mean, const float* __restrict__ logstd,
                                                           const float* __restrict__ eps, const float* __restrict__ value,
                                                           float* __restrict__ actions, float* __restrict__ logp,
                                                           float* __restrict__ value_preds, long long R, int N, int A, int K) {
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    float lp = 0.f;
#pragma unroll
    for (int d = 0; d < kAMax; ++d)
        if (d < A) {
            const float ls = logstd[d], sd = expf(ls), mu = mean[r * A + d];
            const float a = __fadd_rn(mu, __fmul_rn(sd, eps[r * A + d]));   // FixedNormal.sample: mean + std * randn (no fma)
            actions[r * A + d] = a;
            const float dev = a - mu;                             // Normal.log_prob on the rounded action, like torch
            lp += -(dev * dev) / (2.f * sd * sd) - ls - 0.91893853320467274178f;
        }
    for (int k = 0; k < K; ++k) logp[r * K + k] = lp;
    if (value_preds) value_preds[r] = value[r / N];              // one critic value per env, broadcast over its agents
}

__global__ __launch_bounds__(kBlock) void rollout_record_k(const float* __restrict__ reward, const unsigned char* __restrict__ done,
                                                           float* __restrict__ rewards, float* __restrict__ masks_next,
                                                           long long R, int N, const float* __restrict__ coverage,
                                                           double* __restrict__ rew_acc, float* __restrict__ cov_max) {
    const long long r = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (r >= R) return;
    const long long e = r / N;
    const float rw = reward[e];
    rewards[r] = rw;
    masks_next[r] = done[e] ? 0.f : 1.f;                         // masks = 1 - done (learner.py:262-264)
    if (r == e * N) {   // the env's first agent row also keeps the env's logged statistics (learner.py:187-193): element-wise, no reduction
        if (rew_acc) rew_acc[e] += (double)rw;
        if (cov_max && coverage) cov_max[e] = fmaxf(cov_max[e], coverage[e]);
    }
}

// ---- actor first block from compact features -------------------------------------------------------------------
// z[r,c] = rstd_in[r] * (sum_k head[r,k] Wh[c,k] + G[e,c] - mean_in[r] s[c]) + cb[c];   h = LayerNorm(ReLU(z))
// One wave per env: G[e] is loaded once for its N agent rows and (backward) dG[e] is summed in registers.
// Wh^T lives in LDS as Wt[k][c] (conflict-free float4 reads, shared by the block's waves).
// With many UAVs (HD = 34 at 16, 66 at 32) those HD LDS reads per lane and row bound the kernel (4.2 ms for 2.5 M rows at 16
// UAVs where the 8-UAV register kernel needs 1.4 ms for 4.9 M), so the same kernels also take the per-row term READY-MADE:
// pre[r,c] = sum_k head[r,k] Wh[c,k] from a library GEMM (Zp != NULL, HD = 0; dcc_actor_l1_pre_fwd / _bwd).
template <int VEC, int VPL>
__device__ __forceinline__ void l1_row_z(const float* __restrict__ Wt, int H, int HD, float hv, float hv2, float mean_in,
                                         float rstd_in, const float (&Gv)[VPL][VEC], const float (&sv)[VPL][VEC],
                                         const float (&cv)[VPL][VEC], const int (&cb)[VPL], const bool (&ok)[VPL],
                                         const float (&u0)[VPL][VEC], float (&zr)[VPL][VEC]) {
    float u[VPL][VEC];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j) u[v][j] = u0[v][j];
    // head value k of the row sits in lane k of hv (k < 64) or lane k - 64 of hv2 (up to 128 head columns: 63 UAVs)
    for (int k = 0; k < HD; ++k) {
        const float x = k < 64 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), k))
                               : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv2), k - 64));
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (ok[v]) {
                float w[VEC];
                ld<VEC>(Wt + k * H + cb[v], w);
#pragma unroll
                for (int j = 0; j < VEC; ++j) u[v][j] += x * w[j];
            }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            zr[v][j] = ok[v] ? rstd_in * (u[v][j] + Gv[v][j] - mean_in * sv[v][j]) + cv[v][j] : 0.f;
}

__device__ __forceinline__ void in_stats(const double* __restrict__ stats, long long r, int D, float eps_in,
                                         float& mean_in, float& rstd_in) {
    mean_in = 0.f; rstd_in = 1.f;
    if (stats) {
        const double m = stats[2 * r], m2 = stats[2 * r + 1];
        mean_in = (float)m;
        rstd_in = 1.0f / sqrtf((float)(m2 * (1.0 / (double)D)) + eps_in);
    }
}

template <int VEC, int VPL>
__global__ __launch_bounds__(kBlock) void actor_l1_fwd_k(const float* __restrict__ head, const float* __restrict__ Zp,
                                                         const float* __restrict__ G, const double* __restrict__ stats,
                                                         const float* __restrict__ Wh, const float* __restrict__ s,
                                                         const float* __restrict__ c, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps_in, float eps_ln,
                                                         int D, float* __restrict__ h, long long n, int N, int HD,
                                                         int H) {
    extern __shared__ __attribute__((aligned(16))) float Wt[];   // [HD][H]
    for (int i = threadIdx.x; i < HD * H; i += kBlock) { const int k = i / H, cc = i - k * H; Wt[i] = Wh[cc * HD + k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], b[VPL][VEC], sv[VPL][VEC], cv[VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[v][j] = 0.f; b[v][j] = 0.f; sv[v][j] = 0.f; cv[v][j] = 0.f; }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); ld<VEC>(beta + cb[v], b[v]); ld<VEC>(s + cb[v], sv[v]); ld<VEC>(c + cb[v], cv[v]); }
    }
    const float invH = 1.0f / (float)H;
    float nhv = 0.f, nhv2 = 0.f, nz[VPL][VEC];
    double nm = 0.0, nm2 = 0.0;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int j = 0; j < VEC; ++j) nz[v][j] = 0.f;
    auto fetch = [&](long long r) {
        nhv = lane < HD ? head[r * HD + lane] : 0.f;
        nhv2 = lane + 64 < HD ? head[r * HD + 64 + lane] : 0.f;
        if (stats) { nm = stats[2 * r]; nm2 = stats[2 * r + 1]; }
        if (Zp) {
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (ok[v]) ld<VEC>(Zp + r * H + cb[v], nz[v]);
        }
    };
    if (gw < n) fetch(gw * N);
    for (long long e = gw; e < n; e += nw) {
        float Gv[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) Gv[v][j] = 0.f;
            if (ok[v]) ld<VEC>(G + e * H + cb[v], Gv[v]);
        }
        for (int i = 0; i < N; ++i) {
            const long long r = e * N + i;
            const float hv = nhv, hv2 = nhv2;
            float mean_in = 0.f, rstd_in = 1.f, u0[VPL][VEC];
            if (stats) { mean_in = (float)nm; rstd_in = 1.0f / sqrtf((float)(nm2 * (1.0 / (double)D)) + eps_in); }
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) u0[v][j] = nz[v][j];
            const long long rn = (i + 1 < N) ? r + 1 : ((e + nw < n) ? (e + nw) * N : -1);
            if (rn >= 0) fetch(rn);     // the next row's head values / moments / ready-made term travel while this row is computed
            float a[VPL][VEC];
            l1_row_z<VEC, VPL>(Wt, H, HD, hv, hv2, mean_in, rstd_in, Gv, sv, cv, cb, ok, u0, a);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) a[v][j] = fmaxf(a[v][j], 0.f);
            float mean, rstd;
            row_stats<VEC, VPL>(a, ok, invH, eps_ln, mean, rstd);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
                if (ok[v]) {
                    float o[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] = (a[v][j] - mean) * rstd * g[v][j] + b[v][j];
                    st<VEC>(h + r * H + cb[v], o);
                }
        }
    }
}

// ---- actor first block, BASELINE sizes: NR agents per env fixed at compile time (4 / 8 UAVs, H <= 256, float4 lanes) ----
// Two things bound the generic kernels above at 4.9 M rows: (1) Wh^T in LDS costs HD ds_read_b128 per lane and row (18 KB of
// LDS reads per row at 8 UAVs: ~1.2 ms of pure LDS bandwidth per pass), (2) a row's inputs (72 B of head values, its moments)
// are fetched one row ahead, which covers a fraction of the HBM latency at 4 waves per SIMD.  Here Wh^T lives in REGISTERS
// (HD x 4 columns per lane), the env's G row is fetched a whole ENV ahead, and what the lanes of a row share (its HD head
// values, its two moments) is read with SCALAR loads a row ahead into scalar registers.
#ifndef DCC_L1F_WAVES
#define DCC_L1F_WAVES 3     // waves per SIMD the env kernels are compiled for (register budget 512 / waves): at 4 the
#endif                      // forward spills its prefetch registers to scratch, which serialises the HBM latency again
#ifndef DCC_L1B_WAVES
#define DCC_L1B_WAVES 2
#endif
template <int NR>
struct EnvIn {
    static constexpr int HD = 4 + 2 * (NR - 1);
    static constexpr int HW = (NR * HD + 63) / 64;
    float h[HW];     // the env's NR*HD head values, one coalesced load each: fetched only to pull their lines into L2
    float G[4];
};

template <int NR>
__device__ __forceinline__ void fetch_env(const float* __restrict__ head, const float* __restrict__ G, long long e, int lane,
                                          EnvIn<NR>& in) {
    constexpr int HD = EnvIn<NR>::HD;
#pragma unroll
    for (int w = 0; w < EnvIn<NR>::HW; ++w) {
        const int idx = w * 64 + lane;
        in.h[w] = idx < NR * HD ? head[e * (NR * HD) + idx] : 0.f;
    }
    ld<4>(G + e * 256 + lane * 4, in.G);
}
template <int NR>
__device__ __forceinline__ void keep_alive(const EnvIn<NR>& in) {
#pragma unroll
    for (int w = 0; w < EnvIn<NR>::HW; ++w) asm volatile("" ::"v"(in.h[w]));
}

// What a row's lanes share -- its HD head values and its input moments -- lives in SCALAR registers, loaded with scalar
// loads (the addresses are wave-uniform) one row ahead: no gather, no readlane, and every multiply-add below takes its
// multiplier straight from a scalar register.  A scalar load that misses L2 costs more than one row of work, which is why
// fetch_env's vector loads still touch the env's head lines an env ahead.
typedef const float __attribute__((address_space(4)))* cfloat_p;    // constant address space: selects s_load
typedef const double __attribute__((address_space(4)))* cdouble_p;
template <int HD>
struct RowIn {
    float x[HD];
    double mean, m2;
};
template <int HD>
__device__ __forceinline__ void fetch_row(cfloat_p head, cdouble_p stats, long long r, RowIn<HD>& o) {
#pragma unroll
    for (int k = 0; k < HD; ++k) o.x[k] = head[r * HD + k];
    o.mean = 0.0; o.m2 = 0.0;
    if (stats) { o.mean = stats[2 * r]; o.m2 = stats[2 * r + 1]; }
}

typedef float v2f __attribute__((ext_vector_type(2)));   // two adjacent columns: one v_pk_fma_f32 per pair

// ReLU + LayerNorm moments of one full row of 256 (float4 per lane, every lane valid)
__device__ __forceinline__ void row_stats_full(const float (&a)[4], float eps, float& mean, float& rstd) {
    mean = wave_sum((a[0] + a[1]) + (a[2] + a[3])) * (1.0f / 256.0f);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = a[j] - mean; q += d * d; }
    rstd = fast_rsqrt(wave_sum(q) * (1.0f / 256.0f) + eps);
}

// z = rstd_in * (head_row . Wh^T + G - mean_in s) + c of one row, the k-loop fully unrolled on scalar multipliers.  The
// row loop itself stays rolled: unrolling it lets the scheduler overlap rows and blows the register budget (measured: 256
// VGPRs / 189 spills for the backward at 8 UAVs).
template <int NR>
__device__ __forceinline__ void env_row_z(const RowIn<EnvIn<NR>::HD>& row, const float (&Gv)[4], const v2f (&w)[EnvIn<NR>::HD][2],
                                          const bool has_stats, float invD, float eps_in, const float (&sv)[4],
                                          const float (&cv)[4], float (&zr)[4], float& mean_in, float& rstd_in) {
    constexpr int HD = EnvIn<NR>::HD;
    // the column pairs are explicit 2-vectors: left to itself the vectoriser pairs over k for part of the loop (a
    // v_pk_mul + 2 adds per pair) and leaves the rest scalar -- 88 VALU instructions per row instead of 36
    v2f u0 = {0.f, 0.f}, u1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HD; ++k) {
        const v2f xx = {row.x[k], row.x[k]};
        u0 = xx * w[k][0] + u0;
        u1 = xx * w[k][1] + u1;
    }
    const float u[4] = {u0.x, u0.y, u1.x, u1.y};
    mean_in = 0.f; rstd_in = 1.f;
    if (has_stats) {
        mean_in = (float)row.mean;
        rstd_in = fast_rsqrt((float)row.m2 * invD + eps_in);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) zr[j] = rstd_in * (u[j] + Gv[j] - mean_in * sv[j]) + cv[j];
}

// H == 256 (every lane owns 4 valid columns: no predication anywhere in the row loop)
template <int NR>
__global__ __launch_bounds__(kBlock, DCC_L1F_WAVES) void actor_l1_fwd_env_k(const float* __restrict__ head, const float* __restrict__ G,
                                                                const double* __restrict__ stats, const float* __restrict__ Wh,
                                                                const float* __restrict__ s, const float* __restrict__ c,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps_in, float eps_ln, int D, float* __restrict__ h,
                                                                long long n) {
    constexpr int HD = EnvIn<NR>::HD, H = 256;
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    const int cb = lane * 4;
    float g[4], b[4], sv[4], cv[4];
    v2f w[HD][2];
    ld<4>(gamma + cb, g); ld<4>(beta + cb, b); ld<4>(s + cb, sv); ld<4>(c + cb, cv);
#pragma unroll
    for (int k = 0; k < HD; ++k) {
        w[k][0] = v2f{Wh[cb * HD + k], Wh[(cb + 1) * HD + k]};
        w[k][1] = v2f{Wh[(cb + 2) * HD + k], Wh[(cb + 3) * HD + k]};
    }
    const float invD = 1.0f / (float)D;
    const bool has_stats = stats != nullptr;
    const cfloat_p head_s = (cfloat_p)head;
    const cdouble_p stats_s = (cdouble_p)stats;
    EnvIn<NR> cur, nxt;
    RowIn<HD> row, rown;
    if (gw < n) { fetch_env<NR>(head, G, gw, lane, cur); fetch_row<HD>(head_s, stats_s, gw * NR, row); }
    for (long long e = gw; e < n; e += nw) {
        const bool more = e + nw < n;
        if (more) fetch_env<NR>(head, G, e + nw, lane, nxt);   // a whole env ahead
        float* hrow = h + e * NR * H + cb;
#pragma unroll 1
        for (int i = 0; i < NR; ++i) {
            // the next row of this wave: the env's next agent, or the first agent of the wave's next env
            fetch_row<HD>(head_s, stats_s, i + 1 < NR ? e * NR + i + 1 : (more ? e + nw : e) * NR, rown);
            float a[4], mi, ri;
            env_row_z<NR>(row, cur.G, w, has_stats, invD, eps_in, sv, cv, a, mi, ri);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
            float mean, rstd;
            row_stats_full(a, eps_ln, mean, rstd);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (a[j] - mean) * rstd * g[j] + b[j];
            st<4>(hrow + (long long)i * H, o);
            row = rown;
        }
        keep_alive<NR>(cur);
        cur = nxt;
    }
}

// Backward; the dh rows are streamed three rows ahead.  ACCW = false: q = rstd_in * dz is stored and dWh = q^T head is the
// caller's GEMM; ACCW = true: dWh is accumulated in registers next to Wh^T (no [rows, H] write, no GEMM, no reduction pass
// over q), per-wave partials reduced in a fixed order like the other parameter gradients.
template <int NR, bool ACCW>
__global__ __launch_bounds__(kBlock, DCC_L1B_WAVES) void actor_l1_bwd_env_k(const float* __restrict__ head, const float* __restrict__ G,
                                                                const double* __restrict__ stats, const float* __restrict__ Wh,
                                                                const float* __restrict__ s, const float* __restrict__ c,
                                                                const float* __restrict__ gamma, const float* __restrict__ dh,
                                                                float eps_in, float eps_ln, int D, float* __restrict__ dG,
                                                                float* __restrict__ dq, float* __restrict__ ws, long long n) {
    constexpr int HD = EnvIn<NR>::HD, H = 256;
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    const int cb = lane * 4;
    float g[4], sv[4], cv[4];
    v2f w[HD][2];
    float acc_g[4] = {0, 0, 0, 0}, acc_b[4] = {0, 0, 0, 0}, acc_s[4] = {0, 0, 0, 0}, acc_c[4] = {0, 0, 0, 0};
    v2f aw[ACCW ? HD : 1][2];
#pragma unroll
    for (int k = 0; k < (ACCW ? HD : 1); ++k) aw[k][0] = aw[k][1] = v2f{0.f, 0.f};
    ld<4>(gamma + cb, g); ld<4>(s + cb, sv); ld<4>(c + cb, cv);
#pragma unroll
    for (int k = 0; k < HD; ++k) {
        w[k][0] = v2f{Wh[cb * HD + k], Wh[(cb + 1) * HD + k]};
        w[k][1] = v2f{Wh[(cb + 2) * HD + k], Wh[(cb + 3) * HD + k]};
    }
    const float invD = 1.0f / (float)D;
    const bool has_stats = stats != nullptr;
    // look-ahead cursor over this wave's dh rows: (pe, pi) is the next row to fetch
    long long pe = gw;
    int pi = 0;
    auto fetch_dh = [&](float (&dd)[4]) {
        if (pe < n) {
            ld<4>(dh + (pe * NR + pi) * H + cb, dd);
            if (++pi == NR) { pi = 0; pe += nw; }
        }
    };
    float nd[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    fetch_dh(nd[0]); fetch_dh(nd[1]); fetch_dh(nd[2]);
    const cfloat_p head_s = (cfloat_p)head;
    const cdouble_p stats_s = (cdouble_p)stats;
    EnvIn<NR> cur, nxt;
    RowIn<HD> row, rown;
    if (gw < n) { fetch_env<NR>(head, G, gw, lane, cur); fetch_row<HD>(head_s, stats_s, gw * NR, row); }
    for (long long e = gw; e < n; e += nw) {
        const bool more = e + nw < n;
        if (more) fetch_env<NR>(head, G, e + nw, lane, nxt);
        float dGv[4] = {0.f, 0.f, 0.f, 0.f};
        float* qrow = ACCW ? nullptr : dq + e * NR * H + cb;
#pragma unroll 1
        for (int i = 0; i < NR; ++i) {
            float d[4], zr[4], a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { d[j] = nd[0][j]; nd[0][j] = nd[1][j]; nd[1][j] = nd[2][j]; }
            fetch_dh(nd[2]);                                   // row t+3
            fetch_row<HD>(head_s, stats_s, i + 1 < NR ? e * NR + i + 1 : (more ? e + nw : e) * NR, rown);
            float mean_in, rstd_in;
            env_row_z<NR>(row, cur.G, w, has_stats, invD, eps_in, sv, cv, zr, mean_in, rstd_in);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaxf(zr[j], 0.f);
            float mean, rstd;
            row_stats_full(a, eps_ln, mean, rstd);
            // LayerNorm + ReLU backward of the row (row_bwd, all lanes valid): d <- dL/dz
            float xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = (a[j] - mean) * rstd;
                acc_g[j] += d[j] * xh[j];
                acc_b[j] += d[j];
                d[j] *= g[j];
                s1 += d[j];
                s2 += d[j] * xh[j];
            }
            const float m1 = wave_sum(s1) * (1.0f / 256.0f), m2 = wave_sum(s2) * (1.0f / 256.0f);
            float q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[j] = (zr[j] > 0.f) ? rstd * (d[j] - m1 - xh[j] * m2) : 0.f;
                q[j] = rstd_in * d[j];
                dGv[j] += q[j];
                acc_s[j] -= mean_in * q[j];
                acc_c[j] += d[j];
            }
            if constexpr (ACCW) {
                const v2f q0 = {q[0], q[1]}, q1 = {q[2], q[3]};
#pragma unroll
                for (int k = 0; k < HD; ++k) {
                    const v2f xx = {row.x[k], row.x[k]};
                    aw[k][0] = xx * q0 + aw[k][0];
                    aw[k][1] = xx * q1 + aw[k][1];
                }
            } else {
                st<4>(qrow + (long long)i * H, q);
            }
            row = rown;
        }
        st<4>(dG + e * H + cb, dGv);
        keep_alive<NR>(cur);
        cur = nxt;
    }
    // per-wave partials in l1_reduce_k's layout: [dWt rows (HD x H, ACCW only) | ds | dc | dgamma | dbeta]
    constexpr int KW = ACCW ? HD : 0;
    float* wv = ws + gw * (KW + 4) * H;
    if constexpr (ACCW) {
#pragma unroll
        for (int k = 0; k < HD; ++k) {
            const float t[4] = {aw[k][0].x, aw[k][0].y, aw[k][1].x, aw[k][1].y};
            st<4>(wv + k * H + cb, t);
        }
    }
    st<4>(wv + KW * H + cb, acc_s); st<4>(wv + (KW + 1) * H + cb, acc_c); st<4>(wv + (KW + 2) * H + cb, acc_g);
    st<4>(wv + (KW + 3) * H + cb, acc_b);
}

// per-wave partial vector of the L1 backward: [dWt (HDP*H) | ds (H) | dc (H) | dgamma (H) | dbeta (H)].
// HDP = 0: no dWh accumulators (they are what limits the kernel to 2 waves/SIMD); q = rstd_in * dz is stored instead.
template <int VEC, int VPL, int HDP>
__global__ __launch_bounds__(kBlock) void actor_l1_bwd_k(const float* __restrict__ head, const float* __restrict__ Zp,
                                                         const float* __restrict__ G, const double* __restrict__ stats,
                                                         const float* __restrict__ Wh, const float* __restrict__ s,
                                                         const float* __restrict__ c, const float* __restrict__ gamma,
                                                         const float* __restrict__ dh, float eps_in, float eps_ln,
                                                         int D, float* __restrict__ dG, float* __restrict__ dq,
                                                         float* __restrict__ ws, long long n, int N, int HD, int H) {
    extern __shared__ __attribute__((aligned(16))) float Wt[];   // [HD][H]
    for (int i = threadIdx.x; i < HD * H; i += kBlock) { const int k = i / H, cc = i - k * H; Wt[i] = Wh[cc * HD + k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * kWavesPerBlock;
    float g[VPL][VEC], sv[VPL][VEC], cv[VPL][VEC];
    float acc_g[VPL][VEC], acc_b[VPL][VEC], acc_s[VPL][VEC], acc_c[VPL][VEC], acc_w[HDP > 0 ? HDP : 1][VPL][VEC];
    bool ok[VPL];
    int cb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        cb[v] = (v * 64 + lane) * VEC;
        ok[v] = cb[v] < H;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            g[v][j] = 0.f; sv[v][j] = 0.f; cv[v][j] = 0.f;
            acc_g[v][j] = 0.f; acc_b[v][j] = 0.f; acc_s[v][j] = 0.f; acc_c[v][j] = 0.f;
        }
        if (ok[v]) { ld<VEC>(gamma + cb[v], g[v]); ld<VEC>(s + cb[v], sv[v]); ld<VEC>(c + cb[v], cv[v]); }
    }
#pragma unroll
    for (int k = 0; k < HDP; ++k)
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc_w[k][v][j] = 0.f;
    const float invH = 1.0f / (float)H;
    // software pipeline, two rows deep: with ~100 accumulator registers per lane only two waves fit a SIMD, far too
    // few to hide HBM latency, so the loads of row t+2 (dh, head, input moments) are issued before row t is processed
    float nd[2][VPL][VEC], nz[2][VPL][VEC], nhv[2] = {0.f, 0.f}, nhw[2] = {0.f, 0.f};
    double nm[2] = {0.0, 0.0}, nm2[2] = {0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int j = 0; j < VEC; ++j) nz[t][v][j] = 0.f;
    auto fetch = [&](long long r, float (&dd)[VPL][VEC], float (&zz)[VPL][VEC], float& hv_, float& hw_, double& m_, double& m2_) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) dd[v][j] = 0.f;
            if (ok[v]) { ld<VEC>(dh + r * H + cb[v], dd[v]); if (Zp) ld<VEC>(Zp + r * H + cb[v], zz[v]); }
        }
        hv_ = lane < HD ? head[r * HD + lane] : 0.f;
        hw_ = lane + 64 < HD ? head[r * HD + 64 + lane] : 0.f;
        if (stats) { m_ = stats[2 * r]; m2_ = stats[2 * r + 1]; }
    };
    // look-ahead cursor over this wave's row order (e, 0..N-1), (e + nw, 0..N-1), ...: (pe, pi) is the next row to fetch
    long long pe = gw;
    int pi = 0;
    auto fetch_next = [&](float (&dd)[VPL][VEC], float (&zz)[VPL][VEC], float& hv_, float& hw_, double& m_, double& m2_) {
        if (pe < n) {
            fetch(pe * N + pi, dd, zz, hv_, hw_, m_, m2_);
            if (++pi == N) { pi = 0; pe += nw; }
        }
    };
    fetch_next(nd[0], nz[0], nhv[0], nhw[0], nm[0], nm2[0]);
    fetch_next(nd[1], nz[1], nhv[1], nhw[1], nm[1], nm2[1]);
    for (long long e = gw; e < n; e += nw) {
        float Gv[VPL][VEC], dGv[VPL][VEC];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { Gv[v][j] = 0.f; dGv[v][j] = 0.f; }
            if (ok[v]) ld<VEC>(G + e * H + cb[v], Gv[v]);
        }
        for (int i = 0; i < N; ++i) {
            const long long r = e * N + i;
            float d[VPL][VEC], u0[VPL][VEC];
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    d[v][j] = nd[0][v][j]; nd[0][v][j] = nd[1][v][j];
                    u0[v][j] = nz[0][v][j]; nz[0][v][j] = nz[1][v][j];
                }
            const float hv = nhv[0], hw = nhw[0];
            float mean_in = 0.f, rstd_in = 1.f;
            if (stats) { mean_in = (float)nm[0]; rstd_in = 1.0f / sqrtf((float)(nm2[0] * (1.0 / (double)D)) + eps_in); }
            nhv[0] = nhv[1]; nhw[0] = nhw[1]; nm[0] = nm[1]; nm2[0] = nm2[1];
            fetch_next(nd[1], nz[1], nhv[1], nhw[1], nm[1], nm2[1]);   // row t+2 (row t+1 is already in flight)
            float zr[VPL][VEC], a[VPL][VEC];
            l1_row_z<VEC, VPL>(Wt, H, HD, hv, hw, mean_in, rstd_in, Gv, sv, cv, cb, ok, u0, zr);
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) a[v][j] = fmaxf(zr[v][j], 0.f);
            float mean, rstd;
            row_stats<VEC, VPL>(a, ok, invH, eps_ln, mean, rstd);
            row_bwd<VEC, VPL>(zr, a, d, g, ok, invH, mean, rstd, acc_g, acc_b);   // d = dL/dz
            float q[VPL][VEC];
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    q[v][j] = rstd_in * d[v][j];
                    dGv[v][j] += q[v][j];
                    acc_s[v][j] -= mean_in * q[v][j];
                    acc_c[v][j] += d[v][j];
                }
            if constexpr (HDP == 0) {   // two-kernel variant: q goes to memory, the caller forms dWh = q^T head as a GEMM
                if (dq) {               // (HD = 0, the critic's first block: there is no head term and q = dG is all there is)
#pragma unroll
                    for (int v = 0; v < VPL; ++v)
                        if (ok[v]) st<VEC>(dq + r * H + cb[v], q[v]);
                }
            }
#pragma unroll
            for (int k = 0; k < HDP; ++k) {
                if (k < HD) {
                    const float x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hv), k));
#pragma unroll
                    for (int v = 0; v < VPL; ++v)
#pragma unroll
                        for (int j = 0; j < VEC; ++j) acc_w[k][v][j] += q[v][j] * x;
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v)
            if (ok[v]) st<VEC>(dG + e * H + cb[v], dGv[v]);
    }
    const int P = (HDP + 4) * H;
    float* w = ws + gw * P;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
        if (ok[v]) {
#pragma unroll
            for (int k = 0; k < HDP; ++k) st<VEC>(w + k * H + cb[v], acc_w[k][v]);
            st<VEC>(w + (HDP + 0) * H + cb[v], acc_s[v]);
            st<VEC>(w + (HDP + 1) * H + cb[v], acc_c[v]);
            st<VEC>(w + (HDP + 2) * H + cb[v], acc_g[v]);
            st<VEC>(w + (HDP + 3) * H + cb[v], acc_b[v]);
        }
}

// partial layout [k][c] -> dWh [c][k], plus the four H-vectors
__global__ __launch_bounds__(kBlock) void l1_reduce_k(const float* __restrict__ ws, long long nw, long long wstride, int HDP, int HD, int H,
                                                      float* __restrict__ dWh, float* __restrict__ ds,
                                                      float* __restrict__ dc, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
    const int P = (HDP + 4) * H;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int k = p / H, cc = p - k * H;
    if (k < HDP && k >= HD) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long long w = 0;
    for (; w + 3 < nw; w += 4) {
        s0 += ws[w * wstride + p]; s1 += ws[(w + 1) * wstride + p]; s2 += ws[(w + 2) * wstride + p]; s3 += ws[(w + 3) * wstride + p];
    }
    for (; w < nw; ++w) s0 += ws[w * wstride + p];
    const float t = (s0 + s1) + (s2 + s3);
    if (k < HDP) dWh[cc * HD + k] = t;
    else if (k == HDP) ds[cc] = t;
    else if (k == HDP + 1) dc[cc] = t;
    else if (k == HDP + 2) dgamma[cc] = t;
    else dbeta[cc] = t;
}

// ---- host side --------------------------------------------------------------------------------------------------
struct Shape { int vec, vpl; };
bool pick_shape(int H, Shape& sh) {
    if (H < 1) return false;
    if (H % 4 == 0 && H <= 256) { sh = {4, 1}; return true; }
    if (H % 4 == 0 && H <= 512) { sh = {4, 2}; return true; }
    if (H <= 64) { sh = {1, 1}; return true; }
    if (H <= 128) { sh = {1, 2}; return true; }
    return false;
}
constexpr int kHdQOnly = 1000;   // 40 < HD <= 128: no in-register dWh accumulators, the backward stores q (dq != NULL)
constexpr int kHdMax = 128;      // two head registers per lane: up to 63 UAVs
int pad_hd(int HD) {
    if (HD <= 0) return 0;
    for (int p : {8, 16, 24, 40})
        if (HD <= p) return p;
    return HD <= kHdMax ? kHdQOnly : -1;
}
constexpr size_t kLdsMax = 160 * 1024;
// Wh^T beyond 64 KB of LDS (HD > 64 at H = 256): raise the kernel's dynamic-LDS limit (160 KB per CU on gfx950)
template <typename K>
bool allow_lds(K kernel, size_t lds) {
    return lds <= 64 * 1024 ||
           hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
}
long long waves_for(long long units, int blocks) {
    long long b = (units + kWavesPerBlock - 1) / kWavesPerBlock;
    if (b > blocks) b = blocks;
    if (b < 1) b = 1;
    return b;
}

#define LAUNCH_SHAPE(KERNEL, grid, lds, ...)                                                                       \
    do {                                                                                                           \
        if (sh.vec == 4 && sh.vpl == 1) hipLaunchKernelGGL((KERNEL<4, 1>), dim3(grid), dim3(kBlock), lds, st_, __VA_ARGS__); \
        else if (sh.vec == 4) hipLaunchKernelGGL((KERNEL<4, 2>), dim3(grid), dim3(kBlock), lds, st_, __VA_ARGS__);  \
        else if (sh.vpl == 1) hipLaunchKernelGGL((KERNEL<1, 1>), dim3(grid), dim3(kBlock), lds, st_, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<1, 2>), dim3(grid), dim3(kBlock), lds, st_, __VA_ARGS__);                   \
    } while (0)

template <int HDP, typename... A>
void launch_l1_bwd(const Shape& sh, int grid, size_t lds, hipStream_t st_, A... a) {
    if (sh.vec == 4 && sh.vpl == 1) hipLaunchKernelGGL((actor_l1_bwd_k<4, 1, HDP>), dim3(grid), dim3(kBlock), lds, st_, a...);
    else if (sh.vec == 4) hipLaunchKernelGGL((actor_l1_bwd_k<4, 2, HDP>), dim3(grid), dim3(kBlock), lds, st_, a...);
    else if (sh.vpl == 1) hipLaunchKernelGGL((actor_l1_bwd_k<1, 1, HDP>), dim3(grid), dim3(kBlock), lds, st_, a...);
    else hipLaunchKernelGGL((actor_l1_bwd_k<1, 2, HDP>), dim3(grid), dim3(kBlock), lds, st_, a...);
}

// Waves walk envs with a grid stride, so the grid must not exceed what is co-resident (a partly filled second round of
// workgroups would add its whole duration): blocks = min(wanted, occupancy x CUs), occupancy queried once per kernel.
int resident_blocks(const void* fn) {
    int per_cu = 0, dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return per_cu * cus;
}

constexpr int kSegWaves = 64;   // waves per stage-1 segment of the partial-sum reduction
// stage 1 in place; returns the number of segments (the "waves" stage 2 sees, at stride kSegWaves * stride)
long long reduce_stage1(float* ws, long long nw, int P, int stride, hipStream_t st_) {
    const long long nseg = (nw + kSegWaves - 1) / kSegWaves;
    hipLaunchKernelGGL(reduce_segments_k, dim3((P + kBlock - 1) / kBlock, (unsigned)nseg), dim3(kBlock), 0, st_, ws, nw,
                       kSegWaves, P, stride);
    return nseg;
}

constexpr int kEINVAL = -1, kEHIP = -2, kEUNSUPPORTED = -4;
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
int launch_status(const char* fn) {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : dcc_fail(kEHIP, std::string(fn) + ": " + hipGetErrorString(e));
}

}  // namespace

extern "C" {

DCC_API int64_t dcc_mlp_workspace_floats(int32_t H, int32_t HD) {
    Shape sh;
    if (!pick_shape(H, sh)) return 0;
    int hdp = pad_hd(HD);
    if (hdp < 0) return 0;
    if (hdp == kHdQOnly) hdp = 0;
    if ((size_t)HD * H * sizeof(float) > kLdsMax) return 0;
    const int64_t a = (int64_t)kReluLnBlocks * kWavesPerBlock * (3 + kAMax) * H;
    int64_t b = HD > 0 ? (int64_t)kL1Blocks * kWavesPerBlock * (hdp + 4) * H : 0;
    const int64_t b2 = HD > 0 ? (int64_t)kL1Blocks * 2 * kWavesPerBlock * 4 * H : 0;
    if (b2 > b) b = b2;
    return a > b ? a : b;
}

DCC_API int dcc_relu_ln_fwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps, float* h,
                            int64_t R, int32_t H, void* stream) {
    if (!z || !gamma || !beta || !h || R < 0) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    if (!pick_shape(H, sh)) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (sh.vec == 4 && !(aligned16(z) && aligned16(h) && aligned16(gamma) && aligned16(beta))) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    if (R == 0) return 0;
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)waves_for(R, kReluLnBlocks);
    LAUNCH_SHAPE(relu_ln_fwd_k, grid, 0, z, bias, gamma, beta, eps, h, (long long)R, (int)H);
    return launch_status(__func__);
}

DCC_API int dcc_relu_ln_bwd(const float* z, const float* bias, const float* gamma, const float* dh, float eps, float* dz,
                            float* dparams, float* workspace, int64_t R, int32_t H, void* stream) {
    if (!z || !gamma || !dh || !dz || !dparams || !workspace || R < 1) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    if (!pick_shape(H, sh)) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (sh.vec == 4 && !(aligned16(z) && aligned16(dh) && aligned16(dz) && aligned16(gamma) && aligned16(workspace) &&
                         aligned16(bias)))
        return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)waves_for(R, kReluLnBlocks);
    LAUNCH_SHAPE(relu_ln_bwd_k, grid, 0, z, bias, gamma, dh, eps, dz, workspace, (long long)R, (int)H);
    const int P = 3 * H;
    const long long nseg = reduce_stage1(workspace, (long long)grid * kWavesPerBlock, P, P, st_);
    hipLaunchKernelGGL(reduce_partials_k, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, workspace, nseg, P,
                       kSegWaves * P, dparams);
    return launch_status(__func__);
}

DCC_API int dcc_relu_ln_head_fwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps,
                                 const float* Wo, const float* bo, float* y, int64_t R, int32_t H, int32_t A,
                                 void* stream) {
    if (!z || !gamma || !beta || !Wo || !y || R < 0) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    if (!pick_shape(H, sh) || A < 1 || A > kAMax) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (sh.vec == 4 && !(aligned16(z) && aligned16(gamma) && aligned16(beta) && aligned16(Wo))) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    if (R == 0) return 0;
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)waves_for(R, kReluLnBlocks);
    LAUNCH_SHAPE(relu_ln_head_fwd_k, grid, 0, z, bias, gamma, beta, eps, Wo, bo, y, (long long)R, (int)H, (int)A);
    return launch_status(__func__);
}

DCC_API int dcc_relu_ln_head_bwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps,
                                 const float* Wo, const float* dy, float* dz, float* dparams, float* workspace, int64_t R,
                                 int32_t H, int32_t A, void* stream) {
    if (!z || !gamma || !beta || !Wo || !dy || !dz || !dparams || !workspace || R < 1) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    if (!pick_shape(H, sh) || A < 1 || A > kAMax) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (sh.vec == 4 && !(aligned16(z) && aligned16(dz) && aligned16(gamma) && aligned16(beta) && aligned16(Wo) &&
                         aligned16(workspace)))
        return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)waves_for(R, kReluLnBlocks);
    LAUNCH_SHAPE(relu_ln_head_bwd_k, grid, 0, z, bias, gamma, beta, eps, Wo, dy, dz, workspace, (long long)R, (int)H, (int)A);
    const int P = (3 + A) * H, stride = (3 + kAMax) * (int)H;
    const long long nseg = reduce_stage1(workspace, (long long)grid * kWavesPerBlock, P, stride, st_);
    hipLaunchKernelGGL(reduce_partials_k, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, workspace, nseg, P,
                       kSegWaves * stride, dparams);
    return launch_status(__func__);
}

DCC_API int dcc_ppo_policy_loss(const float* mean, const float* logstd, const float* actions, const float* old_logp,
                                const float* adv, const float* active, float clip, float* dmean, float* sums,
                                float* workspace, int64_t R, int32_t A, int32_t K, void* stream) {
    if (!mean || !logstd || !actions || !old_logp || !adv || !dmean || !sums || !workspace || R < 1) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    if (A < 1 || A > kAMax || K < 1 || K > kAMax) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    long long blocks = (R + kBlock - 1) / kBlock;
    if (blocks > kReluLnBlocks * 2) blocks = kReluLnBlocks * 2;
    hipLaunchKernelGGL(ppo_policy_loss_k, dim3((unsigned)blocks), dim3(kBlock), 0, st_, mean, logstd, actions, old_logp, adv,
                       active, clip, dmean, workspace, (long long)R, (int)A, (int)K);
    const long long nseg = reduce_stage1(workspace, blocks, kPpoP, kPpoP, st_);
    hipLaunchKernelGGL(reduce_partials_k, dim3(1), dim3(kBlock), 0, st_, workspace, nseg, kPpoP, kSegWaves * kPpoP, sums);
    return launch_status(__func__);
}

DCC_API int dcc_ppo_value_loss(const float* values, const float* value_preds, const float* returns, const float* active,
                               const float* norm, float clip, float delta, int32_t use_clipped, float* dvalues, float* sums,
                               float* workspace, int64_t n, int32_t N, void* stream) {
    if (!values || !value_preds || !returns || !dvalues || !sums || !workspace || n < 1 || N < 1) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    long long blocks = (n + kBlock - 1) / kBlock;
    if (blocks > kReluLnBlocks * 2) blocks = kReluLnBlocks * 2;
    hipLaunchKernelGGL(ppo_value_loss_k, dim3((unsigned)blocks), dim3(kBlock), 0, st_, values, value_preds, returns, active,
                       norm, clip, delta, (int)use_clipped, dvalues, workspace, (long long)n, (int)N);
    const long long nseg = reduce_stage1(workspace, blocks, 2, 2, st_);
    hipLaunchKernelGGL(reduce_partials_k, dim3(1), dim3(kBlock), 0, st_, workspace, nseg, 2, kSegWaves * 2, sums);
    return launch_status(__func__);
}

DCC_API int dcc_rollout_sample(const float* mean, const float* logstd, const float* eps, const float* value, float* actions,
                               float* logp, float* value_preds, int64_t R, int32_t N, int32_t A, int32_t K, void* stream) {
    if (!mean || !logstd || !eps || !actions || !logp || R < 1 || N < 1 || (value_preds && !value)) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    if (A < 1 || A > kAMax || K < 1 || K > kAMax) return dcc_fail(kEUNSUPPORTED, std::string(__func__) + ": shape outside the compiled variants (hidden width / head width / output width)");
    hipLaunchKernelGGL(rollout_sample_k, dim3((unsigned)((R + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), mean, logstd, eps, value, actions, logp, value_preds, (long long)R,
                       (int)N, (int)A, (int)K);
    return launch_status(__func__);
}

DCC_API int dcc_rollout_record(const float* reward, const uint8_t* done, float* rewards, float* masks_next, int64_t R,
                               int32_t N, void* stream) {
    if (!reward || !done || !rewards || !masks_next || R < 1 || N < 1) return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipLaunchKernelGGL(rollout_record_k, dim3((unsigned)((R + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reward, done, rewards, masks_next, (long long)R, (int)N,
                       (const float*)nullptr, (double*)nullptr, (float*)nullptr);
    return launch_status(__func__);
}

DCC_API int dcc_rollout_record_stats(const float* reward, const uint8_t* done, const float* coverage, float* rewards,
                                     float* masks_next, double* rew_acc, float* cov_max, int64_t R, int32_t N, void* stream) {
    if (!reward || !done || !rewards || !masks_next || R < 1 || N < 1 || (cov_max && !coverage))
        return dcc_fail(kEINVAL, std::string(__func__) + ": invalid argument (null pointer or size)");
    hipLaunchKernelGGL(rollout_record_k, dim3((unsigned)((R + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       reinterpret_cast<hipStream_t>(stream), reward, done, rewards, masks_next, (long long)R, (int)N, coverage,
                       rew_acc, cov_max);
    return launch_status(__func__);
}

}  // extern "C"

namespace {

// pre == NULL: the per-row term is head . Wh^T, formed in the kernel; pre != NULL (HD = 0): it is read from pre [n*N, H]
int l1_fwd_impl(const char* fn, const float* head, const float* pre, const float* G, const double* stats, const float* Wh,
                const float* s, const float* c, const float* gamma, const float* beta, float eps_in, float eps_ln,
                int32_t D, float* h, int64_t n, int32_t N, int32_t HD, int32_t H, void* stream) {
    if ((!pre && (!head || !Wh)) || !G || !s || !c || !gamma || !beta || !h || n < 1 || N < 1 || D < 1) return dcc_fail(kEINVAL, std::string(fn) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    if (!pick_shape(H, sh) || HD < 0 || HD > kHdMax || pad_hd(HD) < 0) return dcc_fail(kEUNSUPPORTED, std::string(fn) + ": shape outside the compiled variants (hidden width / head width / output width)");
    const size_t lds = (size_t)HD * H * sizeof(float);
    if (lds > kLdsMax) return dcc_fail(kEUNSUPPORTED, std::string(fn) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (lds > 64 * 1024) {
        const bool okl = sh.vec == 4 ? (sh.vpl == 1 ? allow_lds(actor_l1_fwd_k<4, 1>, lds) : allow_lds(actor_l1_fwd_k<4, 2>, lds))
                                     : (sh.vpl == 1 ? allow_lds(actor_l1_fwd_k<1, 1>, lds) : allow_lds(actor_l1_fwd_k<1, 2>, lds));
        if (!okl) return dcc_fail(kEHIP, std::string(fn) + ": could not raise the dynamic LDS limit");
    }
    if (pre && HD != 0) return dcc_fail(kEINVAL, std::string(fn) + ": a ready-made per-row term excludes head columns (HD must be 0)");
    if (sh.vec == 4 && !(aligned16(G) && aligned16(h) && aligned16(gamma) && aligned16(beta) && aligned16(s) && aligned16(c) && aligned16(pre)))
        return dcc_fail(kEINVAL, std::string(fn) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    if (!pre && H == 256 && HD == 4 + 2 * (N - 1) && (N == 8 || N == 4) && aligned16(stats) && aligned16(head)) {
        // BASELINE sizes: Wh^T in registers, inputs fetched an env ahead (actor_l1_fwd_env_k)
        static int res8 = 0, res4 = 0;
        if (N == 8) {
            if (!res8) res8 = resident_blocks(reinterpret_cast<const void*>(&actor_l1_fwd_env_k<8>));
            hipLaunchKernelGGL((actor_l1_fwd_env_k<8>), dim3((int)waves_for(n, res8)), dim3(kBlock), 0, st_, head, G, stats, Wh, s, c, gamma,
                               beta, eps_in, eps_ln, (int)D, h, (long long)n);
        } else {
            if (!res4) res4 = resident_blocks(reinterpret_cast<const void*>(&actor_l1_fwd_env_k<4>));
            hipLaunchKernelGGL((actor_l1_fwd_env_k<4>), dim3((int)waves_for(n, res4)), dim3(kBlock), 0, st_, head, G, stats, Wh, s, c, gamma,
                               beta, eps_in, eps_ln, (int)D, h, (long long)n);
        }
        return launch_status(fn);
    }
    const int grid = (int)waves_for(n, kL1Blocks * 2);
    LAUNCH_SHAPE(actor_l1_fwd_k, grid, lds, head, pre, G, stats, Wh, s, c, gamma, beta, eps_in, eps_ln, (int)D, h,
                 (long long)n, (int)N, (int)HD, (int)H);
    return launch_status(fn);
}


int l1_bwd_impl(const char* fn, const float* head, const float* pre, const float* G, const double* stats, const float* Wh,
                const float* s, const float* c, const float* gamma, const float* dh, float eps_in, float eps_ln, int32_t D,
                float* dG, float* dWh, float* dq, float* ds, float* dc, float* dgamma, float* dbeta, float* workspace,
                int64_t n, int32_t N, int32_t HD, int32_t H, void* stream) {
    if ((!pre && (!head || !Wh)) || (pre && (HD != 0 || !dq)) || !G || !s || !c || !gamma || !dh || !dG || (HD > 0 && !dWh && !dq) || !ds || !dc || !dgamma || !dbeta ||
        !workspace || n < 1 || N < 1 || D < 1)
        return dcc_fail(kEINVAL, std::string(fn) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    Shape sh;
    const int hdp = pad_hd(HD);
    if (!pick_shape(H, sh) || HD < 0 || HD > kHdMax || hdp < 0) return dcc_fail(kEUNSUPPORTED, std::string(fn) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (hdp == kHdQOnly && !dq) return dcc_fail(kEUNSUPPORTED, std::string(fn) + ": more than 40 head columns need the q-storing form (dq != NULL)");
    const size_t lds = (size_t)HD * H * sizeof(float);
    if (lds > kLdsMax) return dcc_fail(kEUNSUPPORTED, std::string(fn) + ": shape outside the compiled variants (hidden width / head width / output width)");
    if (lds > 64 * 1024) {
        const bool okl = sh.vec == 4 ? (sh.vpl == 1 ? allow_lds(actor_l1_bwd_k<4, 1, 0>, lds) : allow_lds(actor_l1_bwd_k<4, 2, 0>, lds))
                                     : (sh.vpl == 1 ? allow_lds(actor_l1_bwd_k<1, 1, 0>, lds) : allow_lds(actor_l1_bwd_k<1, 2, 0>, lds));
        if (!okl) return dcc_fail(kEHIP, std::string(fn) + ": could not raise the dynamic LDS limit");
    }
    if (sh.vec == 4 && !(aligned16(G) && aligned16(dh) && aligned16(dG) && aligned16(gamma) && aligned16(s) &&
                         aligned16(c) && aligned16(workspace) && aligned16(pre)))
        return dcc_fail(kEINVAL, std::string(fn) + ": invalid argument (null pointer, size, or an array that is not 16-byte aligned)");
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const int grid = (int)waves_for(n, kL1Blocks);
    int hdp_used = hdp;
    int grid_used = grid;
    const bool env_ok = !pre && H == 256 && HD == 4 + 2 * (N - 1) && (N == 8 || N == 4) && aligned16(stats) && aligned16(head);
    bool done_launch = false;
    if (env_ok) {   // BASELINE sizes: actor_l1_bwd_env_k (Wh^T in registers, env-ahead prefetch), grid = what is co-resident
        if (dq && sh.vec == 4 && !aligned16(dq)) return dcc_fail(kEINVAL, std::string(fn) + ": dq must be 16-byte aligned");
        static int res[4] = {0, 0, 0, 0};
        const int v = (N == 8 ? 0 : 1) + (dq ? 0 : 2);
        const void* fns[4] = {reinterpret_cast<const void*>(&actor_l1_bwd_env_k<8, false>), reinterpret_cast<const void*>(&actor_l1_bwd_env_k<4, false>),
                              reinterpret_cast<const void*>(&actor_l1_bwd_env_k<8, true>), reinterpret_cast<const void*>(&actor_l1_bwd_env_k<4, true>)};
        if (!res[v]) res[v] = resident_blocks(fns[v]);
        grid_used = (int)waves_for(n, dq ? kL1Blocks * 2 : kL1Blocks);       // bounded by the workspace the caller sized
        if (res[v] < grid_used) grid_used = res[v];
#define L1B_ENV(NRV, ACC) hipLaunchKernelGGL((actor_l1_bwd_env_k<NRV, ACC>), dim3(grid_used), dim3(kBlock), 0, st_, head, G, stats, Wh, s, c, \
                                             gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n)
        if (N == 8) { if (dq) L1B_ENV(8, false); else L1B_ENV(8, true); }
        else { if (dq) L1B_ENV(4, false); else L1B_ENV(4, true); }
#undef L1B_ENV
        hdp_used = dq ? 0 : HD;
        done_launch = true;
    } else if (dq) {   // two-kernel variant: light registers -> full occupancy, so use the larger grid too
        if (sh.vec == 4 && !aligned16(dq)) return dcc_fail(kEINVAL, std::string(fn) + ": dq must be 16-byte aligned");
        grid_used = (int)waves_for(n, kL1Blocks * 2);
        launch_l1_bwd<0>(sh, grid_used, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H);
    } else {
        switch (hdp) {
            case 0: launch_l1_bwd<0>(sh, grid, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H); break;
            case 8: launch_l1_bwd<8>(sh, grid, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H); break;
            case 16: launch_l1_bwd<16>(sh, grid, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H); break;
            case 24: launch_l1_bwd<24>(sh, grid, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H); break;
            default: launch_l1_bwd<40>(sh, grid, lds, st_, head, pre, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, (int)D, dG, dq, workspace, (long long)n, (int)N, (int)HD, (int)H); break;
        }
    }
    (void)done_launch;
    if (!env_ok && dq) hdp_used = 0;
    const int P = (hdp_used + 4) * H;
    const long long nseg = reduce_stage1(workspace, (long long)grid_used * kWavesPerBlock, P, P, st_);
    hipLaunchKernelGGL(l1_reduce_k, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, st_, workspace, nseg,
                       (long long)kSegWaves * P, hdp_used, (int)HD, (int)H, dWh, ds, dc, dgamma, dbeta);
    return launch_status(fn);
}

}  // namespace

extern "C" {

DCC_API int dcc_actor_l1_fwd(const float* head, const float* G, const double* stats, const float* Wh, const float* s,
                             const float* c, const float* gamma, const float* beta, float eps_in, float eps_ln,
                             int32_t D, float* h, int64_t n, int32_t N, int32_t HD, int32_t H, void* stream) {
    return l1_fwd_impl(__func__, head, nullptr, G, stats, Wh, s, c, gamma, beta, eps_in, eps_ln, D, h, n, N, HD, H, stream);
}

DCC_API int dcc_actor_l1_bwd(const float* head, const float* G, const double* stats, const float* Wh, const float* s,
                             const float* c, const float* gamma, const float* dh, float eps_in, float eps_ln, int32_t D,
                             float* dG, float* dWh, float* dq, float* ds, float* dc, float* dgamma, float* dbeta,
                             float* workspace, int64_t n, int32_t N, int32_t HD, int32_t H, void* stream) {
    return l1_bwd_impl(__func__, head, nullptr, G, stats, Wh, s, c, gamma, dh, eps_in, eps_ln, D, dG, dWh, dq, ds, dc, dgamma,
                       dbeta, workspace, n, N, HD, H, stream);
}

DCC_API int dcc_actor_l1_pre_fwd(const float* pre, const float* G, const double* stats, const float* s, const float* c,
                                 const float* gamma, const float* beta, float eps_in, float eps_ln, int32_t D, float* h,
                                 int64_t n, int32_t N, int32_t H, void* stream) {
    if (!pre) return dcc_fail(kEINVAL, std::string(__func__) + ": pre must not be NULL");
    return l1_fwd_impl(__func__, nullptr, pre, G, stats, nullptr, s, c, gamma, beta, eps_in, eps_ln, D, h, n, N, 0, H, stream);
}

DCC_API int dcc_actor_l1_pre_bwd(const float* pre, const float* G, const double* stats, const float* s, const float* c,
                                 const float* gamma, const float* dh, float eps_in, float eps_ln, int32_t D, float* dG,
                                 float* dpre, float* ds, float* dc, float* dgamma, float* dbeta, float* workspace, int64_t n,
                                 int32_t N, int32_t H, void* stream) {
    if (!pre || !dpre) return dcc_fail(kEINVAL, std::string(__func__) + ": pre and dpre must not be NULL");
    return l1_bwd_impl(__func__, nullptr, pre, G, stats, nullptr, s, c, gamma, dh, eps_in, eps_ln, D, dG, nullptr, dpre, ds, dc,
                       dgamma, dbeta, workspace, n, N, 0, H, stream);
}

}  // extern "C"
