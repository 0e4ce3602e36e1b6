// dcc_optim.hip -- gradient-norm clipping + Adam on flat parameter storage (include/dcc_optim.h).
//
// Bound: HBM (4 arrays read, 3 written per element; 0.9 M - 44 M elements, i.e. launch- to bandwidth-bound).
// Layout: thread t of block b owns the float4 groups {b*256 + t + k * gridDim*256}: every access of a wave is 1 KB of
// consecutive bytes.  The norm is reduced in a FIXED order (per-thread strided sum -> DPP wave sum -> block sum ->
// per-block partials -> one block adds the partials sequentially), so the clip coefficient -- and with it the whole
// update -- is bit-reproducible, which the checkpoint/resume test relies on.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "dcc_internal.h"
#include "dcc_optim.h"

#pragma clang fp contract(fast)

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;
constexpr int kEINVAL = -1, kEHIP = -2;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(kBlock) void sqnorm_partial_k(const float* __restrict__ g, long long n, float* __restrict__ partials) {
    __shared__ float red[kBlock / 64];
    const long long n4 = n >> 2;
    float s0 = 0.f, s1 = 0.f;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        const float4 v = g4[i];
        s0 += v.x * v.x + v.y * v.y;
        s1 += v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s0 += v * v; }
    const float w = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void norm_finish_k(const float* __restrict__ partials, int nblk, float max_norm, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partials[i];
    const float norm = sqrtf(s);
    out[0] = norm;
    float c = 1.f;
    if (max_norm > 0.f) { c = max_norm / (norm + 1e-6f); c = c > 1.f ? 1.f : c; }   // clip_grad_norm_: clamp(max=1.0)
    out[1] = c;
}

// omb1 / omb2 = 1 - beta1 / 1 - beta2 rounded from DOUBLE precision, as torch passes them (1.f - 0.999f is off by 1.3e-5 relative)
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float clip, float step_size, float bc2_sqrt,
                                      float omb1, float beta2, float omb2, float eps, float wd) {
    g *= clip;
    if (wd != 0.f) g += wd * p;
    m = m + (g - m) * omb1;                         // exp_avg.lerp_(grad, 1 - beta1)
    v = v * beta2 + omb2 * g * g;                   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);                // param.addcdiv_(exp_avg, denom, value = -step_size)
}

__global__ __launch_bounds__(kBlock) void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long long n, float step_size, float bc2_sqrt,
                                                 float omb1, float beta2, float omb2, float eps, float wd,
                                                 const float* __restrict__ clip_p) {
    const float clip = clip_p ? *clip_p : 1.f;
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, clip, step_size, bc2_sqrt, omb1, beta2, omb2, eps, wd);
        adam1(pp.y, gg.y, mm.y, vv.y, clip, step_size, bc2_sqrt, omb1, beta2, omb2, eps, wd);
        adam1(pp.z, gg.z, mm.z, vv.z, clip, step_size, bc2_sqrt, omb1, beta2, omb2, eps, wd);
        adam1(pp.w, gg.w, mm.w, vv.w, clip, step_size, bc2_sqrt, omb1, beta2, omb2, eps, wd);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        adam1(p[i], g[i], m[i], v[i], clip, step_size, bc2_sqrt, omb1, beta2, omb2, eps, wd);
    }
}

int blocks_for(long long n) {
    long long b = ((n >> 2) + kBlock - 1) / kBlock;
    if (b > kMaxBlocks) b = kMaxBlocks;
    if (b < 1) b = 1;
    return (int)b;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" {

DCC_API int64_t dcc_grad_norm_workspace_floats(int64_t n) { return n < 1 ? 0 : blocks_for(n); }

DCC_API int dcc_grad_norm_clip(const float* grad, int64_t n, float max_norm, float* out, float* workspace, void* stream) {
    if (!grad || !out || !workspace || n < 1) return dcc_fail(kEINVAL, "dcc_grad_norm_clip: null pointer or n < 1");
    if (!aligned16(grad)) return dcc_fail(kEINVAL, "dcc_grad_norm_clip: grad must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nb = blocks_for(n);
    hipLaunchKernelGGL(sqnorm_partial_k, dim3(nb), dim3(kBlock), 0, st, grad, (long long)n, workspace);
    hipLaunchKernelGGL(norm_finish_k, dim3(1), dim3(64), 0, st, workspace, nb, max_norm, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : dcc_fail(kEHIP, std::string("dcc_grad_norm_clip: ") + hipGetErrorString(e));
}

DCC_API int dcc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float step_size,
                          float bc2_sqrt, double beta1, double beta2, float eps, float weight_decay, const float* clip,
                          void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1) return dcc_fail(kEINVAL, "dcc_adam_step: null pointer or n < 1");
    if (!(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq)))
        return dcc_fail(kEINVAL, "dcc_adam_step: arrays must be 16-byte aligned");
    if (!(bc2_sqrt > 0.f)) return dcc_fail(kEINVAL, "dcc_adam_step: bc2_sqrt must be > 0 (step >= 1)");
    hipLaunchKernelGGL(adam_k, dim3(blocks_for(n)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), param, grad, exp_avg,
                       exp_avg_sq, (long long)n, step_size, bc2_sqrt, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), eps, weight_decay, clip);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : dcc_fail(kEHIP, std::string("dcc_adam_step: ") + hipGetErrorString(e));
}

}  // extern "C"
