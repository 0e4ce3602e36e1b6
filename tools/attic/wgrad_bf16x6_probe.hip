// Split-bf16 ("bf16 x 6") fp32-equivalent WEIGHT-GRADIENT product of the policy trunks:  dW[256,256] = dZ[M,256]^T . H[M,256], fp32 in / out.
//
// The one GEMM of the PPO update whose operands are both streamed once and whose output is tiny: no weight image to re-stream from
// L2, no [M,256] result to store -- the two things that kept the forward form (tools/gemm_bf16x6_probe.hip) at the library's speed.
// 10 GB of HBM reads (1.6 ms at 6.3 TB/s) against 6 x 0.64 TFLOP of bf16 MFMA work (1.55 ms at the 2.5 PF/s peak); the library's
// fp32-MFMA split-K kernel takes 4.4 ms (146 TF/s = 0.93 of the fp32 matrix peak).
//
// Every fp32 value is the exact sum of three bf16 values (round to nearest), the product is accumulated in fp32 from the six bf16 MFMA
// products that matter (hh + hm + mh + hl + lh + mm; the dropped ones are <= 2^-23 of the product).  The contraction runs over the ROWS
// of both operands, so an MFMA fragment (8 consecutive contraction indices of one column per lane) is a COLUMN piece of a row-major
// matrix: a wave loads 8 full rows (one 1 KB-contiguous instruction each), every lane keeps the 8 x 2 block of its two columns, splits
// it in registers and writes complete 16-byte fragment entries to LDS -- the transpose costs nothing.
//
// One persistent workgroup of 8 waves per CU walks its share of the rows in steps of 16.  All waves are alike: each owns a 64 x 128
// block of dW (2 x 4 MFMA tiles, 128 accumulator registers), and each stages 1/8 of a step's operands (operand, row group, column
// half).  Loads are issued three steps ahead into a register ring, LDS is double-buffered, one barrier per step.  Partial sums of the
// workgroups go to HBM and are added in a fixed order by a second kernel (deterministic).
//   hipcc --offload-arch=gfx950 -O3 tools/wgrad_bf16x6_probe.hip -o /tmp/wgrad_x6 && /tmp/wgrad_x6 [rows]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

constexpr int H = 256;
#ifndef VAR
#define VAR 0               // timing experiments (results wrong): 1 = every step re-reads the first three steps' rows (L2-resident operands)
#endif

__device__ __forceinline__ unsigned pack_rne(float a, float b) {
    v2f v = {a, b};
    v2bf r = __builtin_convertvector(v, v2bf);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ v16f mfma(const uint4& a, const uint4& b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
// 8 values of one column (8 consecutive rows) -> the three bf16 fragment entries
__device__ __forceinline__ void split_col(const float (&x)[8], uint4& h, uint4& m, uint4& l) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        hh[i] = pack_rne(a, b);
        const float ra = a - lo_f(hh[i]), rb = b - hi_f(hh[i]);
        mm[i] = pack_rne(ra, rb);
        ll[i] = pack_rne(ra - lo_f(mm[i]), rb - hi_f(mm[i]));
    }
    h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    m = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

constexpr int OPB = 3 * 8 * 1024;        // one operand image of a step: [piece][column tile][lane] x 16 B = 24 KB
constexpr int SLOT = 2 * OPB;            // dZ image | H image

// steps [s0, s1) of 16 rows each for this workgroup; partial [gridDim.x][256 n][256 k]
__global__ __launch_bounds__(512, 1) void wgrad_x6(const float* __restrict__ dZ, const float* __restrict__ Hm, float* __restrict__ partial,
                                                    long long M, int steps_total, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // [2][SLOT]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per = (steps_total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int s0 = (int)blockIdx.x * per, s1 = min(steps_total, s0 + per);
    const int S = max(0, s1 - s0);
    // staging role: operand, row group of the step, column half
    const int op = w >> 2, kg = (w >> 1) & 1, ch = w & 1;
    const float* src = op ? Hm : dZ;
    const int c0 = 128 * ch + 2 * lane;                      // this lane's two columns
    // MFMA role: dW rows (n) 64 wn .. +63, columns (k) 128 wk .. +127
    const int wn = w >> 1, wk = w & 1;
    const int j = lane & 31, g = lane >> 5;

    float2 r0[8], r1[8];                                     // register ring: step s lives in slot s & 1
    auto load = [&](int s, float2 (&r)[8]) __attribute__((always_inline)) {
        const long long m0 = (long long)(s0 + ((VAR & 1) ? (s & 3) : s)) * 16 + 8 * kg;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const float2*>(src + (m0 + e) * H + c0);
    };
    auto stage = [&](int s, const float2 (&r)[8]) __attribute__((always_inline)) {
        unsigned char* img = lds + (s & 1) * SLOT + op * OPB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = q ? r[e].y : r[e].x;
            uint4 h, m, l;
            split_col(x, h, m, l);
            const int c = c0 + q;
            uint4* ent = reinterpret_cast<uint4*>(img) + ((c >> 5) * 64 + (c & 31) + 32 * kg);
            ent[0] = h; ent[8 * 64] = m; ent[16 * 64] = l;
        }
    };
    v16f acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    auto mma = [&](int s) __attribute__((always_inline)) {
        const uint4* za = reinterpret_cast<const uint4*>(lds + (s & 1) * SLOT);            // dZ image: A operand (tile rows = n)
        const uint4* hb = reinterpret_cast<const uint4*>(lds + (s & 1) * SLOT + OPB);      // H image: B operand (tile columns = k)
        uint4 af[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) af[p][t] = za[(p * 8 + 2 * wn + t) * 64 + lane];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const uint4 bh = hb[(0 * 8 + 4 * wk + kt) * 64 + lane];
            const uint4 bm = hb[(1 * 8 + 4 * wk + kt) * 64 + lane];
            const uint4 bl = hb[(2 * 8 + 4 * wk + kt) * 64 + lane];
            v16f a0 = acc[0][kt], a1 = acc[1][kt];
            a0 = mfma(af[2][0], bh, a0); a1 = mfma(af[2][1], bh, a1);     // small products first, the leading one last
            a0 = mfma(af[0][0], bl, a0); a1 = mfma(af[0][1], bl, a1);
            a0 = mfma(af[1][0], bm, a0); a1 = mfma(af[1][1], bm, a1);
            a0 = mfma(af[1][0], bh, a0); a1 = mfma(af[1][1], bh, a1);
            a0 = mfma(af[0][0], bm, a0); a1 = mfma(af[0][1], bm, a1);
            a0 = mfma(af[0][0], bh, a0); a1 = mfma(af[0][1], bh, a1);
            acc[0][kt] = a0; acc[1][kt] = a1;
        }
    };
    // iteration s: request step s+3, stage step s+1 (requested two iterations ago) into the other LDS slot, multiply step s.
    // Staging (VALU + LDS writes) and multiplying (LDS reads + MFMAs) are independent, and the 8 waves run in lock step (one barrier
    // per step): if all of them stage and then all of them multiply, the matrix pipes idle while everybody stages.
    // Waves w and w + 4 share a SIMD.  One of them multiplies first and stages afterwards, the other the other way round, so that on
    // every SIMD one wave keeps the matrix pipe busy while the other one occupies the VALU.
    // iteration s: multiply step s, stage step s+1 (slot (s+1) & 1, requested two iterations ago) and re-use its registers for step s+3
    auto iter_fast = [&](int s, float2 (&slot)[8], const bool mma_first) __attribute__((always_inline)) {
        const bool rec = dbg && blockIdx.x == 3 && (w == 0 || w == 4) && lane == 0 && s >= 64 && s < 80;
        long long t0 = 0, t1 = 0, t2 = 0;
        if (rec) t0 = clock64();
#ifndef STAGE_PRIO
#define STAGE_PRIO 3        // the staging phase outranks the other wave's MFMA stream on the SIMD (it is the one with many short instructions)
#endif
        if (mma_first) { mma(s); if (rec) t1 = clock64(); __builtin_amdgcn_s_setprio(STAGE_PRIO); stage(s + 1, slot); load(s + 3, slot); __builtin_amdgcn_s_setprio(0); }
        else { __builtin_amdgcn_s_setprio(STAGE_PRIO); stage(s + 1, slot); load(s + 3, slot); __builtin_amdgcn_s_setprio(0); if (rec) t1 = clock64(); mma(s); }
        if (rec) t2 = clock64();
        __syncthreads();
        if (rec) { long long* d = dbg + ((w >> 2) * 16 + (s - 64)) * 4; d[0] = t0; d[1] = t1; d[2] = t2; d[3] = clock64(); }
    };
    auto iter = [&](int s, float2 (&slot)[8]) __attribute__((always_inline)) {
        if (s >= S) return;
        if (s + 1 < S) stage(s + 1, slot);
        if (s + 3 < S) load(s + 3, slot);
        mma(s);
        __syncthreads();
    };
    if (S > 0) load(0, r0);
    if (S > 0) stage(0, r0);
    if (S > 1) load(1, r1);
    if (S > 2) load(2, r0);
    __syncthreads();
    int s = 0;
    if (w < 4) {
        for (; s + 5 < S; s += 2) { iter_fast(s, r1, true); iter_fast(s + 1, r0, true); }
    } else {
        for (; s + 5 < S; s += 2) { iter_fast(s, r1, false); iter_fast(s + 1, r0, false); }
    }
    for (; s < S; s += 2) { iter(s, r1); iter(s + 1, r0); }
    // D[i = n][j = k]: lane (j, g) holds column k = j of rows n = (v & 3) + 8 (v >> 2) + 4 g
    float* P = partial + (size_t)blockIdx.x * H * H;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                P[(size_t)(64 * wn + 32 * t + (v & 3) + 8 * (v >> 2) + 4 * g) * H + 128 * wk + 32 * kt + j] = acc[t][kt][v];
}

// rows [m_lo, M): dW[n][k] += sum_m dZ[m][n] H[m][k]  (the <= 15 rows beyond the last whole step; plain fp32)
__global__ void wgrad_tail(const float* __restrict__ dZ, const float* __restrict__ Hm, float* __restrict__ dW, long long m_lo, long long M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = i >> 8, k = i & 255;
    float s = 0.f;
    for (long long m = m_lo; m < M; ++m) s = fmaf(dZ[m * H + n], Hm[m * H + k], s);
    dW[i] += s;
}

// dW[i] (+)= sum over the workgroups' partials, in workgroup order (deterministic)
__global__ void wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dW, int nparts, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float s = accumulate ? dW[i] : 0.f;
    for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * H * H + i];
    dW[i] = s;
}

int main(int argc, char** argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 4915200LL;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = argc > 2 ? atoi(argv[2]) : prop.multiProcessorCount;
    std::vector<float> hz((size_t)4096 * H), hh((size_t)4096 * H);
    srand(11);
    auto nrm = [] { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
    for (auto& v : hz) v = (float)(nrm() * 2e-7);          // gradients of a mean loss over millions of rows are tiny
    for (auto& v : hh) v = (float)nrm();                    // LayerNorm outputs
    float *dZ, *Hm, *P, *dW;
    CK(hipMalloc(&dZ, (size_t)M * H * 4)); CK(hipMalloc(&Hm, (size_t)M * H * 4)); CK(hipMalloc(&P, (size_t)grid * H * H * 4)); CK(hipMalloc(&dW, H * H * 4));
    for (long long r = 0; r < M; r += 4096) {               // the same 4096 rows repeated: the exact result is (M / 4096) x one block's
        const long long n = (M - r) < 4096 ? (M - r) : 4096;
        CK(hipMemcpy(dZ + r * H, hz.data(), (size_t)n * H * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(Hm + r * H, hh.data(), (size_t)n * H * 4, hipMemcpyHostToDevice));
    }
    const int steps = (int)(M / 16);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_x6), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT));
    long long* dbg = nullptr; if (argc > 3) { CK(hipMalloc(&dbg, 2 * 16 * 4 * 8)); CK(hipMemset(dbg, 0, 2 * 16 * 4 * 8)); }
    auto run = [&] { wgrad_x6<<<grid, 512, 2 * SLOT>>>(dZ, Hm, P, M, steps, dbg); wgrad_reduce<<<H * H / 256, 256>>>(P, dW, grid, 0);
                      if (M % 16) wgrad_tail<<<H * H / 256, 256>>>(dZ, Hm, dW, (long long)steps * 16, M); };
    run(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<float> out((size_t)H * H);
    CK(hipMemcpy(out.data(), dW, out.size() * 4, hipMemcpyDeviceToHost));
    // float64 reference of the first 4096 rows' contribution, scaled (M is a multiple of 4096 in the default run)
    const double scale = (double)M / 4096.0;
    double worst = 0, sum = 0, worst32 = 0; int cnt = 0;
    for (int n = 0; n < H; n += 5)
        for (int k = 0; k < H; k += 7) {
            double ref = 0, sabs = 0; float f = 0.f;
            for (int m = 0; m < 4096; ++m) {
                const double p = (double)hz[(size_t)m * H + n] * (double)hh[(size_t)m * H + k];
                ref += p; sabs += std::fabs(p);
                f = std::fmaf(hz[(size_t)m * H + n], hh[(size_t)m * H + k], f);
            }
            const double e = std::fabs((double)out[(size_t)n * H + k] - ref * scale) / (sabs * scale);
            const double e32 = std::fabs((double)f - ref) / sabs;
            worst = e > worst ? e : worst; worst32 = e32 > worst32 ? e32 : worst32; sum += e; ++cnt;
        }
    if (dbg) {
        long long h[2 * 16 * 4]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int r = 0; r < 2; ++r) {
            printf("wave %d (%s first), steps 64..79: [first phase | second phase | barrier wait] cycles\n ", 4 * r, r ? "stage" : "mma");
            for (int i = 0; i < 16; ++i) printf(" [%lld|%lld|%lld]", h[(r * 16 + i) * 4 + 1] - h[(r * 16 + i) * 4], h[(r * 16 + i) * 4 + 2] - h[(r * 16 + i) * 4 + 1], h[(r * 16 + i) * 4 + 3] - h[(r * 16 + i) * 4 + 2]);
            printf("\n");
        }
    }
    printf("rows %lld grid %d VAR=%d: %.3f ms  %.1f TF/s fp32-equivalent (%.0f TF/s of bf16 MFMA work, %.0f GB/s of operand reads)\n", M, grid, VAR, ms,
           2.0 * M * H * H / ms / 1e9, 12.0 * M * H * H / ms / 1e9, 2.0 * M * H * 4 / ms / 1e6);
    printf("error / sum|dz.h| vs float64 over %d entries: bf16x6 max %.3e mean %.3e   |   a 4096-row fp32 FMA chain: max %.3e\n", cnt, worst, sum / cnt, worst32);
    return 0;
}
