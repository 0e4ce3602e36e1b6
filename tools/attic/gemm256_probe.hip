// fp32 MFMA GEMM core for the policy trunks: C[R,256] = A[R,256] . W[256,256]^T (row-major, W = nn.Linear.weight).
// Stage 1 of a fused trunk kernel: is a hand-written core within reach of the library's 129-146 TF/s?
//   hipcc --offload-arch=gfx950 -O3 tools/gemm256_probe.hip -o /tmp/gemm256 && /tmp/gemm256 [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
#ifndef VAR
#define VAR 0                   // timing experiments: 1 = no C stores, 2 = A loaded once, 3 = both
#endif
constexpr int H = 256;          // K and N of the GEMM
constexpr int BM = 128;         // rows per workgroup
constexpr int KC = 16;          // k per LDS stage
constexpr int LS = 20;          // floats per LDS row (16 + 4 pad: conflict-free b128 reads)
constexpr int NT = 256;         // threads: 4 waves, wave (wm, wn) owns 64 rows x 128 columns = 2 x 4 tiles of 32 x 32

// Which two k values an MFMA's two k-slots carry is free as long as A and B agree: slot 0 (lanes 0-31) takes k = s of the
// chunk, slot 1 (lanes 32-63) k = 8 + s, so a lane reads its 8 values of a chunk (k-steps s = 0..7) as two b128 from a
// plain row-major row.
__device__ __forceinline__ void stage_store(float* dst_row, int kq, const float4 v) {
    *reinterpret_cast<float4*>(dst_row + 4 * kq) = v;
}

__global__ __launch_bounds__(NT, 2) void gemm256_k(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, long long R) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                          // [2][BM][LS]
    float* Bs = lds + 2 * BM * LS;            // [2][H][LS]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long long row0 = (long long)blockIdx.x * BM;
    // global -> register staging: A: 2 float4 per thread (rows t/4 + 64 i), B: 4 float4 (rows t/4 + 64 i)
    const int lr = t >> 2, kq = t & 3;
    float4 ga[2], gb[4];
    auto gload = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long r = row0 + lr + 64 * i;
            ga[i] = r < R ? *reinterpret_cast<const float4*>(A + r * H + kc * KC + kq * 4) : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) gb[i] = *reinterpret_cast<const float4*>(W + (size_t)(lr + 64 * i) * H + kc * KC + kq * 4);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) stage_store(As + (buf * BM + lr + 64 * i) * LS, kq, ga[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_store(Bs + (buf * H + lr + 64 * i) * LS, kq, gb[i]);
    };
    v16f acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    gload(0);
    lstore(0);
    gload(1);
    __syncthreads();
    const int l32 = lane & 31, par = lane >> 5;
    constexpr int NKC = H / KC;
    for (int kc = 0; kc < NKC; ++kc) {
        const int buf = kc & 1;
        // registers hold chunk kc + 1 (loaded a whole iteration ago): into the other LDS buffer (free since the barrier that
        // ended iteration kc - 1), then the loads of chunk kc + 2 start their trip before this chunk's MFMAs are issued
        if (kc + 1 < NKC) lstore(buf ^ 1);
        if (kc + 2 < NKC && !((VAR & 2) && kc > 0)) gload(kc + 2);
        float4 fa[2][2], fb[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* p = As + (buf * BM + wm * 64 + i * 32 + l32) * LS + par * 8;
            fa[i][0] = *reinterpret_cast<const float4*>(p); fa[i][1] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* p = Bs + (buf * H + wn * 128 + j * 32 + l32) * LS + par * 8;
            fb[j][0] = *reinterpret_cast<const float4*>(p); fb[j][1] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float a = reinterpret_cast<const float*>(&fa[i][0])[s];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b = reinterpret_cast<const float*>(&fb[j][0])[s];
                    // M = output column (W row), N = data row: a lane ends up with 4 CONSECUTIVE COLUMNS of one row per
                    // accumulator quad (float4 stores; a row's 256 outputs sit in 2 lanes x 2 waves: cheap row reductions)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: lane = row l32 of the i-th 32-row tile; accumulator quad q of column tile j = columns 8q + 4 par .. + 3.
    // Stored straight from the registers that is 16 bytes per row and instruction (partial lines: 0.55 ms of 5.6); instead
    // each wave transposes 32-row x 64-column pieces through its share of the (now idle) LDS and stores 256 contiguous bytes
    // per row: full 128-byte lines only.
    float* tp = lds + w * (32 * 68);          // 8.5 KB per wave
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = jp * 2 + jj;
                    *reinterpret_cast<float4*>(tp + l32 * 68 + jj * 32 + q * 8 + par * 4) =
                        make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int rr = g * 4 + (lane >> 4);
                const long long r = row0 + wm * 64 + i * 32 + rr;
                const float4 o = *reinterpret_cast<const float4*>(tp + rr * 68 + (lane & 15) * 4);
                if ((VAR & 1) ? (o.x == 123.456f) : (r < R))
                    *reinterpret_cast<float4*>(C + r * H + wn * 128 + jp * 64 + (lane & 15) * 4) = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
}

int main(int argc, char** argv) {
    const long long R = argc > 1 ? atoll(argv[1]) : 4915200;
    float *A, *W, *C;
    CK(hipMalloc(&A, R * H * 4)); CK(hipMalloc(&W, H * H * 4)); CK(hipMalloc(&C, R * H * 4));
    std::vector<float> hA((size_t)4096 * H), hW((size_t)H * H);
    srand(1);
    for (auto& x : hA) x = (float)rand() / RAND_MAX - 0.5f;
    for (auto& x : hW) x = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    // fill A: the first 4096 rows with hA, the rest by repeating device-side copies
    for (long long off = 0; off < R; off += 4096) {
        const long long n = (R - off) < 4096 ? (R - off) : 4096;
        CK(hipMemcpy(A + off * H, hA.data(), n * H * 4, hipMemcpyHostToDevice));
        if (off >= 65536) { // after a few host copies use D2D doubling
            long long have = off + n;
            while (have < R) { const long long m = (R - have) < have ? (R - have) : have; CK(hipMemcpy(A + have * H, A, m * H * 4, hipMemcpyDeviceToDevice)); have += m; }
            break;
        }
    }
    CK(hipMemcpy(W, hW.data(), H * H * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)(2 * BM * LS + 2 * H * LS) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (int)((R + BM - 1) / BM);
    hipLaunchKernelGGL(gemm256_k, dim3(grid), dim3(NT), lds, 0, A, W, C, R);
    CK(hipDeviceSynchronize());
    // check rows 0..63 and the last 64 rows against float64 on the host
    std::vector<float> hC((size_t)64 * H), hC2((size_t)64 * H);
    CK(hipMemcpy(hC.data(), C, 64 * H * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC2.data(), C + (R - 64) * H, 64 * H * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int r = 0; r < 64; ++r)
        for (int o = 0; o < H; ++o) {
            double s = 0, s2 = 0;
            const long long rr = (R - 64 + r) % 4096;
            for (int k = 0; k < H; ++k) { s += (double)hA[(size_t)r * H + k] * hW[(size_t)o * H + k]; s2 += (double)hA[(size_t)rr * H + k] * hW[(size_t)o * H + k]; }
            maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * H + o]));
            maxerr = fmax(maxerr, fabs(s2 - hC2[(size_t)r * H + o]));
        }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm256_k, dim3(grid), dim3(NT), lds, 0, A, W, C, R);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("rows %lld: %.3f ms  %.1f TF/s  (max abs err vs float64 %.2e; lds %zu B, grid %d)\n", R, ms, 2.0 * R * H * H / ms / 1e9, maxerr, lds, grid);
    return 0;
}
