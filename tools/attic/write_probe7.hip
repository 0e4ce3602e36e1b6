// Team sweep: the chip writes inside a SMALL address window that sweeps the [K, E] observation blocks in index order, with
// every hand-off inside a workgroup.  `teams` persistent workgroups; team b stores env block (t * teams + b) at iteration t, its
// `tw` writer waves each taking either a contiguous 1/tw of the block (split = 0: what produce_obs_rows emits, a range of agent
// rows per wave) or interleaved 1 KB pieces (split = 1).  Window = teams x 10,816 B (256 teams: 2.7 MB; write_probe6's
// one-block-per-wave sweep had 1024 blocks = 11 MB in flight and streamed at the scattered rate).  `pw` extra waves per team burn
// valu_p dependent float64 FMAs per env block of their share (the physics waves, running ahead through an LDS ring in the real
// kernel), the writers valu_w per block share.
// Build: hipcc --offload-arch=gfx950 -O3 -o write_probe7 write_probe7.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double spin_fma(double x, int n, double a, double b) {
    double y = x + 1.0, z = x + 2.0, w = x + 3.0;
    for (int i = 0; i < n; i += 4) {
        x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w = __builtin_fma(w, a, b);
    }
    return (x + y) + (z + w);
}

__global__ void fill(float4* p, long total_blocks, int blk4, int tw, int pw, int split, int valu_w, int valu_p, double a, double b,
                     float* sink) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int teams = gridDim.x;
    double acc = (double)lane;
    if (wv >= tw) {                       // physics wave q of pw: every pw-th block of the team
        const int q = wv - tw;
        for (long c = blockIdx.x + (long)q * teams; c < total_blocks; c += (long)teams * pw) acc = spin_fma(acc, valu_p, a, b);
    } else {
        for (long c = blockIdx.x; c < total_blocks; c += teams) {
            float4* g = p + (size_t)c * blk4;
            acc = spin_fma(acc, valu_w, a, b);
            const float v = (float)acc;
            const float4 x = make_float4(v, v, v, v);
            if (split == 0) {
                const int per = (blk4 + tw - 1) / tw;
                const int lo = wv * per, hi = (lo + per) < blk4 ? (lo + per) : blk4;
                for (int i = lo + lane; i < hi; i += 64) g[i] = x;
            } else {
                for (int i = wv * 64 + lane; i < blk4; i += tw * 64) g[i] = x;
            }
        }
    }
    if (acc == 12345.678) sink[0] = (float)acc;
}

template <typename F> double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 3.0 * bytes / ms / 1e6;
}

int main(int argc, char** argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 4;
    const int K = 150, E = 4096; const size_t blk = 10816, bytes = (size_t)K * E * blk;
    float* sink; CK(hipMalloc(&sink, 64));
    const double a = 0.999999, b = 1e-9;
    for (int bi = 0; bi < nbuf; ++bi) {
        float4* buf; CK(hipMalloc(&buf, bytes));
        printf("buffer %d: memset %5.0f\n", bi, timeit([&] { CK(hipMemsetAsync(buf, 0x5a, bytes, 0)); }, bytes));
        auto run = [&](int teams, int tw, int pw, int split, int vw, int vp) {
            const double r = timeit([&] { fill<<<teams, (tw + pw) * 64>>>(buf, (long)K * E, (int)(blk / 16), tw, pw, split, vw, vp, a, b, sink); }, bytes);
            printf("  teams=%4d writers=%2d physics=%2d %s valu_w=%3d valu_p=%4d window=%5.1f MB : %5.0f GB/s  (%.3f ms per 150 steps)\n", teams, tw,
                   pw, split ? "1KB-interleaved" : "contiguous-1/tw ", vw, vp, teams * (double)blk / 1e6, r, bytes / r / 1e6);
        };
        for (int split : {0, 1})
            for (int teams : {64, 128, 256, 512, 1024})
                for (int tw : {1, 2, 4, 8}) {
                    if (teams * tw < 256 || teams * tw > 4096) continue;
                    run(teams, tw, 0, split, 0, 0);
                }
        // with compute: writers 40 valu per quarter block, physics 300 per block
        for (int teams : {128, 256, 512}) {
            run(teams, 4, 0, 0, 40, 0);
            run(teams, 4, 4, 0, 40, 300);
            run(teams, 4, 4, 0, 40, 600);
            run(teams, 4, 4, 0, 40, 1200);
            run(teams, 4, 8, 0, 40, 1200);
            run(teams, 8, 8, 0, 20, 1200);
        }
        fflush(stdout);
    }
    return 0;
}
