// Address-ordered chunk sweep: the store pattern of a PERSISTENT env kernel whose W writer waves walk the [K, E] observation
// blocks in index order -- writer P stores chunk (t * W + P) at iteration t, a chunk being G adjacent env blocks (G x 10,816 B
// at c2) -- against the current pattern (every writer keeps its own 2 envs: 2048 writers spread over the 44 MB of a step).
// Variants: W in {512, 1024, 2048} writers, G in {1, 2, 4} envs per chunk, and `valu` dependent float64 FMAs per env block issued
// (a) by the writer itself between its stores or (b) by a second, non-storing wave per writer (the physics wave of a role pair).
// Build: hipcc --offload-arch=gfx950 -O3 -o write_probe6 write_probe6.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double spin_fma(double x, int n, double a, double b) {
    double y = x + 1.0, z = x + 2.0, w = x + 3.0;
    for (int i = 0; i < n; i += 4) {      // four independent chains: what an unrolled physics step offers the scheduler
        x = __builtin_fma(x, a, b); y = __builtin_fma(y, a, b); z = __builtin_fma(z, a, b); w = __builtin_fma(w, a, b);
    }
    return (x + y) + (z + w);
}

// mode 0: current pattern (writer t owns envs [t*G, t*G+G) for all K steps)
// mode 1: sweep (writer P, iteration t -> chunk t*W + P)
// roles: 1 = every wave writes; 2 = even waves write, odd waves only compute (valu_phys per env block)
__global__ void fill(float4* p, int K, int E, int blk4, int G, int mode, int roles, int valu_w, int valu_p, double a, double b,
                     float* sink) {
    __shared__ float4 lds[16 * 64 * 4];
    const int lane = threadIdx.x & 63;
    const int wave_in_block = threadIdx.x >> 6;
    const int waves_per_block = blockDim.x >> 6;
    const int gw = blockIdx.x * waves_per_block + wave_in_block;
    const int writer = roles == 2 ? gw >> 1 : gw;
    const bool is_writer = roles == 2 ? (gw & 1) == 0 : true;
    const int W = (gridDim.x * waves_per_block) / roles;
    const int chunks_per_step = E / G;
    const size_t run4 = (size_t)blk4 * G;
    double acc = (double)lane;
    if (mode == 0) {
        if (writer >= chunks_per_step) return;
        for (int k = 0; k < K; ++k) {
            if (!is_writer) { acc = spin_fma(acc, valu_p * G, a, b); continue; }
            float4* g = p + ((size_t)k * chunks_per_step + writer) * run4;
            for (int e = 0; e < G; ++e) {
                acc = spin_fma(acc, valu_w, a, b);
                const float v = (float)acc;
                const float4 x = make_float4(v, v, v, v);
                for (int i = lane; i < blk4; i += 64) g[(size_t)e * blk4 + i] = x;
            }
        }
    } else {
        const long total = (long)K * chunks_per_step;
        for (long c = writer; c < total; c += W) {
            if (!is_writer) { acc = spin_fma(acc, valu_p * G, a, b); continue; }
            float4* g = p + (size_t)c * run4;
            for (int e = 0; e < G; ++e) {
                acc = spin_fma(acc, valu_w, a, b);
                const float v = (float)acc;
                const float4 x = make_float4(v, v, v, v);
                for (int i = lane; i < blk4; i += 64) g[(size_t)e * blk4 + i] = x;
            }
        }
    }
    if (acc == 12345.678) sink[0] = (float)acc + lds[lane].x;
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
        auto run = [&](const char* name, int W, int G, int mode, int roles, int block, int vw, int vp) {
            const int waves = W * roles;
            const int grid = (waves * 64 + block - 1) / block;
            const double r = timeit([&] { fill<<<grid, block>>>(buf, K, E, (int)(blk / 16), G, mode, roles, vw, vp, a, b, sink); }, bytes);
            printf("  %-34s W=%4d G=%d block=%4d valu_w=%3d valu_p=%3d : %5.0f GB/s  (%.3f ms per 150 steps)\n", name, W, G, block, vw, vp, r,
                   bytes / r / 1e6);
        };
        run("current pattern, stores only", 2048, 2, 0, 1, 256, 0, 0);
        run("current pattern + own valu", 2048, 2, 0, 1, 256, 160, 0);
        run("current pattern, role pairs", 2048, 2, 0, 2, 128, 160, 300);
        for (int W : {512, 1024, 2048})
            for (int G : {1, 2, 4}) {
                run("sweep, stores only", W, G, 1, 1, 256, 0, 0);
            }
        for (int W : {512, 1024})
            for (int G : {1, 2}) {
                run("sweep + own valu 160", W, G, 1, 1, 256, 160, 0);
                run("sweep + own valu 460", W, G, 1, 1, 256, 460, 0);
                run("sweep, role pairs 160/300", W, G, 1, 2, 512, 160, 300);
                run("sweep, role pairs 160/600", W, G, 1, 2, 512, 160, 600);
            }
        run("sweep, role pairs 160/300 b1024", 1024, 2, 1, 2, 1024, 160, 300);
        run("sweep, role pairs 160/300 b256", 1024, 2, 1, 2, 256, 160, 300);
        fflush(stdout);
    }
    return 0;
}
