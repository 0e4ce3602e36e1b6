// Few, fat streams: the c2 K-step store pattern with G envs per writer (run = G x 10,816 B per step) and `team` waves sweeping each run
// together (1 KB pieces, interleaved) -- what a big-workgroup env kernel (all hand-offs through LDS, no cross-workgroup sync) would emit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill_steps(float4* p, int K, int E, size_t blk4, int G, int team, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t t = wave / team; const int q = (int)(wave % team);
    if ((t + 1) * G > (size_t)E) return;
    const float4 x = make_float4(v, v, v, v);
    const size_t run4 = blk4 * G;
    for (int k = 0; k < K; ++k) {
        float4* g = p + ((size_t)k * E + t * G) * blk4;
        for (size_t i = (size_t)q * 64 + lane; i < run4; i += (size_t)team * 64) g[i] = x;
    }
}
template <typename F> double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 3.0 * bytes / ms / 1e6;
}
int main() {
    const int K = 150, E = 4096; const size_t blk = 10816, bytes = (size_t)K * E * blk;
    for (int b = 0; b < 3; ++b) {
        float4* a; CK(hipMalloc(&a, bytes));
        printf("buffer %d: memset %5.0f | G2/t1 %5.0f |", b, timeit([&] { CK(hipMemsetAsync(a, 0x5a, bytes, 0)); }, bytes),
               timeit([&] { fill_steps<<<512, 256>>>(a, K, E, blk / 16, 2, 1, 1.f); }, bytes));
        for (int G : {16, 32, 64, 128, 256, 512, 1024, 4096})
            for (int team : {4, 8, 16, 32}) {
                const int writers = E / G; const long waves = (long)writers * team;
                if (waves < 128 || waves > 4096) continue;
                const int block = team >= 4 ? (team >= 16 ? 1024 : team * 64) : 256;
                const int grid = (int)((waves * 64 + block - 1) / block);
                printf(" G%d/t%d(%ldw) %5.0f", G, team, waves, timeit([&] { fill_steps<<<grid, block>>>(a, K, E, blk / 16, G, team, 1.f); }, bytes));
            }
        printf("\n");
    }
    return 0;
}
