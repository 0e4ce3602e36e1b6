// Pure-write probe of the c2 observation stream WITH the pacing round 6 found (a writer waits for its stores in flight after
// every P store instructions of 1 KB): does another geometry -- envs per writer G, pieces per pacing window P, writers mapped
// XCD-aware or not, one or two writer waves per workgroup -- beat what ships (G = 2, P = 4, XCD-aware)?
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe9.hip -o /tmp/write_probe9 && /tmp/write_probe9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// region = [K][E] blocks of blk4 float4; writer w owns envs [w*G, (w+1)*G) and at step k streams its G adjacent blocks.
// One writer wave per 64-thread workgroup (workgroup b runs on XCD b % 8); SWZ: the workgroups of one XCD take consecutive writers.
// env_lo / env_n: the launch covers envs [env_lo, env_lo + env_n) of the E-env layout (part launches: `halves` below).
template <int P, bool SWZ>
__global__ void fill_steps(float4* p, int K, int E, int blk4, int G, float v, int env_lo = 0, int env_n = -1) {
    const int lane = threadIdx.x & 63;
    const int nb = gridDim.x * (blockDim.x >> 6);
    int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (SWZ) { const int full = nb & ~7; if (b < full) b = (b & 7) * (full >> 3) + (b >> 3); }
    if (env_n < 0) env_n = E;
    if ((size_t)(b + 1) * G > (size_t)env_n) return;
    const float4 x = make_float4(v, v, v, v);
    const int run4 = blk4 * G;
    for (int k = 0; k < K; ++k) {
        float4* g = p + ((size_t)k * E + env_lo + (size_t)b * G) * blk4;
        int cnt = 0;
        for (int i = lane; i < run4; i += 64) {
            if (P > 0 && cnt == P) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); cnt = 0; }
            g[i] = x;
            ++cnt;
        }
    }
}

// Variant: every store instruction covers ONE 1 KB-aligned piece of the absolute address space (first / last piece of a run partial),
// and the pacing wait falls on absolute AL-byte boundaries (AL = 4096: a burst never straddles a 4 KB line of the address space).
template <int AL>
__global__ void fill_steps_aligned(float4* p, int K, int E, int blk4, int G, float v) {
    const int lane = threadIdx.x & 63;
    const int nb = gridDim.x;
    int b = blockIdx.x;
    { const int full = nb & ~7; if (b < full) b = (b & 7) * (full >> 3) + (b >> 3); }
    if ((size_t)(b + 1) * G > (size_t)E) return;
    const float4 x = make_float4(v, v, v, v);
    const long long run4 = (long long)blk4 * G;
    for (int k = 0; k < K; ++k) {
        const long long g0 = ((long long)k * E + (long long)b * G) * blk4;       // float4 index of the run's start
        const long long first = g0 & ~63LL;                                       // 1 KB-aligned piece that contains it
        for (long long q = first; q < g0 + run4; q += 64) {
            if ((q & (AL / 16 - 1)) == 0 && q != first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long i = q + lane;
            if (i >= g0 && i < g0 + run4) p[i] = x;
        }
    }
}

template <typename F>
double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 4; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 4.0 * bytes / ms / 1e6;
}

template <int P, bool SWZ>
double run(float4* a, int K, int E, size_t blk, int G, int wpb) {
    const int writers = E / G;
    const int grid = (writers + wpb - 1) / wpb;
    return timeit([&] { fill_steps<P, SWZ><<<grid, 64 * wpb>>>(a, K, E, (int)(blk / 16), G, 1.f); }, (size_t)K * writers * G * blk);
}

int main() {
    const size_t bytes = (size_t)150 * 4096 * 676 * 16;   // 6.6 GB = the c2 observation stream of one launch
    float4* a; CK(hipMalloc(&a, bytes));
    for (int rep = 0; rep < (getenv("WP9_PARTS_ONLY") ? 0 : 2); ++rep) {
        printf("hipMemsetAsync   %6.0f GB/s\n", timeit([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); }, bytes));
        printf("c2 stream (K = 150, E = 4096, 10816 B per env-step): GB/s by envs per writer G, waves per workgroup, pacing window P (0 = none), XCD-aware mapping\n");
        for (int G : {1, 2, 3, 4, 8, 16})
            for (int wpb : {1, 2}) {
                printf("G=%2d wpb=%d writers=%4d  |  swz: P0 %5.0f  P2 %5.0f  P4 %5.0f  P8 %5.0f  P16 %5.0f  |  plain: P0 %5.0f  P4 %5.0f\n", G, wpb, 4096 / G,
                       run<0, true>(a, 150, 4096, 10816, G, wpb), run<2, true>(a, 150, 4096, 10816, G, wpb), run<4, true>(a, 150, 4096, 10816, G, wpb),
                       run<8, true>(a, 150, 4096, 10816, G, wpb), run<16, true>(a, 150, 4096, 10816, G, wpb),
                       run<0, false>(a, 150, 4096, 10816, G, wpb), run<4, false>(a, 150, 4096, 10816, G, wpb));
                fflush(stdout);
            }
    }
    // the same stream as 1 / 2 / 4 part launches over env ranges, back to back (each part: all K steps of its envs)
    for (int rep = 0; rep < 3; ++rep)
        for (int parts : {1, 2, 4}) {
            const int En = 4096 / parts, writers = En / 2;
            auto go = [&] { for (int q = 0; q < parts; ++q) fill_steps<4, true><<<writers, 64>>>(a, 150, 4096, 10816 / 16, 2, 1.f, q * En, En); };
            printf("G=2 paced P4, %d part launch(es) of %d envs: %5.0f GB/s\n", parts, En, timeit(go, (size_t)150 * 4096 * 10816));
        }
    for (int rep = 0; rep < 3; ++rep) {
        const size_t used = (size_t)150 * 4096 * 10816;
        printf("G=2: paced P4 from the run start %5.0f | 1 KB-aligned pieces, wait on absolute 2 KB %5.0f  4 KB %5.0f  8 KB %5.0f  16 KB %5.0f boundaries\n",
               timeit([&] { fill_steps<4, true><<<2048, 64>>>(a, 150, 4096, 676, 2, 1.f); }, used),
               timeit([&] { fill_steps_aligned<2048><<<2048, 64>>>(a, 150, 4096, 676, 2, 1.f); }, used),
               timeit([&] { fill_steps_aligned<4096><<<2048, 64>>>(a, 150, 4096, 676, 2, 1.f); }, used),
               timeit([&] { fill_steps_aligned<8192><<<2048, 64>>>(a, 150, 4096, 676, 2, 1.f); }, used),
               timeit([&] { fill_steps_aligned<16384><<<2048, 64>>>(a, 150, 4096, 676, 2, 1.f); }, used));
    }
    return 0;
}
