// Pure-write bandwidth of MI355X for the store patterns an env kernel can choose between (round 3): which contiguous run
// length per writer, and how many waves co-operate on one run, reach the 6.0-6.4 TB/s that 64 KB runs / memset reach.
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe2.hip -o /tmp/write_probe2 && /tmp/write_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// A TEAM of `team` consecutive waves owns a contiguous chunk of chunk4 float4 and streams it in 1 KB pieces: wave q of the
// team stores pieces q, q + team, ...  Chunks are dealt round-robin to the teams.  team = 1: one wave per chunk.
template <bool NT>
__global__ void fill_team(float4* p, size_t n4, size_t chunk4, int team, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t t = wave / team, nt = nw / team;
    const int q = (int)(wave % team);
    const float4 x = make_float4(v, v, v, v);
    for (size_t c = t; (c + 1) * chunk4 <= n4; c += nt) {
        float4* g = p + c * chunk4;
        for (size_t i = (size_t)q * 64 + lane; i < chunk4; i += (size_t)team * 64) {
            if (NT) { v4f y = {v, v, v, v}; __builtin_nontemporal_store(y, reinterpret_cast<v4f*>(g + i)); } else g[i] = x;
        }
    }
}

// K-step structure of the env kernels: region = [K][E] blocks of blk4 float4; writer w (a team) owns envs [w*G, (w+1)*G) and
// at step k streams its G adjacent blocks (G*blk4 float4 contiguous); all writers go k = 0..K-1.
template <bool NT>
__global__ void fill_steps(float4* p, int K, int E, size_t blk4, int G, int team, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t t = wave / team;
    const int q = (int)(wave % team);
    if ((t + 1) * G > (size_t)E) return;
    const float4 x = make_float4(v, v, v, v);
    const size_t run4 = blk4 * G;
    for (int k = 0; k < K; ++k) {
        float4* g = p + ((size_t)k * E + t * G) * blk4;
        for (size_t i = (size_t)q * 64 + lane; i < run4; i += (size_t)team * 64) {
            if (NT) { v4f y = {v, v, v, v}; __builtin_nontemporal_store(y, reinterpret_cast<v4f*>(g + i)); } else g[i] = x;
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

int main() {
    const size_t bytes = (size_t)150 * 4096 * 676 * 16;   // 6.6 GB = the c2 observation stream of one launch
    float4* a; CK(hipMalloc(&a, bytes));
    const size_t n4 = bytes / 16;
    printf("hipMemsetAsync   %6.0f GB/s\n", timeit([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); }, bytes));
    printf("== K-step env structure: K=150 steps; block bytes per env-step, envs, envs per writer (G), waves per writer (team) -> GB/s\n");
    struct Cfg { int K, E; size_t blk; } cfgs[] = {{150, 4096, 10816}, {75, 1024, 84096}, {4, 2048, 663808}};   // each <= 6.6 GB
    const bool quick = getenv("WP2_QUICK") != nullptr;     // box classification only: the kernels' own patterns + memset + 64 KB chunks
    for (auto c : cfgs)
        for (int G : {1, 2, 3, 4, 6, 8, 12, 16})
            for (int team : {1, 2, 4}) {
                if (quick && !((c.blk == 10816 && G == 2 && team == 1) || (c.blk != 10816 && G == 1 && team != 2))) continue;
                if (c.blk > 100000 && G > 2) continue;
                if (c.blk > 50000 && G > 4) continue;
                const int writers = c.E / G, waves = writers * team;
                if (waves < 256) continue;
                const int grid = (waves * 64 + 255) / 256;
                const size_t used = (size_t)c.K * writers * G * c.blk;
                printf("blk=%7zu E=%4d G=%2d team=%d waves=%5d run=%7zu B  %6.0f\n", c.blk, c.E, G, team, waves, G * c.blk,
                       timeit([&] { fill_steps<false><<<grid, 256>>>(a, c.K, c.E, c.blk / 16, G, team, 1.f); }, used));
                fflush(stdout);
            }
    printf("== round-robin chunks: chunk bytes, waves per chunk (team), total waves -> GB/s (plain | nontemporal)\n");
    const size_t chunks[] = {10816, 21632, 43264, 64896, 4096, 8192, 16384, 32768, 65536, 131072, 1048576, 84096, 663808};
    for (size_t cb : chunks)
        for (int team : {1, 4})
            for (int waves : {2048, 4096}) {
                if (quick && !(cb == 65536 && team == 1)) continue;
                const int grid = waves / 4;
                const size_t used = (n4 / (cb / 16)) * cb;
                printf("chunk=%7zu team=%d waves=%4d  %6.0f | %6.0f\n", cb, team, waves,
                       timeit([&] { fill_team<false><<<grid, 256>>>(a, n4, cb / 16, team, 1.f); }, used),
                       timeit([&] { fill_team<true><<<grid, 256>>>(a, n4, cb / 16, team, 1.f); }, used));
                fflush(stdout);
            }
    return 0;
}
