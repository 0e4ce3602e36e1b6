// Is the compute/store overlap penalty a per-wave serialisation effect or a chip-level one?
// Mode A: every wave does n FMAs then its 10.8 KB of stores (like overlap_probe).
// Mode B: specialised waves: 4096 store-only waves + 4096 compute-only waves (same total work) in one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(512) void probe(float4* p, double* sink, int E, int K, int n, double seed, int mode) {
    int lane = threadIdx.x & 63;
    int blk = blockIdx.x;
    bool do_store = true, do_comp = true;
    int w = threadIdx.x >> 6;
    if (mode == 1) { do_store = w < 4; do_comp = !do_store; w &= 3; }   // wave-level specialisation inside each block
    int e = blk * 4 + w;
    if (e >= E) return;
    double a0 = seed + lane, a1 = seed * 2 + lane, a2 = seed * 3 + lane, a3 = seed * 4 + lane;
    for (int k = 0; k < K; ++k) {
        if (do_comp) {
            // 4 independent chains (ILP) of n/4 f32-rate-free f64 FMAs, unrolled so that the scalar unit is not the limiter
#pragma unroll 8
            for (int i = 0; i < n / 4; ++i) {
                a0 = __builtin_fma(a0, 1.0000001, 1e-9); a1 = __builtin_fma(a1, 1.0000001, 1e-9);
                a2 = __builtin_fma(a2, 1.0000001, 1e-9); a3 = __builtin_fma(a3, 1.0000001, 1e-9);
            }
        }
        if (do_store) {
            float f = (float)(a0 + a1 + a2 + a3);
            float4 x = make_float4(f, f, f, f);
            float4* g = p + ((size_t)k * E + e) * 676;
            for (int i = lane; i < 676; i += 64) g[i] = x;
        }
    }
    if (do_comp && !do_store) sink[(size_t)e * 64 + lane] = a0 + a1 + a2 + a3;
}
int main() {
    const int E = 4096, K = 150;
    size_t bytes = (size_t)K * E * 676 * 16;
    float4* a; CK(hipMalloc(&a, bytes));
    double* sink; CK(hipMalloc(&sink, (size_t)E * 64 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int mode : {0, 1}) for (int n : {0, 400, 800, 1600, 3200}) {
        int grid = 1024, blk = mode ? 512 : 256;
        probe<<<grid, blk>>>(a, sink, E, K, n, 1.0, mode); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) probe<<<grid, blk>>>(a, sink, E, K, n, 2.0, mode); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("mode=%s n=%4d  %.2f us/step  %.0f GB/s\n", mode ? "specialised" : "fused      ", n, ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
    // compute-only reference
    return 0;
}
