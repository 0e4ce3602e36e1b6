// Does per-wave compute overlap with other waves' streaming stores?  4096 waves, each per step:
// a dependent chain of `n` f64 FMAs (compute phase) then 11 x 1 KiB stores (the env obs block).
// If time(n) stays flat while n grows, compute hides under the store stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void probe(float4* p, int E, int K, int n, double seed, int stagger) {
    int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    double a = seed + lane;
    if (stagger) { int d = (e * 2654435761u >> 20) % stagger; for (int i = 0; i < d; ++i) a = __builtin_fma(a, 1.0000001, 1e-9); }
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < n; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
        float f = (float)a;
        float4 x = make_float4(f, f, f, f);
        float4* g = p + ((size_t)k * E + e) * 676;
        for (int i = lane; i < 676; i += 64) g[i] = x;
    }
}
int main() {
    const int E = 4096, K = 150;
    size_t bytes = (size_t)K * E * 676 * 16;
    float4* a; CK(hipMalloc(&a, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int stagger : {0, 4000}) for (int n : {0, 100, 200, 400, 800, 1600, 3200}) {
        probe<<<1024, 256>>>(a, E, K, n, 1.0, stagger); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) probe<<<1024, 256>>>(a, E, K, n, 2.0, stagger); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stagger=%4d n=%4d  %.2f us/step  %.0f GB/s\n", stagger, n, ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
    return 0;
}
