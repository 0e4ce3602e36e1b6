// Store-flavour probe: same pattern as overlap_probe (4096 waves, n dependent FMAs + 10.8 KB block
// per step) with plain / non-temporal stores and 16 B / 8 B per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* p, int E, int K, int n, double seed) {
    int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    double a = seed + lane;
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < n; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
        float f = (float)a;
        float* g = p + ((size_t)k * E + e) * 2704;
        if (MODE == 0) { float4 x = make_float4(f, f, f, f); for (int i = lane; i < 676; i += 64) reinterpret_cast<float4*>(g)[i] = x; }
        if (MODE == 1) { typedef float v4 __attribute__((ext_vector_type(4))); v4 x = {f, f, f, f}; for (int i = lane; i < 676; i += 64) __builtin_nontemporal_store(x, reinterpret_cast<v4*>(g) + i); }
        if (MODE == 2) { float2 x = make_float2(f, f); for (int i = lane; i < 1352; i += 64) reinterpret_cast<float2*>(g)[i] = x; }
        if (MODE == 3) { for (int i = lane; i < 2704; i += 64) g[i] = f; }
    }
}
template <int MODE> void run(float* a, const char* name, int blk) {
    const int E = 4096, K = 150; size_t bytes = (size_t)K * E * 2704 * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); float ms;
    for (int n : {0, 200, 400}) {
        probe<MODE><<<E / (blk / 64), blk>>>(a, E, K, n, 1.0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) probe<MODE><<<E / (blk / 64), blk>>>(a, E, K, n, 2.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-22s blk=%3d n=%4d  %.2f us/step  %.0f GB/s\n", name, blk, n, ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
}
int main() {
    float* a; CK(hipMalloc(&a, (size_t)150 * 4096 * 2704 * 4));
    run<0>(a, "plain dwordx4", 256); run<1>(a, "nontemporal dwordx4", 256); run<2>(a, "plain dwordx2", 256); run<3>(a, "plain dword", 256);
    run<0>(a, "plain dwordx4", 64); run<1>(a, "nontemporal dwordx4", 64);
    return 0;
}
