// Wave-count scaling of the compute/store overlap: E waves (one 10.8 KB block per wave per step),
// n dependent FMAs per step.  Total bytes fixed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void probe(float4* p, int E, int K, int n, double seed, int per) {
    int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    double a = seed + lane;
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < n; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
        float f = (float)a;
        float4 x = make_float4(f, f, f, f);
        float4* g = p + ((size_t)k * E + e) * per;
        for (int i = lane; i < per; i += 64) g[i] = x;
    }
}
int main() {
    size_t bytes = (size_t)150 * 4096 * 676 * 16;
    float4* a; CK(hipMalloc(&a, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // (E, K, per): same total bytes; "per" = float4 per wave per step
    int cfg[4][3] = {{2048, 150, 1352}, {4096, 150, 676}, {8192, 150, 338}, {16384, 150, 169}};
    for (auto& c : cfg) for (int n : {0, 100, 200, 400}) {
        int E = c[0], K = c[1], per = c[2];
        int nn = n * 4096 / E;   // same total compute per step: n scaled so that E*nn is constant
        probe<<<E / 4, 256>>>(a, E, K, nn, 1.0, per); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) probe<<<E / 4, 256>>>(a, E, K, nn, 2.0, per); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("waves=%5d per_wave=%5d B n=%4d(x%d)  %.2f us/step  %.0f GB/s\n", E, per * 16, nn, E / 4096 ? E / 4096 : 1, ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
    return 0;
}
