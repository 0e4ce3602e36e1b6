// Role-specialised workgroup probe: 4 waves = 2 "physics" waves (np dependent-ish FMAs per step, no big stores)
// + 2 "observation" waves (no FMAs of their own beyond no, 2 env blocks of 10.8 KB per step), one
// __syncthreads per step (pipeline depth 1).  Compared with the fused form (every wave: n FMAs + 1 block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__device__ __forceinline__ double work(double a, int n) {
    double a0 = a, a1 = a + 1, a2 = a + 2, a3 = a + 3;
#pragma unroll 8
    for (int i = 0; i < n / 4; ++i) {
        a0 = __builtin_fma(a0, 1.0000001, 1e-9); a1 = __builtin_fma(a1, 1.0000001, 1e-9);
        a2 = __builtin_fma(a2, 1.0000001, 1e-9); a3 = __builtin_fma(a3, 1.0000001, 1e-9);
    }
    return a0 + a1 + a2 + a3;
}
__global__ __launch_bounds__(256) void fused(float4* p, int E, int K, int n, double seed) {
    int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    double a = seed + lane;
    for (int k = 0; k < K; ++k) {
        a = work(a, n);
        float f = (float)a; float4 x = make_float4(f, f, f, f);
        float4* g = p + ((size_t)k * E + e) * 676;
        for (int i = lane; i < 676; i += 64) g[i] = x;
    }
}
__global__ __launch_bounds__(256) void roles(float4* p, int E, int K, int np, int no, double seed) {
    __shared__ double hand[2][4][64];
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool phys = w < 2;
    const int pair = w & 1;               // this wave serves envs 2*pair, 2*pair+1 of the block
    double a = seed + lane;
    for (int k = 0; k < K; ++k) {
        if (phys) {
            for (int s = 0; s < 2; ++s) { a = work(a, np); hand[k & 1][2 * pair + s][lane] = a; }
            __syncthreads();
        } else {
            __syncthreads();
            for (int s = 0; s < 2; ++s) {
                double b = hand[k & 1][2 * pair + s][lane];
                b = work(b, no);
                float f = (float)b; float4 x = make_float4(f, f, f, f);
                int e = blockIdx.x * 4 + 2 * pair + s;
                float4* g = p + ((size_t)k * E + e) * 676;
                for (int i = lane; i < 676; i += 64) g[i] = x;
            }
        }
    }
}
int main() {
    const int E = 4096, K = 150;
    size_t bytes = (size_t)K * E * 676 * 16;
    float4* a; CK(hipMalloc(&a, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int n : {0, 600, 1000}) {
        fused<<<1024, 256>>>(a, E, K, n, 1.0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) fused<<<1024, 256>>>(a, E, K, n, 2.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fused n=%4d             %.2f us/step  %.0f GB/s\n", n, ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
    int cfgs[5][2] = {{0, 0}, {400, 200}, {700, 300}, {1000, 0}, {1400, 400}};
    for (auto& c : cfgs) {
        roles<<<1024, 256>>>(a, E, K, c[0], c[1], 1.0); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) roles<<<1024, 256>>>(a, E, K, c[0], c[1], 2.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("roles np=%4d no=%4d (per env; total/env=%4d)  %.2f us/step  %.0f GB/s\n", c[0], c[1], c[0] + c[1], ms / 3 / K * 1e3, 3.0 * bytes / ms / 1e6);
    }
    return 0;
}
