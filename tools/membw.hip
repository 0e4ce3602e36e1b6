// HBM bandwidth ceilings on the box: pure-write (fill), copy, and the env kernel's own store
// pattern (one wave streams a contiguous 10,816-byte block, 4096 blocks per "step").
// Build: hipcc --offload-arch=gfx950 -O3 tools/membw.hip -o gpurun_out/membw ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void fill4(float4* p, size_t n4, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    float4 x = make_float4(v, v, v, v);
    for (; i < n4; i += st) p[i] = x;
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += st) b[i] = a[i];
}
// one wave per 676-float4 block per step, K steps: block (k, e) at ((k*E+e)*676)
__global__ __launch_bounds__(256) void wave_blocks(float4* p, int E, int K, float v) {
    int lane = threadIdx.x & 63, e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    float4 x = make_float4(v, v, v, v);
    for (int k = 0; k < K; ++k) {
        float4* g = p + ((size_t)k * E + e) * 676;
        for (int i = lane; i < 676; i += 64) g[i] = x;
    }
}
int main() {
    size_t bytes = (size_t)150 * 4096 * 676 * 16;  // 6.6 GB = one c2 rollout of obs
    float4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    size_t n4 = bytes / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int grid : {2048, 4096, 16384}) {
        fill4<<<grid, 256>>>(a, n4, 1.f); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) fill4<<<grid, 256>>>(a, n4, 2.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("fill  grid=%5d  %.0f GB/s\n", grid, 5.0 * bytes / ms / 1e6);
        copy4<<<grid, 256>>>(a, b, n4); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) copy4<<<grid, 256>>>(a, b, n4); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("copy  grid=%5d  %.0f GB/s (read+write)\n", grid, 2 * 5.0 * bytes / ms / 1e6);
    }
    wave_blocks<<<1024, 256>>>(a, 4096, 150, 1.f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) wave_blocks<<<1024, 256>>>(a, 4096, 150, 3.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("wave_blocks (env store pattern, 4096 waves) %.0f GB/s  %.2f us/step\n", 5.0 * bytes / ms / 1e6, ms / 5 / 150 * 1e3);
    wave_blocks<<<2048, 256>>>(a, 8192, 75, 1.f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) wave_blocks<<<2048, 256>>>(a, 8192, 75, 3.f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); printf("wave_blocks (8192 waves) %.0f GB/s\n", 5.0 * bytes / ms / 1e6);
    return 0;
}
