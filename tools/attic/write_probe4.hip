#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ void fill_stride(float4* p, size_t n4, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const float4 x = make_float4(v, v, v, v);
    for (; i < n4; i += st) p[i] = x;
}
template <typename F> double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 4; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return 4.0 * bytes / ms / 1e6;
}
int main() {
    const size_t bytes = (size_t)150 * 4096 * 10816; const size_t n4 = bytes / 16;
    for (int b = 0; b < 3; ++b) {
        float4* a; CK(hipMalloc(&a, bytes));
        printf("buffer %d: memset %6.0f |", b, timeit([&] { CK(hipMemsetAsync(a, 0x5a, bytes, 0)); }, bytes));
        for (int grid : {64, 128, 256, 512, 1024, 2048}) for (int block : {256, 1024})
            printf(" g%d/b%d %5.0f", grid, block, timeit([&] { fill_stride<<<grid, block>>>(a, n4, 1.f); }, bytes));
        printf("\n");
    }
    return 0;
}
