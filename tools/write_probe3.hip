// Does the DATA written change the write bandwidth?  The c2 K-step store pattern (2 envs = 21.6 KB per writer and step) with
// (a) one constant, (b) zeros, (c) a different pseudo-random word per store, into the same buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe3.hip -o /tmp/wp3 && /tmp/wp3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int MODE>
__global__ void fill_steps(float4* p, int K, int E, size_t blk4, int G, float v) {
    const int lane = threadIdx.x & 63;
    const size_t t = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    if ((t + 1) * G > (size_t)E) return;
    const size_t run4 = blk4 * G;
    unsigned s = 0x9E3779B9u * (unsigned)(t * 64 + lane + 1);
    for (int k = 0; k < K; ++k) {
        float4* g = p + ((size_t)k * E + t * G) * blk4;
        for (size_t i = lane; i < run4; i += 64) {
            float4 x;
            if (MODE == 0) x = make_float4(v, v, v, v);
            else if (MODE == 1) x = make_float4(0.f, 0.f, 0.f, 0.f);
            else {
                s = s * 1664525u + 1013904223u; const unsigned a = s;
                s = s * 1664525u + 1013904223u; const unsigned b = s;
                x = make_float4(__uint_as_float((a >> 9) | 0x3f800000u), __uint_as_float((b >> 9) | 0x3f800000u),
                                __uint_as_float((a << 7 >> 9) | 0x3f800000u), __uint_as_float((b << 5 >> 9) | 0x3f800000u));
            }
            g[i] = x;
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
    const int K = 150, E = 4096, G = 2; const size_t blk = 10816, bytes = (size_t)K * E * blk;
    for (int b = 0; b < 4; ++b) {
        float4* a; CK(hipMalloc(&a, bytes));
        const int grid = (E / G * 64 + 255) / 256;
        printf("buffer %d: constant %6.0f  zeros %6.0f  random %6.0f  constant %6.0f GB/s | hipMemsetAsync(0) %6.0f  (0x5a) %6.0f\n", b,
               timeit([&] { fill_steps<0><<<grid, 256>>>(a, K, E, blk / 16, G, 1.f); }, bytes),
               timeit([&] { fill_steps<1><<<grid, 256>>>(a, K, E, blk / 16, G, 1.f); }, bytes),
               timeit([&] { fill_steps<2><<<grid, 256>>>(a, K, E, blk / 16, G, 1.f); }, bytes),
               timeit([&] { fill_steps<0><<<grid, 256>>>(a, K, E, blk / 16, G, 1.f); }, bytes),
               timeit([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); }, bytes), timeit([&] { CK(hipMemsetAsync(a, 0x5a, bytes, 0)); }, bytes));
    }
    return 0;
}
