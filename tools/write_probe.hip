// Pure-write bandwidth of MI355X under different store patterns (what bounds the env kernel: its obs stream).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o /tmp/write_probe && /tmp/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// (a) grid-stride float4
__global__ void fill_stride(float4* p, size_t n4, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const float4 x = make_float4(v, v, v, v);
    for (; i < n4; i += st) p[i] = x;
}
// (b) every WAVE owns a contiguous chunk of `chunk4` float4 and streams it (1 KB per instruction), chunks assigned round-robin
template <bool NT>
__global__ void fill_wave_chunks(float4* p, size_t n4, size_t chunk4, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float4 x = make_float4(v, v, v, v);
    for (size_t c = wave; c * chunk4 < n4; c += nw) {
        float4* g = p + c * chunk4;
        const size_t lim = (c + 1) * chunk4 <= n4 ? chunk4 : n4 - c * chunk4;
        for (size_t i = lane; i < lim; i += 64) {
            if (NT) { typedef float v4f __attribute__((ext_vector_type(4))); v4f y = {v, v, v, v}; __builtin_nontemporal_store(y, reinterpret_cast<v4f*>(g + i)); } else g[i] = x;
        }
    }
}
// (c) every BLOCK owns a contiguous chunk
__global__ void fill_block_chunks(float4* p, size_t n4, size_t chunk4, float v) {
    const float4 x = make_float4(v, v, v, v);
    for (size_t c = blockIdx.x; c * chunk4 < n4; c += gridDim.x) {
        float4* g = p + c * chunk4;
        const size_t lim = (c + 1) * chunk4 <= n4 ? chunk4 : n4 - c * chunk4;
        for (size_t i = threadIdx.x; i < lim; i += blockDim.x) g[i] = x;
    }
}
// (d) 8 bytes per lane
__global__ void fill_stride2(float2* p, size_t n2, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const float2 x = make_float2(v, v);
    for (; i < n2; i += st) p[i] = x;
}

// (g) wave chunks whose 1 KB store instructions are aligned to 1 KB in GLOBAL addresses although the chunk base is only
// 128-byte aligned: a partial head store, aligned full stores, a partial tail.  off4: float4 offset added to every chunk base.
template <bool ALIGN>
__global__ void fill_wave_chunks_al(float4* p, size_t n4, size_t chunk4, size_t off4, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float4 x = make_float4(v, v, v, v);
    for (size_t c = wave; (c + 1) * chunk4 + off4 <= n4; c += nw) {
        const size_t base = c * chunk4 + off4;
        const long long m = ALIGN ? (long long)(base & 63) : 0;
        for (long long b = -m; b < (long long)chunk4; b += 64) {
            const long long i = b + lane;
            if (i >= 0 && i < (long long)chunk4) p[base + i] = x;
        }
    }
}

// (e) grid-stride, 4 stores in flight per thread
__global__ void fill_stride_u4(float4* p, size_t n4, float v) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    const float4 x = make_float4(v, v, v, v);
    for (; i + 3 * st < n4; i += 4 * st) { p[i] = x; p[i + st] = x; p[i + 2 * st] = x; p[i + 3 * st] = x; }
    for (; i < n4; i += st) p[i] = x;
}
// (f) 64 contiguous bytes per lane (a wave covers 4 KB with 4 instructions of 16 B at lane stride 64 B)
__global__ void fill_lane64(float4* p, size_t n4, float v) {
    size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4, st = (size_t)gridDim.x * blockDim.x * 4;
    const float4 x = make_float4(v, v, v, v);
    for (; i + 3 < n4; i += st) { p[i] = x; p[i + 1] = x; p[i + 2] = x; p[i + 3] = x; }
}

template <typename F>
double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 5.0 * bytes / ms / 1e6;
}

int main() {
    const size_t bytes = (size_t)150 * 4096 * 676 * 16;   // 6.6 GB
    float4* a; CK(hipMalloc(&a, bytes));
    const size_t n4 = bytes / 16;
    for (int block : {256, 512, 1024})
        for (int grid : {1024, 2048, 4096, 8192, 16384, 65536})
            printf("stride float4  block=%4d grid=%6d  %6.0f GB/s\n", block, grid, timeit([&] { fill_stride<<<grid, block>>>(a, n4, 1.f); }, bytes));
    for (int grid : {2048, 16384})
        printf("stride float2  grid=%6d  %6.0f GB/s\n", grid, timeit([&] { fill_stride2<<<grid, 256>>>((float2*)a, n4 * 2, 1.f); }, bytes));
    for (size_t chunk : {(size_t)64, (size_t)676, (size_t)1352, (size_t)4096, (size_t)65536})
        for (int grid : {1024, 2048, 4096})
            printf("wave chunks    chunk=%7zu B grid=%5d  %6.0f GB/s   nontemporal %6.0f GB/s\n", chunk * 16, grid,
                   timeit([&] { fill_wave_chunks<false><<<grid, 256>>>(a, n4, chunk, 1.f); }, bytes),
                   timeit([&] { fill_wave_chunks<true><<<grid, 256>>>(a, n4, chunk, 1.f); }, bytes));
    for (size_t chunk : {(size_t)4096, (size_t)65536, (size_t)1 << 20})
        for (int grid : {256, 512, 1024, 2048})
            printf("block chunks   chunk=%8zu B grid=%5d  %6.0f GB/s\n", chunk * 16, grid, timeit([&] { fill_block_chunks<<<grid, 256>>>(a, n4, chunk, 1.f); }, bytes));
    for (int grid : {1024, 4096, 16384, 65536}) {
        printf("stride u4      grid=%6d  %6.0f GB/s\n", grid, timeit([&] { fill_stride_u4<<<grid, 256>>>(a, n4, 1.f); }, bytes));
        printf("lane 64 B      grid=%6d  %6.0f GB/s\n", grid, timeit([&] { fill_lane64<<<grid, 256>>>(a, n4, 1.f); }, bytes));
    }
    for (size_t off : {(size_t)0, (size_t)8, (size_t)24})
        printf("64 KB chunks at +%3zu B   %6.0f GB/s   stores re-aligned to 1 KB: %6.0f GB/s\n", off * 16,
               timeit([&] { fill_wave_chunks_al<false><<<1024, 256>>>(a, n4, 4096, off, 1.f); }, bytes),
               timeit([&] { fill_wave_chunks_al<true><<<1024, 256>>>(a, n4, 4096, off, 1.f); }, bytes));
    for (size_t chunk : {(size_t)676, (size_t)1352, (size_t)2704})
        for (int grid : {512, 1024})
            printf("env chunks %6zu B waves=%5d  as is %6.0f GB/s   stores re-aligned to 1 KB: %6.0f GB/s\n", chunk * 16, grid * 4,
                   timeit([&] { fill_wave_chunks_al<false><<<grid, 256>>>(a, n4, chunk, 0, 1.f); }, bytes),
                   timeit([&] { fill_wave_chunks_al<true><<<grid, 256>>>(a, n4, chunk, 0, 1.f); }, bytes));
    // few writer waves with long contiguous runs (what a restructured env kernel could do: one observation wave per 4-6 envs)
    for (size_t chunk : {(size_t)1352, (size_t)2704, (size_t)4056, (size_t)5408, (size_t)8112})
        for (int grid : {128, 171, 256, 342, 512, 1024})
            printf("few waves      chunk=%7zu B waves=%5d  %6.0f GB/s   nontemporal %6.0f GB/s\n", chunk * 16, grid * 4,
                   timeit([&] { fill_wave_chunks<false><<<grid, 256>>>(a, n4, chunk, 1.f); }, bytes),
                   timeit([&] { fill_wave_chunks<true><<<grid, 256>>>(a, n4, chunk, 1.f); }, bytes));
    printf("hipMemsetAsync %6.0f GB/s\n", timeit([&] { CK(hipMemsetAsync(a, 0, bytes, 0)); }, bytes));
    return 0;
}
