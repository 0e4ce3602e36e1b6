// What property of hipMemset's store stream makes it immune to the buffer placement?  Two sweeps, every store line-aligned:
//  (1) lock-step grid-stride with W waves x P-byte pieces (P = 1 KB is the memset geometry; window = W x P);
//  (2) the env kernels' own pattern (a writer keeps 2 adjacent env blocks = 21,632 B per step) and the chunk sweep of write_probe6,
//      with every store instruction ALIGNED to A bytes of the buffer (A = 0: instructions start at the chunk start, i.e. at any
//      128-byte multiple -- what the kernels do today; A = 1024: a masked head instruction up to the next 1 KB boundary, then
//      aligned 1 KB instructions, then a masked tail).
// Build: hipcc --offload-arch=gfx950 -O3 -o write_probe8 write_probe8.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// (1) W waves, pieces of P4 float4 (P4 multiple of 64): wave w, iteration t -> piece t*W + w
__global__ void fill_pieces(float4* p, size_t n4, int P4) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t W = ((size_t)gridDim.x * blockDim.x) >> 6;
    const float4 x = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t base = wave * P4; base < n4; base += W * P4)
        for (int i = lane; i < P4 && base + i < n4; i += 64) p[base + i] = x;
}

// (2) chunk writers: mode 0 = own chunk per step, mode 1 = chunk sweep; A4 = alignment of the store instructions in float4 (0 = none)
__global__ void fill_chunks(float4* p, int K, int chunks_per_step, int run4, int mode, int A4) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
    const long W = ((long)gridDim.x * blockDim.x) >> 6;
    const float4 x = make_float4(1.f, 2.f, 3.f, 4.f);
    const long total = (long)K * chunks_per_step;
    long c = mode == 0 ? wave : wave;
    const long stride = mode == 0 ? chunks_per_step : W;
    if (mode == 0 && wave >= chunks_per_step) return;
    for (; c < total; c += stride) {
        const size_t b = (size_t)c * run4;             // chunk start in float4 units from the buffer base
        int i0 = 0;
        if (A4 > 0) {
            const int head = (int)((A4 - (b % A4)) % A4);   // float4s up to the next aligned boundary
            if (lane < head && lane < run4) p[b + lane] = x;
            i0 = head < run4 ? head : run4;
        }
        for (int i = i0 + lane; i < run4; i += 64) p[b + i] = x;
    }
}

template <typename F> double timeit(F f, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 3.0 * bytes / ms / 1e6;
}

int main(int argc, char** argv) {
    const int nbuf = argc > 1 ? atoi(argv[1]) : 4;
    const int K = 150, E = 4096; const size_t blk = 10816, bytes = (size_t)K * E * blk, n4 = bytes / 16;
    for (int bi = 0; bi < nbuf; ++bi) {
        float4* buf; CK(hipMalloc(&buf, bytes));
        printf("buffer %d (base %% 2MB = %zu): memset %5.0f\n", bi, (size_t)((uintptr_t)buf & ((2u << 20) - 1)),
               timeit([&] { CK(hipMemsetAsync(buf, 0x5a, bytes, 0)); }, bytes));
        printf("  lock-step pieces, GB/s   P=  1KB   2KB   4KB   8KB  16KB  32KB  64KB\n");
        for (int W : {256, 512, 1024, 2048, 4096}) {
            printf("    W=%4d waves (b256)     ", W);
            for (int P : {1, 2, 4, 8, 16, 32, 64})
                printf(" %5.0f", timeit([&] { fill_pieces<<<W / 4, 256>>>(buf, n4, P * 64); }, bytes));
            printf("\n");
        }
        printf("  2-env chunks (21,632 B), GB/s   A= none  128B  256B  512B   1KB\n");
        for (int mode : {0, 1})
            for (int W : {1024, 2048}) {
                if (mode == 0 && W != 2048) continue;
                printf("    %s W=%4d          ", mode ? "chunk sweep " : "own 2 envs  ", W);
                for (int A : {0, 128, 256, 512, 1024})
                    printf(" %5.0f", timeit([&] { fill_chunks<<<W / 4, 256>>>(buf, K, E / 2, (int)(2 * blk / 16), mode, A / 16); }, bytes));
                printf("\n");
            }
        fflush(stdout);
    }
    return 0;
}
