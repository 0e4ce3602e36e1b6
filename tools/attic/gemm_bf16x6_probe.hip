// Split-bf16 ("bf16 x 6") fp32-equivalent GEMM core for the policy trunks:  Y[M,256] = X[M,256] . W[256,256]^T, fp32 in / fp32 out.
//
// gfx950's f32-input MFMA runs at the f32 VECTOR rate (157 TF/s, 1/16 of the bf16 MFMA rate), which caps the PPO update's three
// 4.9 M x 256 x 256 products at 4.4-5.0 ms each however good the library kernel is.  Here every fp32 operand value is written as
// the exact sum of three bf16 values  x = h + m + l  (round-to-nearest: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|; 3 x 9 significand bits
// cover fp32's 24), and the product is accumulated in fp32 from the six bf16 MFMA products that matter
//     x.w = hh + hm + mh + hl + lh + mm          (dropped: ml + lm + ll <= 2^-23 |x.w|, random-signed)
// each of which is exact in fp32 before accumulation.  W is split once per call into an MFMA-fragment-ordered image (393 KB,
// L2-resident); X is split IN REGISTERS by the wave that loaded it (no extra HBM traffic: X stays 4 B per element).
//
// Workgroup = 4 waves, tile = 128 rows x 256 columns; wave (wm, wn) owns 64 rows x 128 columns = 2 x 4 MFMA tiles
// (v_mfma_f32_32x32x16_bf16, W as the A operand so that a lane ends up with 4 consecutive output columns of one row: float4 stores).
// K is walked in chunks of 16; a chunk's W fragments (24 KB) and, every second chunk, a 32-column stage of the 128 X rows (16 KB, full
// 128-byte lines, bank-swizzled on the source side) are copied global -> LDS by LDS-DMA into double buffers.
//   hipcc --offload-arch=gfx950 -O3 tools/gemm_bf16x6_probe.hip -o /tmp/gemm_x6 && /tmp/gemm_x6 [rows]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

constexpr int H = 256;      // K and N
constexpr int BM = 128;     // rows per workgroup

__device__ __forceinline__ unsigned pack_rne(float a, float b) {   // two fp32 -> packed bf16 pair, round to nearest even
    v2f v = {a, b};
    v2bf r = __builtin_convertvector(v, v2bf);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float lo_f(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi_f(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// x[8] (fp32) -> three bf16 fragments h, m, l with h + m + l == x exactly (barring underflow)
__device__ __forceinline__ void split3(const float (&x)[8], uint4& h, uint4& m, uint4& l) {
    unsigned hh[4], mm[4], ll[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        hh[i] = pack_rne(a, b);
        const float ra = a - lo_f(hh[i]), rb = b - hi_f(hh[i]);
        mm[i] = pack_rne(ra, rb);
        const float sa = ra - lo_f(mm[i]), sb = rb - hi_f(mm[i]);
        ll[i] = pack_rne(sa, sb);
    }
    h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    m = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// W [256 n, 256 k] fp32 -> fragment image: chunk c (16 k values) = [piece p][n-tile nt][lane], uint4 index ((c*3 + p)*8 + nt)*64 + lane,
// holding W[n = 32 nt + (lane & 31)][k = 32 (c >> 1) + 16 (lane >> 5) + 8 (c & 1) + e], e = 0..7.  (Which 16 k values a chunk contracts is
// free as long as both operands agree: with this choice a lane's X values of two consecutive chunks are 64 contiguous bytes of its row.)
__global__ void pack_w(const float* __restrict__ W, uint4* __restrict__ img) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (c, nt, lane)
    if (idx >= 16 * 8 * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) & 7, c = idx >> 9;
    const int n = 32 * nt + (lane & 31), k0 = 32 * (c >> 1) + 16 * (lane >> 5) + 8 * (c & 1);
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = W[n * H + k0 + e];
    uint4 h, m, l;
    split3(x, h, m, l);
    img[((c * 3 + 0) * 8 + nt) * 64 + lane] = h;
    img[((c * 3 + 1) * 8 + nt) * 64 + lane] = m;
    img[((c * 3 + 2) * 8 + nt) * 64 + lane] = l;
}

__device__ __forceinline__ v16f mfma(const uint4& a, const uint4& b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

#ifndef OCC
#define OCC 2
#endif
#ifndef VAR
#define VAR 0               // timing experiments (results wrong): 1 = no Y stores, 2 = X staged for stage 0 only, 4 = W copied for chunks 0 / 1 only
#endif
constexpr int WCH = 3 * 8 * 1024;          // bytes of one W chunk
constexpr int XST = BM * 128;              // bytes of one X stage: 128 rows x 32 floats
// LDS: W chunk double buffer | X stage double buffer = 48 KB + 32 KB: two workgroups per CU fill the 160 KB exactly
__global__ __launch_bounds__(256, OCC) void gemm_x6(const float* __restrict__ X, const uint4* __restrict__ Wimg, float* __restrict__ Y,
                                                     long long M) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wbuf = lds;
    unsigned char* xbuf = lds + 2 * WCH;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int j = lane & 31, kg = lane >> 5;
    const long long wg_row0 = (long long)blockIdx.x * BM;

    auto w_copy = [&](int c) {                    // this wave's share of W chunk c: 1 KB pieces w, w + 4, ...
        const unsigned char* src = reinterpret_cast<const unsigned char*>(Wimg) + (size_t)c * WCH;
#pragma unroll
        for (int q = 0; q < 6; ++q) glds16(src + (w + 4 * q) * 1024 + lane * 16, wbuf + (c & 1) * WCH + (w + 4 * q) * 1024);
    };
    // X stage S = columns [32 S, 32 S + 32) of the workgroup's 128 rows, one 128-byte line per row.  A 1 KB piece = 8 rows; lane t
    // fetches 16-byte column (t & 7) ^ ((row >> 1) & 7) of row 8 P + (t >> 3) and lands at LDS offset 16 t of the piece (LDS-DMA
    // destinations are lane-linear), i.e. LDS column c16 of a row holds global column c16 ^ ((row >> 1) & 7): fragment reads of
    // 16 consecutive rows then spread over all 64 banks.
    auto x_copy = [&](int S) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int P = w + 4 * q, row = 8 * P + (lane >> 3);
            const long long r = wg_row0 + row;
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            glds16(reinterpret_cast<const unsigned char*>(X + (r < M ? r : 0) * H + 32 * S) + 16 * g, xbuf + (S & 1) * XST + P * 1024);
        }
    };

    v16f acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

    w_copy(0);
    x_copy(0);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (c + 1 < 16 && (!(VAR & 4) || c == 0)) w_copy(c + 1);
        if ((c & 1) == 0 && c + 2 < 16 && !(VAR & 2)) x_copy((c >> 1) + 1);
        const uint4* wl = reinterpret_cast<const uint4*>(wbuf + (c & 1) * WCH);
        const int S = (VAR & 2) ? 0 : (c >> 1);
        uint4 xh[2], xm[2], xl[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = 64 * wm + 32 * mt + j, g = 4 * kg + 2 * (c & 1), f = (row >> 1) & 7;
            const unsigned char* xrow = xbuf + (S & 1) * XST + row * 128;
            const float4 u = *reinterpret_cast<const float4*>(xrow + 16 * (g ^ f));
            const float4 v = *reinterpret_cast<const float4*>(xrow + 16 * ((g + 1) ^ f));
            const float x8[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
            split3(x8, xh[mt], xm[mt], xl[mt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = 4 * wn + nt;
            const uint4 wh = wl[(0 * 8 + t) * 64 + lane];
            const uint4 wmid = wl[(1 * 8 + t) * 64 + lane];
            const uint4 wlo = wl[(2 * 8 + t) * 64 + lane];
            v16f a0 = acc[0][nt], a1 = acc[1][nt];
            a0 = mfma(wlo, xh[0], a0);  a1 = mfma(wlo, xh[1], a1);      // small terms first, the leading product last;
            a0 = mfma(wh, xl[0], a0);   a1 = mfma(wh, xl[1], a1);       // the two row tiles alternate (independent accumulators)
            a0 = mfma(wmid, xm[0], a0); a1 = mfma(wmid, xm[1], a1);
            a0 = mfma(wmid, xh[0], a0); a1 = mfma(wmid, xh[1], a1);
            a0 = mfma(wh, xm[0], a0);   a1 = mfma(wh, xm[1], a1);
            a0 = mfma(wh, xh[0], a0);   a1 = mfma(wh, xh[1], a1);
            acc[0][nt] = a0; acc[1][nt] = a1;
        }
    }
    // D[i = n][j = m]: lane (j, kg) holds for row m = j the columns n = 8 q + 4 kg + (0..3), q = reg >> 2
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const long long r = wg_row0 + 64 * wm + 32 * mt + j;
        if (r < M && (!(VAR & 1) || acc[mt][0][0] == 123.456f)) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v16f a = acc[mt][nt];
                    *reinterpret_cast<float4*>(Y + r * H + 128 * wn + 32 * nt + 8 * q + 4 * kg) =
                        make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                }
        }
    }
}


// ---- v3: persistent producer / consumer form -------------------------------------------------------------------------------------
// One workgroup of 8 waves per CU, looping over 128-row tiles.  Waves 0-3 (one per SIMD) issue nothing but LDS fragment reads and
// MFMAs; waves 4-7 stream the operands: W chunks by LDS-DMA, X rows in full 128-byte lines (8 lanes per row), split into the three
// bf16 pieces in registers and written to LDS in MFMA-fragment order.  Chunk = 32 k values (96 MFMAs per consumer wave); the
// producers fill the buffers of chunk g+1 while the consumers work on chunk g; ONE workgroup barrier per chunk is the only hand-off.
constexpr int WCH3 = 2 * 3 * 8 * 1024;      // W chunk: [s][p][nt][lane] x 16 B = 48 KB
constexpr int XCH3 = 2 * 3 * 4 * 1024;      // X chunk: [s][p][mt][lane] x 16 B = 24 KB
__global__ void pack_w3(const float* __restrict__ W, uint4* __restrict__ img, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (c, s, nt, lane)
    if (idx >= 8 * 2 * 8 * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) & 7, s = (idx >> 9) & 1, c = idx >> 10;
    const int n = 32 * nt + (lane & 31), k0 = 32 * c + 16 * (lane >> 5) + 8 * s;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = transpose ? W[(k0 + e) * H + n] : W[n * H + k0 + e];
    uint4 h, m, l;
    split3(x, h, m, l);
    const size_t base = (size_t)(c * 2 + s) * 3 * 8 * 64;
    img[base + (0 * 8 + nt) * 64 + lane] = h;
    img[base + (1 * 8 + nt) * 64 + lane] = m;
    img[base + (2 * 8 + nt) * 64 + lane] = l;
}

__global__ __launch_bounds__(512, 1) void gemm_x6_pc(const float* __restrict__ X, const uint4* __restrict__ Wimg, float* __restrict__ Y,
                                                      long long M, int ntiles, int ldy, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wbuf = lds;                       // [2][WCH3]
    unsigned char* xbuf = lds + 2 * WCH3;            // [2][XCH3]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * 8;                      // chunks this workgroup walks
#ifndef STAGGER
#define STAGGER 3500
#endif
    if (STAGGER > 0) {
        // Persistent workgroups started together run in lock step: all 256 CUs would reach their store burst (128 KB per tile) at the
        // same moment and then leave the HBM write path idle for a whole tile.  Eight start phases spread the bursts over the tile time.
        const long long t0 = clock64(), wait = (long long)((blockIdx.x >> 3) & 7) * STAGGER;
        while (clock64() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    if (wv >= 4) {
        // ------------------------------------------------ producers ------------------------------------------------------------
        const int pw = wv - 4;
        const int r8 = lane >> 3, g8 = lane & 7;
        const int kg = g8 >> 2, sx = (g8 >> 1) & 1, half = g8 & 1;
        auto w_copy = [&](int g) {
            const unsigned char* src = reinterpret_cast<const unsigned char*>(Wimg) + (size_t)(g & 7) * WCH3;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                // every CU wants the same 48 KB at about the same time: start at a different piece per workgroup, so that the requests
                // of the 32 CUs of an XCD spread over its L2 channels instead of queueing on one
#ifndef WROT
#define WROT 1
#endif
                const int piece = WROT ? (pw + 4 * q + 4 * (int)(blockIdx.x >> 3)) % 48 : pw + 4 * q;
                glds16(src + piece * 1024 + lane * 16, wbuf + (g & 1) * WCH3 + piece * 1024);
            }
        };
        float4 x0[4], x1[4], x2[4];               // X chunks in flight: a ring of three (chunk g lives in slot g % 3)
        auto x_load = [&](int g, float4 (&xr)[4]) {
            const long long row0 = ((long long)blockIdx.x + (long long)(g >> 3) * gridDim.x) * BM + 32 * pw;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long r = row0 + 8 * q + r8;
                xr[q] = *reinterpret_cast<const float4*>(X + (r < M ? r : 0) * H + 32 * (g & 7) + 4 * g8);
            }
        };
        auto x_split_store = [&](int g, const float4 (&xr)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = xr[q];
                const unsigned h0 = pack_rne(v.x, v.y), h1 = pack_rne(v.z, v.w);
                const float r0 = v.x - lo_f(h0), r1 = v.y - hi_f(h0), r2 = v.z - lo_f(h1), r3 = v.w - hi_f(h1);
                const unsigned m0 = pack_rne(r0, r1), m1 = pack_rne(r2, r3);
                const unsigned l0 = pack_rne(r0 - lo_f(m0), r1 - hi_f(m0)), l1 = pack_rne(r2 - lo_f(m1), r3 - hi_f(m1));
                const int pos = ((8 * q + r8) ^ (kg << 3)) + 32 * kg;          // fragment lane (row & 31) + 32 kg, bank-swizzled
                unsigned char* e = xbuf + (g & 1) * XCH3 + (size_t)((sx * 3) * 4 + pw) * 1024 + pos * 16 + half * 8;
                *reinterpret_cast<uint2*>(e) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(e + 4 * 1024) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(e + 8 * 1024) = make_uint2(l0, l1);
            }
        };
        // Iteration g (the consumers work on chunk g): split chunk g+1 (loaded two iterations ago) into the other X buffer, start
        // the W copy of chunk g+1 and the X loads of chunk g+3, then wait until only those X loads are still in flight (loads
        // return in order: X(g+2), W(g+1) are older) -- an X load has two whole iterations to arrive, a W copy (L2-resident) one.
#ifndef WREG
#define WREG 1              // 1: W chunks travel global -> registers -> LDS (plain loads, counted by the compiler); 0: LDS-DMA
#endif
#if WREG
        // Every load of a producer is a plain register load, so the compiler's own counted waits are exact: a W chunk is requested one
        // whole iteration before it is written to LDS (12 x 16 B per lane), an X chunk two.
        uint4 w0[12], w1[12];
        auto w_load = [&](int g, uint4 (&wr)[12]) __attribute__((always_inline)) {
            const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(Wimg) + (size_t)(g & 7) * WCH3);
#pragma unroll
            for (int q = 0; q < 12; ++q) wr[q] = src[(pw + 4 * q) * 64 + lane];
        };
        auto w_store = [&](int g, const uint4 (&wr)[12]) __attribute__((always_inline)) {
            uint4* dst = reinterpret_cast<uint4*>(wbuf + (g & 1) * WCH3);
#pragma unroll
            for (int q = 0; q < 12; ++q) dst[(pw + 4 * q) * 64 + lane] = wr[q];
        };
        auto step = [&](int g, const float4 (&xu)[4], float4 (&xf)[4], const uint4 (&wu)[12], uint4 (&wf)[12]) __attribute__((always_inline)) {
            if (g >= G) return;
            if (g + 2 < G) w_load(g + 2, wf);
            if (g + 3 < G) x_load(g + 3, xf);
            if (g + 1 < G) { w_store(g + 1, wu); x_split_store(g + 1, xu); }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        w_load(0, w0); x_load(0, x0);
        if (G > 1) { x_load(1, x1); w_load(1, w1); }
        if (G > 2) x_load(2, x2);
        w_store(0, w0); x_split_store(0, x0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int g = 0; g < G; g += 6) {      // a step beyond the walk's end does nothing (G is a multiple of 8, not of 6)
            step(g, x1, x0, w1, w0);
            step(g + 1, x2, x1, w0, w1);
            step(g + 2, x0, x2, w1, w0);
            step(g + 3, x1, x0, w0, w1);
            step(g + 4, x2, x1, w1, w0);
            step(g + 5, x0, x2, w0, w1);
        }
#else
        auto step = [&](int g, const float4 (&use)[4], float4 (&fill)[4]) {
            if (g + 1 < G) x_split_store(g + 1, use);
            if (VAR & 32) {     // timing experiment: the tile's 128 KB of output leave from the PRODUCERS, 16 KB per iteration, 1 KB-contiguous stores
                const long long row = ((long long)blockIdx.x + (long long)(g >> 3) * gridDim.x) * BM + 16 * (g & 7) + 4 * pw;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (row + q < M) *reinterpret_cast<float4*>(Y + (row + q) * ldy + 4 * lane) = use[q];
            }
            if (g + 1 < G && (!(VAR & 4) || g < 8)) w_copy(g + 1);
            if (g + 3 < G) {
                x_load(g + 3, fill);
                if (VAR & 32) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        };
        if (G > 0) { w_copy(0); x_load(0, x0); }
        if (G > 1) x_load(1, x1);
        if (G > 2) x_load(2, x2);
        if (G > 0) x_split_store(0, x0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int g = 0; g < G; g += 3) {
            step(g, x1, x0);                       // uses chunk g+1 (slot (g+1) % 3 = 1), refills slot g % 3 = 0 with chunk g+3
            if (g + 1 >= G) break;
            step(g + 1, x2, x1);
            if (g + 2 >= G) break;
            step(g + 2, x0, x2);
        }
#endif
    } else {
        // ------------------------------------------------ consumers ------------------------------------------------------------
        const int wm = wv & 1, wn = wv >> 1;
        const int j = lane & 31, kg = lane >> 5;
        const int xpos = ((j ^ (kg << 3)) + 32 * kg) * 16;
        v16f acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence would wait for the output stores
        for (int g = 0; g < G; ++g) {
            if (dbg && blockIdx.x == 7 && wv == 0 && lane == 0 && g < 64) dbg[g] = clock64();
            const unsigned char* wl = wbuf + (g & 1) * WCH3;
            const unsigned char* xl = xbuf + (g & 1) * XCH3;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 xf[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        xf[p][mt] = *reinterpret_cast<const uint4*>(xl + ((s * 3 + p) * 4 + 2 * wm + mt) * 1024 + xpos);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int t = 4 * wn + nt;
                    const uint4 wh = *reinterpret_cast<const uint4*>(wl + (((s * 3 + 0) * 8 + t) * 64 + lane) * 16);
                    const uint4 wmid = *reinterpret_cast<const uint4*>(wl + (((s * 3 + 1) * 8 + t) * 64 + lane) * 16);
                    const uint4 wlo = *reinterpret_cast<const uint4*>(wl + (((s * 3 + 2) * 8 + t) * 64 + lane) * 16);
                    v16f a0 = acc[0][nt], a1 = acc[1][nt];
                    // X is the A operand (tile rows = batch rows), W the B operand (tile columns = output columns): lane (n, kg) ends up
                    // with 16 rows of ONE column, so a store instruction writes 32 consecutive floats of a row = one full 128-byte line per
                    // half-wave.  (With W as the A operand a lane holds 4 consecutive columns -- float4 stores, but every instruction then
                    // touches 32 lines 1 KB apart with 32 bytes each: measured 4.7 ms against 2.0 ms without the stores.)
                    a0 = mfma(xf[0][0], wlo, a0);  a1 = mfma(xf[0][1], wlo, a1);     // small terms first, the leading product last
                    a0 = mfma(xf[2][0], wh, a0);   a1 = mfma(xf[2][1], wh, a1);
                    a0 = mfma(xf[1][0], wmid, a0); a1 = mfma(xf[1][1], wmid, a1);
                    a0 = mfma(xf[0][0], wmid, a0); a1 = mfma(xf[0][1], wmid, a1);
                    a0 = mfma(xf[1][0], wh, a0);   a1 = mfma(xf[1][1], wh, a1);
                    a0 = mfma(xf[0][0], wh, a0);   a1 = mfma(xf[0][1], wh, a1);
                    acc[0][nt] = a0; acc[1][nt] = a1;
                }
            }
            if ((g & 7) == 7) {
                const long long row0 = ((long long)blockIdx.x + (long long)(g >> 3) * gridDim.x) * BM + 64 * wm;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const v16f a = acc[mt][nt];
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const long long r = row0 + 32 * mt + (v & 3) + 8 * (v >> 2) + 4 * kg;
                            if (r < M && !(VAR & 1) && (!(VAR & 8) || v == 0)) {
                                if (VAR & 16) Y[(long long)(blockIdx.x * 4 + wv) * 64 + lane] = a[v];      // timing only: same line every time
#ifdef NTSTORE
                                else __builtin_nontemporal_store(a[v], &Y[r * ldy + 128 * wn + 32 * nt + j]);
#else
                                else Y[r * ldy + 128 * wn + 32 * nt + j] = a[v];
#endif
                            }
                        }
#pragma unroll
                        for (int v = 0; v < 16; ++v) acc[mt][nt][v] = 0.f;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence would wait for the output stores
        }
    }
}


// ---- v4: units of 128 rows x 128 columns, double accumulators, three-deep operand rings ------------------------------------------------
// What v3 taught (MI355X, 4.9 M rows): its consumer loop runs at the bf16 MFMA peak (1.54 ms = 2.5 PF/s with the W copies and
// the output stores off), but (a) a chunk's operands had ONE iteration to arrive and (b) the tile's 128 KB of output left in one burst
// from the MFMA waves, which then sat on their 63-deep memory counter and on the write-after-read of their accumulators: 4.4 ms.
// Here a workgroup walks (row tile, column half) units.  A consumer wave owns 64 rows x 64 columns = 64 accumulator registers, so it can
// keep TWO sets: while unit u+1 accumulates into one, unit u's results leave from the other, 8 store instructions per iteration.
// LDS holds THREE slots of each operand (3 x 24 KB W half-chunks + 3 x 24 KB split X chunks = 144 KB): a W copy issued in iteration g
// has until the end of iteration g+1 to land, an X row (registers) three iterations to arrive.
constexpr int WCH4 = 2 * 3 * 4 * 1024;      // W chunk of one column half: [s][p][nt (4)][lane] x 16 B
constexpr int XCH4 = 2 * 3 * 4 * 1024;      // X chunk: [s][p][mt (4)][lane] x 16 B
__global__ void pack_w4(const float* __restrict__ W, uint4* __restrict__ img, int transpose) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (half, c, s, nt, lane)
    if (idx >= 2 * 8 * 2 * 4 * 64) return;
    const int lane = idx & 63, nt = (idx >> 6) & 3, s = (idx >> 8) & 1, c = (idx >> 9) & 7, half = idx >> 12;
    const int n = 128 * half + 32 * nt + (lane & 31), k0 = 32 * c + 16 * (lane >> 5) + 8 * s;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = transpose ? W[(k0 + e) * H + n] : W[n * H + k0 + e];
    uint4 h, m, l;
    split3(x, h, m, l);
    const size_t base = ((size_t)((half * 8 + c) * 2 + s) * 3) * 4 * 64;
    img[base + (0 * 4 + nt) * 64 + lane] = h;
    img[base + (1 * 4 + nt) * 64 + lane] = m;
    img[base + (2 * 4 + nt) * 64 + lane] = l;
}

#define X_LOAD_ASM(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

__global__ __launch_bounds__(512, 1) void gemm_x6_v4(const float* __restrict__ X, const uint4* __restrict__ Wimg, float* __restrict__ Y,
                                                      long long M, int ntiles, int ldy, long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* wbuf = lds;                       // [3][WCH4]
    unsigned char* xbuf = lds + 3 * WCH4;            // [3][XCH4]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int U = 2 * my_tiles;                      // units: (tile, column half), both halves of a tile back to back
    const int G = 8 * U;                             // chunk iterations
    if (G == 0) return;
    if (wv >= 4) {
        // ------------------------------------------------ producers ------------------------------------------------------------
        // Two kinds, so that each wave's memory counter holds ONE kind of operation: waves 4-5 load X rows into registers (plain loads:
        // the compiler counts them and waits exactly for the chunk it splits), waves 6-7 copy W chunks by LDS-DMA (counted by hand).
        // (In-flight loads hidden from the compiler in inline asm are not an option: it is free to copy or re-use their destination
        // registers before the data has landed -- measured: memory faults on a full grid.)
        if (wv < 6) {
            const int pw = wv - 4;                   // X rows 64 pw .. 64 pw + 63 of the tile = fragment blocks mt = 2 pw, 2 pw + 1
            const int r8 = lane >> 3, g8 = lane & 7;
            const int kg = g8 >> 2, sx = (g8 >> 1) & 1, half8 = g8 & 1;
            float4 x0[8], x1[8], x2[8];              // register ring: chunk c of the walk lives in slot c % 3
            auto x_load = [&](int g, float4 (&xr)[8]) {
                const long long row0 = ((long long)blockIdx.x + (long long)(g >> 4) * gridDim.x) * BM + 64 * pw;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const long long r = row0 + 8 * q + r8;
                    xr[q] = *reinterpret_cast<const float4*>(X + (r < M ? r : 0) * H + 32 * (g & 7) + 4 * g8);
                }
            };
            auto x_split_store = [&](int g, const float4 (&xr)[8]) {
                unsigned char* dst = xbuf + (g % 3) * XCH4;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 v = xr[q];
                    const unsigned h0 = pack_rne(v.x, v.y), h1 = pack_rne(v.z, v.w);
                    const float r0 = v.x - lo_f(h0), r1 = v.y - hi_f(h0), r2 = v.z - lo_f(h1), r3 = v.w - hi_f(h1);
                    const unsigned m0 = pack_rne(r0, r1), m1 = pack_rne(r2, r3);
                    const unsigned l0 = pack_rne(r0 - lo_f(m0), r1 - hi_f(m0)), l1 = pack_rne(r2 - lo_f(m1), r3 - hi_f(m1));
                    const int mt = 2 * pw + (q >> 2);
                    const int pos = ((8 * (q & 3) + r8) ^ (kg << 3)) + 32 * kg;
                    // LDS writes as inline asm: a compiler-visible LDS access next to LDS-DMA traffic makes hipcc drain vmcnt(0) in front of it
                    const unsigned e = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(dst + (size_t)((sx * 3) * 4 + mt) * 1024 + pos * 16 + half8 * 8);
                    const v2u hv = {h0, h1}, mv = {m0, m1}, lv = {l0, l1};
                    asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:4096\n\tds_write_b64 %0, %3 offset:8192"
                                 :: "v"(e), "v"(hv), "v"(mv), "v"(lv) : "memory");
                }
            };
            // prologue: chunks 0 and 1 split into LDS slots 0 / 1, chunks 2..4 on their way
            x_load(0, x0); if (G > 1) x_load(1, x1);
            x_split_store(0, x0); if (G > 1) x_split_store(1, x1);
            if (G > 2) x_load(2, x2);
            if (G > 3) x_load(3, x0);
            if (G > 4) x_load(4, x1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // iteration g (the consumers read slot g % 3): split chunk g+2 (issued three iterations ago) into slot (g+2) % 3 and re-use its
            // registers for chunk g+5
            auto step = [&](int g, float4 (&ring)[8]) {
                if (g + 2 < G) x_split_store(g + 2, ring);
                if (g + 5 < G) x_load(g + 5, ring);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            };
            for (int g = 0; g < G; g += 3) {
                step(g, x2);
                if (g + 1 >= G) break;
                step(g + 1, x0);
                if (g + 2 >= G) break;
                step(g + 2, x1);
            }
        } else {
            const int ww = wv - 6;
            auto w_copy = [&](int g) {               // chunk g of the walk = image chunk g & 15 (half = (g >> 3) & 1, c = g & 7)
                const unsigned char* src = reinterpret_cast<const unsigned char*>(Wimg) + (size_t)(g & 15) * WCH4;
                unsigned char* dst = wbuf + (g % 3) * WCH4;
#pragma unroll
                for (int q = 0; q < 12; ++q) glds16(src + (ww + 2 * q) * 1024 + lane * 16, dst + (ww + 2 * q) * 1024);
            };
            w_copy(0); if (G > 1) w_copy(1);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            // iteration g: start the copy of chunk g+2 into slot (g+2) % 3 (read last in iteration g-1); chunk g+1 must have landed by the
            // end of this iteration, the copy just issued may stay in flight: it has until the end of iteration g+1
            for (int g = 0; g < G; ++g) {
                if (g + 2 < G && (!(VAR & 4) || g < 16)) {
                    w_copy(g + 2);
                    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
        }
    } else {
        // ------------------------------------------------ consumers ------------------------------------------------------------
        const int wm = wv & 1, wn = wv >> 1;
        const int j = lane & 31, kg = lane >> 5;
        const int xpos = ((j ^ (kg << 3)) + 32 * kg) * 16;
        v16f accA[2][2], accB[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) { accA[a][b][v] = 0.f; accB[a][b][v] = 0.f; }
        // one chunk of MFMAs of unit-iteration g into `acc`
        auto mma = [&](int g, v16f (&acc)[2][2]) {
            const unsigned char* wl = wbuf + (g % 3) * WCH4;
            const unsigned char* xl = xbuf + (g % 3) * XCH4;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 xf[3][2], wf[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        xf[p][t] = *reinterpret_cast<const uint4*>(xl + ((s * 3 + p) * 4 + 2 * wm + t) * 1024 + xpos);
                        wf[p][t] = *reinterpret_cast<const uint4*>(wl + (((s * 3 + p) * 4 + 2 * wn + t) * 64 + lane) * 16);
                    }
                // X = A operand (tile rows = batch rows), W = B operand (tile columns = output columns); four independent accumulators
                // take turns; per accumulator the small products go first, the leading one last
#define X6_TERM(PX, PW)                                                                                                      \
                acc[0][0] = mfma(xf[PX][0], wf[PW][0], acc[0][0]); acc[0][1] = mfma(xf[PX][0], wf[PW][1], acc[0][1]);        \
                acc[1][0] = mfma(xf[PX][1], wf[PW][0], acc[1][0]); acc[1][1] = mfma(xf[PX][1], wf[PW][1], acc[1][1]);
                X6_TERM(0, 2) X6_TERM(2, 0) X6_TERM(1, 1) X6_TERM(0, 1) X6_TERM(1, 0) X6_TERM(0, 0)
#undef X6_TERM
            }
        };
        // slice c (0..7) of the 64 store instructions of a finished unit u: block (mt, nt) = (c >> 2, (c >> 1) & 1), registers 8 (c & 1) ..
        auto store_slice = [&](int u, int c, v16f (&acc)[2][2]) {
            const long long tile = (long long)blockIdx.x + (long long)(u >> 1) * gridDim.x;
            const int mt = c >> 2, nt = (c >> 1) & 1;
            float* yb = Y + (tile * BM + 64 * wm + 32 * mt + 4 * kg) * ldy + 128 * (u & 1) + 64 * wn + 32 * nt + j;
#pragma unroll
            for (int v8 = 0; v8 < 8; ++v8) {
                const int v = 8 * (c & 1) + v8;
                const long long r = tile * BM + 64 * wm + 32 * mt + (v & 3) + 8 * (v >> 2) + 4 * kg;
                float val = 0.f;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (a == mt && b == nt) val = acc[a][b][v];
                if (r < M && !(VAR & 1)) yb[(long long)((v & 3) + 8 * (v >> 2)) * ldy] = val;
            }
        };
        auto zero_slice = [&](int c, v16f (&acc)[2][2]) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v)
                        if (a == (c >> 2) && b == ((c >> 1) & 1) && (v >> 3) == (c & 1)) acc[a][b][v] = 0.f;
        };
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int u = 0; u < U; u += 2) {
            // unit u accumulates into A while unit u-1 (in B) leaves; then unit u+1 into B while A leaves
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int g = 8 * u + c;
                if (dbg && blockIdx.x == 7 && wv == 0 && lane == 0 && g < 64) dbg[g] = clock64();
                mma(g, accA);
                if (u > 0) { store_slice(u - 1, c, accB); zero_slice(c, accB); }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (u + 1 < U) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int g = 8 * (u + 1) + c;
                    if (dbg && blockIdx.x == 7 && wv == 0 && lane == 0 && g < 64) dbg[g] = clock64();
                    mma(g, accB);
                    store_slice(u, c, accA); zero_slice(c, accA);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
        }
        // the last unit's results
        if (U & 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) store_slice(U - 1, c, accA);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) store_slice(U - 1, c, accB);
        }
    }
}

int main(int argc, char** argv) {
    const long long M = argc > 1 ? atoll(argv[1]) : 4915200LL;
    std::vector<float> hx((size_t)1024 * H), hw((size_t)H * H);
    srand(7);
    auto nrm = [] { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
    for (auto& v : hw) v = (float)(nrm() * 0.0884);        // ~ orthogonal init with gain sqrt 2
    float *X, *W, *Y; uint4* img;
    CK(hipMalloc(&X, (size_t)M * H * 4)); CK(hipMalloc(&Y, (size_t)M * 320 * 4)); CK(hipMalloc(&W, H * H * 4));
    CK(hipMalloc(&img, 16 * WCH));
    // X: the first 1024 rows are known on the host (normal(0,1), like LayerNorm outputs; rows 512.. scaled over 12 decades), the rest a device fill
    for (size_t i = 0; i < hx.size(); ++i) {
        const size_t r = i / H;
        hx[i] = (float)(nrm() * (r < 512 ? 1.0 : std::pow(10.0, -6.0 + 12.0 * ((r - 512) / 511.0))));
    }
    CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    for (long long r = 0; r < M; r += 1024) {
        const long long n = (M - r) < 1024 ? (M - r) : 1024;
        CK(hipMemcpy(X + r * H, hx.data(), (size_t)n * H * 4, hipMemcpyHostToDevice));
    }
    uint4* img3; CK(hipMalloc(&img3, 8 * WCH3));
    long long* dbg = nullptr; if (argc > 5) CK(hipMalloc(&dbg, 64 * 8));
    const int ldy = argc > 4 ? atoi(argv[4]) : H;
    const int form = argc > 2 ? atoi(argv[2]) : 4;       // 2 = one tile per workgroup (v2), 3 = persistent producer / consumer (v3)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = argc > 3 ? atoi(argv[3]) : prop.multiProcessorCount;
    const int ntiles = (int)((M + BM - 1) / BM);
    const size_t lds3 = 2 * WCH3 + 2 * XCH3;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_pc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    pack_w<<<32, 256>>>(W, img);
    const size_t ldsb = 2 * WCH + 2 * XST;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    const int grid = (int)((M + BM - 1) / BM);
    uint4* img4; CK(hipMalloc(&img4, 16 * WCH4));
    const size_t lds4 = 3 * WCH4 + 3 * XCH4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x6_v4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    auto run = [&] {
        if (form == 4) { pack_w4<<<32, 256>>>(W, img4, 0); gemm_x6_v4<<<ncu < ntiles ? ncu : ntiles, 512, lds4>>>(X, img4, Y, M, ntiles, ldy, dbg); }
        else if (form == 3) { pack_w3<<<32, 256>>>(W, img3, 0); gemm_x6_pc<<<ncu < ntiles ? ncu : ntiles, 512, lds3>>>(X, img3, Y, M, ntiles, ldy, dbg); }
        else { pack_w<<<32, 256>>>(W, img); gemm_x6<<<grid, 256, ldsb>>>(X, img, Y, M); }
    };
    run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) run();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    // accuracy on the 1024 known rows (taken from the LAST row block that holds them completely, so that late workgroups are checked)
    const long long rb = M >= 1024 ? ((M - 1024) / 1024) * 1024 : 0;
    std::vector<float> hy((size_t)1024 * H);
    CK(hipMemcpy(hy.data(), Y + rb * H, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst_x6 = 0, worst_f32 = 0, sum_x6 = 0, sum_f32 = 0; long cnt = 0;
    for (int r = 0; r < 1024; ++r)
        for (int n = 0; n < H; ++n) {
            double ref = 0, sabs = 0; float f = 0.f;
            for (int k = 0; k < H; ++k) {
                const double p = (double)hx[(size_t)r * H + k] * (double)hw[(size_t)n * H + k];
                ref += p; sabs += std::fabs(p);
                f = std::fmaf(hx[(size_t)r * H + k], hw[(size_t)n * H + k], f);     // a k-ordered fp32 FMA chain = the f32 MFMA's numerics
            }
            const double e6 = std::fabs((double)hy[(size_t)r * H + n] - ref) / sabs, e32 = std::fabs((double)f - ref) / sabs;
            worst_x6 = e6 > worst_x6 ? e6 : worst_x6; worst_f32 = e32 > worst_f32 ? e32 : worst_f32;
            sum_x6 += e6; sum_f32 += e32; ++cnt;
        }
    if (dbg) { long long h[64]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost)); printf("cycles per chunk iteration (consumer wave 0 of workgroup 7):"); for (int g = 1; g < 64; ++g) printf("%s%lld", (g & 7) == 1 ? "\n  " : " ", h[g] - h[g - 1]); printf("\n"); }
    printf("rows %lld form %d VAR=%d: %.3f ms  %.1f TF/s fp32-equivalent (%.0f TF/s of bf16 MFMA work)\n", M, form, VAR, ms, 2.0 * M * H * H / ms / 1e9,
           12.0 * M * H * H / ms / 1e9);
    printf("error / sum|x.w| vs float64:  bf16x6 max %.3e mean %.3e   |   fp32 FMA chain max %.3e mean %.3e\n", worst_x6, sum_x6 / cnt, worst_f32,
           sum_f32 / cnt);
    return 0;
}
