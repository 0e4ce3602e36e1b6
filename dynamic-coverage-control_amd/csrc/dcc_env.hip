// dcc_env.hip -- batched multi-agent coverage environment for MI355X (gfx950 / CDNA4).
//
// One environment per 64-lane wavefront, 4 wavefronts (4 envs) per workgroup:
//   * lane i < N owns UAV i: pos/vel live in that lane's registers in float64 for all K fused
//     steps and are mirrored into LDS (apos/avel) so that every lane can broadcast-read them;
//   * lane l owns PoIs {l, l+64, ...} (PPL per lane): energy / done bits live in registers;
//   * UAVxUAV adjacency: lane <-> (a,b) pair, one __ballot per 64 pairs gives adjacency rows as
//     bitmasks; connectivity is a bitmask BFS driven by __ballot (the env is wave-uniform, so the
//     connectivity-preserving force branch is a scalar branch, never divergent);
//   * UAVxPoI distances, coverage count, min-distance reward and the PoI-assignment index are
//     computed lane-per-PoI against the LDS-resident UAV positions, reduced with ballots /
//     shuffles;
//   * observations (91 % of the bytes) are produced into a per-wave LDS staging window in the
//     reference's feature order and streamed to HBM as fully coalesced 16-byte-per-lane stores.
//
// Arithmetic follows the reference statement by statement in float64 (compiled with
// -ffp-contract=off; the only fused multiply-add is the one numpy/OpenBLAS itself performs in
// np.linalg.norm).  Reference = zhaozijie2022/dynamic-coverage-control, paths relative to
// uav_dcc_control/: "CW" envs/mpe/multiagent/CoverageWorld.py, "SC" .../scenarios/coverage.py,
// "EN" .../environment.py, "WR" envs/wrappers.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "dcc_env.h"
#include "dcc_internal.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kTileFloats = 5 * 64;  // one PoI tile of one agent row: 64 PoIs x 5 features
#ifndef DCC_STAGE_C
#define DCC_STAGE_C 1024
#endif
constexpr int kStageC = DCC_STAGE_C;  // floats in the per-wave LDS staging window (4 KB; ~3 obs rows at c2:
                                      // flushes are small and frequent, which keeps the store stream smooth)

struct KParams {
    int E, N, M, D, L, H;       // L = N*D floats per env, H = 4 + 2(N-1) header floats per agent row
    int K;                      // fused steps in this launch
    int mode;                   // 0 = step, 1 = reset
    int vec_ok;                 // obs rows may be stored as float4
    int use_connect, use_force;
    int roles_envs;             // role-specialised kernel: envs per workgroup (2; 1 for small batches: twice the workgroups, half the chain)
    int roles_pairs;            // role-specialised kernel: (physics, observation) wave pairs per workgroup: 1, or 2 (a 4-wave workgroup: one wave per SIMD)
    int roles_lds;              // ... and the LDS bytes of one pair
    int roles_slots;            // hand-off slots per env (a power of two): the physics wave may run that many steps ahead of the observation wave
    int obs_drain;              // store pacing of the row-producing waves: 2 = wait for the wave's stores in flight before every staging-window
                                // flush (default), 0 = only at the start of an env-step (role-specialised kernel), -1 = never (DCC_OBS_DRAIN, A/B)
    unsigned magicN;            // ceil(2^20 / N): p / N == (p * magicN) >> 20 for p < 4096
    double sq_cover, sq_thr, sq_thr_s, sq_speed;  // radicand bounds of the threshold tests (see kernel)
    double thr2, dmax, contact_force, contact_margin;
    double dt, keep, max_speed, sens, mass, m_energy;
    double rew_cover, rew_done, rew_out, bound_soft, bound_hard;
    float sens_f, mass_f, dt_f, m_energy_f;
    // state (library owned)
    const double2* poi;
    double2* pos;
    double2* vel;
    float* energy;
    uint8_t* done_poi;
    // inputs
    const void* actions;        // [K,E,N,2]
    unsigned long long seed;
    unsigned step0;
    int env0, env_total;
    // outputs (leading K)
    float* obs;
    float* reward;
    uint8_t* done;
    uint8_t* connect;
    uint8_t* connect_s;
    float* coverage;
    uint8_t* assign;
    double* reward64;
    // compact post-step state outputs (leading K) / inputs of the expansion kernel
    double2* st_pos;
    double2* st_vel;
    float* st_energy;
    uint8_t* st_done;
};

__device__ __forceinline__ void wave_fence() {
    // LDS traffic of one wave is issued and serviced in order; this only stops the compiler from
    // moving LDS accesses across the point where other lanes' data is consumed.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// np.linalg.norm of a 2-vector as numpy evaluates it: x.dot(x) -> OpenBLAS ddot -> fma (see oracle).
__device__ __forceinline__ double norm2(double a, double b) { return __builtin_sqrt(__builtin_fma(b, b, a * a)); }

__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// Wave-wide float64 sum on the VALU (DPP moves of the two 32-bit halves: 4 steps leave every 16-lane row with its row sum,
// the 4 row sums are combined through readlane) instead of 6 dependent ds_bpermute (LDS) round trips per reduction.
// Deterministic; used by the env step's reward (latency chain of small batches) and by the feature producer (a chain of
// 2N + 2 such reductions per state, purely latency-bound: 17 us per 4096 states with the ds_bpermute form).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
    v += dpp_mov_f64<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);   // row_half_mirror
    v += dpp_mov_f64<0x140>(v);   // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        double w = __shfl_xor(v, o, 64);
        v = (w < v) ? w : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// numpy's npy_logaddexp(0, y) (CW:136).
__device__ __forceinline__ double logaddexp0(double y) {
    if (y == 0.0) return 0.6931471805599453094172321214581766;
    if (y < 0.0) return log1p(exp(y));
    return y + log1p(exp(-y));
}

// Per-wave staging window over the flat [N*D] observation block of one env-step.
struct Stager {
    float* stg;    // LDS, 16-byte aligned, C floats
    float* gout;   // HBM base of this env-step's obs block
    int w0;        // flat index held by stg[0] (multiple of 4 in vector mode)
    int vec;
    int drain = 0; // 2: a flush starts only when the wave's earlier stores have been taken by the L2 (KParams::obs_drain)

    // Stream out [w0, end) and slide the window so that `s` (next flat index to be produced) fits.
    __device__ __forceinline__ void flush(int s, int lane) {
        wave_fence();
        // Paced store stream: at most one window (4 KB = four 1 KB store instructions) of this wave is in flight.  Measured on
        // MI355X (profiles/r06/obs_store_pacing.txt): c2 x 4096 envs +3.0-4.5 %, c4 / c5 shards +4.9 / +3.6 %, small batches and
        // the 8192- / 16384-env legs unchanged; windows of 2 / 8 / 16 KB, a deeper queue (vmcnt >= 4) or non-temporal stores lose.
        if (drain == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int end = vec ? (s & ~3) : s;
        const int n = end - w0;
        if (vec) {
            const int nv = n >> 2;
            const float4* s4 = reinterpret_cast<const float4*>(stg);
            float4* g4 = reinterpret_cast<float4*>(gout + w0);
            // Four 1 KB pieces per trip: the LDS reads are issued back to back (one LDS round trip per trip instead of one per
            // piece -- a read past `nv` stays inside the window and is never stored), then the stores.  Trip count is wave-uniform.
            static_assert(kStageC % 1024 == 0, "flush() reads whole groups of 4 x 64 float4 from the staging window");
            for (int b = 0; b < nv; b += 256) {
                const int v = b + lane;
                const float4 a0 = s4[v], a1 = s4[v + 64], a2 = s4[v + 128], a3 = s4[v + 192];
                if (v < nv) g4[v] = a0;
                if (v + 64 < nv) g4[v + 64] = a1;
                if (v + 128 < nv) g4[v + 128] = a2;
                if (v + 192 < nv) g4[v + 192] = a3;
            }
        } else {
            for (int v = lane; v < n; v += 64) gout[w0 + v] = stg[v];
        }
        const int tail = s - end;  // 0..3 floats already produced beyond the last full float4
        float tv = 0.f;
        if (lane < tail) tv = stg[n + lane];
        wave_fence();
        if (lane < tail) stg[lane] = tv;
        w0 = end;
        wave_fence();
    }
    __device__ __forceinline__ float* reserve(int s, int len, int lane) {
        if (s + len - w0 > kStageC) flush(s, lane);
        return stg + (s - w0);
    }
};

// ACT: 0 = float32 actions from HBM, 1 = float64 actions from HBM, 2 = drawn in-kernel (float32).
// FORCE: the connectivity-preserving pull force (CW:100-140) is compiled in.
//
// Exact sqrt-free comparisons: every threshold test on a distance d = sqrt_rn(s) (IEEE, correctly
// rounded, monotone in s) is rewritten as a test on the radicand s against a host-computed bound
//   d <= r  <=>  s <= max{ t : sqrt_rn(t) <= r }      d < r  <=>  s <= max{ t : sqrt_rn(t) < r }
// and min_i sqrt_rn(s_i) == sqrt_rn(min_i s_i), so the results are bit-identical to taking the
// square roots (the argmin tie rule is handled by an exact slow path, see `near`).
//
// NC / MC > 0: N and M are compile-time constants (the BASELINE configs): every loop over agents
// and PoI tiles unrolls, the staging-window flush points become static and the scalar unit (one per
// CU, shared by the resident waves) is relieved of loop control and index arithmetic.  NC = 0 is
// the generic runtime-size code.

// XCD-aware workgroup -> env-group mapping.  Workgroup b runs on XCD b % 8 (observed dispatch order; used for
// speed only, never for correctness), and each XCD has its own L2.  The small per-step outputs ([K,E] arrays:
// 1-4 bytes per env) of neighbouring envs share cache lines, so neighbouring envs should be written from the
// SAME XCD, where the partial lines merge in one L2 before they reach HBM: workgroups b, b+8, b+16, ... (one
// XCD) take consecutive env groups.
__device__ __forceinline__ int xcd_swizzle(int b, int nblocks) {
#ifdef DCC_NO_XCD_SWIZZLE
    return b;
#else
    const int full = nblocks & ~7;          // blocks beyond the last multiple of 8 keep their index
    if (b >= full) return b;
    return (b & 7) * (full >> 3) + (b >> 3);
#endif
}

// action loads per lane and chunk: 4 (x 64/N steps) for float32 actions when there is one PoI per lane, no pull-force
// path and hence registers to spare (8 spills), else 2; 2 for float64 actions (twice the registers)
template <int PPL, bool FORCE> constexpr int act_rf() { return (PPL == 1 && !FORCE) ? 3 : 2; }
constexpr int ACT_RD = 2;

// Per-lane registers of one env: lane i < N holds UAV i, lane l holds PoIs {l, l+64, ...}.
template <int PPL>
struct EnvRegs {
    double px, py, vx, vy;
    float en[PPL];
    unsigned dmask;  // bit q: PoI q*64+lane is done
};

// Chunked action prefetch: lane (s*N + i) holds the action of agent i at step (chunk start + s);
// R loads per lane -> R*(64/N) steps per chunk, so that the unavoidable vmcnt wait (which also
// drains the wave's older stores) is paid once per chunk instead of once per step.
// The chunk AFTER the current one is already in flight (fn / dn), so its HBM latency is hidden behind a
// whole chunk of steps.
template <int RF>
struct ActFetch {
    float2 f[RF], fn[RF];
    double2 d[ACT_RD], dn[ACT_RD];
    int kc, r_sel, s_sel;
    bool primed;      // the first chunk's loads were issued before the step loop (prefetch_actions)
};

// Per-step outputs of one env, handed from the physics wave to the observation wave (role-specialised kernel):
// the physics wave then issues no HBM store at all and never queues behind the observation stream.
struct StepRec {
    double R;
    float cov;
    unsigned flags;            // bit 0 done, bit 1 connect, bit 2 connect_s
    unsigned char assign[64];  // PoI-assignment index of PoI `lane` (one PoI per lane)
};

// PoI coordinates of this lane: registers for <= 4 PoIs per lane, the LDS table otherwise.
template <int PPL>
struct PoiLane {
    static constexpr bool REG = PPL <= 4;
    static constexpr int NR = REG ? PPL : 1;
    double x[NR], y[NR];
    const double2* table;
    int lane, M;
    __device__ __forceinline__ void init(const double2* s_poi, int lane_, int M_) {
        table = s_poi; lane = lane_; M = M_;
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int j = q * 64 + lane;
            x[q] = 0; y[q] = 0;
            if (REG && j < M) { const double2 pj = s_poi[j]; x[q] = pj.x; y[q] = pj.y; }
        }
    }
    __device__ __forceinline__ double2 get(int q) const {
        if (REG) return make_double2(x[REG ? q : 0], y[REG ? q : 0]);
        const int j = q * 64 + lane;
        return table[j < M ? j : 0];
    }
};

// HBM stores of the per-step outputs of one env-step (ko = k*E + env).  amin: this lane's PoI-assignment
// indices.  Flags: one byte-store instruction (lanes 0-2, one array each); reward / coverage: one dword-store
// instruction (lanes 0-1); assignment: four neighbouring lanes pack their bytes into one dword (two
// quad-permute DPP moves) -> a 16-lane dword store instead of a 64-lane byte store.
template <int PPL>
__device__ __forceinline__ void write_step_outputs(const KParams& p, const size_t ko, const int M, const int lane,
                                                   const double R, const float cov, const bool env_done,
                                                   const bool connect, const bool connect_s, const int (&amin)[PPL],
                                                   const bool skip_assign = false) {
    if (p.assign && !skip_assign) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int j = q * 64 + lane;
            const bool valid = j < M;
            if ((M & 3) == 0) {
                const int t1 = amin[q] | (__builtin_amdgcn_update_dpp(0, amin[q], 0xF9, 0xF, 0xF, false) << 8);   // lane+1
                const int t2 = t1 | (__builtin_amdgcn_update_dpp(0, t1, 0xEE, 0xF, 0xF, false) << 16);            // lane+2
                if (valid && (lane & 3) == 0) *reinterpret_cast<unsigned*>(p.assign + ko * M + j) = (unsigned)t2;
            } else if (valid) {
                p.assign[ko * M + j] = (uint8_t)amin[q];
            }
        }
    }
    // The pointers are copied to (scalar) locals first: `lane == 0 ? p.done : ...` on the struct MEMBERS is a conditional
    // lvalue, i.e. a per-lane ADDRESS into the kernel-argument segment followed by a vector load + `s_waitcnt vmcnt(0)` --
    // which on gfx9 also waits for every row store the wave still has in flight, once per env-step.
    uint8_t* const d_done = p.done;
    uint8_t* const d_connect = p.connect;
    uint8_t* const d_connect_s = p.connect_s;
    float* const d_reward = p.reward;
    float* const d_coverage = p.coverage;
    if (lane < 3) {
        uint8_t* dst = lane == 0 ? d_done : lane == 1 ? d_connect : d_connect_s;
        const bool v = lane == 0 ? env_done : lane == 1 ? connect : connect_s;
        if (dst) dst[ko] = v ? 1 : 0;
    }
    if (lane < 2) {
        float* dst = lane == 0 ? d_reward : d_coverage;
        const float v = lane == 0 ? (float)R : cov;
        if (dst) dst[ko] = v;
    }
    if (lane == 0 && p.reward64) p.reward64[ko] = R;
}

// One reference env.step (phases A-F, H, I of SURVEY.md 3.3) of env `env` at fused step `k`.
// apos_in: UAV positions BEFORE the move (LDS, [N]); apos_out / avel_out: where the post-step (and
// post-auto-reset) positions / velocities go (may alias apos_in).  Per-step outputs go to HBM, or into the LDS
// record `rec` (one PoI per lane only) from which the observation wave stores them.
template <int PPL, int ACT, bool FORCE, int NC, int MC>
__device__ __forceinline__ void env_physics_step(const KParams& p, const int env, const int k, const int lane,
                                                 EnvRegs<PPL>& r, ActFetch<act_rf<PPL, FORCE>()>& af, const PoiLane<PPL>& poi,
                                                 const double2* apos_in, double2* apos_out, double2* avel_out,
                                                 StepRec* rec = nullptr) {
    constexpr bool SPEC = NC > 0;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    constexpr int UNR_F = SPEC ? 4 : 2;  // unroll of the energy-pass loop over agents
    const size_t ko = (size_t)k * p.E + env;  // index of this env-step in [K,E] outputs
    const unsigned long long fullN = (N >= 64) ? ~0ULL : ((1ULL << N) - 1ULL);

    // ---- (A) EN:153-201 u = action; u *= 5.0 in the action's dtype ------------------------------
    float uxf = 0.f, uyf = 0.f;
    double uxd = 0, uyd = 0;
    if constexpr (ACT == 2) {
        if (lane < N) {
            const unsigned long long idx =
                ((unsigned long long)(p.step0 + (unsigned)k) * (unsigned long long)p.env_total +
                 (unsigned long long)(p.env0 + env)) * (unsigned long long)N + (unsigned long long)lane;
            const unsigned long long z = splitmix64(p.seed + 0x9E3779B97F4A7C15ULL * (idx + 1ULL));
            const unsigned hi = (unsigned)(z >> 40), lo = (unsigned)((z & 0xFFFFFFFFULL) >> 8);
            uxf = ((float)hi * (1.0f / 8388608.0f) - 1.0f) * p.sens_f;
            uyf = ((float)lo * (1.0f / 8388608.0f) - 1.0f) * p.sens_f;
        }
    } else {
        constexpr int ACT_R = (ACT == 1) ? ACT_RD : act_rf<PPL, FORCE>();
        const int steps_per_load = 64 / N;  // >= 1 because N <= 64
        const int chunk = ACT_R * steps_per_load;
        if (af.kc == chunk) { af.kc = 0; af.r_sel = 0; af.s_sel = 0; }
        if (af.kc == 0) {
            const int my_s = SPEC ? (lane / (SPEC ? NC : 1)) : (int)(((unsigned)lane * p.magicN) >> 20);  // lane / N
            const int my_i = lane - my_s * N;                                                              // lane % N
            auto issue = [&](int k0) {   // loads of the chunk that starts at step k0 -> fn / dn
#pragma unroll
                for (int rr = 0; rr < ACT_R; ++rr) {
                    const int ks = k0 + rr * steps_per_load + my_s;
                    if (my_s < steps_per_load && ks < p.K) {
                        const size_t ai = ((size_t)ks * p.E + env) * N + my_i;
                        if (ACT == 1) af.dn[rr] = reinterpret_cast<const double2*>(p.actions)[ai];
                        else af.fn[rr] = reinterpret_cast<const float2*>(p.actions)[ai];
                    }
                }
            };
            // float32 actions (the production dtype): the chunk AFTER the current one is kept in flight, so the
            // consume below finds loads that were issued a whole chunk of steps ago.  float64 actions (twice the
            // registers) are loaded and consumed on the spot.
            constexpr bool AHEAD = (ACT == 0);
            if ((!AHEAD || k == 0) && !(k == 0 && af.primed)) issue(k);
            // consume inside this branch: the vmcnt wait is paid only on chunk boundaries
#pragma unroll
            for (int rr = 0; rr < ACT_R; ++rr) {
                if (ACT == 1) { asm volatile("" : "+v"(af.dn[rr].x), "+v"(af.dn[rr].y)); af.d[rr] = af.dn[rr]; }
                else { asm volatile("" : "+v"(af.fn[rr].x), "+v"(af.fn[rr].y)); af.f[rr] = af.fn[rr]; }
            }
            if (AHEAD && k + chunk < p.K) issue(k + chunk);
        }
        if (af.s_sel == steps_per_load) { af.s_sel = 0; ++af.r_sel; }
        const int src = af.s_sel * N + lane;  // lane holding (step kc, agent = lane)
        ++af.kc; ++af.s_sel;
        if (ACT == 1) {
            double2 a = af.d[0];
#pragma unroll
            for (int rr = 1; rr < ACT_R; ++rr) if (af.r_sel == rr) a = af.d[rr];
            const double axd = __shfl(a.x, src & 63, 64), ayd = __shfl(a.y, src & 63, 64);
            if (lane < N) { uxd = axd * p.sens; uyd = ayd * p.sens; }
        } else {
            float2 a = af.f[0];
#pragma unroll
            for (int rr = 1; rr < ACT_R; ++rr) if (af.r_sel == rr) a = af.f[rr];
            const float axf = __shfl(a.x, src & 63, 64), ayf = __shfl(a.y, src & 63, 64);
            if (lane < N) { uxf = axf * p.sens_f; uyf = ayf * p.sens_f; }
        }
    }

    // ---- (B) CW:70-93 update_connect on PRE-move positions --------------------------------------
    unsigned long long rowA = 0, rowS = 0;  // lane a: adjacency rows (bit b)
    bool connect = false, connect_s = false;
    if (p.use_connect) {
        const int npairs = N * N;
        for (int r0 = 0; r0 < npairs; r0 += 64) {
            const int pidx = r0 + lane;
            const int a = SPEC ? (pidx / (SPEC ? NC : 1)) : (int)(((unsigned)pidx * p.magicN) >> 20);
            const int b = pidx - a * N;
            bool adj = false, adjs = false;
            if (pidx < npairs && a != b) {
                const double2 pa = apos_in[a], pb = apos_in[b];
                const double dx = pa.x - pb.x, dy = pa.y - pb.y;
                const double s = __builtin_fma(dy, dy, dx * dx);
                adj = s <= p.sq_thr;                 // d < r_a + r_b              (CW:77)
                adjs = adj && (s <= p.sq_thr_s);     // d < comm_r_scale*(r_a+r_b) (CW:79)
            }
            const unsigned long long mA = __ballot(adj), mS = __ballot(adjs);
            if (lane < N) {
                const int rs = lane * N;  // first pair index of my row
                const int lo = rs > r0 ? rs : r0;
                const int hi = (rs + N) < (r0 + 64) ? (rs + N) : (r0 + 64);
                if (lo < hi) {
                    const int w = hi - lo;
                    const unsigned long long msk = (w >= 64) ? ~0ULL : ((1ULL << w) - 1ULL);
                    rowA |= ((mA >> (lo - r0)) & msk) << (lo - rs);
                    rowS |= ((mS >> (lo - r0)) & msk) << (lo - rs);
                }
            }
        }
        // connect  = all(sum_k A^k > 0)  <=>  graph(A) connected: BFS from node 0 (A symmetric)
        unsigned long long visited = 1ULL;
        for (int it = 1; it < N; ++it) {
            const unsigned long long nv = visited | __ballot(lane < N && (rowA & visited) != 0ULL);
            if (nv == visited) break;
            visited = nv;
        }
        connect = (visited == fullN);
        // connect_ = all(I + sum_{k>=1} A^k A_ > 0) (CW:90 multiplies the just-appended A^k):
        // N=1 -> True; N=2 -> always False; N>=3 -> connected and every node has an A_ neighbour.
        const unsigned long long iso = __ballot(lane < N && rowS == 0ULL);
        connect_s = (N == 1) ? true : (N == 2) ? false : (connect && iso == 0ULL);

        // ---- (D) CW:100-140 connectivity-preserving pull force (wave-uniform branch) ----------------
        if (FORCE && !connect_s) {
            double best = 0.0, fx = 0.0, fy = 0.0;
            int bi = 0;
            const bool branch1 = (iso != 0ULL);
            const bool mine = (lane < N) && (branch1 ? (rowS == 0ULL) : true);
            if (mine) {
                const double2 pa = apos_in[lane];
                for (int b = 0; b < N; ++b) {
                    const double2 pb = apos_in[b];
                    double d = norm2(pa.x - pb.x, pa.y - pb.y);
                    if (b == lane) d = 1e5;                       // CW:81
                    else if (!branch1 && d < p.thr2) d = 1e5;     // CW:119-120
                    if (b == 0 || d < best) { best = d; bi = b; } // np.argmin: first minimum
                }
            }
            unsigned long long todo;
            if (branch1) {
                todo = iso;  // CW:110-116: every isolated agent, ascending
            } else {
                // CW:121-123: argmin over the flattened matrix = smallest row minimum, lowest row first
                const double g = wave_min_f64(mine ? best : 1.7976931348623157e308);
                const unsigned long long eq = __ballot(mine && best == g);
                todo = 1ULL << __builtin_ctzll(eq);
            }
            if (mine && bi != lane && ((todo >> lane) & 1ULL)) {
                // CW:129-140 get_connect_force(a = me, b = bi)
                const double2 pa = apos_in[lane], pb = apos_in[bi];
                const double dx = pa.x - pb.x, dy = pa.y - pb.y;
                const double dist = norm2(dx, dy);
                const double pen = logaddexp0((dist - p.dmax) / p.contact_margin) * p.contact_margin;
                fx = p.contact_force * dx / dist * pen;
                fy = p.contact_force * dy / dist * pen;
            }
            // sequential accumulation in the reference's order (float32 round trip per add when the
            // action is float32: `p_force[a] += f_a` on a float32 array)
            while (todo) {
                const int a = __builtin_ctzll(todo);
                todo &= todo - 1ULL;
                const int b = __builtin_amdgcn_readlane(bi, a);
                const double fax = readlane_f64(fx, a), fay = readlane_f64(fy, a);
                if (a == b) continue;  // get_connect_force returns [0, 0] (CW:130-131)
                if (lane == a) {
                    if (ACT == 1) { uxd += -fax; uyd += -fay; }
                    else { uxf = (float)((double)uxf + (-fax)); uyf = (float)((double)uyf + (-fay)); }
                }
                if (lane == b) {
                    if (ACT == 1) { uxd += fax; uyd += fay; }
                    else { uxf = (float)((double)uxf + fax); uyf = (float)((double)uyf + fay); }
                }
            }
        }
    }
    wave_fence();  // every read of the pre-move positions is done before they may be overwritten

    // ---- (E) CW:142-155 integrate_state -----------------------------------------------------------
    if (lane < N) {
        r.vx = r.vx * p.keep; r.vy = r.vy * p.keep;
        if (ACT == 1) {
            r.vx += (uxd / p.mass) * p.dt; r.vy += (uyd / p.mass) * p.dt;
        } else {
            float ax = uxf, ay = uyf;
            if (p.mass_f != 1.0f) { ax = ax / p.mass_f; ay = ay / p.mass_f; }   // x / 1.0f == x exactly
            ax *= p.dt_f; ay *= p.dt_f;
            r.vx += (double)ax; r.vy += (double)ay;
        }
        const double s2 = r.vx * r.vx + r.vy * r.vy;  // np.square + np.square: no fusion (CW:150)
        if (s2 > p.sq_speed) {                        // sqrt(s2) > max_speed
            const double speed = __builtin_sqrt(s2);
            r.vx = r.vx / speed * p.max_speed; r.vy = r.vy / speed * p.max_speed;
        }
        r.px += r.vx * p.dt; r.py += r.vy * p.dt;
        apos_out[lane] = make_double2(r.px, r.py);
        avel_out[lane] = make_double2(r.vx, r.vy);
    }
    wave_fence();

    // ---- (F) CW:157-174 update_energy + SC:80-97 reward terms on POST-move positions ---------------
    int n_done = 0, n_just = 0;
    int amin_q[PPL];
    double part = 0.0;  // per-lane share of (OOB terms - sum of min distances)
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int j = q * 64 + lane;
        const bool valid = j < M;
        const double2 pj = poi.get(q);
        int cnt = 0, amin = 0;
        // smin: running minimum of the radicand; sprev: the running minimum just before its last
        // displacement = the smallest radicand among the agents in front of the argmin.  If sprev is
        // within a few ulp of smin the rounded distances could tie (np.argmin would then keep the
        // earlier agent): that case, practically never taken, is resolved exactly below.
        double smin = 1.7976931348623157e308, sprev = 1.7976931348623157e308;
#pragma unroll UNR_F
        for (int i = 0; i < N; ++i) {
            const double2 xa = apos_out[i];   // (LDS broadcast read; fetching x_i with v_readlane from its owner lane is no faster, r05 A/B)
            const double dx = pj.x - xa.x, dy = pj.y - xa.y;
            const double s = __builtin_fma(dy, dy, dx * dx);
            cnt += (s <= p.sq_cover) ? 1 : 0;  // ||p_j - x_i|| <= r_cover (CW:164-165)
            const bool lt = s < smin;
            sprev = lt ? smin : sprev;
            smin = lt ? s : smin;
            amin = lt ? i : amin;
        }
        const bool near = sprev <= smin * 1.0000000000000009;
        if (near && p.assign) {  // exact first-minimum on the rounded distances (practically never taken)
            double dmn = 0.0;
            for (int i = 0; i < N; ++i) {
                const double2 xa = apos_out[i];
                const double d = norm2(pj.x - xa.x, pj.y - xa.y);
                if (i == 0 || d < dmn) { dmn = d; amin = i; }
            }
        }
        bool dn = (r.dmask >> q) & 1u, just = false;
        if (valid && !dn) {
            r.en[q] += (float)cnt;
            if (r.en[q] >= p.m_energy_f) { dn = true; just = true; r.dmask |= 1u << q; }
        }
        n_done += __popcll(__ballot(valid && dn));
        n_just += __popcll(__ballot(just));
        if (valid && !dn) part -= __builtin_sqrt(smin);
        amin_q[q] = amin;
    }
    bool oob = false;
    if (lane < N) {
        const double ax = fabs(r.px), ay = fabs(r.py);
        double s = 0.0;
        if (ax > p.bound_soft) s += ax - p.bound_soft;
        if (ay > p.bound_soft) s += ay - p.bound_soft;
        part += s * p.rew_out;
        oob = (ax > p.bound_hard) || (ay > p.bound_hard);
        if (oob) part += p.rew_out;
    }
    const bool any_oob = __ballot(oob) != 0ULL;
    const bool all_done = (n_done == M);
    // VALU-only (DPP) wave sum: the ds_bpermute form is 6 dependent LDS round trips, 0.2 us of the ~2 us step chain that bounds
    // small batches (512 envs: 2.02 -> 1.81 us per step, profiles/r05/ab_small_batch_chain.txt); another summation order, equally
    // far from the reference's sequential `rew -= min(dists)` loop (rewards are compared at 1e-12)
    double base = wave_sum_f64_dpp(part);
    if (all_done) base += p.rew_done;
    // EN:106-108 sum over the N per-agent rewards; `just` bonus is paid once (SC:87-89)
    const double R = (double)N * base + p.rew_cover * (double)n_just;
    const bool env_done = all_done || any_oob;  // SC:112-117
    const float cov = (float)((double)n_done / (double)M);
    if (rec != nullptr) {
        if (lane == 0) { rec->R = R; rec->cov = cov; rec->flags = (env_done ? 1u : 0u) | (connect ? 2u : 0u) | (connect_s ? 4u : 0u); }
        rec->assign[lane] = (unsigned char)amin_q[0];
    } else {
        write_step_outputs<PPL>(p, ko, M, lane, R, cov, env_done, connect, connect_s, amin_q);
    }
    // ---- (I) WR:104-109 auto-reset -> SC:64-78 ------------------------------------------------------
    if (env_done) {
        r.px = r.py = r.vx = r.vy = 0.0;
        r.dmask = 0;
#pragma unroll
        for (int q = 0; q < PPL; ++q) r.en[q] = 0.f;
        if (lane < N) { apos_out[lane] = make_double2(0.0, 0.0); avel_out[lane] = make_double2(0.0, 0.0); }
        wave_fence();
    }
}

// ---- (G) SC:99-110 observation rows of one env-step, streamed through the LDS staging window --------
// apv: flat view [pos | vel][N][2] of the (post-reset) UAV state in LDS; en / dmask: this lane's PoIs.
template <int PPL, bool FORCE, int NC, int MC>
__device__ __forceinline__ void produce_obs(const KParams& p, Stager& st, const double* apv,
                                            const float (&en)[PPL], const unsigned dmask, const PoiLane<PPL>& poi,
                                            const int lane, const int flat_base = 0, const bool finish = true) {
    // st: the staging window over the output stream that starts at st.gout; this env-step's block occupies the
    // flat range [flat_base, flat_base + L) of it.  finish = false leaves the tail of the block in the window:
    // the role-specialised kernel streams the adjacent blocks of its two envs as ONE stream, so the cache line
    // that straddles the two blocks is written by one store instead of two half-line stores.
    constexpr bool SPEC = NC > 0;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    const int H = 4 + 2 * (N - 1), D = H + 5 * M, L = N * D;
    constexpr int UNR_O = SPEC ? ((NC <= 8 && !FORCE) ? NC : 2) : 1;  // full unroll -> static flush points
    const double2* apos = reinterpret_cast<const double2*>(apv);
#pragma unroll UNR_O
    for (int i = 0; i < N; ++i) {
        const double2 xi = apos[i];
        // header: vel(2) pos(2) (x_k - x_i for k != i); branch-free per lane:
        //   f<2 -> avel[i][f&1]; f<4 -> apos[i][f&1]; else apos[k'][f&1] - x_i[f&1], k' skips i
        for (int f0 = 0; f0 < H; f0 += 64) {
            const int len = (H - f0) < 64 ? (H - f0) : 64;
            float* dst = st.reserve(flat_base + i * D + f0, len, lane);
            const int f = f0 + lane;
            if (f < H) {
                const int c = f & 1;
                const int kk = (f - 4) >> 1;
                const bool rel = f >= 4;
                const int src = rel ? (kk + (kk >= i ? 1 : 0)) : (f < 2 ? N + i : i);
                const double val = apv[2 * src + c];
                const double sub = rel ? (c ? xi.y : xi.x) : 0.0;
                dst[lane] = (float)(val - sub);
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (q * 64 < M) {
                const int cntj = (M - q * 64) < 64 ? (M - q * 64) : 64;
                float* dst = st.reserve(flat_base + i * D + H + q * kTileFloats, 5 * cntj, lane);
                if (lane < cntj) {
                    float* d5 = dst + 5 * lane;
                    const double2 pj = poi.get(q);
                    d5[0] = (float)(pj.x - xi.x);
                    d5[1] = (float)(pj.y - xi.y);
                    d5[2] = en[q];
                    d5[3] = p.m_energy_f;
                    d5[4] = ((dmask >> q) & 1u) ? 1.f : 0.f;
                }
            }
        }
    }
    if (finish) st.flush(flat_base + L, lane);
}

template <int PPL>
__device__ __forceinline__ void load_env_state(const KParams& p, int env, int lane, int N, int M, EnvRegs<PPL>& r) {
    r.px = r.py = r.vx = r.vy = 0.0;
    r.dmask = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) r.en[q] = 0.f;
    if (p.mode == 0) {
        if (lane < N) {
            const double2 a = p.pos[(size_t)env * N + lane], b = p.vel[(size_t)env * N + lane];
            r.px = a.x; r.py = a.y; r.vx = b.x; r.vy = b.y;
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const int j = q * 64 + lane;
            if (j < M) {
                r.en[q] = p.energy[(size_t)env * M + j];
                if (p.done_poi[(size_t)env * M + j]) r.dmask |= 1u << q;
            }
        }
    }
    // Every value loaded from HBM is consumed here, once: the step loop then contains no use of a
    // pending load, so the compiler never has to drain the obs stores (s_waitcnt vmcnt(0)) of earlier
    // steps in the middle of a step.
    asm volatile("" : "+v"(r.px), "+v"(r.py), "+v"(r.vx), "+v"(r.vy), "+v"(r.dmask));
#pragma unroll
    for (int q = 0; q < PPL; ++q) asm volatile("" : "+v"(r.en[q]));
}

// Per-step compact state output (post-reset): what the observations of this env-step are a function of.
template <int PPL>
__device__ __forceinline__ void write_step_state(const KParams& p, const size_t ko, int lane, int N, int M,
                                                 const EnvRegs<PPL>& r) {
    if (p.st_pos && lane < N) p.st_pos[ko * N + lane] = make_double2(r.px, r.py);
    if (p.st_vel && lane < N) p.st_vel[ko * N + lane] = make_double2(r.vx, r.vy);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int j = q * 64 + lane;
        if (j < M) {
            if (p.st_energy) p.st_energy[ko * M + j] = r.en[q];
            if (p.st_done) p.st_done[ko * M + j] = (r.dmask >> q) & 1u;
        }
    }
}

template <int PPL>
__device__ __forceinline__ void store_env_state(const KParams& p, int env, int lane, int N, int M, const EnvRegs<PPL>& r) {
    if (lane < N) {
        p.pos[(size_t)env * N + lane] = make_double2(r.px, r.py);
        p.vel[(size_t)env * N + lane] = make_double2(r.vx, r.vy);
    }
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int j = q * 64 + lane;
        if (j < M) {
            p.energy[(size_t)env * M + j] = r.en[q];
            p.done_poi[(size_t)env * M + j] = (r.dmask >> q) & 1u;
        }
    }
}

template <int RF>
__device__ __forceinline__ void init_act(ActFetch<RF>& af) {
#pragma unroll
    for (int rr = 0; rr < RF; ++rr) af.f[rr] = af.fn[rr] = make_float2(0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < ACT_RD; ++rr) af.d[rr] = af.dn[rr] = make_double2(0.0, 0.0);
    af.kc = 0; af.r_sel = 0; af.s_sel = 0; af.primed = false;
}

// Issue the loads of the first action chunk (the same addresses env_physics_step's `issue(0)` would load) before anything
// else of the launch is waited for; the step at k = 0 then finds them in flight (af.primed).
template <int PPL, int ACT, bool FORCE, int NC>
__device__ __forceinline__ void prefetch_actions(const KParams& p, const int env, const int lane, ActFetch<act_rf<PPL, FORCE>()>& af) {
    if constexpr (ACT != 2) {
        constexpr bool SPEC = NC > 0;
        const int N = SPEC ? NC : p.N;
        constexpr int ACT_R = (ACT == 1) ? ACT_RD : act_rf<PPL, FORCE>();
        const int steps_per_load = 64 / N;
        const int my_s = SPEC ? (lane / (SPEC ? NC : 1)) : (int)(((unsigned)lane * p.magicN) >> 20);
        const int my_i = lane - my_s * N;
#pragma unroll
        for (int rr = 0; rr < ACT_R; ++rr) {
            const int ks = rr * steps_per_load + my_s;
            if (my_s < steps_per_load && ks < p.K) {
                const size_t ai = ((size_t)ks * p.E + env) * N + my_i;
                if (ACT == 1) af.dn[rr] = reinterpret_cast<const double2*>(p.actions)[ai];
                else af.fn[rr] = reinterpret_cast<const float2*>(p.actions)[ai];
            }
        }
        af.primed = true;
    }
}

// ---- kernel 1: fused -- one env per wavefront does physics AND its own observation stores ------------
// Register budget: 4 waves/SIMD (<=128 VGPRs) keeps all 1024 workgroups of a 4096-env batch co-resident;
// kernels that hold >= 8 PoIs per lane or the pull-force path trade occupancy for registers instead of spilling.
template <int PPL, int ACT, bool FORCE, int NC, int MC>
__global__ __launch_bounds__(kBlock, (PPL >= 8 ? 2 : (FORCE ? (PPL >= 4 ? 2 : 3) : 4))) void dcc_env_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env = xcd_swizzle(blockIdx.x, gridDim.x) * kWavesPerBlock + wid;
    constexpr bool SPEC = NC > 0;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    const int L = N * (4 + 2 * (N - 1) + 5 * M);

    // LDS carve: PoI table shared by the block, then per wave: apos[N], avel[N], staging[C].
    double2* s_poi = reinterpret_cast<double2*>(smem);
    const int per_wave = N * 32 + kStageC * 4;
    unsigned char* wbase = smem + ((M * 16 + 15) & ~15) + wid * per_wave;
    double2* apos = reinterpret_cast<double2*>(wbase);
    double2* avel = apos + N;
    float* stg = reinterpret_cast<float*>(avel + N);

    // <= 4 PoIs per lane: their coordinates live in registers, read straight from the (L2-resident) table -- no LDS
    // staging, no workgroup barrier on the way to the first step (the K = 1 launches of a policy-driven rollout are pure
    // latency: 9.6 -> 8.x us).  More PoIs per lane: the table is staged in LDS and read from there.
    if (!PoiLane<PPL>::REG) {
        for (int j = threadIdx.x; j < M; j += kBlock) s_poi[j] = p.poi[j];
        __syncthreads();
    }
    if (env >= p.E) return;

    EnvRegs<PPL> r;
    PoiLane<PPL> poi;
    ActFetch<act_rf<PPL, FORCE>()> af;
    init_act(af);
    if (p.mode == 0) prefetch_actions<PPL, ACT, FORCE, NC>(p, env, lane, af);   // in flight while the state loads are waited for
    poi.init(PoiLane<PPL>::REG ? p.poi : s_poi, lane, M);
    load_env_state<PPL>(p, env, lane, N, M, r);
    if (lane < N) { apos[lane] = make_double2(r.px, r.py); avel[lane] = make_double2(r.vx, r.vy); }
    wave_fence();

    for (int k = 0; k < p.K; ++k) {
        if (p.mode == 0) env_physics_step<PPL, ACT, FORCE, NC, MC>(p, env, k, lane, r, af, poi, apos, apos, avel);
        if (p.st_pos || p.st_vel || p.st_energy || p.st_done) write_step_state<PPL>(p, (size_t)k * p.E + env, lane, N, M, r);
        if (p.obs) {
            Stager st;
            st.stg = stg; st.w0 = 0; st.vec = SPEC ? 1 : p.vec_ok; st.drain = p.obs_drain;
            st.gout = p.obs + ((size_t)k * p.E + env) * (size_t)L;
            produce_obs<PPL, FORCE, NC, MC>(p, st, reinterpret_cast<const double*>(apos), r.en, r.dmask, poi, lane);
        }
    }
    store_env_state<PPL>(p, env, lane, N, M, r);
}

// ---- kernel 3: observation rows from compact state (dcc_obs_expand) ------------------------------------------
// One wavefront per state: the same produce_obs() as the env kernels, fed from state arrays instead of a step.
template <int PPL>
__global__ __launch_bounds__(kBlock, (PPL >= 8 ? 2 : 4)) void dcc_obs_expand_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = xcd_swizzle(blockIdx.x, gridDim.x) * kWavesPerBlock + wid;   // p.E = number of states
    const int N = p.N, M = p.M;
    const int L = N * (4 + 2 * (N - 1) + 5 * M);
    double2* s_poi = reinterpret_cast<double2*>(smem);
    const int per_wave = N * 32 + kStageC * 4;
    unsigned char* wbase = smem + ((M * 16 + 15) & ~15) + wid * per_wave;
    double2* apos = reinterpret_cast<double2*>(wbase);
    double2* avel = apos + N;
    float* stg = reinterpret_cast<float*>(avel + N);
    for (int j = threadIdx.x; j < M; j += kBlock) s_poi[j] = p.poi[j];
    __syncthreads();
    if (n >= p.E) return;
    PoiLane<PPL> poi;
    poi.init(s_poi, lane, M);
    if (lane < N) { apos[lane] = p.st_pos[(size_t)n * N + lane]; avel[lane] = p.st_vel[(size_t)n * N + lane]; }
    float en[PPL];
    unsigned dmask = 0;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int j = q * 64 + lane;
        en[q] = 0.f;
        if (j < M) {
            en[q] = p.st_energy[(size_t)n * M + j];
            if (p.st_done[(size_t)n * M + j]) dmask |= 1u << q;
        }
    }
    wave_fence();
    Stager st;
    st.stg = stg; st.w0 = 0; st.vec = p.vec_ok; st.gout = p.obs + (size_t)n * (size_t)L;
    produce_obs<PPL, false, 0, 0>(p, st, reinterpret_cast<const double*>(apos), en, dmask, poi, lane);
}

// ---- kernel 4: compact policy-input features from compact state (dcc_obs_features) -------------------------------
// The observation row of agent i is  [vel_i, pos_i, (pos_a - pos_i) a!=i | (poi_j - pos_i, energy_j, m_energy, done_j) j].
// Its PoI block differs between the agents of an env only by the translation pos_i, so the first Linear layer of a
// policy can be evaluated from (a) the 4 + 2(N-1) "head" columns per agent, (b) the 2M per-env PoI features
// (energy, done) and (c) the LayerNorm moments of the full row -- ~1/37 of the row at 8 UAV x 64 PoI -- without the
// row ever being written (algos/algo_utils/structured.py has the algebra).  One wavefront per state.
struct FeatParams {
    const double2* pos; const double2* vel; const float* energy; const uint8_t* done; const double2* poi;
    float* head; float* poi_feat; double* stats; double* cstats;
    float* xa; float* xc;       // per-env GEMM inputs [energy | done | 1 | 0..] and [head_0..head_{N-1} | energy | done | 1 | 0..]
    int ka, kc;                 // their row lengths (multiples of 8 floats)
    int n, N, M; float m_energy;
    int stage;                  // 1: the N*HD head values of a state are assembled in LDS and stored as float4 runs
};

// The features of ONE state held in a wave's registers: lane i < N has UAV i (mp, mv), lane l has PoIs {l, l + 64, ...}
// (coordinates qx / qy, energy en, done bit t of dmask).  The PoI coordinates / energies / done flags are re-used by every
// agent row and both moment passes, so they stay in registers (read from global memory inside the agent loop they cost a
// dependent L2 round trip per agent and pass: 16 at 8 UAVs, which made the stand-alone kernel 30 us per 4096 states).
// Shared by dcc_obs_features_kernel (state from HBM) and by the env kernels (dcc_env_step_features: the state the step just
// produced, never re-read).  hrow: N*HD floats of per-wave LDS when p.stage, else NULL.
template <int PPL>
__device__ __forceinline__ void produce_features(const FeatParams& p, const size_t n, const int lane, float* hrow,
                                                 const double2 mp, const double2 mv, const double (&qx)[PPL],
                                                 const double (&qy)[PPL], const float (&en)[PPL], const unsigned dmask) {
    const int N = p.N, M = p.M, HD = 4 + 2 * (N - 1), D = HD + 5 * M;
    // The (energy, m_energy, done) columns of a row are the same for every agent of the env: their sum and sum of squares
    // are formed ONCE per state (se, qe), and a row's moments are  mean = (sum of its own columns + se) / D,
    // m2 = sum (own - mean)^2 + (qe - 2 mean se + 3M mean^2)  -- float64 throughout, so expanding the square of the shared
    // part costs nothing measurable in accuracy (values are O(1), checked to 1e-11 against the two-pass form), and the agent
    // loop touches two columns per PoI instead of five.
    double se = 0.0, qe = 0.0;
    const double me = (double)p.m_energy;
#pragma unroll
    for (int t = 0; t < PPL; ++t) {
        const int j = t * 64 + lane;
        if (j < M) {
            const float e = en[t];
            const float d = ((dmask >> t) & 1u) ? 1.f : 0.f;
            se += ((double)e + me) + (double)d;
            qe += ((double)e * (double)e + me * me) + (double)d * (double)d;
            if (p.poi_feat) { float* f = p.poi_feat + n * 2 * M; f[j] = e; f[M + j] = d; }
            if (p.xa) { float* f = p.xa + n * p.ka; f[j] = e; f[M + j] = d; }
            if (p.xc) { float* f = p.xc + n * p.kc + N * HD; f[j] = e; f[M + j] = d; }
        }
    }
    if (p.xa) { const int c = 2 * M + lane; if (c < p.ka) p.xa[n * p.ka + c] = lane == 0 ? 1.f : 0.f; }          // 1 | zero padding (< 8)
    if (p.xc) { const int c = N * HD + 2 * M + lane; if (c < p.kc) p.xc[n * p.kc + c] = lane == 0 ? 1.f : 0.f; }
    const bool want_stats = p.stats || p.cstats;
    if (want_stats) { se = wave_sum_f64_dpp(se); qe = wave_sum_f64_dpp(qe); }
    double my_mean = 0.0, my_m2 = 0.0;     // lane i keeps the moments of agent row i (for the pooled critic moments)
    for (int i = 0; i < N; ++i) {
        const double px = readlane_f64(mp.x, i), py = readlane_f64(mp.y, i);
        const float rx = (float)(mp.x - px), ry = (float)(mp.y - py);   // what the obs row holds for agent `lane`
        const bool other = lane < N && lane != i;
        const float v0 = (float)mv.x, v1 = (float)mv.y, p0 = (float)mp.x, p1 = (float)mp.y;
        if (hrow) {
            float* h = hrow + i * HD;
            if (lane == i) { h[0] = v0; h[1] = v1; h[2] = p0; h[3] = p1; }
            if (other) { const int k = lane < i ? lane : lane - 1; h[4 + 2 * k] = rx; h[5 + 2 * k] = ry; }
        } else if (p.head || p.xc) {
            float* h = p.head ? p.head + (n * N + i) * HD : nullptr;
            float* x = p.xc ? p.xc + n * p.kc + i * HD : nullptr;
            if (lane == i) {
                if (h) { h[0] = v0; h[1] = v1; h[2] = p0; h[3] = p1; }
                if (x) { x[0] = v0; x[1] = v1; x[2] = p0; x[3] = p1; }
            }
            if (other) {
                const int k = lane < i ? lane : lane - 1;
                if (h) { h[4 + 2 * k] = rx; h[5 + 2 * k] = ry; }
                if (x) { x[4 + 2 * k] = rx; x[5 + 2 * k] = ry; }
            }
        }
        if (!want_stats) continue;
        // moments of the D float32 values of the row, accumulated in float64: two passes over the row's own columns
        double s = 0.0;
        if (lane == i) s = ((double)v0 + (double)v1) + ((double)p0 + (double)p1);
        if (other) s = (double)rx + (double)ry;
        double fx[PPL], fy[PPL];     // the float32 values the row holds, widened once
#pragma unroll
        for (int t = 0; t < PPL; ++t) {
            fx[t] = (double)(float)(qx[t] - px); fy[t] = (double)(float)(qy[t] - py);
            if (t * 64 + lane < M) s += fx[t] + fy[t];
        }
        const double mean = (wave_sum_f64_dpp(s) + se) / (double)D;
        double m2 = 0.0;
        auto sq = [mean](double x) { const double d = x - mean; return d * d; };
        if (lane == i) m2 = sq((double)v0) + sq((double)v1) + sq((double)p0) + sq((double)p1);
        if (other) m2 = sq((double)rx) + sq((double)ry);
#pragma unroll
        for (int t = 0; t < PPL; ++t)
            if (t * 64 + lane < M) m2 += sq(fx[t]) + sq(fy[t]);
        m2 = wave_sum_f64_dpp(m2) + ((qe - 2.0 * mean * se) + (double)(3 * M) * mean * mean);
        if (lane == 0 && p.stats) { p.stats[(n * N + i) * 2] = mean; p.stats[(n * N + i) * 2 + 1] = m2; }
        if (lane == i) { my_mean = mean; my_m2 = m2; }
    }
    if (hrow) {     // N*HD = 2N(N+1) floats: a multiple of 4, rows of head / xc start 16-byte aligned
        wave_fence();
        const int n4 = (N * HD) >> 2;
        const float4* src = reinterpret_cast<const float4*>(hrow);
        float4* dh = p.head ? reinterpret_cast<float4*>(p.head + n * N * HD) : nullptr;
        float4* dx = p.xc ? reinterpret_cast<float4*>(p.xc + n * p.kc) : nullptr;
        for (int v = lane; v < n4; v += 64) {
            const float4 t = src[v];
            if (dh) dh[v] = t;
            if (dx) dx[v] = t;
        }
        wave_fence();
    }
    if (p.cstats) {
        // moments of the centralised row = concatenation of the N agent rows (equal widths D): pooled mean, and
        // sum of squared deviations by the parallel-variance identity
        const double mean_e = wave_sum_f64_dpp(lane < N ? my_mean : 0.0) / (double)N;
        const double dm = my_mean - mean_e;
        const double m2_e = wave_sum_f64_dpp(lane < N ? my_m2 + (double)D * dm * dm : 0.0);
        if (lane == 0) { p.cstats[n * 2] = mean_e; p.cstats[n * 2 + 1] = m2_e; }
    }
}

template <int PPL>
__global__ __launch_bounds__(kBlock) void dcc_obs_features_kernel(const FeatParams p) {
    extern __shared__ __attribute__((aligned(16))) float feat_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * kWavesPerBlock + wid;
    if (n >= p.n) return;
    const int N = p.N, M = p.M, HD = 4 + 2 * (N - 1);
    // The head values of agent i come from different lanes (own vel / pos from lane i, the relative position of UAV a from
    // lane a): written straight to HBM they are 4-byte stores at an 8-byte stride, 4 store instructions per agent that
    // each touch a cache line partially.  Staged per wave in LDS (N*HD floats) they leave as two float4 runs per state.
    float* hrow = p.stage ? feat_lds + (size_t)wid * N * HD : nullptr;
    double2 mp = make_double2(0.0, 0.0), mv = mp;
    if (lane < N) { mp = p.pos[(size_t)n * N + lane]; mv = p.vel[(size_t)n * N + lane]; }
    double qx[PPL], qy[PPL];
    float en[PPL];
    unsigned dmask = 0;
#pragma unroll
    for (int t = 0; t < PPL; ++t) {
        const int j = t * 64 + lane;
        qx[t] = qy[t] = 0.0; en[t] = 0.f;
        if (j < M) {
            const double2 q = p.poi[j];
            qx[t] = q.x; qy[t] = q.y;
            en[t] = p.energy[(size_t)n * M + j];
            if (p.done[(size_t)n * M + j]) dmask |= 1u << t;
        }
    }
    produce_features<PPL>(p, (size_t)n, lane, hrow, mp, mv, qx, qy, en, dmask);
}

// ---- kernel 6: env step(s) + the policy-input features of the state each step leaves (dcc_env_step_features) -------------
// The policy-driven rollout alternates a K = 1 env launch and the policy forward, and with structured first layers the
// forward reads the FEATURES of the new state, not its rows.  Produced by a second launch they cost a launch, a round trip of
// the state through HBM / L2 and a latency-bound kernel of their own (23 us per step at c3, 131 us at the c5 shard); here the
// wave that stepped the env derives them from the registers it still holds.  float32 actions only (the rollout's dtype).
template <int PPL, bool FORCE, int NC, int MC>
__global__ __launch_bounds__(kBlock, (PPL >= 8 ? 2 : (FORCE ? (PPL >= 4 ? 2 : 3) : 4))) void dcc_env_feat_kernel(const KParams p, const FeatParams f) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int env = xcd_swizzle(blockIdx.x, gridDim.x) * kWavesPerBlock + wid;
    constexpr bool SPEC = NC > 0;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    double2* s_poi = reinterpret_cast<double2*>(smem);
    const int per_wave = N * 32 + kStageC * 4;
    unsigned char* wbase = smem + ((M * 16 + 15) & ~15) + wid * per_wave;
    double2* apos = reinterpret_cast<double2*>(wbase);
    double2* avel = apos + N;
    float* stg = reinterpret_cast<float*>(avel + N);
    if (!PoiLane<PPL>::REG) {
        for (int j = threadIdx.x; j < M; j += kBlock) s_poi[j] = p.poi[j];
        __syncthreads();
    }
    if (env >= p.E) return;
    EnvRegs<PPL> r;
    PoiLane<PPL> poi;
    ActFetch<act_rf<PPL, FORCE>()> af;
    init_act(af);
    prefetch_actions<PPL, 0, FORCE, NC>(p, env, lane, af);
    poi.init(PoiLane<PPL>::REG ? p.poi : s_poi, lane, M);
    load_env_state<PPL>(p, env, lane, N, M, r);
    if (lane < N) { apos[lane] = make_double2(r.px, r.py); avel[lane] = make_double2(r.vx, r.vy); }
    wave_fence();
    for (int k = 0; k < p.K; ++k) {
        env_physics_step<PPL, 0, FORCE, NC, MC>(p, env, k, lane, r, af, poi, apos, apos, avel);
        const size_t ko = (size_t)k * p.E + env;
        if (p.st_pos || p.st_vel || p.st_energy || p.st_done) write_step_state<PPL>(p, ko, lane, N, M, r);
        double qx[PPL], qy[PPL];
#pragma unroll
        for (int q = 0; q < PPL; ++q) { const double2 pj = poi.get(q); qx[q] = pj.x; qy[q] = pj.y; }
        double2 mp = make_double2(0.0, 0.0), mv = mp;
        if (lane < N) { mp = make_double2(r.px, r.py); mv = make_double2(r.vx, r.vy); }
        produce_features<PPL>(f, ko, lane, f.stage ? stg : nullptr, mp, mv, qx, qy, r.en, r.dmask);
    }
    store_env_state<PPL>(p, env, lane, N, M, r);
}

// ---- kernel 2: role-specialised -- a PHYSICS wave and an OBSERVATION wave per workgroup --------------------
// Measured on MI355X (tools/overlap_probe*.hip): when every wave alternates compute and its 10.8 KB of obs
// stores, the compute overlaps only ~2/3 with the chip-wide store stream; waves that do nothing but stream
// stores keep HBM saturated while other waves compute in their shadow.  Here a workgroup is two waves serving
// two envs: wave 0 runs the physics of both envs and hands the post-step state (positions, velocities, PoI
// energies, done mask: ~0.5 KB) to wave 1 through a ring of LDS slots (KParams::roles_slots); wave 1 expands it into the
// observation rows and streams them to HBM.  ready/consumed counters in LDS (workgroup-scope release/acquire)
// let the physics wave run up to two steps ahead, so the observation wave always has a backlog.
#ifndef DCC_ROLES_OBS
#define DCC_ROLES_OBS 1            // observation waves per workgroup: 1 = one wave streams both envs as ONE stream; 2 = one wave per env
#endif
constexpr int kRolesObs = DCC_ROLES_OBS;
constexpr int kRolesBlock = 64 * (1 + kRolesObs);   // wave 0 = physics, waves 1.. = observation
struct Handoff {  // one env, one slot; laid out in LDS as: apos[N] | avel[N] | en[64] | dmask(u64) | pad | StepRec
    double2* apos; double2* avel; float* en; unsigned long long* dmask; StepRec* rec;
};
__device__ __forceinline__ Handoff handoff_at(unsigned char* base, int N) {
    Handoff h;
    h.apos = reinterpret_cast<double2*>(base);
    h.avel = h.apos + N;
    h.en = reinterpret_cast<float*>(h.avel + N);
    h.dmask = reinterpret_cast<unsigned long long*>(h.en + 64);
    h.rec = reinterpret_cast<StepRec*>(h.en + 64 + 4);
    return h;
}
__device__ __forceinline__ int handoff_bytes(int N) { return N * 32 + 64 * 4 + 16 + (int)sizeof(StepRec); }

__device__ __forceinline__ void spin_until_ge(unsigned* flag, unsigned v) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(2);
}
__device__ __forceinline__ void publish(unsigned* flag, unsigned v, int lane) {
    wave_fence();
    if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Up to round 5 every env-step of the observation wave held an accidental `global_load` + `s_waitcnt vmcnt(0)`: write_step_outputs
// selected its output pointers with `lane == 0 ? p.done : ...` on the struct MEMBERS, a conditional lvalue, i.e. a per-lane address
// into the kernel-argument segment and a vector load from it.  The load's round trip cost 12 % at 256 envs; its full drain of
// the store queue, on the other hand, GAINED 1.5 % at 4096 envs -- which is how the pacing of KParams::obs_drain was found.
__device__ __forceinline__ void obs_store_queue_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int ACT, bool FORCE, int NC, int MC>
__global__ __launch_bounds__(2 * kRolesBlock, (FORCE ? 3 : 4)) void dcc_env_roles_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
    constexpr int PPL = 1;
    constexpr bool SPEC = NC > 0;
    const int lane = threadIdx.x & 63;
    // A workgroup is one (physics, observation) pair of waves, or TWO independent pairs side by side (p.roles_pairs = 2): the
    // hardware spreads the four waves of such a workgroup over the four SIMDs of its CU, whereas two 2-wave workgroups on one CU
    // land on three SIMDs (tools/hwid_probe.hip: the second workgroup's physics wave shares a SIMD with the first one's
    // observation wave and one SIMD stays empty) -- what a 512-workgroup launch looks like on 256 CUs.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = (kRolesObs == 1 && p.roles_pairs == 2) ? (wave >> 1) : 0;
    const int role = wave - pair * (1 + kRolesObs);                        // 0 = physics, 1 = observation
    const int tid = threadIdx.x - pair * kRolesBlock;                      // thread index inside the pair
    unsigned char* smem = smem_wg + (size_t)pair * p.roles_lds;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    const int L = N * (4 + 2 * (N - 1) + 5 * M);
    const int epw = p.roles_envs;                                        // envs of this pair of waves: 2 (adjacent), or 1
    const int kSlots = p.roles_slots;
    const int env_base = (xcd_swizzle(blockIdx.x, gridDim.x) * p.roles_pairs + pair) * epw;

    // LDS: PoI table | hand-off [env 0..1][slot 0..1] | flags ready[2], consumed[2] | staging window
    double2* s_poi = reinterpret_cast<double2*>(smem);
    const int hb = handoff_bytes(N);
    unsigned char* hbase = smem + ((M * 16 + 15) & ~15);
    unsigned* flags = reinterpret_cast<unsigned*>(hbase + 2 * kSlots * hb);
    float* stg = reinterpret_cast<float*>(hbase + 2 * kSlots * hb + 16) + (role > 1 ? (role - 1) * kStageC : 0);

    for (int j = tid; j < M; j += kRolesBlock) s_poi[j] = p.poi[j];
    if (tid < 4) flags[tid] = 0u;
    __syncthreads();

    PoiLane<PPL> poi;
    poi.init(s_poi, lane, M);

    if (role == 0) {
        // ---------------- physics wave: both envs of the workgroup, up to two steps ahead -------------------
#ifdef DCC_ROLES_PHYS_PRIO
        __builtin_amdgcn_s_setprio(DCC_ROLES_PHYS_PRIO);
#endif
        EnvRegs<PPL> r[2];
        ActFetch<act_rf<PPL, FORCE>()> af[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int env = env_base + s;
            init_act(af[s]);
            if (env < p.E && s < epw) {
                load_env_state<PPL>(p, env, lane, N, M, r[s]);
                // the "previous" slot (1) holds the pre-move positions of step 0
                Handoff h = handoff_at(hbase + (kSlots * s + kSlots - 1) * hb, N);
                if (lane < N) { h.apos[lane] = make_double2(r[s].px, r[s].py); h.avel[lane] = make_double2(r[s].vx, r[s].vy); }
            }
        }
        wave_fence();
        for (int k = 0; k < p.K; ++k) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int env = env_base + s;
                if (env >= p.E || s >= epw) continue;
                const int slot = k & (kSlots - 1);
                Handoff out = handoff_at(hbase + (kSlots * s + slot) * hb, N);
                Handoff in = handoff_at(hbase + (kSlots * s + ((slot + kSlots - 1) & (kSlots - 1))) * hb, N);
                // slot `slot` was published at step k - kSlots: wait until the observation wave has read it
                if (k >= kSlots) spin_until_ge(&flags[2 + s], (unsigned)(k - kSlots + 1));
                if (p.mode == 0) {
                    env_physics_step<PPL, ACT, FORCE, NC, MC>(p, env, k, lane, r[s], af[s], poi, in.apos, out.apos, out.avel, out.rec);
                } else if (lane < N) {
                    out.apos[lane] = make_double2(r[s].px, r[s].py); out.avel[lane] = make_double2(r[s].vx, r[s].vy);
                }
                out.en[lane] = r[s].en[0];
                const unsigned long long dm = __ballot((r[s].dmask & 1u) != 0u);
                if (lane == 0) *out.dmask = dm;
                publish(&flags[s], (unsigned)(k + 1), lane);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (env_base + s < p.E && s < epw) store_env_state<PPL>(p, env_base + s, lane, N, M, r[s]);
    } else {
        // ---------------- observation wave: expand + stream, one env-step at a time ---------------------------
        // It is the wave that feeds HBM: it outranks the physics waves on its SIMD (they have slack).
#ifndef DCC_ROLES_OBS_PRIO
#define DCC_ROLES_OBS_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(DCC_ROLES_OBS_PRIO);
        unsigned assign_w0 = 0;   // env 0's packed assignment row, held until env 1's is ready
        Stager st;
        st.stg = stg; st.w0 = 0; st.vec = SPEC ? 1 : p.vec_ok; st.gout = p.obs; st.drain = p.obs_drain;
        for (int k = 0; k < p.K; ++k) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int env = env_base + s;
                if (env >= p.E || s >= epw) continue;
                if (kRolesObs == 2 && s != role - 1) continue;       // one observation wave per env
                spin_until_ge(&flags[s], (unsigned)(k + 1));
                if (p.obs_drain == 0) obs_store_queue_drain();
                Handoff h = handoff_at(hbase + (kSlots * s + (k & (kSlots - 1))) * hb, N);
                float en[1];
                en[0] = h.en[lane];
                const unsigned dmask = (unsigned)((*h.dmask >> lane) & 1ULL);
                if (p.mode == 0) {   // the physics wave's per-step outputs leave through this wave's store stream
                    const unsigned fl = h.rec->flags;
                    int am[1];
                    am[0] = (int)h.rec->assign[lane];
                    // The assignment rows of the workgroup's two envs are adjacent in HBM (2 x M bytes): with one
                    // observation wave they are written by ONE store (a full 128-byte line at M = 64) instead of
                    // two half-line stores a microsecond apart.
                    const bool pair = kRolesObs == 1 && (M & 3) == 0 && p.assign != nullptr;
                    if (pair) {
                        const int nd = M >> 2;   // dwords per row (<= 16)
                        const int t1 = am[0] | (__builtin_amdgcn_update_dpp(0, am[0], 0xF9, 0xF, 0xF, false) << 8);
                        const int t2 = t1 | (__builtin_amdgcn_update_dpp(0, t1, 0xEE, 0xF, 0xF, false) << 16);
                        const int rel = lane < nd ? lane : lane - nd;
                        const unsigned w = (unsigned)__shfl(t2, 4 * (rel < nd ? rel : nd - 1), 64);   // dword `rel` of this row
                        unsigned* row0 = reinterpret_cast<unsigned*>(p.assign + ((size_t)k * p.E + env_base) * M);
                        const bool last_of_pair = (s == 1) || epw == 1 || (env_base + 1 >= p.E);
                        if (s == 0) assign_w0 = w;
                        if (last_of_pair) {
                            const int n_rows = (s == 1) ? 2 : 1;
                            if (lane < n_rows * nd) row0[lane] = (s == 1 && lane >= nd) ? w : assign_w0;
                        }
                    }
                    write_step_outputs<1>(p, (size_t)k * p.E + env, M, lane, h.rec->R, h.rec->cov, (fl & 1u) != 0u,
                                          (fl & 2u) != 0u, (fl & 4u) != 0u, am, pair);
                }
                if (p.st_pos || p.st_vel || p.st_energy || p.st_done) {   // compact post-step (post-reset) state, from the slot
                    const size_t ko = (size_t)k * p.E + env;
                    if (p.st_pos && lane < N) p.st_pos[ko * N + lane] = h.apos[lane];
                    if (p.st_vel && lane < N) p.st_vel[ko * N + lane] = h.avel[lane];
                    if (lane < M) {
                        if (p.st_energy) p.st_energy[ko * M + lane] = en[0];
                        if (p.st_done) p.st_done[ko * M + lane] = (uint8_t)dmask;
                    }
                }
                if (kRolesObs == 2) {   // own block, own stream
                    st.w0 = 0; st.gout = p.obs + ((size_t)k * p.E + env) * (size_t)L;
                    produce_obs<PPL, FORCE, NC, MC>(p, st, reinterpret_cast<const double*>(h.apos), en, dmask, poi, lane, 0, true);
                } else {
                    // both envs of the workgroup: one output stream of 2 L floats
                    if (s == 0) { st.w0 = 0; st.gout = p.obs + ((size_t)k * p.E + env_base) * (size_t)L; }
                    produce_obs<PPL, FORCE, NC, MC>(p, st, reinterpret_cast<const double*>(h.apos), en, dmask, poi, lane,
                                                    s * L, (s == 1) || epw == 1 || (env_base + 1 >= p.E));
                }
                publish(&flags[2 + s], (unsigned)(k + 1), lane);
            }
        }
    }
}

// ---- kernel 5: split -- one env per workgroup: a physics wave + kSplitObs observation waves (M > 64) ------------
// The BASELINE shards with many PoIs per env are small in env count (c4: 1024 envs per GPU, c5: 2048), so the fused
// kernel runs one wave per SIMD: nothing hides the latency of its 84 KB (c4) of row stores per step, and physics and
// stores alternate instead of overlapping.  Here the rows of an env are produced by kSplitObs waves in parallel (each a
// contiguous, float4-aligned range of agent rows with its own staging window) while the physics wave is already on
// the next step; the hand-off is the double-buffered LDS slot of the role-specialised kernel, sized for any M.
#ifndef DCC_SPLIT_OBS
#define DCC_SPLIT_OBS 2     // observation waves per env.  c4 shard, placement-probed buffers, same box: 1 -> 2.27-2.30 ms per 150-step
#endif                      // launch, 2 -> 2.25-2.27, 3 -> 2.29-2.31, 4 -> 2.79 (a fifth wave per workgroup no longer fits one round)
constexpr int kSplitObs = DCC_SPLIT_OBS;
constexpr int kSplitBlock = 64 * (1 + kSplitObs);
struct SplitSlot { double2* apos; double2* avel; float* en; unsigned* dm; };
__device__ __forceinline__ int split_slot_bytes(int N, int ppl) { return N * 32 + ppl * 256 + 256; }
__device__ __forceinline__ SplitSlot split_slot_at(unsigned char* base, int N, int ppl) {
    SplitSlot h;
    h.apos = reinterpret_cast<double2*>(base);
    h.avel = h.apos + N;                                   // contiguous with apos: produce_obs indexes both through one pointer
    h.en = reinterpret_cast<float*>(h.avel + N);
    h.dm = reinterpret_cast<unsigned*>(h.en + ppl * 64);
    return h;
}

// produce_obs() restricted to the agent rows [i0, i1): the stream of `st` starts at row i0.
template <int PPL, bool FORCE, int NC, int MC>
__device__ __forceinline__ void produce_obs_rows(const KParams& p, Stager& st, const double* apv, const float (&en)[PPL],
                                                 const unsigned dmask, const PoiLane<PPL>& poi, const int lane,
                                                 const int i0, const int i1) {
    constexpr bool SPEC = NC > 0;
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    const int H = 4 + 2 * (N - 1), D = H + 5 * M;
    const double2* apos = reinterpret_cast<const double2*>(apv);
    for (int i = i0; i < i1; ++i) {
        const double2 xi = apos[i];
        const int rb = (i - i0) * D;
        for (int f0 = 0; f0 < H; f0 += 64) {
            const int len = (H - f0) < 64 ? (H - f0) : 64;
            float* dst = st.reserve(rb + f0, len, lane);
            const int f = f0 + lane;
            if (f < H) {
                const int c = f & 1;
                const int kk = (f - 4) >> 1;
                const bool rel = f >= 4;
                const int src = rel ? (kk + (kk >= i ? 1 : 0)) : (f < 2 ? N + i : i);
                const double val = apv[2 * src + c];
                const double sub = rel ? (c ? xi.y : xi.x) : 0.0;
                dst[lane] = (float)(val - sub);
            }
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (q * 64 < M) {
                const int cntj = (M - q * 64) < 64 ? (M - q * 64) : 64;
                float* dst = st.reserve(rb + H + q * kTileFloats, 5 * cntj, lane);
                if (lane < cntj) {
                    float* d5 = dst + 5 * lane;
                    const double2 pj = poi.get(q);
                    d5[0] = (float)(pj.x - xi.x);
                    d5[1] = (float)(pj.y - xi.y);
                    d5[2] = en[q];
                    d5[3] = p.m_energy_f;
                    d5[4] = ((dmask >> q) & 1u) ? 1.f : 0.f;
                }
            }
        }
    }
    st.flush((i1 - i0) * D, lane);
}

template <int PPL, int ACT, bool FORCE, int NC, int MC>
__global__ __launch_bounds__(kSplitBlock, (PPL >= 8 || FORCE ? 1 : 2)) void dcc_env_split_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool SPEC = NC > 0;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 = physics, 1.. = observation waves
    const int N = SPEC ? NC : p.N, M = SPEC ? MC : p.M;
    const int D = 4 + 2 * (N - 1) + 5 * M, L = N * D;
    const int env = xcd_swizzle(blockIdx.x, gridDim.x);                  // grid = E: one env per workgroup

    // LDS: PoI table | slot 0, slot 1 | flags: ready, consumed[kSplitObs] | kSplitObs staging windows
    double2* s_poi = reinterpret_cast<double2*>(smem);
    const int sb = split_slot_bytes(N, PPL);
    unsigned char* hbase = smem + ((M * 16 + 15) & ~15);
    unsigned* flags = reinterpret_cast<unsigned*>(hbase + 2 * sb);                 // ready, consumed[kSplitObs] (<= 8 words)
    float* stg_base = reinterpret_cast<float*>(hbase + 2 * sb + 32);

    for (int j = threadIdx.x; j < M; j += kSplitBlock) s_poi[j] = p.poi[j];
    if (threadIdx.x < 8) flags[threadIdx.x] = 0u;
    __syncthreads();
    PoiLane<PPL> poi;
    poi.init(s_poi, lane, M);

    if (role == 0) {
        // The physics wave is the producer every observation wave of the workgroup waits for: it outranks them (it blocks on
        // the slot flags after running two steps ahead, so it cannot hog the SIMD).  Same-box A/B at the c4 shard, well-placed
        // output buffer: 2.297 vs 2.372 ms per 150-step launch with the priorities the other way round (round 3).
#ifndef DCC_SPLIT_PHYS_PRIO
#define DCC_SPLIT_PHYS_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(DCC_SPLIT_PHYS_PRIO);
        EnvRegs<PPL> r;
        ActFetch<act_rf<PPL, FORCE>()> af;
        init_act(af);
        load_env_state<PPL>(p, env, lane, N, M, r);
        {   // the "previous" slot (1) holds the pre-move positions of step 0
            SplitSlot h = split_slot_at(hbase + sb, N, PPL);
            if (lane < N) { h.apos[lane] = make_double2(r.px, r.py); h.avel[lane] = make_double2(r.vx, r.vy); }
        }
        wave_fence();
        for (int k = 0; k < p.K; ++k) {
            const int slot = k & 1;
            SplitSlot out = split_slot_at(hbase + slot * sb, N, PPL);
            SplitSlot in = split_slot_at(hbase + (slot ^ 1) * sb, N, PPL);
            if (k >= 2) {   // slot `slot` was published at step k-2: every observation wave must have read it
#pragma unroll
                for (int w = 0; w < kSplitObs; ++w) spin_until_ge(&flags[1 + w], (unsigned)(k - 1));
            }
            if (p.mode == 0) {
                env_physics_step<PPL, ACT, FORCE, NC, MC>(p, env, k, lane, r, af, poi, in.apos, out.apos, out.avel);
            } else if (lane < N) {      // observation producer only (dcc_env_obs_write_probe): the reset state, K times
                out.apos[lane] = make_double2(r.px, r.py); out.avel[lane] = make_double2(r.vx, r.vy);
            }
            if (p.st_pos || p.st_vel || p.st_energy || p.st_done) write_step_state<PPL>(p, (size_t)k * p.E + env, lane, N, M, r);
#pragma unroll
            for (int q = 0; q < PPL; ++q) out.en[q * 64 + lane] = r.en[q];
            out.dm[lane] = r.dmask;
            publish(&flags[0], (unsigned)(k + 1), lane);
        }
        store_env_state<PPL>(p, env, lane, N, M, r);
    } else {
#ifndef DCC_SPLIT_OBS_PRIO
#define DCC_SPLIT_OBS_PRIO 1
#endif
        __builtin_amdgcn_s_setprio(DCC_SPLIT_OBS_PRIO);
        const int w = role - 1;
        const int vec = SPEC ? 1 : p.vec_ok;
        // contiguous row ranges whose first float index i0 * D is a multiple of 4 in float4 mode
        const int ra = vec ? ((D & 3) == 0 ? 1 : ((D & 1) == 0 ? 2 : 4)) : 1;
        const int i0 = ((w * N) / kSplitObs) / ra * ra;
        const int i1 = (w == kSplitObs - 1) ? N : (((w + 1) * N) / kSplitObs) / ra * ra;
        float* stg = stg_base + w * kStageC;
        for (int k = 0; k < p.K; ++k) {
            spin_until_ge(&flags[0], (unsigned)(k + 1));
            SplitSlot h = split_slot_at(hbase + (k & 1) * sb, N, PPL);
            float en[PPL];
#pragma unroll
            for (int q = 0; q < PPL; ++q) en[q] = h.en[q * 64 + lane];
            const unsigned dmask = h.dm[lane];
            if (i1 > i0) {
                Stager st;
                st.stg = stg; st.w0 = 0; st.vec = vec; st.drain = p.obs_drain;
                st.gout = p.obs + ((size_t)k * p.E + env) * (size_t)L + (size_t)i0 * D;
                produce_obs_rows<PPL, FORCE, NC, MC>(p, st, reinterpret_cast<const double*>(h.apos), en, dmask, poi, lane, i0, i1);
            }
            publish(&flags[1 + w], (unsigned)(k + 1), lane);
        }
    }
}

// ================================ host side =====================================================

thread_local std::string g_err;

int fail(int code, const std::string& msg) { return dcc_fail(code, msg); }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(DCC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    } while (0)

}  // namespace

// the one error slot of the library: dcc_env_*, dcc_obs_*, dcc_gae_*, the dcc_mlp.h and dcc_optim.h entry points all report here
int dcc_fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

struct dcc_env {
    dcc_env_cfg cfg;
    int device;
    int D, L, PPL;
    KParams base;
    double2* d_poi = nullptr;
    double2* d_pos = nullptr;
    double2* d_vel = nullptr;
    float* d_energy = nullptr;
    uint8_t* d_done = nullptr;
    size_t lds_bytes = 0, lds_bytes_roles = 0, lds_bytes_split = 0;
    bool no_spec = false, no_roles = false, force_roles = false, no_split = false, force_split = false;
    int roles_envs_forced = 0;  // DCC_ROLES_ENVS = 1 / 2 (tests, A/B); 0 = by batch size
    int roles1_max = 1600;      // batches up to this many envs run one env per (physics, observation) wave pair (DCC_ROLES1_MAX); measured
                                // crossover on MI355X between 1536 (one env per pair +7 %) and 1792 (two envs +4 %): profiles/r06/small_batch_shapes.txt
    int n_cus = 256;            // compute units of the device (launch-shape policy only)
    long fused_lds_pad = -1;    // DCC_FUSED_LDS_PAD (bytes of unused LDS per fused workgroup, A/B); < 0 = the residency policy of launch()
    size_t lds_two_per_cu = 0;  // LDS per workgroup above which at most two workgroups fit a CU (0: unknown -> no cap)
    int roles_slots = 2;        // hand-off slots per env of the role-specialised kernel (DCC_ROLES_SLOTS = 2 / 4 / 8: A/B)
    int roles_pairs_forced = 0; // DCC_ROLES_PAIRS = 1 / 2 (A/B); 0 = by batch size
    int obs_drain_forced = -2;  // DCC_OBS_DRAIN = -1 / 0 / 2 (A/B); -2 = the default (2)
    // create-time choice between the role-specialised and the fused kernel for obs-writing multi-step launches with one PoI
    // per lane (which of the two streams faster depends on the box: DESIGN.md 4.1); tune_us: measured us per step of each
    bool prefer_fused = false;
    int tuned = 0;              // 0: not measured (shape not eligible, disabled, or the measurement failed), 1: measured
    float tune_us[2] = {0.f, 0.f};   // [0] role-specialised, [1] fused
};

namespace {

typedef void (*kernel_fn)(const KParams);
constexpr int kSplitMaxEnvs = 1280;   // above this the fused kernel (two workgroups resident per CU) streams as fast or faster: 16 x 256, split vs fused at
                                      // 1024 / 1536 / 2048 / 3072 envs 0.718 / 0.632 / 0.724 / 0.674 vs 0.665 / 0.651 / 0.733 / 0.680 (round 6; it was 3072)

template <int ACT, bool FORCE>
kernel_fn pick_ppl(int ppl) {
    switch (ppl) {
        case 1: return dcc_env_kernel<1, ACT, FORCE, 0, 0>;
        case 2: return dcc_env_kernel<2, ACT, FORCE, 0, 0>;
        case 4: return dcc_env_kernel<4, ACT, FORCE, 0, 0>;
        case 8: return dcc_env_kernel<8, ACT, FORCE, 0, 0>;
        default: return dcc_env_kernel<16, ACT, FORCE, 0, 0>;
    }
}

// compile-time (N, M) specialisations: BASELINE configs c1 (4,16), shipped (4,20), c2/c3 (8,64), c4 (16,256)
template <int ACT, bool FORCE>
kernel_fn pick_spec(int n, int m) {
    if (n == 8 && m == 64) return dcc_env_kernel<1, ACT, FORCE, 8, 64>;
    if (n == 4 && m == 16) return dcc_env_kernel<1, ACT, FORCE, 4, 16>;
    if (n == 4 && m == 20) return dcc_env_kernel<1, ACT, FORCE, 4, 20>;
    if (n == 16 && m == 256) return dcc_env_kernel<4, ACT, FORCE, 16, 256>;
    // (32, 1024) -- c5 -- deliberately has NO compile-time instantiation: measured in round 5, `<16, ACT, true, 32, 1024>` needs 256
    // VGPRs + 16-22 spilled against the generic kernel's 226 / 0 and streams 4 % SLOWER at the 2048-env shard (0.652-0.664 vs
    // 0.690-0.691 of the HBM peak, three interleaved runs each; profiles/r05/c5_spec_ab.txt)
    return nullptr;
}

kernel_fn pick_kernel(int ppl, int act, bool force, int n, int m, bool allow_spec) {
    kernel_fn f = nullptr;
    if (allow_spec) {
        if (force) f = act == 0 ? pick_spec<0, true>(n, m) : act == 1 ? pick_spec<1, true>(n, m) : pick_spec<2, true>(n, m);
        else f = act == 0 ? pick_spec<0, false>(n, m) : act == 1 ? pick_spec<1, false>(n, m) : pick_spec<2, false>(n, m);
        if (f) return f;
    }
    if (force) {
        return act == 0 ? pick_ppl<0, true>(ppl) : act == 1 ? pick_ppl<1, true>(ppl) : pick_ppl<2, true>(ppl);
    }
    return act == 0 ? pick_ppl<0, false>(ppl) : act == 1 ? pick_ppl<1, false>(ppl) : pick_ppl<2, false>(ppl);
}

// role-specialised kernels (one PoI per lane only): generic size + the small BASELINE configs
template <int ACT, bool FORCE>
kernel_fn pick_roles(int n, int m, bool allow_spec) {
    if (allow_spec) {
        if (n == 8 && m == 64) return dcc_env_roles_kernel<ACT, FORCE, 8, 64>;
        if (n == 4 && m == 16) return dcc_env_roles_kernel<ACT, FORCE, 4, 16>;
        if (n == 4 && m == 20) return dcc_env_roles_kernel<ACT, FORCE, 4, 20>;
    }
    return dcc_env_roles_kernel<ACT, FORCE, 0, 0>;
}

kernel_fn pick_roles_kernel(int act, bool force, int n, int m, bool allow_spec) {
    if (force) return act == 0 ? pick_roles<0, true>(n, m, allow_spec) : act == 1 ? pick_roles<1, true>(n, m, allow_spec)
                                                                                  : pick_roles<2, true>(n, m, allow_spec);
    return act == 0 ? pick_roles<0, false>(n, m, allow_spec) : act == 1 ? pick_roles<1, false>(n, m, allow_spec)
                                                                          : pick_roles<2, false>(n, m, allow_spec);
}

// split kernels (several PoIs per lane): float32 / in-kernel actions, generic sizes + BASELINE config c4
template <int ACT, bool FORCE>
kernel_fn pick_split(int ppl, int n, int m, bool allow_spec) {
    if (allow_spec && n == 16 && m == 256) return dcc_env_split_kernel<4, ACT, FORCE, 16, 256>;
    switch (ppl) {
        case 2: return dcc_env_split_kernel<2, ACT, FORCE, 0, 0>;
        case 4: return dcc_env_split_kernel<4, ACT, FORCE, 0, 0>;
        case 8: return dcc_env_split_kernel<8, ACT, FORCE, 0, 0>;
        default: return dcc_env_split_kernel<16, ACT, FORCE, 0, 0>;
    }
}

kernel_fn pick_split_kernel(int ppl, int act, bool force, int n, int m, bool allow_spec) {
    if (force) return act == 0 ? pick_split<0, true>(ppl, n, m, allow_spec) : pick_split<2, true>(ppl, n, m, allow_spec);
    return act == 0 ? pick_split<0, false>(ppl, n, m, allow_spec) : pick_split<2, false>(ppl, n, m, allow_spec);
}

typedef void (*feat_kernel_fn)(const KParams, const FeatParams);

template <bool FORCE>
feat_kernel_fn pick_feat_kernel(int ppl, int n, int m, bool allow_spec) {
    if (allow_spec) {
        if (n == 8 && m == 64) return dcc_env_feat_kernel<1, FORCE, 8, 64>;
        if (n == 4 && m == 16) return dcc_env_feat_kernel<1, FORCE, 4, 16>;
        if (n == 4 && m == 20) return dcc_env_feat_kernel<1, FORCE, 4, 20>;
        if (n == 16 && m == 256) return dcc_env_feat_kernel<4, FORCE, 16, 256>;
    }
    switch (ppl) {
        case 1: return dcc_env_feat_kernel<1, FORCE, 0, 0>;
        case 2: return dcc_env_feat_kernel<2, FORCE, 0, 0>;
        case 4: return dcc_env_feat_kernel<4, FORCE, 0, 0>;
        case 8: return dcc_env_feat_kernel<8, FORCE, 0, 0>;
        default: return dcc_env_feat_kernel<16, FORCE, 0, 0>;
    }
}

// act: 0 = f32 actions, 1 = f64 actions, 2 = in-kernel generator
int launch(dcc_env* env, KParams& p, int act, void* stream) {
    // specialised kernels assume float4-aligned obs rows; DCC_NO_SPEC=1 forces the generic kernels (tests)
    const bool allow_spec = (p.obs == nullptr || p.vec_ok) && !env->no_spec;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    p.obs_drain = (env->obs_drain_forced > -2) ? env->obs_drain_forced : 2;
    // observation-producing launches with one PoI per lane use the role-specialised kernel
    // (DCC_NO_ROLES=1 forces the fused kernel: tests, A/B)
    // and only for fused multi-step launches: with K = 1 there is nothing to pipeline and the hand-off only adds
    // latency (13.4 vs 14.3 us per single-step launch); DCC_FORCE_ROLES=1 overrides (tests)
    if (p.obs != nullptr && env->PPL == 1 && !env->no_roles && (p.K >= 2 || env->force_roles) &&
        !(env->prefer_fused && !env->force_roles)) {
        kernel_fn fn = pick_roles_kernel(act, p.use_force != 0, p.N, p.M, allow_spec);
        // Small batches are latency-bound: with two envs per workgroup a 512-env launch is 256 workgroups whose physics wave walks
        // two envs per step; one env per workgroup fills every CU twice over and halves the dependent chain of a step.
        p.roles_envs = (env->roles_envs_forced > 0) ? env->roles_envs_forced : (p.E <= env->roles1_max ? 1 : 2);
        const int n_pairs = (p.E + p.roles_envs - 1) / p.roles_envs;
        // Two wave pairs per workgroup where the one-pair form would put 2-3 workgroups on a CU: the dispatcher spreads a 4-wave
        // workgroup over the four SIMDs, but packs two 2-wave workgroups onto three of them (tools/hwid_probe.hip) -- 512 envs
        // 1.78 -> 1.62 us per step; at <= one workgroup per CU and from four per CU on the one-pair form is as good or better.
        const bool two_pairs = p.roles_envs == 1 && n_pairs > env->n_cus && n_pairs < 4 * env->n_cus;
        p.roles_pairs = (kRolesObs == 1) ? ((env->roles_pairs_forced > 0) ? env->roles_pairs_forced : (two_pairs ? 2 : 1)) : 1;
        p.roles_lds = (int)env->lds_bytes_roles;
        p.roles_slots = env->roles_slots;
        const int grid = (n_pairs + p.roles_pairs - 1) / p.roles_pairs;
        hipLaunchKernelGGL(fn, dim3(grid), dim3(kRolesBlock * p.roles_pairs), env->lds_bytes_roles * p.roles_pairs, s, p);
        HIP_TRY(hipGetLastError());
        return DCC_OK;
    }
    // several PoIs per lane and few envs (the c4 / c5 shards): one env per workgroup, rows produced by kSplitObs waves
    // while the physics wave runs ahead.  With many envs the fused kernel already fills the chip (c4 x 8192: 0.82 of
    // peak) and stays the choice, and so it does for 16 PoIs per lane with the pull force on (c5: 398 vs 245 us/step --
    // the split form is not resident in one round there: its LDS admits 4 of the 8 workgroups a CU needs at 2048 envs, and the
    // 226 VGPRs of that physics allow one wave per env chip-wide).  Measured (us/step split vs fused, same box): 16 x 256 x
    // 1024 envs 16.4 vs 19.8; 16 x 128 x 1024 9.2 vs 16.0; 9 x 500 x 1024 22.7 vs 27.9; 32 x 1024 x 2048 (no force) 232 vs
    // 233.  DCC_NO_SPLIT=1 / DCC_FORCE_SPLIT=1: tests, A/B.
    const bool split_pays = !(env->PPL >= 16 && p.use_force != 0);
    // (mode 1 with K >= 2 is the write probe: it must stream through the same kernel shape as the rollouts it stands for)
    if (p.obs != nullptr && env->PPL > 1 && act != 1 && (p.mode == 0 || p.K >= 2) && !env->no_split &&
        ((p.E <= kSplitMaxEnvs && split_pays) || env->force_split)) {   // single steps too: 24.8 -> 22.2 us at the c4 shard
        kernel_fn fn = pick_split_kernel(env->PPL, act, p.use_force != 0, p.N, p.M, allow_spec);
        hipLaunchKernelGGL(fn, dim3(p.E), dim3(kSplitBlock), env->lds_bytes_split, s, p);
        HIP_TRY(hipGetLastError());
        return DCC_OK;
    }
    kernel_fn fn = pick_kernel(env->PPL, act, p.use_force != 0, p.N, p.M, allow_spec);
    const int grid = (p.E + kWavesPerBlock - 1) / kWavesPerBlock;
    if (env->lds_bytes > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)env->lds_bytes));
    }
    // Row-writing multi-step launches stream best with about two 4-wave workgroups (8 writer waves) resident per CU -- c2 through this kernel
    // (DCC_NO_ROLES / a create-time choice of the fused shape) 0.726 -> 0.770 at 4096 envs, the c4 leg (16 x 256 x 8192 envs) 0.762 -> 0.787 of 8 TB/s with two instead of four, three change nothing, c5 is there by its
    // registers (profiles/r06/small_batch_shapes.txt): unused LDS caps the residency.  Single steps (latency-bound) and grids that fit
    // in two workgroups per CU anyway are left alone; DCC_FUSED_LDS_PAD=<bytes> (A/B) replaces the policy.
    size_t lds = env->lds_bytes;
    if (env->fused_lds_pad >= 0) lds += (size_t)env->fused_lds_pad;
    else if (p.obs != nullptr && p.K >= 2 && grid > 2 * env->n_cus && lds < env->lds_two_per_cu) lds = env->lds_two_per_cu;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), lds, s, p);
    HIP_TRY(hipGetLastError());
    return DCC_OK;
}

// max{ t >= 0 : sqrt_rn(t) <= r } (strict = false) or max{ t : sqrt_rn(t) < r } (strict = true);
// -1 when no t >= 0 qualifies.  Host sqrt is IEEE correctly rounded, like the device's.
double radicand_bound(double r, bool strict) {
    auto ok = [&](double t) { const double q = std::sqrt(t); return strict ? (q < r) : (q <= r); };
    if (!(r >= 0.0) || !ok(0.0)) return -1.0;
    const double big = 1.7976931348623157e308;
    if (std::isinf(r)) return big;
    double t = r * r;
    if (!std::isfinite(t)) t = big;
    while (!ok(t)) t = std::nextafter(t, 0.0);
    while (t < big && ok(std::nextafter(t, INFINITY))) t = std::nextafter(t, INFINITY);
    return t;
}

int fill_out(KParams& p, const dcc_env_out* out) {
    p.obs = out ? out->obs : nullptr;
    p.reward = out ? out->reward : nullptr;
    p.done = out ? out->done : nullptr;
    p.connect = out ? out->connect : nullptr;
    p.connect_s = out ? out->connect_s : nullptr;
    p.coverage = out ? out->coverage : nullptr;
    p.assign = out ? out->assign : nullptr;
    p.reward64 = out ? out->reward64 : nullptr;
    p.st_pos = out ? reinterpret_cast<double2*>(out->state_pos) : nullptr;
    p.st_vel = out ? reinterpret_cast<double2*>(out->state_vel) : nullptr;
    p.st_energy = out ? out->state_energy : nullptr;
    p.st_done = out ? out->state_done : nullptr;
    p.vec_ok = (p.L % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.obs) & 15u) == 0);
    return DCC_OK;
}

// Create-time A/B of the two kernel shapes an obs-writing K-step launch can take when there is one PoI per lane: the same
// short rollout (in-kernel action stream, observation rows into a scratch buffer) through each, timed with HIP events on
// a private stream; the faster one is used from then on (both are bit-identical: tests/test_env_hip_parity.py runs every
// golden case through both).  The env state is reset afterwards, i.e. left exactly as dcc_env_create leaves it.
// The choice is a property of (device, shape): measured once per process and shape, reused by later dcc_env_create calls (every
// test / every learner creates envs of the same few shapes), so that a library user pays the transient scratch allocation once.
// ... and of everything else that selects the two instantiations being timed: the FORCE template argument (pull force on: other
// launch bounds / occupancy), generic vs compile-time-specialised (DCC_NO_SPEC), envs per role-specialised workgroup.
struct TuneKey {
    int dev, E, N, M, force, no_spec, roles_envs;
    bool operator==(const TuneKey& o) const {
        return dev == o.dev && E == o.E && N == o.N && M == o.M && force == o.force && no_spec == o.no_spec && roles_envs == o.roles_envs;
    }
};
struct TuneVal { bool prefer_fused; float us[2]; };
std::mutex g_tune_mu;
std::vector<std::pair<TuneKey, TuneVal>> g_tune_cache;
constexpr size_t kTuneScratchMax = (size_t)4 << 30;   // larger batches keep the default shape (role-specialised): no multi-GB spike next to torch's allocator

void autotune_kernel_shape(dcc_env* e) {
    const char* at = std::getenv("DCC_AUTOTUNE");
    if ((at && at[0] == '0') || e->PPL != 1 || e->no_roles || e->force_roles) return;
    const size_t step_bytes = (size_t)e->cfg.n_envs * (size_t)e->L * sizeof(float);
    if (step_bytes < ((size_t)8 << 20)) return;          // small batches are latency-bound: keep the default
    // K: the role-specialised shape pays a pipeline fill / drain of about two steps per launch (the observation wave
    // trails the physics wave), which a short measurement would count against it: 64 steps keep that bias at 3 %.
    const int K = 64;
    if (step_bytes * K > kTuneScratchMax) return;
    const TuneKey key{e->device, e->cfg.n_envs, e->cfg.n_agents, e->cfg.n_pois, e->base.use_force != 0, e->no_spec ? 1 : 0,
                      (e->roles_envs_forced > 0) ? e->roles_envs_forced : (e->cfg.n_envs <= e->roles1_max ? 1 : 2)};
    {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        for (const auto& kv : g_tune_cache)
            if (kv.first == key) {
                e->prefer_fused = kv.second.prefer_fused; e->tune_us[0] = kv.second.us[0]; e->tune_us[1] = kv.second.us[1]; e->tuned = 1;
                return;
            }
    }
    float* scratch = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    if (hipMalloc(&scratch, step_bytes * K) != hipSuccess) { (void)hipGetLastError(); return; }
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&ev[0]) == hipSuccess &&
              hipEventCreate(&ev[1]) == hipSuccess;
    float best[2] = {1e30f, 1e30f};
    for (int rep = 0; ok && rep < 4; ++rep) {              // rep 0 warms both up (first touch of the scratch pages, clocks)
        for (int v = 0; ok && v < 2; ++v) {
            KParams p = e->base;
            p.mode = 0; p.K = K; p.actions = nullptr; p.seed = 0x5eedULL + rep; p.step0 = 0; p.env0 = 0; p.env_total = e->cfg.n_envs;
            dcc_env_out o;
            std::memset(&o, 0, sizeof(o));
            o.obs = scratch;
            fill_out(p, &o);
            e->prefer_fused = (v == 1);
            ok = hipEventRecord(ev[0], st) == hipSuccess && launch(e, p, 2, st) == DCC_OK && hipEventRecord(ev[1], st) == hipSuccess &&
                 hipEventSynchronize(ev[1]) == hipSuccess;
            float ms = 0.f;
            if (ok) ok = hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess;
            if (ok && rep > 0 && ms < best[v]) best[v] = ms;
        }
    }
    e->prefer_fused = false;
    if (ok) {
        e->tune_us[0] = best[0] * 1e3f / K; e->tune_us[1] = best[1] * 1e3f / K;
        // asymmetric on purpose: where the role-specialised shape wins it wins by 9-11 %, where the fused one wins it is by
        // ~1 % (profiles/r02), so a wrong "fused" costs ten times what a wrong "roles" costs: the fused shape has to be ahead
        // by more than the measurement's own bias and noise
        e->prefer_fused = best[1] < 0.94f * best[0];
        e->tuned = 1;
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_tune_cache.push_back({key, TuneVal{e->prefer_fused, {e->tune_us[0], e->tune_us[1]}}});
    } else {
        (void)hipGetLastError();
    }
    // Back to the reset state -- after everything issued on the private stream has finished (on the error path a kernel may still
    // be running there, and the memsets below go to the null stream, which a non-blocking stream does not synchronise with).
    if (st) (void)hipStreamSynchronize(st);
    const size_t E = e->cfg.n_envs, N = e->cfg.n_agents, M = e->cfg.n_pois;
    (void)hipMemset(e->d_pos, 0, sizeof(double2) * E * N); (void)hipMemset(e->d_vel, 0, sizeof(double2) * E * N);
    (void)hipMemset(e->d_energy, 0, sizeof(float) * E * M); (void)hipMemset(e->d_done, 0, E * M);
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(scratch);
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        target = dev;
    }
    ~DeviceGuard() { if (prev >= 0 && prev != target) (void)hipSetDevice(prev); }
    int target = -1;
};

}  // namespace

extern "C" {

int dcc_abi_version(void) { return DCC_ABI_VERSION; }
const char* dcc_last_error(void) { return g_err.c_str(); }

void dcc_env_cfg_default(dcc_env_cfg* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->device = -1;
    c->r_cover = 0.25; c->r_comm = 0.5;          // SC:20 defaults (dcc.yaml overrides: 0.2 / 0.4)
    c->comm_r_scale = 0.9; c->comm_force_scale = 0.0;  // CW:7
    c->dt = 0.1; c->damping = 0.25; c->max_speed = 0.5; c->sensitivity = 5.0; c->mass = 1.0;
    c->contact_margin = 1e-3; c->m_energy = 5.0;
    c->rew_cover = 75.0; c->rew_done = 1500.0; c->rew_out = -100.0;
    c->bound_soft = 1.0; c->bound_hard = 1.5;
}

int64_t dcc_env_bytes_per_step(int32_t N, int32_t M, int32_t with_actions, int32_t with_obs) {
    const int64_t D = 4 + 2 * (int64_t)(N - 1) + 5 * (int64_t)M;
    int64_t b = 40 * (int64_t)N + 11 * (int64_t)M + 11;
    if (!with_actions) b -= 8 * (int64_t)N;
    if (with_obs) b += 4 * (int64_t)N * D;
    return b;
}

int dcc_env_create(const dcc_env_cfg* c, dcc_env** out) {
    if (!c || !out) return fail(DCC_EINVAL, "dcc_env_create: null argument");
    *out = nullptr;
    if (c->n_envs < 1) return fail(DCC_EINVAL, "dcc_env_create: n_envs must be >= 1");
    if (c->n_agents < 1 || c->n_agents > DCC_MAX_AGENTS)
        return fail(DCC_EINVAL, "dcc_env_create: n_agents must be in 1..64 (one UAV per wavefront lane)");
    if (c->n_pois < 1 || c->n_pois > DCC_MAX_POIS) return fail(DCC_EINVAL, "dcc_env_create: n_pois must be in 1..1024");
    if (!c->poi_xy) return fail(DCC_EINVAL, "dcc_env_create: poi_xy is NULL");
    if (!(c->mass > 0) || !(c->dt > 0) || !(c->r_cover >= 0) || !(c->r_comm >= 0) || !(c->contact_margin > 0))
        return fail(DCC_EINVAL, "dcc_env_create: non-positive mass/dt/contact_margin or negative radius");
    if (c->comm_force_scale > 0 && !(c->comm_r_scale > 0))
        return fail(DCC_EINVAL, "dcc_env_create: comm_force_scale > 0 requires comm_r_scale > 0");
    if (c->comm_force_scale < 0) return fail(DCC_EINVAL, "dcc_env_create: comm_force_scale < 0");

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(DCC_EHIP, "dcc_env_create: no HIP device visible (this library has no CPU path)");
    int dev = c->device;
    if (dev < 0) HIP_TRY(hipGetDevice(&dev));
    if (dev >= ndev) return fail(DCC_EINVAL, "dcc_env_create: device ordinal out of range");
    DeviceGuard guard(dev);
    if (!guard.ok) return fail(DCC_EHIP, "dcc_env_create: hipSetDevice failed");

    dcc_env* e = new (std::nothrow) dcc_env();
    if (!e) return fail(DCC_ENOMEM, "dcc_env_create: out of host memory");
    e->cfg = *c;
    e->cfg.poi_xy = nullptr;
    e->device = dev;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) e->n_cus = cus; else (void)hipGetLastError(); }
    {   // LDS of a CU -> the smallest workgroup allocation of which only two fit (must stay within the 64 KB a launch may ask for)
        int lds_cu = 0;
        if (hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && lds_cu >= 96 * 1024) {
            const size_t third = (size_t)lds_cu / 3 + 1024;
            if (third <= 64 * 1024) e->lds_two_per_cu = (third + 255) & ~(size_t)255;
        } else {
            (void)hipGetLastError();
        }
    }
    const int E = c->n_envs, N = c->n_agents, M = c->n_pois;
    e->D = 4 + 2 * (N - 1) + 5 * M;
    e->L = N * e->D;
    int ppl = (M + 63) / 64, p2 = 1;
    while (p2 < ppl) p2 <<= 1;
    e->PPL = p2;

    KParams& p = e->base;
    std::memset(&p, 0, sizeof(p));
    p.E = E; p.N = N; p.M = M; p.D = e->D; p.L = e->L; p.H = 4 + 2 * (N - 1);
    p.K = 1; p.mode = 0; p.roles_envs = 2; p.roles_pairs = 1; p.roles_lds = 0; p.roles_slots = 2; p.obs_drain = 2;
    p.use_connect = c->comm_r_scale > 0;
    const double contact_force = 1e+2 * c->comm_force_scale;  // core.py:109 scaled at CW:16
    p.use_force = contact_force > 0;
    p.magicN = ((1u << 20) + (unsigned)N - 1u) / (unsigned)N;
    p.sq_cover = radicand_bound(c->r_cover, false);                                   // d <= r_cover (CW:165)
    p.sq_thr = radicand_bound(c->r_comm + c->r_comm, true);                           // d <  r_a + r_b (CW:77)
    p.sq_thr_s = radicand_bound(c->comm_r_scale * (c->r_comm + c->r_comm), true);     // CW:79
    p.sq_speed = radicand_bound(c->max_speed, false);                                 // speed > max_speed (CW:151)
    p.thr2 = c->comm_r_scale * 2 * c->r_comm;               // CW:119 (its own rounding order)
    p.dmax = (c->r_comm + c->r_comm) * c->comm_r_scale;     // CW:134
    p.contact_force = contact_force; p.contact_margin = c->contact_margin;
    p.dt = c->dt; p.keep = 1 - c->damping; p.max_speed = c->max_speed; p.sens = c->sensitivity; p.mass = c->mass;
    p.m_energy = c->m_energy;
    p.rew_cover = c->rew_cover; p.rew_done = c->rew_done; p.rew_out = c->rew_out;
    p.bound_soft = c->bound_soft; p.bound_hard = c->bound_hard;
    p.sens_f = (float)c->sensitivity; p.mass_f = (float)c->mass; p.dt_f = (float)c->dt; p.m_energy_f = (float)c->m_energy;
    p.env0 = 0; p.env_total = E;

    { const char* ns = std::getenv("DCC_NO_SPEC"); e->no_spec = ns && ns[0] == '1'; }
    { const char* nr = std::getenv("DCC_NO_ROLES"); e->no_roles = nr && nr[0] == '1'; }
    { const char* fr = std::getenv("DCC_FORCE_ROLES"); e->force_roles = fr && fr[0] == '1'; }
    { const char* ns = std::getenv("DCC_NO_SPLIT"); e->no_split = ns && ns[0] == '1'; }
    { const char* fs = std::getenv("DCC_FORCE_SPLIT"); e->force_split = fs && fs[0] == '1'; }
    { const char* re = std::getenv("DCC_ROLES_ENVS"); if (re && (re[0] == '1' || re[0] == '2')) e->roles_envs_forced = re[0] - '0'; }
    { const char* rm = std::getenv("DCC_ROLES1_MAX"); if (rm) e->roles1_max = std::atoi(rm); }
    { const char* fp = std::getenv("DCC_FUSED_LDS_PAD"); if (fp && fp[0]) e->fused_lds_pad = std::atol(fp); }
    { const char* rp = std::getenv("DCC_ROLES_PAIRS"); if (rp && (rp[0] == '1' || rp[0] == '2')) e->roles_pairs_forced = rp[0] - '0'; }
    { const char* od = std::getenv("DCC_OBS_DRAIN"); if (od && od[0]) e->obs_drain_forced = std::atoi(od); }
    // hand-off depth: 8 slots per env while they are small (N <= 16: <= 1.1 KB each), else the minimum of 2.  Measured at c2 (profiles/r06/
    // small_batch_shapes.txt): 2 -> 4 -> 8 slots = 0.793 -> 0.799 -> 0.802 of 8 TB/s at 4096 envs (HBM actions 0.732 -> 0.737 -> 0.739), no effect below 1024 envs
    { const char* sl = std::getenv("DCC_ROLES_SLOTS"); const int v = sl ? std::atoi(sl) : 0; e->roles_slots = (v == 2 || v == 4 || v == 8) ? v : (N <= 16 ? 8 : 2); }
    e->lds_bytes_roles = (size_t)((M * 16 + 15) & ~15) + 2 * (size_t)e->roles_slots * ((size_t)N * 32 + 64 * 4 + 16 + sizeof(StepRec)) + 16 + (size_t)kRolesObs * kStageC * 4;
    e->lds_bytes = (size_t)((M * 16 + 15) & ~15) + (size_t)kWavesPerBlock * ((size_t)N * 32 + (size_t)kStageC * 4);
    e->lds_bytes_split = (size_t)((M * 16 + 15) & ~15) + 2 * ((size_t)N * 32 + (size_t)p2 * 256 + 256) + 32 +
                         (size_t)kSplitObs * kStageC * 4;

    auto cleanup = [&](int code, const std::string& m) { dcc_env_destroy(e); return fail(code, m); };
    hipError_t err;
    if ((err = hipMalloc(&e->d_poi, sizeof(double2) * M)) != hipSuccess ||
        (err = hipMalloc(&e->d_pos, sizeof(double2) * (size_t)E * N)) != hipSuccess ||
        (err = hipMalloc(&e->d_vel, sizeof(double2) * (size_t)E * N)) != hipSuccess ||
        (err = hipMalloc(&e->d_energy, sizeof(float) * (size_t)E * M)) != hipSuccess ||
        (err = hipMalloc(&e->d_done, (size_t)E * M)) != hipSuccess)
        return cleanup(DCC_ENOMEM, std::string("dcc_env_create: hipMalloc: ") + hipGetErrorString(err));
    if ((err = hipMemcpy(e->d_poi, c->poi_xy, sizeof(double2) * M, hipMemcpyHostToDevice)) != hipSuccess ||
        (err = hipMemset(e->d_pos, 0, sizeof(double2) * (size_t)E * N)) != hipSuccess ||
        (err = hipMemset(e->d_vel, 0, sizeof(double2) * (size_t)E * N)) != hipSuccess ||
        (err = hipMemset(e->d_energy, 0, sizeof(float) * (size_t)E * M)) != hipSuccess ||
        (err = hipMemset(e->d_done, 0, (size_t)E * M)) != hipSuccess)
        return cleanup(DCC_EHIP, std::string("dcc_env_create: init copy: ") + hipGetErrorString(err));
    p.poi = e->d_poi; p.pos = e->d_pos; p.vel = e->d_vel; p.energy = e->d_energy; p.done_poi = e->d_done;
    autotune_kernel_shape(e);
    *out = e;
    return DCC_OK;
}

int dcc_env_kernel_choice(const dcc_env* e, float* us_roles, float* us_fused) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_kernel_choice: null env");
    if (us_roles) *us_roles = e->tune_us[0];
    if (us_fused) *us_fused = e->tune_us[1];
    if (!e->tuned) return 0;
    return e->prefer_fused ? 2 : 1;
}

int dcc_env_destroy(dcc_env* e) {
    if (!e) return DCC_OK;
    {
        DeviceGuard guard(e->device);
        (void)hipFree(e->d_poi); (void)hipFree(e->d_pos); (void)hipFree(e->d_vel);
        (void)hipFree(e->d_energy); (void)hipFree(e->d_done);
    }
    delete e;
    return DCC_OK;
}

int dcc_env_obs_dim(const dcc_env* e) { return e ? e->D : DCC_EINVAL; }

int dcc_env_reset(dcc_env* e, float* obs, void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_reset: null env");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.mode = 1; p.K = 1;
    dcc_env_out o;
    std::memset(&o, 0, sizeof(o));
    o.obs = obs;
    fill_out(p, &o);
    return launch(e, p, 0, stream);
}

// K repetitions of the observation producer only (reset state, no physics): the env kernels' own store pattern -- isolates
// the LDS staging + HBM store pipeline, and is what a caller times to tell a well-placed output buffer from a badly placed one.
int dcc_env_obs_write_probe(dcc_env* e, int32_t K, float* obs, void* stream) {
    if (!e || K < 1) return fail(DCC_EINVAL, "dcc_env_obs_write_probe: null env or K < 1");
    if (!obs) return fail(DCC_EINVAL, "dcc_env_obs_write_probe: obs is NULL");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.mode = 1; p.K = K;
    dcc_env_out o;
    std::memset(&o, 0, sizeof(o));
    o.obs = obs;
    fill_out(p, &o);
    return launch(e, p, 0, stream);
}

int dcc_env_step(dcc_env* e, const void* actions, int act_dtype, const dcc_env_out* out, void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_step: null env");
    if (!actions) return fail(DCC_EINVAL, "dcc_env_step: actions is NULL");
    if (act_dtype != DCC_ACT_F32 && act_dtype != DCC_ACT_F64) return fail(DCC_EINVAL, "dcc_env_step: bad act_dtype");
    const uintptr_t align = act_dtype == DCC_ACT_F64 ? 15u : 7u;
    if (reinterpret_cast<uintptr_t>(actions) & align) return fail(DCC_EINVAL, "dcc_env_step: actions pointer misaligned");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.mode = 0; p.K = 1; p.actions = actions;
    fill_out(p, out);
    return launch(e, p, act_dtype == DCC_ACT_F64 ? 1 : 0, stream);
}

int dcc_env_step_features(dcc_env* e, const void* actions, int act_dtype, const dcc_env_out* out, const dcc_obs_feat* feat,
                          void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_step_features: null env");
    if (!actions || !feat) return fail(DCC_EINVAL, "dcc_env_step_features: actions / feat is NULL");
    if (act_dtype != DCC_ACT_F32) return fail(DCC_EUNSUPPORTED, "dcc_env_step_features: float32 actions only (use dcc_env_step + dcc_obs_features)");
    if (out && out->obs) return fail(DCC_EINVAL, "dcc_env_step_features: observation rows are not written by this entry point (out->obs must be NULL)");
    if (reinterpret_cast<uintptr_t>(actions) & 7u) return fail(DCC_EINVAL, "dcc_env_step_features: actions pointer misaligned");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.mode = 0; p.K = 1; p.actions = actions;
    fill_out(p, out);
    FeatParams f;
    std::memset(&f, 0, sizeof(f));
    f.poi = e->d_poi;
    f.head = feat->head; f.poi_feat = feat->poi_feat; f.stats = feat->stats; f.cstats = feat->cstats; f.xa = feat->xa; f.xc = feat->xc;
    f.n = p.E; f.N = p.N; f.M = p.M; f.m_energy = (float)e->cfg.m_energy;
    f.ka = (2 * p.M + 1 + 7) / 8 * 8;
    f.kc = (p.N * (4 + 2 * (p.N - 1)) + 2 * p.M + 1 + 7) / 8 * 8;
    const auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    f.stage = (f.head || f.xc) && p.N * (4 + 2 * (p.N - 1)) <= kStageC && a16(f.head) && a16(f.xc);
    feat_kernel_fn fn = p.use_force ? pick_feat_kernel<true>(e->PPL, p.N, p.M, !e->no_spec) : pick_feat_kernel<false>(e->PPL, p.N, p.M, !e->no_spec);
    const int grid = (p.E + kWavesPerBlock - 1) / kWavesPerBlock;
    if (e->lds_bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), e->lds_bytes, reinterpret_cast<hipStream_t>(stream), p, f);
    HIP_TRY(hipGetLastError());
    return DCC_OK;
}

int dcc_env_rollout(dcc_env* e, int32_t K, const float* actions, uint64_t seed, uint32_t step0, int32_t env0,
                    int32_t env_total, const dcc_env_out* out, void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_rollout: null env");
    if (K < 1) return fail(DCC_EINVAL, "dcc_env_rollout: K must be >= 1");
    if (actions && (reinterpret_cast<uintptr_t>(actions) & 7u)) return fail(DCC_EINVAL, "dcc_env_rollout: actions misaligned");
    if (!actions && (env0 < 0 || env_total < env0 + e->cfg.n_envs))
        return fail(DCC_EINVAL, "dcc_env_rollout: env0/env_total do not cover this shard");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.mode = 0; p.K = K; p.actions = actions;
    p.seed = seed; p.step0 = step0; p.env0 = env0; p.env_total = env_total;
    fill_out(p, out);
    return launch(e, p, actions ? 0 : 2, stream);
}

int dcc_obs_expand(dcc_env* e, int64_t n, const double* pos, const double* vel, const float* energy,
                   const uint8_t* done, float* obs, void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_obs_expand: null env");
    if (n < 1 || n > 0x7fffffffLL) return fail(DCC_EINVAL, "dcc_obs_expand: n out of range");
    if (!pos || !vel || !energy || !done || !obs) return fail(DCC_EINVAL, "dcc_obs_expand: null pointer");
    if ((reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(vel)) & 15u)
        return fail(DCC_EINVAL, "dcc_obs_expand: pos / vel must be 16-byte aligned");
    DeviceGuard guard(e->device);
    KParams p = e->base;
    p.E = (int)n; p.K = 1; p.mode = 1;
    dcc_env_out o;
    std::memset(&o, 0, sizeof(o));
    o.obs = obs;
    fill_out(p, &o);
    p.st_pos = reinterpret_cast<double2*>(const_cast<double*>(pos));
    p.st_vel = reinterpret_cast<double2*>(const_cast<double*>(vel));
    p.st_energy = const_cast<float*>(energy);
    p.st_done = const_cast<uint8_t*>(done);
    kernel_fn fn;
    switch (e->PPL) {
        case 1: fn = dcc_obs_expand_kernel<1>; break;
        case 2: fn = dcc_obs_expand_kernel<2>; break;
        case 4: fn = dcc_obs_expand_kernel<4>; break;
        case 8: fn = dcc_obs_expand_kernel<8>; break;
        default: fn = dcc_obs_expand_kernel<16>; break;
    }
    const int grid = (int)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kBlock), e->lds_bytes, reinterpret_cast<hipStream_t>(stream), p);
    HIP_TRY(hipGetLastError());
    return DCC_OK;
}

int dcc_obs_features_x(dcc_env* e, int64_t n, const double* pos, const double* vel, const float* energy,
                       const uint8_t* done, float* head, float* poi_feat, double* stats, double* cstats, float* xa, float* xc,
                       void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_obs_features: null env");
    if (n < 1 || n > 0x7fffffffLL) return fail(DCC_EINVAL, "dcc_obs_features: n out of range");
    if (!pos || !vel || !energy || !done) return fail(DCC_EINVAL, "dcc_obs_features: null state pointer");
    if ((reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(vel)) & 15u)
        return fail(DCC_EINVAL, "dcc_obs_features: pos / vel must be 16-byte aligned");
    if (!head && !poi_feat && !stats && !cstats && !xa && !xc) return DCC_OK;
    DeviceGuard guard(e->device);
    FeatParams p;
    p.pos = reinterpret_cast<const double2*>(pos); p.vel = reinterpret_cast<const double2*>(vel);
    p.energy = energy; p.done = done; p.poi = e->d_poi;
    p.head = head; p.poi_feat = poi_feat; p.stats = stats; p.cstats = cstats;
    p.n = (int)n; p.N = e->cfg.n_agents; p.M = e->cfg.n_pois; p.m_energy = (float)e->cfg.m_energy;
    p.xa = xa; p.xc = xc;
    p.ka = (2 * p.M + 1 + 7) / 8 * 8;
    p.kc = (p.N * (4 + 2 * (p.N - 1)) + 2 * p.M + 1 + 7) / 8 * 8;
    const int grid = (int)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t head_floats = (size_t)p.N * (4 + 2 * (p.N - 1));
    size_t lds = (size_t)kWavesPerBlock * head_floats * sizeof(float);
    const auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; };
    p.stage = (head || xc) && lds <= 48 * 1024 && a16(head) && a16(xc);
    if (!p.stage) lds = 0;
    switch (e->PPL) {
        case 1: hipLaunchKernelGGL(dcc_obs_features_kernel<1>, dim3(grid), dim3(kBlock), lds, s, p); break;
        case 2: hipLaunchKernelGGL(dcc_obs_features_kernel<2>, dim3(grid), dim3(kBlock), lds, s, p); break;
        case 4: hipLaunchKernelGGL(dcc_obs_features_kernel<4>, dim3(grid), dim3(kBlock), lds, s, p); break;
        case 8: hipLaunchKernelGGL(dcc_obs_features_kernel<8>, dim3(grid), dim3(kBlock), lds, s, p); break;
        default: hipLaunchKernelGGL(dcc_obs_features_kernel<16>, dim3(grid), dim3(kBlock), lds, s, p); break;
    }
    HIP_TRY(hipGetLastError());
    return DCC_OK;
}

int dcc_obs_features(dcc_env* e, int64_t n, const double* pos, const double* vel, const float* energy,
                     const uint8_t* done, float* head, float* poi_feat, double* stats, double* cstats, void* stream) {
    return dcc_obs_features_x(e, n, pos, vel, energy, done, head, poi_feat, stats, cstats, nullptr, nullptr, stream);
}

int dcc_env_get_state(dcc_env* e, double* pos, double* vel, float* energy, uint8_t* done, void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_get_state: null env");
    DeviceGuard guard(e->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t E = e->cfg.n_envs, N = e->cfg.n_agents, M = e->cfg.n_pois;
    if (pos) HIP_TRY(hipMemcpyAsync(pos, e->d_pos, sizeof(double2) * E * N, hipMemcpyDeviceToDevice, s));
    if (vel) HIP_TRY(hipMemcpyAsync(vel, e->d_vel, sizeof(double2) * E * N, hipMemcpyDeviceToDevice, s));
    if (energy) HIP_TRY(hipMemcpyAsync(energy, e->d_energy, sizeof(float) * E * M, hipMemcpyDeviceToDevice, s));
    if (done) HIP_TRY(hipMemcpyAsync(done, e->d_done, E * M, hipMemcpyDeviceToDevice, s));
    return DCC_OK;
}

int dcc_env_set_state(dcc_env* e, const double* pos, const double* vel, const float* energy, const uint8_t* done,
                      void* stream) {
    if (!e) return fail(DCC_EINVAL, "dcc_env_set_state: null env");
    DeviceGuard guard(e->device);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t E = e->cfg.n_envs, N = e->cfg.n_agents, M = e->cfg.n_pois;
    if (pos) HIP_TRY(hipMemcpyAsync(e->d_pos, pos, sizeof(double2) * E * N, hipMemcpyDeviceToDevice, s));
    if (vel) HIP_TRY(hipMemcpyAsync(e->d_vel, vel, sizeof(double2) * E * N, hipMemcpyDeviceToDevice, s));
    if (energy) HIP_TRY(hipMemcpyAsync(e->d_energy, energy, sizeof(float) * E * M, hipMemcpyDeviceToDevice, s));
    if (done) HIP_TRY(hipMemcpyAsync(e->d_done, done, E * M, hipMemcpyDeviceToDevice, s));
    return DCC_OK;
}

}  // extern "C"
