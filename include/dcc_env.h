/*
 * dcc_env.h -- C-ABI of libdcc_hip.so: the batched multi-agent coverage environment on MI355X.
 *
 * The reference (zhaozijie2022/dynamic-coverage-control) is pure Python and has no FFI; its only
 * extension seam is duck typing: `make_env(cfg)` (uav_dcc_control/envs/make_env.py:15-49) returns a
 * vec-env with reset()/step() (uav_dcc_control/envs/wrappers.py:133-261).  This header is the
 * boundary a ctypes binding of that seam calls; INTEGRATION.md shows the binding.
 * Paths below are relative to uav_dcc_control/ in the reference.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DCC_E* code; dcc_last_error() gives the
 *     message of the last failure on the calling thread.  No exceptions / Python objects cross.
 *   - all I/O buffers are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr());
 *     the library owns only its internal env state (pos, vel, energy, done) and the PoI table.
 *   - calls are asynchronous and ordered on the hipStream_t passed as `stream` (void*; NULL = the
 *     default stream).  No hidden device synchronisation.  A handle is not thread-safe.
 *   - plain C types only; no torch types.
 */
#ifndef DCC_ENV_H
#define DCC_ENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define DCC_API __attribute__((visibility("default")))
#else
#define DCC_API
#endif

/* Bumped when a struct layout or the meaning of an existing entry point changes; entry points ADDED since (round 2:
 * dcc_obs_features_x, dcc_grad_norm_clip / dcc_adam_step, dcc_actor_l1_pre_fwd / _bwd) leave it as it is -- a binding that needs
 * one of them checks for the symbol. */
#define DCC_ABI_VERSION 2

#define DCC_OK 0
#define DCC_EINVAL (-1)   /* bad argument / shape */
#define DCC_EHIP (-2)     /* HIP runtime error (message in dcc_last_error) */
#define DCC_ENOMEM (-3)
#define DCC_EUNSUPPORTED (-4)

#define DCC_MAX_AGENTS 64   /* one UAV per lane of a wavefront */
#define DCC_MAX_POIS 1024   /* <= 16 PoIs per lane */

/* action dtypes accepted by step/rollout: the reference keeps the caller's dtype through
 * `u *= 5.0` and the force accumulation (envs/mpe/multiagent/environment.py:186-190,
 * CoverageWorld.py:115-116,147); float32 is what Learner.collect passes (learner.py:240-250). */
#define DCC_ACT_F32 0
#define DCC_ACT_F64 1

/* Scenario + world constants.  Defaults of the reference are given by dcc_env_cfg_default():
 * envs/mpe/multiagent/scenarios/coverage.py:20-31 (r_cover, r_comm, m_energy, rew_*),
 * core.py:105-110 (dt, damping, contact_force, contact_margin), coverage.py:54 (max_speed),
 * environment.py:186 (sensitivity), CoverageWorld.py:7-23 (comm_r_scale, comm_force_scale, dt). */
typedef struct dcc_env_cfg {
    int32_t n_envs;            /* E: independent env instances on this device */
    int32_t n_agents;          /* N: UAVs per env, 1..DCC_MAX_AGENTS */
    int32_t n_pois;            /* M: PoIs per env, 1..DCC_MAX_POIS */
    int32_t device;            /* HIP device ordinal, -1 = current device */
    double r_cover;            /* coverage radius (<=) */
    double r_comm;             /* communication radius; adjacency iff d < r_a + r_b */
    double comm_r_scale;       /* >0: connectivity flags computed; radius scale for the pull force */
    double comm_force_scale;   /* contact_force = 1e2 * comm_force_scale; 0 disables the force */
    double dt, damping, max_speed, sensitivity, mass;
    double contact_margin, m_energy;
    double rew_cover, rew_done, rew_out;
    double bound_soft, bound_hard;   /* 1.0 (penalty) and 1.5 (done) in coverage.py:93-96,112-116 */
    const double* poi_xy;      /* HOST pointer, [M,2] float64 (pos_pois.npy rows); copied */
} dcc_env_cfg;

typedef struct dcc_env dcc_env;   /* opaque */

/* Per-step outputs.  Any pointer may be NULL (that output is skipped).  With dcc_env_rollout
 * every array has a leading K (step) dimension. */
typedef struct dcc_env_out {
    float*   obs;        /* [E,N,D] float32: what the vec-env returns (reset obs for envs that just
                            finished, wrappers.py:104-109), cast as SharedReplayBuffer stores it */
    float*   reward;     /* [E]  shared reward every agent of the env receives (environment.py:106-108) */
    uint8_t* done;       /* [E]  np.all(done_n) (coverage.py:112-117); terminal-step value */
    uint8_t* connect;    /* [E]  world.connect  (CoverageWorld.py:92), from PRE-move positions */
    uint8_t* connect_s;  /* [E]  world.connect_ (CoverageWorld.py:93) */
    float*   coverage;   /* [E]  info["coverage_rate"] (mpe/uav_dcc.py:48) */
    uint8_t* assign;     /* [E,M] PoI-assignment index argmin_i ||x_i - p_j|| (first min), terminal positions */
    double*  reward64;   /* [E]  the same reward before the float32 cast (optional) */
    /* ABI v2 -- compact state the observations are a function of (the state AFTER the step and after the
     * auto-reset, i.e. exactly what `obs` was built from): 32N + 5M bytes per env-step instead of 4*N*D.
     * A rollout buffer can store these and regenerate observations with dcc_obs_expand(). */
    double*  state_pos;     /* [E,N,2] float64 */
    double*  state_vel;     /* [E,N,2] float64 */
    float*   state_energy;  /* [E,M]   float32 */
    uint8_t* state_done;    /* [E,M]   uint8   */
} dcc_env_out;

DCC_API int         dcc_abi_version(void);
DCC_API const char* dcc_last_error(void);

/* Fill cfg with the reference's constants (n_envs/n_agents/n_pois/poi_xy left 0/NULL). */
DCC_API void dcc_env_cfg_default(dcc_env_cfg* cfg);

/* Replaces: make_env + DCEnv.__init__ + Scenario.make_world/reset_world
 * (envs/make_env.py:15-49, envs/mpe/uav_dcc.py:8-44, scenarios/coverage.py:33-78). */
DCC_API int dcc_env_create(const dcc_env_cfg* cfg, dcc_env** out);
DCC_API int dcc_env_destroy(dcc_env* env);

/* D = 4 + 2(N-1) + 5M (coverage.py:99-110); share-obs dim is N*D (uav_dcc.py:40-43). */
DCC_API int dcc_env_obs_dim(const dcc_env* env);

/* Replaces: ShareVecEnv.reset (wrappers.py:167-171,236-238) -> reset_world (coverage.py:64-78).
 * obs: [E,N,D] float32 device pointer or NULL. */
DCC_API int dcc_env_reset(dcc_env* env, float* obs, void* stream);

/* Replaces: SubprocVecEnv.step_async/step_wait + worker auto-reset (wrappers.py:97-110,156-165),
 * DCEnv.step (uav_dcc.py:46-49), MultiAgentEnv.step/_set_action (environment.py:86-110,153-201),
 * CoverageWorld.step and everything it calls (CoverageWorld.py:57-174), Scenario.reward/
 * observation/done (coverage.py:80-117).
 * actions: [E,N,2] device pointer of act_dtype; never written. */
DCC_API int dcc_env_step(dcc_env* env, const void* actions, int act_dtype, const dcc_env_out* out, void* stream);

/* K fused steps in one launch (the learner-free rollout used for BASELINE config 2).
 * actions: [K,E,N,2] float32 device pointer, or NULL to draw them in-kernel from the
 * counter-based generator (uniform [-1,1), keyed by seed, step0+k, env0+e, agent; `env_total`
 * is the job-wide env count so that shards of one job draw disjoint, reproducible streams).
 * out arrays carry a leading K dimension; out->obs == NULL skips the obs write. */
DCC_API int dcc_env_rollout(dcc_env* env, int32_t K, const float* actions, uint64_t seed, uint32_t step0,
                    int32_t env0, int32_t env_total, const dcc_env_out* out, void* stream);

/* Checkpoint / fixture access to the internal state (device pointers, NULL = skip):
 * pos [E,N,2] f64, vel [E,N,2] f64, energy [E,M] f32, done [E,M] u8. */
DCC_API int dcc_env_get_state(dcc_env* env, double* pos, double* vel, float* energy, uint8_t* done, void* stream);
DCC_API int dcc_env_set_state(dcc_env* env, const double* pos, const double* vel, const float* energy,
                      const uint8_t* done, void* stream);

/* Observation rows from compact state: obs[n] = Scenario.observation of every agent (coverage.py:99-110) for n
 * independent (env-)states laid out like the state_* outputs above (pos/vel [n,N,2] f64, energy [n,M] f32,
 * done [n,M] u8) -> obs [n,N,D] float32.  Uses the PoI table and sizes of `env`; does not touch its state.
 * Bit-identical to the obs the step that produced the state wrote. */
DCC_API int dcc_obs_expand(dcc_env* env, int64_t n, const double* pos, const double* vel, const float* energy,
                           const uint8_t* done, float* obs, void* stream);

/* Compact policy-input features of n states (same state layout as dcc_obs_expand); any output may be NULL:
 *   head     [n,N,4+2(N-1)] float32  the first columns of every observation row (own vel, own pos, other UAVs'
 *                                    relative positions; coverage.py:100-104), bit-identical to obs[..., :4+2(N-1)]
 *   poi_feat [n,2M]         float32  PoI energies then PoI done flags (the obs columns that do not depend on the agent)
 *   stats    [n,N,2]        float64  (mean, sum of squared deviations) of the D float32 values of every row, i.e. the
 *                                    moments an input LayerNorm over the row needs (algos/algo_utils/mlp.py:45-49)
 *   cstats   [n,2]          float64  the same two moments of the centralised row = the N agent rows concatenated
 *                                    (learner.py:217-220), pooled from the per-row moments
 * With them the first Linear layer after the input LayerNorm is evaluated without materialising the rows
 * (dynamic-coverage-control_amd/algos/algo_utils/structured.py). */
DCC_API int dcc_obs_features(dcc_env* env, int64_t n, const double* pos, const double* vel, const float* energy,
                             const uint8_t* done, float* head, float* poi_feat, double* stats, double* cstats,
                             void* stream);

/* dcc_obs_features plus the input matrices of the per-env GEMMs of the structured first layers (any output may be NULL),
 * each row zero-padded to a multiple of 8 floats so that rows stay 16-byte aligned:
 *   xa [n, pad8(2M + 1)]                 float32   [energy | done | 1 | 0..]                       (actor: the term its N agents share)
 *   xc [n, pad8(N(4+2(N-1)) + 2M + 1)]   float32   [head_0 .. head_{N-1} | energy | done | 1 | 0..] (centralised critic)
 * The trailing 1 lets the constant term of the layer ride in the GEMM.  Written by the same kernel that produces the
 * features: no concatenation passes on the caller's side. */
DCC_API int dcc_obs_features_x(dcc_env* env, int64_t n, const double* pos, const double* vel, const float* energy,
                               const uint8_t* done, float* head, float* poi_feat, double* stats, double* cstats,
                               float* xa, float* xc, void* stream);

/* One env step AND the compact policy-input features of the state it leaves, in one launch: what dcc_env_step followed by
 * dcc_obs_features_x on the emitted state computes (bit-identical), without the second launch and without re-reading the
 * state -- the wave that stepped an env derives the features from its registers.  This is the step of a policy-driven
 * rollout with structured first layers (reference learner.py:178-214: collect -> envs.step -> insert).
 * out: like dcc_env_step, but out->obs must be NULL (rows are exactly what this path avoids).  feat: destinations as in
 * dcc_obs_features_x with n = n_envs; any may be NULL.  float32 actions only (DCC_EUNSUPPORTED otherwise). */
typedef struct dcc_obs_feat {
    float* head;       /* [E, N, 4+2(N-1)] */
    float* poi_feat;   /* [E, 2M]          */
    double* stats;     /* [E, N, 2]        */
    double* cstats;    /* [E, 2]           */
    float* xa;         /* [E, pad8(2M+1)]  */
    float* xc;         /* [E, pad8(N(4+2(N-1)) + 2M + 1)] */
} dcc_obs_feat;
DCC_API int dcc_env_step_features(dcc_env* env, const void* actions, int act_dtype, const dcc_env_out* out,
                                  const dcc_obs_feat* feat, void* stream);

/* Algorithmic HBM bytes of one env-step (SURVEY.md section 8d, fp32 I/O contract):
 * 40N + 11M + 11 + 4*N*D; with_actions=0 drops 8N; with_obs=0 drops 4*N*D. */
DCC_API int64_t dcc_env_bytes_per_step(int32_t n_agents, int32_t n_pois, int32_t with_actions, int32_t with_obs);

/* The observation producer alone, K times from the reset state into obs [K, E, N, D] (no physics, no other output): the env
 * kernels' own store pattern.  Purpose: WHERE an output buffer lies in HBM decides how fast this pattern streams into it -- the
 * same 150-step launch takes 1.20 ms into one 6.6 GB buffer and 1.28 ms into another one of the same process, reproducibly per
 * buffer, while a sequential memset is indifferent (tools/placement_probe.py) -- so a caller that is about to stream many
 * launches into one buffer can time this call on a few candidate allocations and keep the best (dcc_hip.py:
 * HipCoverageEnv.alloc_out(placed=...)).  RESETS the env state (like dcc_env_reset); asynchronous on `stream`. */
DCC_API int dcc_env_obs_write_probe(dcc_env* env, int32_t K, float* obs, void* stream);

/* Which kernel shape obs-writing multi-step launches of this env use, as measured by dcc_env_create on THIS device (a short
 * rollout of 64 steps through each shape into a scratch buffer; batches of >= 8 MB of observation rows per step with <= 64 PoIs;
 * the fused shape is taken only when it is more than 6 % ahead, the margin covering the role-specialised shape's pipeline fill
 * in a short launch; DCC_AUTOTUNE=0 disables):
 * returns 0 = not measured (the built-in default applies), 1 = role-specialised (a physics wave + an observation wave per
 * two envs), 2 = fused (one wave per env); the measured microseconds per batched step go to us_roles / us_fused (may be
 * NULL).  Both shapes produce bit-identical outputs. */
DCC_API int dcc_env_kernel_choice(const dcc_env* env, float* us_roles, float* us_fused);

#ifdef __cplusplus
}
#endif
#endif /* DCC_ENV_H */
