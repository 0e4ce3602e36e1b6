/*
 * dcc_env_cpu.c -- the CPU restatement behind the SAME C-ABI as the HIP library ("a `_cpu` twin of each
 * entry point taking host pointers").  TEST INFRASTRUCTURE: built into oracle/libdcc_oracle.so, never linked into or
 * loaded by the product (libdcc_hip.so has no CPU path).  Each dcc_env_*_cpu function has the signature of its
 * include/dcc_env.h namesake -- the same dcc_env_cfg / dcc_env_out structs -- with HOST pointers in place of device
 * pointers and `stream` ignored, and runs the reference restatement of dcc_oracle.c (float64 state, the reference's
 * operation order).  Uses: bench.py's cpu_baseline leg times the CPU through the very ABI the GPU numbers go through, and
 * tests/test_cpu_twin_abi.py drives both libraries with one binding.  The reference's in-process analogue is DummyVecEnv
 * (uav_dcc_control/envs/wrappers.py:204-261).
 */
#include "dcc_oracle.c"
#include "dcc_gae_cpu.c"   /* the twin of include/dcc_gae.h, same library */

#include "../include/dcc_env.h"

typedef struct dcc_env_cpu {
    dcc_oracle *o;
    double *obs64, *rew64, *cov64;   /* per-step scratch in the oracle's types */
    int32_t *assign32;
    float *act;                      /* generated actions of one step (rollout with actions == NULL) */
} dcc_env_cpu;

static _Thread_local char g_cpu_err[256];
static int cpu_fail(int code, const char *msg)
{
    strncpy(g_cpu_err, msg, sizeof(g_cpu_err) - 1);
    g_cpu_err[sizeof(g_cpu_err) - 1] = 0;
    return code;
}

DCC_API const char *dcc_last_error_cpu(void) { return g_cpu_err; }

DCC_API int dcc_env_destroy_cpu(dcc_env *env)
{
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return DCC_OK;
    dcc_oracle_destroy(c->o);
    free(c->obs64); free(c->rew64); free(c->cov64); free(c->assign32); free(c->act);
    free(c);
    return DCC_OK;
}

DCC_API int dcc_env_create_cpu(const dcc_env_cfg *cfg, dcc_env **out)
{
    if (!cfg || !out) return cpu_fail(DCC_EINVAL, "dcc_env_create_cpu: null argument");
    *out = NULL;
    if (cfg->n_envs < 1 || cfg->n_agents < 1 || cfg->n_agents > DCC_MAX_AGENTS || cfg->n_pois < 1 ||
        cfg->n_pois > DCC_MAX_POIS || !cfg->poi_xy)
        return cpu_fail(DCC_EINVAL, "dcc_env_create_cpu: bad sizes or poi_xy is NULL");
    if (cfg->bound_soft != 1.0 || cfg->bound_hard != 1.5)
        return cpu_fail(DCC_EUNSUPPORTED, "dcc_env_create_cpu: the restatement keeps the reference's arena bounds (1.0 / 1.5)");
    dcc_oracle_cfg oc = {cfg->n_envs, cfg->n_agents, cfg->n_pois, cfg->r_cover, cfg->r_comm, cfg->comm_r_scale,
                         cfg->comm_force_scale};
    dcc_env_cpu *c = (dcc_env_cpu *)calloc(1, sizeof(*c));
    if (!c) return cpu_fail(DCC_ENOMEM, "dcc_env_create_cpu: out of memory");
    c->o = dcc_oracle_create(&oc, cfg->poi_xy);
    if (!c->o) { free(c); return cpu_fail(DCC_ENOMEM, "dcc_env_create_cpu: dcc_oracle_create failed"); }
    dcc_oracle *o = c->o;           /* the remaining constants of dcc_env_cfg (the oracle's defaults are the reference's) */
    o->dt = cfg->dt; o->damping = cfg->damping; o->max_speed = cfg->max_speed; o->sensitivity = cfg->sensitivity;
    o->mass = cfg->mass; o->contact_margin = cfg->contact_margin; o->m_energy = cfg->m_energy;
    o->rew_cover = cfg->rew_cover; o->rew_done = cfg->rew_done; o->rew_out = cfg->rew_out;
    const size_t E = o->E, N = o->N, M = o->M, D = o->D;
    c->obs64 = (double *)malloc(sizeof(double) * E * N * D);
    c->rew64 = (double *)malloc(sizeof(double) * E);
    c->cov64 = (double *)malloc(sizeof(double) * E);
    c->assign32 = (int32_t *)malloc(sizeof(int32_t) * E * M);
    c->act = (float *)malloc(sizeof(float) * E * N * 2);
    if (!c->obs64 || !c->rew64 || !c->cov64 || !c->assign32 || !c->act) {
        dcc_env_destroy_cpu((dcc_env *)c);
        return cpu_fail(DCC_ENOMEM, "dcc_env_create_cpu: out of memory");
    }
    *out = (dcc_env *)c;
    return DCC_OK;
}

DCC_API int dcc_env_obs_dim_cpu(const dcc_env *env) { return env ? ((const dcc_env_cpu *)env)->o->D : DCC_EINVAL; }

DCC_API int dcc_env_reset_cpu(dcc_env *env, float *obs, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_env_reset_cpu: null env");
    dcc_oracle_reset(c->o, c->obs64);
    if (obs) {
        const size_t n = (size_t)c->o->E * c->o->N * c->o->D;
        for (size_t i = 0; i < n; i++) obs[i] = (float)c->obs64[i];
    }
    return DCC_OK;
}

/* one batched step; outputs written at step offset k of arrays with a leading step dimension */
static int step_k(dcc_env_cpu *c, const void *actions, int act_f32, const dcc_env_out *out, size_t k)
{
    dcc_oracle *o = c->o;
    const size_t E = o->E, N = o->N, M = o->M, D = o->D;
    const int want_obs = out && out->obs, want_assign = out && out->assign;
    const int rc = dcc_oracle_step(o, actions, act_f32, want_obs ? c->obs64 : NULL, c->rew64,
                                   out && out->done ? out->done + k * E : NULL, out && out->connect ? out->connect + k * E : NULL,
                                   out && out->connect_s ? out->connect_s + k * E : NULL, c->cov64,
                                   want_assign ? c->assign32 : NULL, NULL, NULL, NULL, NULL, NULL);
    if (rc != 0) return cpu_fail(DCC_EINVAL, "dcc_env_step_cpu: dcc_oracle_step failed");
    if (!out) return DCC_OK;
    if (want_obs) { float *dst = out->obs + k * E * N * D; for (size_t i = 0; i < E * N * D; i++) dst[i] = (float)c->obs64[i]; }
    if (want_assign) { uint8_t *dst = out->assign + k * E * M; for (size_t i = 0; i < E * M; i++) dst[i] = (uint8_t)c->assign32[i]; }
    for (size_t e = 0; e < E; e++) {
        if (out->reward) out->reward[k * E + e] = (float)c->rew64[e];
        if (out->reward64) out->reward64[k * E + e] = c->rew64[e];
        if (out->coverage) out->coverage[k * E + e] = (float)c->cov64[e];
    }
    /* compact state AFTER the auto-reset: what the observations were built from */
    if (out->state_pos) memcpy(out->state_pos + k * E * N * 2, o->pos, sizeof(double) * E * N * 2);
    if (out->state_vel) memcpy(out->state_vel + k * E * N * 2, o->vel, sizeof(double) * E * N * 2);
    if (out->state_energy) for (size_t i = 0; i < E * M; i++) out->state_energy[k * E * M + i] = (float)o->energy[i];
    if (out->state_done) memcpy(out->state_done + k * E * M, o->done, E * M);
    return DCC_OK;
}

DCC_API int dcc_env_step_cpu(dcc_env *env, const void *actions, int act_dtype, const dcc_env_out *out, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_env_step_cpu: null env");
    if (!actions) return cpu_fail(DCC_EINVAL, "dcc_env_step_cpu: actions is NULL");
    if (act_dtype != DCC_ACT_F32 && act_dtype != DCC_ACT_F64) return cpu_fail(DCC_EINVAL, "dcc_env_step_cpu: bad act_dtype");
    return step_k(c, actions, act_dtype == DCC_ACT_F32, out, 0);
}

DCC_API int dcc_env_rollout_cpu(dcc_env *env, int32_t K, const float *actions, uint64_t seed, uint32_t step0, int32_t env0,
                                int32_t env_total, const dcc_env_out *out, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_env_rollout_cpu: null env");
    if (K < 1) return cpu_fail(DCC_EINVAL, "dcc_env_rollout_cpu: K must be >= 1");
    if (!actions && (env0 < 0 || env_total < env0 + c->o->E))
        return cpu_fail(DCC_EINVAL, "dcc_env_rollout_cpu: env0/env_total do not cover this shard");
    const size_t stride = (size_t)c->o->E * c->o->N * 2;
    for (int32_t k = 0; k < K; k++) {
        const float *a = actions ? actions + (size_t)k * stride : c->act;
        if (!actions) dcc_oracle_rng_actions(seed, step0 + (uint32_t)k, c->o->E, c->o->N, env0, env_total, c->act);
        const int rc = step_k(c, a, 1, out, (size_t)k);
        if (rc != DCC_OK) return rc;
    }
    return DCC_OK;
}

DCC_API int dcc_env_get_state_cpu(dcc_env *env, double *pos, double *vel, float *energy, uint8_t *done, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_env_get_state_cpu: null env");
    const dcc_oracle *o = c->o;
    const size_t E = o->E, N = o->N, M = o->M;
    if (pos) memcpy(pos, o->pos, sizeof(double) * E * N * 2);
    if (vel) memcpy(vel, o->vel, sizeof(double) * E * N * 2);
    if (energy) for (size_t i = 0; i < E * M; i++) energy[i] = (float)o->energy[i];
    if (done) memcpy(done, o->done, E * M);
    return DCC_OK;
}

DCC_API int dcc_env_set_state_cpu(dcc_env *env, const double *pos, const double *vel, const float *energy, const uint8_t *done,
                                  void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_env_set_state_cpu: null env");
    dcc_oracle *o = c->o;
    const size_t E = o->E, N = o->N, M = o->M;
    if (pos) memcpy(o->pos, pos, sizeof(double) * E * N * 2);
    if (vel) memcpy(o->vel, vel, sizeof(double) * E * N * 2);
    if (energy) for (size_t i = 0; i < E * M; i++) o->energy[i] = (double)energy[i];
    if (done) memcpy(o->done, done, E * M);
    return DCC_OK;
}

/* The twin of dcc_obs_expand: Scenario.observation (SC:99-110) of every agent for n independent states given as arrays (pos / vel [n,N,2]
 * f64, energy [n,M] f32, done [n,M] u8) -> obs [n,N,D] float32 (the cast SharedReplayBuffer applies, shared_buffer.py:40).  Same feature order
 * as observe() in dcc_oracle.c: vel_i, pos_i, (pos_a - pos_i) for a != i, then per PoI (poi_j - pos_i, energy_j, m_energy, done_j).  Uses the
 * PoI table and constants of `env`; does not touch its state. */
DCC_API int dcc_obs_expand_cpu(dcc_env *env, int64_t n, const double *pos, const double *vel, const float *energy, const uint8_t *done,
                               float *obs, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_obs_expand_cpu: null env");
    if (n < 1 || !pos || !vel || !energy || !done || !obs) return cpu_fail(DCC_EINVAL, "dcc_obs_expand_cpu: bad argument");
    const dcc_oracle *o = c->o;
    const int N = o->N, M = o->M, D = o->D;
    for (int64_t s = 0; s < n; s++) {
        const double *p = pos + (size_t)s * N * 2, *v = vel + (size_t)s * N * 2;
        const float *en = energy + (size_t)s * M;
        const uint8_t *dn = done + (size_t)s * M;
        for (int i = 0; i < N; i++) {
            float *out = obs + ((size_t)s * N + i) * D;
            int k = 0;
            out[k++] = (float)v[2 * i]; out[k++] = (float)v[2 * i + 1];
            out[k++] = (float)p[2 * i]; out[k++] = (float)p[2 * i + 1];
            for (int a = 0; a < N; a++) {
                if (a == i) continue;
                out[k++] = (float)(p[2 * a] - p[2 * i]);
                out[k++] = (float)(p[2 * a + 1] - p[2 * i + 1]);
            }
            for (int j = 0; j < M; j++) {
                out[k++] = (float)(o->poi[2 * j] - p[2 * i]);
                out[k++] = (float)(o->poi[2 * j + 1] - p[2 * i + 1]);
                out[k++] = (float)(double)en[j];
                out[k++] = (float)o->m_energy;
                out[k++] = dn[j] ? 1.0f : 0.0f;
            }
        }
    }
    return DCC_OK;
}

/* The twins of dcc_obs_features / dcc_obs_features_x (include/dcc_env.h): the compact policy-input features of n states, defined by
 * the observation rows of coverage.py:99-110 -- head = the first 4 + 2(N-1) columns of every row, poi_feat = the agent-independent
 * (energy, done) columns, stats = (mean, sum of squared deviations) of the D float32 values of every row in float64 (two passes),
 * cstats = the same moments of the centralised row pooled from the per-row moments, xa / xc = the input matrices of the per-env
 * GEMMs of the structured first layers ([energy | done | 1 | 0..], [head_0..head_{N-1} | energy | done | 1 | 0..], rows padded to a
 * multiple of 8 floats).  Any output may be NULL.  No counterpart in the reference beyond the rows themselves; here so that a
 * caller of the shipped (state-only, structured-input) configuration can swap libraries like for every other entry point. */
static int pad8(int k) { return (k + 7) / 8 * 8; }

DCC_API int dcc_obs_features_x_cpu(dcc_env *env, int64_t n, const double *pos, const double *vel, const float *energy, const uint8_t *done,
                                   float *head, float *poi_feat, double *stats, double *cstats, float *xa, float *xc, void *stream)
{
    (void)stream;
    dcc_env_cpu *c = (dcc_env_cpu *)env;
    if (!c) return cpu_fail(DCC_EINVAL, "dcc_obs_features_cpu: null env");
    if (n < 1 || !pos || !vel || !energy || !done) return cpu_fail(DCC_EINVAL, "dcc_obs_features_cpu: bad argument");
    const dcc_oracle *o = c->o;
    const int N = o->N, M = o->M, D = o->D, HD = 4 + 2 * (N - 1);
    const int ka = pad8(2 * M + 1), kc = pad8(N * HD + 2 * M + 1);
    float *rows = (float *)malloc(sizeof(float) * (size_t)N * D);
    double *mean_i = (double *)malloc(sizeof(double) * N), *m2_i = (double *)malloc(sizeof(double) * N);
    if (!rows || !mean_i || !m2_i) { free(rows); free(mean_i); free(m2_i); return cpu_fail(DCC_ENOMEM, "dcc_obs_features_cpu: out of memory"); }
    int rc = DCC_OK;
    for (int64_t s = 0; s < n && rc == DCC_OK; s++) {
        rc = dcc_obs_expand_cpu(env, 1, pos + (size_t)s * N * 2, vel + (size_t)s * N * 2, energy + (size_t)s * M, done + (size_t)s * M, rows, NULL);
        if (rc != DCC_OK) break;
        for (int i = 0; i < N; i++) {
            const float *r = rows + (size_t)i * D;
            double sum = 0.0, m2 = 0.0;
            for (int k = 0; k < D; k++) sum += (double)r[k];
            const double mean = sum / (double)D;
            for (int k = 0; k < D; k++) { const double d = (double)r[k] - mean; m2 += d * d; }
            mean_i[i] = mean; m2_i[i] = m2;
            if (head) memcpy(head + ((size_t)s * N + i) * HD, r, sizeof(float) * HD);
            if (stats) { stats[((size_t)s * N + i) * 2] = mean; stats[((size_t)s * N + i) * 2 + 1] = m2; }
            if (xc) memcpy(xc + (size_t)s * kc + (size_t)i * HD, r, sizeof(float) * HD);
        }
        if (cstats) {
            double mc = 0.0, m2c = 0.0;
            for (int i = 0; i < N; i++) mc += mean_i[i];
            mc /= (double)N;
            for (int i = 0; i < N; i++) m2c += m2_i[i] + (double)D * (mean_i[i] - mc) * (mean_i[i] - mc);
            cstats[(size_t)s * 2] = mc; cstats[(size_t)s * 2 + 1] = m2c;
        }
        const float *en = energy + (size_t)s * M;
        const uint8_t *dn = done + (size_t)s * M;
        for (int j = 0; j < M; j++) {
            const float e = en[j], d = dn[j] ? 1.0f : 0.0f;
            if (poi_feat) { poi_feat[(size_t)s * 2 * M + j] = e; poi_feat[(size_t)s * 2 * M + M + j] = d; }
            if (xa) { xa[(size_t)s * ka + j] = e; xa[(size_t)s * ka + M + j] = d; }
            if (xc) { xc[(size_t)s * kc + N * HD + j] = e; xc[(size_t)s * kc + N * HD + M + j] = d; }
        }
        if (xa) { xa[(size_t)s * ka + 2 * M] = 1.0f; for (int k = 2 * M + 1; k < ka; k++) xa[(size_t)s * ka + k] = 0.0f; }
        if (xc) { xc[(size_t)s * kc + N * HD + 2 * M] = 1.0f; for (int k = N * HD + 2 * M + 1; k < kc; k++) xc[(size_t)s * kc + k] = 0.0f; }
    }
    free(rows); free(mean_i); free(m2_i);
    return rc;
}

DCC_API int dcc_obs_features_cpu(dcc_env *env, int64_t n, const double *pos, const double *vel, const float *energy, const uint8_t *done,
                                 float *head, float *poi_feat, double *stats, double *cstats, void *stream)
{
    return dcc_obs_features_x_cpu(env, n, pos, vel, energy, done, head, poi_feat, stats, cstats, NULL, NULL, stream);
}
