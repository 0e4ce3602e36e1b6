/*
 * dcc_oracle.c -- CPU restatement (plain C, float64) of the reference's coverage-env step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker / the reported CPU baseline.  The product path (HIP, libdcc_hip.so) never calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file step by step against
 * golden vectors produced by importing the reference itself (tools/gen_golden.py, container only).
 *
 * Every function cites the reference lines it restates; paths are relative to
 * /root/reference/uav_dcc_control/ ("CW" = envs/mpe/multiagent/CoverageWorld.py,
 * "SC" = envs/mpe/multiagent/scenarios/coverage.py, "EN" = envs/mpe/multiagent/environment.py,
 * "WR" = envs/wrappers.py, "CO" = envs/mpe/multiagent/core.py).
 *
 * Floating-point conventions (must be compiled with -ffp-contract=off):
 *   - np.linalg.norm([a,b]) is x.dot(x) -> cblas_ddot -> sqrt(fma(b,b,a*a)) with the OpenBLAS
 *     Haswell/SkylakeX kernels numpy ships (measured in the fixture container: 0 mismatches in
 *     2e5 samples, 8 % mismatch for the un-fused form).  norm2() below states that.
 *   - np.sqrt(np.square(vx)+np.square(vy)) (CW:150) is three separate ufuncs: no fusion.
 *   - float32 actions stay float32 through `u *= 5.0` (EN:186-190), through the in-place force
 *     accumulation (CW:115-116,125-126: float64 sum rounded back to float32) and through
 *     `(p_force / mass) * dt` (CW:147), exactly as numpy's type promotion does it; float64
 *     actions run the same statements in float64.
 *   - np.sum(list of N float64) is numpy's pairwise sum (8-way unrolled for N >= 8).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__GNUC__)
#define DCC_API __attribute__((visibility("default")))
#else
#define DCC_API
#endif

typedef struct dcc_oracle_cfg {
    int32_t n_envs, n_agents, n_pois;
    double r_cover, r_comm, comm_r_scale, comm_force_scale;
} dcc_oracle_cfg;

typedef struct dcc_oracle {
    int E, N, M, D;
    /* scenario / world constants */
    double r_cover, r_comm, comm_r_scale;
    double contact_force;   /* CO:109 1e2, scaled by comm_force_scale at CW:16 */
    double contact_margin;  /* CO:110 */
    double dt;              /* CW:23 */
    double damping;         /* CO:107 */
    double mass;            /* CO:55,58-60 */
    double max_speed;       /* SC:54 */
    double sensitivity;     /* EN:186 */
    double m_energy;        /* SC:24 */
    double rew_cover, rew_done, rew_out; /* SC:26-29 */
    double *poi;            /* [M,2] */
    /* per-env state */
    double *pos, *vel;      /* [E,N,2] */
    double *energy;         /* [E,M] (python floats in the reference, SC:78) */
    uint8_t *done;          /* [E,M] */
    /* scratch */
    double *dist, *adj, *adj_s, *cm, *cm_s, *acc, *acc_s, *tmp;
} dcc_oracle;

/* np.linalg.norm of a 2-vector as numpy+OpenBLAS evaluate it (see header). */
static double norm2(double a, double b) { return sqrt(fma(b, b, a * a)); }

/* numpy's DOUBLE_pairwise_sum as used by np.sum on a contiguous 1-D array (n <= 128 here). */
static double np_sum(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

/* numpy's npy_logaddexp (np.logaddexp), CW:136. */
static double np_logaddexp(double x, double y)
{
    if (x == y) return x + 0.6931471805599453094172321214581766; /* LOGE2 */
    double t = x - y;
    if (t > 0) return x + log1p(exp(-t));
    if (t <= 0) return y + log1p(exp(t));
    return t; /* NaN */
}

DCC_API int dcc_oracle_obs_dim(int n_agents, int n_pois)
{
    /* SC:99-110: vel(2) pos(2) others 2(N-1) then per PoI (dx,dy,energy,m_energy,done) */
    return 4 + 2 * (n_agents - 1) + 5 * n_pois;
}

DCC_API void dcc_oracle_destroy(dcc_oracle *o)
{
    if (!o) return;
    free(o->poi); free(o->pos); free(o->vel); free(o->energy); free(o->done);
    free(o->dist); free(o->adj); free(o->adj_s); free(o->cm); free(o->cm_s);
    free(o->acc); free(o->acc_s); free(o->tmp);
    free(o);
}

/* SC:64-78 reset_world for env e: agents at the origin, zero velocity, PoIs untouched. */
static void reset_env(dcc_oracle *o, int e)
{
    memset(o->pos + (size_t)e * o->N * 2, 0, sizeof(double) * o->N * 2);
    memset(o->vel + (size_t)e * o->N * 2, 0, sizeof(double) * o->N * 2);
    memset(o->energy + (size_t)e * o->M, 0, sizeof(double) * o->M);
    memset(o->done + (size_t)e * o->M, 0, (size_t)o->M);
}

DCC_API dcc_oracle *dcc_oracle_create(const dcc_oracle_cfg *c, const double *poi_xy)
{
    if (!c || !poi_xy || c->n_envs < 1 || c->n_agents < 1 || c->n_pois < 1) return NULL;
    dcc_oracle *o = (dcc_oracle *)calloc(1, sizeof(*o));
    o->E = c->n_envs; o->N = c->n_agents; o->M = c->n_pois;
    o->D = dcc_oracle_obs_dim(o->N, o->M);
    o->r_cover = c->r_cover; o->r_comm = c->r_comm; o->comm_r_scale = c->comm_r_scale;
    o->contact_force = 1e+2;                 /* CO:109 */
    o->contact_force *= c->comm_force_scale; /* CW:16 */
    o->contact_margin = 1e-3;
    o->dt = 0.1; o->damping = 0.25; o->mass = 1.0; o->max_speed = 0.5; o->sensitivity = 5.0;
    o->m_energy = 5.0; o->rew_cover = 75.0; o->rew_done = 1500.0; o->rew_out = -100.0;
    size_t N = o->N, M = o->M, E = o->E;
    o->poi = (double *)malloc(sizeof(double) * M * 2);
    memcpy(o->poi, poi_xy, sizeof(double) * M * 2);
    o->pos = (double *)malloc(sizeof(double) * E * N * 2);
    o->vel = (double *)malloc(sizeof(double) * E * N * 2);
    o->energy = (double *)malloc(sizeof(double) * E * M);
    o->done = (uint8_t *)malloc(E * M);
    o->dist = (double *)malloc(sizeof(double) * N * N);
    o->adj = (double *)malloc(sizeof(double) * N * N);
    o->adj_s = (double *)malloc(sizeof(double) * N * N);
    o->cm = (double *)malloc(sizeof(double) * N * N);
    o->cm_s = (double *)malloc(sizeof(double) * N * N);
    o->acc = (double *)malloc(sizeof(double) * N * N);
    o->acc_s = (double *)malloc(sizeof(double) * N * N);
    o->tmp = (double *)malloc(sizeof(double) * N * N);
    for (int e = 0; e < o->E; e++) reset_env(o, e);
    return o;
}

/* SC:99-110 observation of agent i of env e into out[D] (float64, as the reference builds it). */
static void observe(const dcc_oracle *o, int e, int i, double *out)
{
    const double *pos = o->pos + (size_t)e * o->N * 2, *vel = o->vel + (size_t)e * o->N * 2;
    const double *en = o->energy + (size_t)e * o->M;
    const uint8_t *dn = o->done + (size_t)e * o->M;
    int k = 0;
    out[k++] = vel[2 * i]; out[k++] = vel[2 * i + 1];
    out[k++] = pos[2 * i]; out[k++] = pos[2 * i + 1];
    for (int a = 0; a < o->N; a++) {
        if (a == i) continue;
        out[k++] = pos[2 * a] - pos[2 * i];
        out[k++] = pos[2 * a + 1] - pos[2 * i + 1];
    }
    for (int j = 0; j < o->M; j++) {
        out[k++] = o->poi[2 * j] - pos[2 * i];
        out[k++] = o->poi[2 * j + 1] - pos[2 * i + 1];
        out[k++] = en[j];
        out[k++] = o->m_energy;
        out[k++] = dn[j] ? 1.0 : 0.0;
    }
}

static void observe_env(const dcc_oracle *o, int e, double *obs /* [N,D] */)
{
    for (int i = 0; i < o->N; i++) observe(o, e, i, obs + (size_t)i * o->D);
}

/* EN:112-122 + WR:226-232: reset every env, return obs [E,N,D]. */
DCC_API int dcc_oracle_reset(dcc_oracle *o, double *obs_out)
{
    for (int e = 0; e < o->E; e++) {
        reset_env(o, e);
        if (obs_out) observe_env(o, e, obs_out + (size_t)e * o->N * o->D);
    }
    return 0;
}

DCC_API int dcc_oracle_get_state(const dcc_oracle *o, double *pos, double *vel, double *energy, uint8_t *done)
{
    if (pos) memcpy(pos, o->pos, sizeof(double) * o->E * o->N * 2);
    if (vel) memcpy(vel, o->vel, sizeof(double) * o->E * o->N * 2);
    if (energy) memcpy(energy, o->energy, sizeof(double) * o->E * o->M);
    if (done) memcpy(done, o->done, (size_t)o->E * o->M);
    return 0;
}

DCC_API int dcc_oracle_set_state(dcc_oracle *o, const double *pos, const double *vel, const double *energy, const uint8_t *done)
{
    if (pos) memcpy(o->pos, pos, sizeof(double) * o->E * o->N * 2);
    if (vel) memcpy(o->vel, vel, sizeof(double) * o->E * o->N * 2);
    if (energy) memcpy(o->energy, energy, sizeof(double) * o->E * o->M);
    if (done) memcpy(o->done, done, (size_t)o->E * o->M);
    return 0;
}

static void matmul(const double *A, const double *B, double *C, int n)
{
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0.0;
            for (int k = 0; k < n; k++) s += A[i * n + k] * B[k * n + j];
            C[i * n + j] = s;
        }
}

/* CW:70-93 update_connect: distances on the PRE-move positions, adjacency at 2r and
 * comm_r_scale*2r (strict <), connectivity flags by summing matrix powers -- including the
 * reference's use of connect_mat[-1] (already A^k) on line 90. */
static void update_connect(dcc_oracle *o, const double *pos, int *connect, int *connect_s)
{
    int N = o->N;
    for (int a = 0; a < N; a++) {
        for (int b = 0; b < N; b++) {
            double d = norm2(pos[2 * a] - pos[2 * b], pos[2 * a + 1] - pos[2 * b + 1]);
            o->dist[a * N + b] = d;
            o->adj[a * N + b] = 0.0; o->adj_s[a * N + b] = 0.0;
            if (d < o->r_comm + o->r_comm) {
                o->adj[a * N + b] = 1.0;
                if (d < o->comm_r_scale * (o->r_comm + o->r_comm)) o->adj_s[a * N + b] = 1.0;
            }
        }
    }
    for (int a = 0; a < N; a++) { /* CW:81-83 (done inside the a-loop in the reference; same effect) */
        o->dist[a * N + a] = 1e5; o->adj[a * N + a] = 0.0; o->adj_s[a * N + a] = 0.0;
    }
    /* CW:86-93 */
    for (int i = 0; i < N * N; i++) { o->cm[i] = 0.0; }
    for (int i = 0; i < N; i++) o->cm[i * N + i] = 1.0;
    memcpy(o->acc, o->cm, sizeof(double) * N * N);
    memcpy(o->acc_s, o->cm, sizeof(double) * N * N);
    for (int t = 0; t < N - 1; t++) {
        matmul(o->cm, o->adj, o->tmp, N);          /* connect_mat.append(cm[-1] @ adj) */
        memcpy(o->cm, o->tmp, sizeof(double) * N * N);
        matmul(o->cm, o->adj_s, o->cm_s, N);        /* connect_mat_.append(connect_mat[-1] @ adj_) */
        for (int i = 0; i < N * N; i++) { o->acc[i] += o->cm[i]; o->acc_s[i] += o->cm_s[i]; }
    }
    int c = 1, cs = 1;
    for (int i = 0; i < N * N; i++) { if (!(o->acc[i] > 0)) c = 0; if (!(o->acc_s[i] > 0)) cs = 0; }
    *connect = c; *connect_s = cs;
}

/* CW:129-140 get_connect_force between agents a and b (positions pos[N,2]). */
static void get_connect_force(const dcc_oracle *o, const double *pos, int a, int b, double fa[2], double fb[2])
{
    if (a == b) { fa[0] = fa[1] = fb[0] = fb[1] = 0.0; return; }
    double dx = pos[2 * a] - pos[2 * b], dy = pos[2 * a + 1] - pos[2 * b + 1];
    double dist = norm2(dx, dy);
    double dist_max = (o->r_comm + o->r_comm) * o->comm_r_scale;
    double k = o->contact_margin;
    double pen = np_logaddexp(0.0, (dist - dist_max) / k) * k;
    double fx = o->contact_force * dx / dist * pen;
    double fy = o->contact_force * dy / dist * pen;
    fa[0] = -fx; fa[1] = -fy; fb[0] = +fx; fb[1] = +fy;
}

/* One reference env.step for env e (EN:86-110 -> CW:57-68 -> SC:80-117), followed by the
 * vec-env auto-reset (WR:104-109 / 226-232).  act points at this env's [N,2] action block. */
static void step_env(dcc_oracle *o, int e, const void *act, int act_f32,
                     double *obs, double *reward, uint8_t *env_done, uint8_t *connect_o, uint8_t *connect_s_o,
                     double *coverage, int32_t *assign, double *pos_t, double *vel_t, double *energy_t,
                     uint8_t *done_t, int32_t *force_pairs)
{
    const int N = o->N, M = o->M;
    double *pos = o->pos + (size_t)e * N * 2, *vel = o->vel + (size_t)e * N * 2;
    double *en = o->energy + (size_t)e * M;
    uint8_t *dn = o->done + (size_t)e * M;
    /* (A) EN:153-201 _set_action: u = action ; u *= 5.0 (in the action's own dtype) */
    float u32[128];
    double u64[128];
    for (int i = 0; i < 2 * N; i++) {
        if (act_f32) u32[i] = ((const float *)act)[i] * (float)o->sensitivity;
        else u64[i] = ((const double *)act)[i] * o->sensitivity;
    }
    /* (B) CW:59-60 */
    int connect = 0, connect_s = 0;
    if (o->comm_r_scale > 0) update_connect(o, pos, &connect, &connect_s);
    /* (C) CW:62-63 p_force aliases the action array; (D) CW:64-65,100-127 */
    int npairs = 0;
    if (o->contact_force > 0 && !connect_s) {
        int any_iso = 0;
        for (int a = 0; a < N; a++) {
            double s = 0.0;
            for (int b = 0; b < N; b++) s += o->adj_s[b * N + a]; /* np.sum(adj_, 0) */
            if (s == 0) any_iso = 1;
        }
        if (any_iso) {
            for (int a = 0; a < N; a++) {
                double s = 0.0;
                for (int b = 0; b < N; b++) s += o->adj_s[b * N + a];
                if (s != 0) continue;
                int b = 0; /* np.argmin(dist_mat[a,:]) -> first minimum */
                for (int k = 1; k < N; k++) if (o->dist[a * N + k] < o->dist[a * N + b]) b = k;
                double fa[2], fb[2];
                get_connect_force(o, pos, a, b, fa, fb);
                for (int c = 0; c < 2; c++) {
                    if (act_f32) { u32[2 * a + c] = (float)((double)u32[2 * a + c] + fa[c]); }
                    else u64[2 * a + c] += fa[c];
                }
                for (int c = 0; c < 2; c++) {
                    if (act_f32) { u32[2 * b + c] = (float)((double)u32[2 * b + c] + fb[c]); }
                    else u64[2 * b + c] += fb[c];
                }
                if (force_pairs && npairs < N) { force_pairs[2 * npairs] = a; force_pairs[2 * npairs + 1] = b; }
                npairs++;
            }
        } else {
            double thr = o->comm_r_scale * 2 * o->r_comm; /* CW:119 (its own rounding order) */
            for (int i = 0; i < N * N; i++) if (o->dist[i] < thr) o->dist[i] = 1e5;
            int idx = 0;
            for (int i = 1; i < N * N; i++) if (o->dist[i] < o->dist[idx]) idx = i;
            int a = idx / N, b = idx % N;
            double fa[2], fb[2];
            get_connect_force(o, pos, a, b, fa, fb);
            for (int c = 0; c < 2; c++) {
                if (act_f32) { u32[2 * a + c] = (float)((double)u32[2 * a + c] + fa[c]); }
                else u64[2 * a + c] += fa[c];
            }
            for (int c = 0; c < 2; c++) {
                if (act_f32) { u32[2 * b + c] = (float)((double)u32[2 * b + c] + fb[c]); }
                else u64[2 * b + c] += fb[c];
            }
            if (force_pairs) { force_pairs[0] = a; force_pairs[1] = b; }
            npairs = 1;
        }
    }
    if (force_pairs) for (int k = npairs; k < N; k++) { force_pairs[2 * k] = -1; force_pairs[2 * k + 1] = -1; }
    /* (E) CW:142-155 integrate_state */
    for (int i = 0; i < N; i++) {
        double vx = vel[2 * i] * (1 - o->damping), vy = vel[2 * i + 1] * (1 - o->damping);
        if (act_f32) {
            float ax = (u32[2 * i] / (float)o->mass) * (float)o->dt;
            float ay = (u32[2 * i + 1] / (float)o->mass) * (float)o->dt;
            vx += (double)ax; vy += (double)ay;
        } else {
            vx += (u64[2 * i] / o->mass) * o->dt; vy += (u64[2 * i + 1] / o->mass) * o->dt;
        }
        double speed = sqrt(vx * vx + vy * vy);
        if (speed > o->max_speed) {
            double s2 = sqrt(vx * vx + vy * vy);
            vx = vx / s2 * o->max_speed; vy = vy / s2 * o->max_speed;
        }
        vel[2 * i] = vx; vel[2 * i + 1] = vy;
        pos[2 * i] += vx * o->dt; pos[2 * i + 1] += vy * o->dt;
    }
    /* (F) CW:157-174 update_energy on POST-move positions */
    uint8_t just[4096];
    int num_done = 0;
    for (int j = 0; j < M; j++) {
        just[j] = 0;
        if (dn[j]) { num_done++; continue; }
        for (int i = 0; i < N; i++) {
            double d = norm2(o->poi[2 * j] - pos[2 * i], o->poi[2 * j + 1] - pos[2 * i + 1]);
            if (d <= o->r_cover) en[j] += 1;
        }
        if (en[j] >= o->m_energy) { dn[j] = 1; just[j] = 1; num_done++; }
    }
    double cov = (double)num_done / (double)M;
    /* PoI-assignment index: argmin_i ||x_i - p_j|| (first minimum), SC:84-85's min() made explicit */
    if (assign) {
        for (int j = 0; j < M; j++) {
            int best = 0; double bd = 0;
            for (int i = 0; i < N; i++) {
                double d = norm2(pos[2 * i] - o->poi[2 * j], pos[2 * i + 1] - o->poi[2 * j + 1]);
                if (i == 0 || d < bd) { bd = d; best = i; }
            }
            assign[j] = best;
        }
    }
    /* (G) SC:80-97 reward, called once per agent in order (EN:98-103); `just` is cleared by the
     * first call (SC:87-89).  (H) EN:106-108: np.sum over the N values, shared by all agents. */
    double rn[128];
    int all_done = 1;
    for (int j = 0; j < M; j++) if (!dn[j]) all_done = 0;
    for (int call = 0; call < N; call++) {
        double rew = 0.0;
        for (int j = 0; j < M; j++) {
            if (!dn[j]) {
                double mn = 0;
                for (int i = 0; i < N; i++) {
                    double d = norm2(pos[2 * i] - o->poi[2 * j], pos[2 * i + 1] - o->poi[2 * j + 1]);
                    if (i == 0 || d < mn) mn = d;
                }
                rew -= mn;
            } else if (just[j]) {
                rew += o->rew_cover;
                just[j] = 0;
            }
        }
        if (all_done) rew += o->rew_done;
        for (int i = 0; i < N; i++) {
            double ax = fabs(pos[2 * i]), ay = fabs(pos[2 * i + 1]);
            double s = 0.0; /* np.sum(abs_pos[abs_pos > 1] - 1) */
            if (ax > 1) s += ax - 1;
            if (ay > 1) s += ay - 1;
            rew += s * o->rew_out;
            if (ax > 1.5 || ay > 1.5) rew += o->rew_out;
        }
        rn[call] = rew;
    }
    double R = np_sum(rn, N);
    /* SC:112-117 done */
    int dflag = all_done;
    for (int i = 0; i < 2 * N; i++) if (fabs(pos[i]) > 1.5) dflag = 1;

    if (reward) *reward = R;
    if (env_done) *env_done = (uint8_t)dflag;
    if (connect_o) *connect_o = (uint8_t)connect;
    if (connect_s_o) *connect_s_o = (uint8_t)connect_s;
    if (coverage) *coverage = cov;
    if (pos_t) memcpy(pos_t, pos, sizeof(double) * N * 2);
    if (vel_t) memcpy(vel_t, vel, sizeof(double) * N * 2);
    if (energy_t) memcpy(energy_t, en, sizeof(double) * M);
    if (done_t) memcpy(done_t, dn, (size_t)M);
    /* (I) WR:104-109: np.all(done) -> reset, and the returned obs is the reset obs */
    if (dflag) reset_env(o, e);
    if (obs) observe_env(o, e, obs);
}

/* Batched step over the E envs (one reference env.step each + auto-reset).
 * actions: [E,N,2] float32 (act_f32=1) or float64.  Any output pointer may be NULL.
 *   obs [E,N,D] f64 (post-reset, what the vec-env returns), reward [E], done [E], connect [E],
 *   connect_s [E], coverage [E], assign [E,M] int32,
 *   pos_t/vel_t [E,N,2], energy_t [E,M], done_t [E,M]: the state AFTER the step but BEFORE the
 *   auto-reset (terminal state), force_pairs [E,N,2] int32 (-1 padded). */
DCC_API int dcc_oracle_step(dcc_oracle *o, const void *actions, int act_f32,
                            double *obs, double *reward, uint8_t *done, uint8_t *connect, uint8_t *connect_s,
                            double *coverage, int32_t *assign,
                            double *pos_t, double *vel_t, double *energy_t, uint8_t *done_t, int32_t *force_pairs)
{
    if (!o || !actions || o->N > 64 || o->M > 4096) return -1;
    const size_t N = o->N, M = o->M, D = o->D;
    const size_t asz = act_f32 ? sizeof(float) : sizeof(double);
    for (int e = 0; e < o->E; e++) {
        step_env(o, e, (const char *)actions + (size_t)e * N * 2 * asz, act_f32,
                 obs ? obs + e * N * D : NULL, reward ? reward + e : NULL, done ? done + e : NULL,
                 connect ? connect + e : NULL, connect_s ? connect_s + e : NULL,
                 coverage ? coverage + e : NULL, assign ? assign + e * M : NULL,
                 pos_t ? pos_t + e * N * 2 : NULL, vel_t ? vel_t + e * N * 2 : NULL,
                 energy_t ? energy_t + e * M : NULL, done_t ? done_t + e * M : NULL,
                 force_pairs ? force_pairs + e * N * 2 : NULL);
    }
    return 0;
}

/* ---- synthetic-action generator (NOT from the reference: the build's own counter-based RNG,
 * restated here so that the in-kernel generator of the HIP rollout can be checked) -------------
 * action(seed, step, env, agent) = two uniforms in [-1, 1) with 24-bit resolution, taken from the
 * high and low words of splitmix64(seed + GOLDEN * (1 + ((step * E + env) * N + agent))). */
static uint64_t splitmix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

DCC_API void dcc_oracle_rng_actions(uint64_t seed, uint32_t step, int n_envs, int n_agents, int env0, int env_total,
                                    float *out /* [n_envs, n_agents, 2] */)
{
    /* env0/env_total: global env index offset and global env count (multi-GPU sharding keeps the
     * stream a function of the GLOBAL env id). */
    for (int e = 0; e < n_envs; e++)
        for (int i = 0; i < n_agents; i++) {
            uint64_t idx = ((uint64_t)step * (uint64_t)env_total + (uint64_t)(env0 + e)) * (uint64_t)n_agents + (uint64_t)i;
            uint64_t z = splitmix64(seed + 0x9E3779B97F4A7C15ULL * (idx + 1));
            uint32_t hi = (uint32_t)(z >> 40), lo = (uint32_t)((z & 0xFFFFFFFFULL) >> 8);
            out[((size_t)e * n_agents + i) * 2 + 0] = (float)hi * (1.0f / 8388608.0f) - 1.0f;
            out[((size_t)e * n_agents + i) * 2 + 1] = (float)lo * (1.0f / 8388608.0f) - 1.0f;
        }
}

/* K fused steps with generated actions (the CPU twin of dcc_env_rollout's rng mode); only the
 * final state and per-step reward/done/coverage are kept.  Used by bench.py's cpu_baseline leg
 * and by the rollout parity tests. */
DCC_API int dcc_oracle_rollout_rng(dcc_oracle *o, int K, uint64_t seed, uint32_t step0, int env0, int env_total,
                                   double *reward /* [K,E] */, uint8_t *done /* [K,E] */, double *coverage /* [K,E] */,
                                   double *obs_last /* [E,N,D] or NULL */)
{
    float *act = (float *)malloc(sizeof(float) * (size_t)o->E * o->N * 2);
    for (int k = 0; k < K; k++) {
        dcc_oracle_rng_actions(seed, step0 + (uint32_t)k, o->E, o->N, env0, env_total, act);
        dcc_oracle_step(o, act, 1, (k == K - 1) ? obs_last : NULL,
                        reward ? reward + (size_t)k * o->E : NULL, done ? done + (size_t)k * o->E : NULL, NULL, NULL,
                        coverage ? coverage + (size_t)k * o->E : NULL, NULL, NULL, NULL, NULL, NULL, NULL);
    }
    free(act);
    return 0;
}
