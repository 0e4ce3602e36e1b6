/*
 * dcc_mlp.h -- C-ABI of the fused element-wise stages of the MAPPO policy trunks in libdcc_hip.so.
 *
 * The reference's actor/critic trunk is  LayerNorm(in) -> [Linear -> ReLU -> LayerNorm] x 2
 * (uav_dcc_control/algos/algo_utils/mlp.py:7-58; called by r_actor_critic.py:43-57,59-79,111-121 from
 * Learner.collect, learner.py:227-252, and MAPPOTrainer.ppo_update, algos/mappo.py:133-187).  With 4.9 M agent rows
 * per PPO epoch (8 UAV x 64 PoI x 4096 envs x 150 steps) every [rows, 256] activation is 5 GB, and the update is
 * bound by the number of passes over such tensors, not by its GEMMs.  These entry points replace chains of
 * element-wise / LayerNorm kernels by single passes:
 *
 *   dcc_relu_ln_fwd / _bwd      h = LayerNorm(ReLU(z))   one read + one write;  backward recomputes the
 *                               LayerNorm statistics from z instead of storing them
 *   dcc_relu_ln_head_fwd / _bwd the last block's tail fused with the narrow output layer that follows it (action
 *                               mean / value): y = LayerNorm(ReLU(z)) Wo^T + bo; h is never stored
 *   dcc_actor_l1_fwd / _bwd     the actor's first block evaluated straight from the compact features of
 *                               dcc_obs_features (include/dcc_env.h):
 *                                   z = rstd_in * (head . Wh^T + G[env] - mean_in * s) + c ;  h = LayerNorm(ReLU(z))
 *                               (algebra in dynamic-coverage-control_amd/algos/algo_utils/structured.py); z is never stored
 *
 * All pointers are DEVICE pointers to contiguous float32 (float64 for `stats`) arrays; calls are asynchronous on
 * `stream`; return 0, or DCC_EINVAL (-1) / DCC_EHIP (-2) / DCC_EUNSUPPORTED (-4: shape outside the compiled
 * variants -- the caller keeps its unfused path).  Parameter gradients are reduced in a fixed order (per-wave
 * partial sums in `workspace`, then one deterministic pass), so results are reproducible run to run.
 */
#ifndef DCC_MLP_H
#define DCC_MLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef DCC_API
#if defined(__GNUC__)
#define DCC_API __attribute__((visibility("default")))
#else
#define DCC_API
#endif
#endif

/* h[R,H] = LayerNorm_{gamma,beta,eps}(max(z + bias, 0)) over the last axis (mlp.py:13-16 block tail: the Linear's bias
 * add, ReLU, LayerNorm); bias [H] may be NULL. */
DCC_API int dcc_relu_ln_fwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps, float* h,
                            int64_t R, int32_t H, void* stream);

/* Floats of workspace the backward calls below need for a given problem (0 if unsupported). */
DCC_API int64_t dcc_mlp_workspace_floats(int32_t H, int32_t HD);

/* Given dh = dL/dh: dz[R,H] = dL/dz; dparams [3,H] = rows dgamma, dbeta, dbias (= column sums of dz, i.e. the gradient of
 * the producing Linear's bias, free here) -- written, not accumulated. */
DCC_API int dcc_relu_ln_bwd(const float* z, const float* bias, const float* gamma, const float* dh, float eps, float* dz,
                            float* dparams, float* workspace, int64_t R, int32_t H, void* stream);

/* y[R,A] = LayerNorm(ReLU(z + bias)) . Wo^T + bo for a narrow head (A <= 4: the Gaussian mean, distributions.py:83-92
 * fc_mean, or the value head, r_actor_critic.py:109 v_out) -- the normalised activations are not stored.  bias and bo may
 * be NULL. */
DCC_API int dcc_relu_ln_head_fwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps,
                                 const float* Wo, const float* bo, float* y, int64_t R, int32_t H, int32_t A,
                                 void* stream);

/* Backward given dy [R,A]: dz [R,H]; dparams [(3+A),H] = rows dgamma, dbeta, dbias, dWo[0..A-1] (d bo = column sums of dy
 * is left to the caller). */
DCC_API int dcc_relu_ln_head_bwd(const float* z, const float* bias, const float* gamma, const float* beta, float eps,
                                 const float* Wo, const float* dy, float* dz, float* dparams, float* workspace, int64_t R,
                                 int32_t H, int32_t A, void* stream);

/*
 * PPO-clip policy surrogate of a diagonal Gaussian policy, value and gradient in one pass over the batch
 * (algos/mappo.py:150-160 + Normal.log_prob via algo_utils/act.py:165-172 and distributions.py:72-81):
 *   logp_r = sum_d Normal(mean[r,d], exp(logstd[d])).log_prob(actions[r,d]);  ratio[r,k] = exp(logp_r - old_logp[r,k])
 *   surr_r = sum_k min(ratio adv_r, clamp(ratio, 1-clip, 1+clip) adv_r)
 * mean, actions [R,A]; logstd [A]; old_logp [R,K] (the reference stores K = A identical columns);
 * adv [R]; active [R] or NULL (= all ones).  A, K <= 4.  Outputs:
 *   dmean [R,A]  = d( -sum_r active_r surr_r ) / d mean        (the caller scales by 1 / sum active, or 1 / R)
 *   sums  [8]    = { sum_r active_r surr_r, sum_r active_r, sum_{r,k} ratio, 0, d(-sum active surr)/d logstd[0..3] }
 * workspace: >= 8 * 2048 floats.  Gradient conventions are PyTorch's (minimum splits ties evenly, clamp passes the
 * gradient on the closed interval), so the result equals autograd on the unfused expression.
 */
DCC_API int dcc_ppo_policy_loss(const float* mean, const float* logstd, const float* actions, const float* old_logp,
                                const float* adv, const float* active, float clip, float* dmean, float* sums,
                                float* workspace, int64_t R, int32_t A, int32_t K, void* stream);

/*
 * Clipped value loss of the centralised critic (MAPPOTrainer.cal_value_loss, algos/mappo.py:103-131, with the one-sided
 * Huber of utils/util.py:36-39 when delta > 0, e^2/2 when delta <= 0), value and gradient in one pass:
 *   values [n]: one critic output per env-step, shared by its N agent rows r = e*N + i;  value_preds, returns, active [n*N]
 *   (active may be NULL = all ones);  norm = {mean, std} of ValueNorm (utils/valuenorm.py:57-66) or NULL
 *   loss_r = max(h(target_r - v_e), h(target_r - (vp_r + clamp(v_e - vp_r, -clip, clip))))  (use_clipped = 0: first term only)
 * Outputs: dvalues [n] = d(sum_r active_r loss_r)/d values;  sums [2] = {sum_r active_r loss_r, sum_r active_r}.
 * workspace: >= 2 * 2048 floats.
 */
DCC_API int dcc_ppo_value_loss(const float* values, const float* value_preds, const float* returns, const float* active,
                               const float* norm, float clip, float delta, int32_t use_clipped, float* dvalues, float* sums,
                               float* workspace, int64_t n, int32_t N, void* stream);

/*
 * Rollout glue (Learner.collect / insert, learner.py:227-276; SharedReplayBuffer.insert, buffer/shared_buffer.py:72-105).
 * dcc_rollout_sample: actions = mean + exp(logstd) * eps (FixedNormal.sample), logp = sum_d Normal.log_prob (written to
 *   all K columns of the buffer's [R,K] log-prob slot), value_preds[r] = value[r / N] (one critic value per env broadcast
 *   over its N agents; value_preds may be NULL).  mean, eps, actions [R,A]; R = E*N rows; A, K <= 4.
 * dcc_rollout_record: rewards[r] = reward[r / N], masks_next[r] = 1 - done[r / N] from the env step's [E] outputs.
 */
DCC_API int dcc_rollout_sample(const float* mean, const float* logstd, const float* eps, const float* value, float* actions,
                               float* logp, float* value_preds, int64_t R, int32_t N, int32_t A, int32_t K, void* stream);
DCC_API int dcc_rollout_record(const float* reward, const uint8_t* done, float* rewards, float* masks_next, int64_t R,
                               int32_t N, void* stream);
/* The same, and the per-env statistics the learner logs (reference learner.py:187-193: episode reward sum, coverage rate)
 * kept in the same launch, element-wise per env: rew_acc[e] += reward[e] (f64), cov_max[e] = max(cov_max[e], coverage[e]).
 * rew_acc / cov_max / coverage may be NULL (cov_max needs coverage). */
DCC_API int dcc_rollout_record_stats(const float* reward, const uint8_t* done, const float* coverage, float* rewards,
                                     float* masks_next, double* rew_acc, float* cov_max, int64_t R, int32_t N, void* stream);

/*
 * Actor first block from compact features of n env states with N agents each (rows r = e*N + i):
 *   head  [n,N,HD] f32, HD = 4 + 2(N-1) <= 128 (two head registers per lane: up to 63 UAVs; Wh^T [HD,H] in LDS, <= 160 KB)
 *                                            dcc_obs_features
 *   G     [n,H]    f32                       per-env term  poi_feat . [We;Wd]^T + const   (shared by the N agents)
 *   stats [n,N,2]  f64 or NULL               (mean, sum sq. dev.) of each observation row; NULL = no input LayerNorm
 *   Wh [H,HD], s [H] (row sums of the folded weight), c [H] (folded bias);  D = observation width, eps_in = input-LN eps
 *   gamma, beta [H], eps_ln                  the block's LayerNorm
 *   h     [n*N,H]  f32 out
 * HD = 0 (N = 1; head / Wh are then never dereferenced, pass any valid pointer): no per-row term -- z = rstd_in * (G - mean_in
 * s) + c with one row per env: the first-block tail of the centralised critic, whose whole per-env GEMM output is G.
 * 4 or 8 UAVs with H = 256 (the BASELINE sizes) take kernels that hold Wh^T in registers and fetch an env ahead.
 */
DCC_API int dcc_actor_l1_fwd(const float* head, const float* G, const double* stats, const float* Wh, const float* s,
                             const float* c, const float* gamma, const float* beta, float eps_in, float eps_ln,
                             int32_t D, float* h, int64_t n, int32_t N, int32_t HD, int32_t H, void* stream);

/* Backward of the above given dh [n*N,H]: dG [n,H], dWh [H,HD], ds [H], dc [H], dgamma [H], dbeta [H].
 * dq == NULL: one kernel, dWh accumulated in registers (generic sizes: ~100 per lane next to the LDS-resident Wh^T, slow;
 * 4 / 8 UAVs with H = 256: Wh^T and the accumulators both in registers, the fastest form -- 2.85 ms at 4.9 M rows).
 * dq != NULL ([n*N,H]): the kernel stores q = rstd_in * dL/dz there instead and leaves dWh untouched -- the caller
 * forms dWh = q^T head as a (split-K) GEMM; without the accumulators the kernel runs at full occupancy, which is
 * faster for long batches even with the extra [rows,H] write (2.7 vs 7.1 ms at 4.9 M rows).  HD = 0: dWh and dq may both be
 * NULL (dG is then the gradient of the per-env GEMM output).  HD > 40: only the q-storing form (dq != NULL). */
DCC_API int dcc_actor_l1_bwd(const float* head, const float* G, const double* stats, const float* Wh, const float* s,
                             const float* c, const float* gamma, const float* dh, float eps_in, float eps_ln, int32_t D,
                             float* dG, float* dWh, float* dq, float* ds, float* dc, float* dgamma, float* dbeta,
                             float* workspace, int64_t n, int32_t N, int32_t HD, int32_t H, void* stream);

/*
 * The same block with the per-row head term supplied by the caller:
 *   pre [n*N,H] f32 = head . Wh^T   (a library GEMM over the [n*N,HD] head values; its autograd owns dWh)
 *   z = rstd_in * (pre[r] + G[e] - mean_in s) + c;   h = LayerNorm(ReLU(z))
 * This is the form for MANY UAVs: the kernels above keep Wh^T in registers up to 8 UAVs and in LDS beyond, where HD reads of
 * it per lane and row bound them (16 UAVs: 4.2 ms for 2.5 M rows); with the product done on the MFMA pipes the kernel is two
 * streaming passes over [rows,H] and knows no limit on the number of UAVs.
 * Backward given dh [n*N,H]: dpre [n*N,H] = rstd_in * dL/dz (the gradient of the GEMM output), dG [n,H], ds, dc, dgamma, dbeta
 * [H]; workspace: dcc_mlp_workspace_floats(H, 1) floats.
 */
DCC_API int dcc_actor_l1_pre_fwd(const float* pre, const float* G, const double* stats, const float* s, const float* c,
                                 const float* gamma, const float* beta, float eps_in, float eps_ln, int32_t D, float* h,
                                 int64_t n, int32_t N, int32_t H, void* stream);
DCC_API int dcc_actor_l1_pre_bwd(const float* pre, const float* G, const double* stats, const float* s, const float* c,
                                 const float* gamma, const float* dh, float eps_in, float eps_ln, int32_t D, float* dG,
                                 float* dpre, float* ds, float* dc, float* dgamma, float* dbeta, float* workspace, int64_t n,
                                 int32_t N, int32_t H, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCC_MLP_H */
