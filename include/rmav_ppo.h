/* rmav_ppo.h - the PPO2 rollout loop behind the boundary: baselines ppo2 Runner.run() = model.step(obs) ->
 * env.step(actions) (gym_reinmav/run.py:63-68) as ONE fused launch with the policy inside the kernel, and the learner-side passes (GAE).
 * Part of the C ABI of librmav.so; included by rmav.h (conventions, rmav_handle, status codes: there). */
#ifndef RMAV_PPO_H
#define RMAV_PPO_H

#include "rmav.h"

#ifdef __cplusplus
extern "C" {
#endif

enum rmav_policy_precision {
    RMAV_POLICY_FP32 = 0,       /* fp32 FMAs on the vector ALU */
    RMAV_POLICY_BF16_MFMA = 1,  /* bf16 operands, fp32 accumulate on the matrix cores */
    RMAV_POLICY_FP32_MFMA = 2,  /* fp32 operands and accumulate on the fp32-input matrix instructions: same precision
                                   class as RMAV_POLICY_FP32 (only the summation order differs), ~2x its speed */
    RMAV_POLICY_F16_MFMA = 3,   /* f16 operands (11-bit mantissa), fp32 accumulate, tanh folded into the next layer's weights
                                   (csrc/rmav_policy_pair.hpp): ~8x closer to the fp32 policy than bf16 and faster.
                                   Weight buffer: rmav_pack_policy_f16 */
    RMAV_POLICY_F16_SHARED = 4  /* a DIFFERENT architecture, same arithmetic as RMAV_POLICY_F16_MFMA: ONE 2x64 tanh trunk with a mean head
                                   and a scalar value head on its latent - baselines' value_network = 'shared', what ppo2 builds for an env
                                   type without a defaults entry (the native envs of gym_reinmav: env_type 'native'); the other precisions
                                   evaluate a policy net and a separate value net (value_network = 'copy', baselines' MuJoCo default).
                                   Weight buffer: rmav_policy_weight_count_shared() floats = ONE net of the bf16 fragment layout with
                                   output rows 0..3 = the mean head, row 4 = the value head, then logstd [4]; built by rmav_pack_policy_f16 */
};

/* PPO2-style rollout with the policy inside the kernel (the caller loop of gym_reinmav/run.py:63-68:
 * baselines ppo2 Runner = model.step(obs) -> env.step(actions), network='mlp').  Policy: two 64-unit tanh
 * layers -> Gaussian mean (state-independent log-std), plus a value net of the same shape.  All pointers
 * are DEVICE pointers, layout is SoA, nothing synchronises (capturable in a hipGraph).
 * weights: rmav_policy_weight_count(kind) floats, 16-byte aligned, layout (H = 64, NSP = nS rounded up
 *   to a multiple of 4), policy net then value net, each:
 *     W1 [H][NSP] (row = hidden unit, zero padded) | b1 [H] | W2T [H][H] (W2T[i][j] = W2[j][i]) | b2 [H] |
 *     W3T [H][4] (W3T[j][k] = W3[k][j], zero padded to 4 outputs) | b3 [4]
 *   then logstd [4] (zero padded).
 * Per step t: a = mean(obs_t) + exp(logstd) * z_t with z_t standard normal from the counter RNG
 * (stream tag 3, Box-Muller; see csrc/rmav_policy.hpp), logp_out[t] = log N(a; mean, std),
 * value_out[t] = V(obs_t); value_out[n_steps] = V(obs after the last step) for bootstrapping.
 * actions_out [n_steps][nA][N], obs_out [n_steps][nS][N], rew_out / done_out [n_steps][N] may be NULL.
 * precision = RMAV_POLICY_BF16_MFMA evaluates the same two nets with v_mfma_f32_32x32x16_bf16 (bf16
 * weights and activations, fp32 accumulation; means / values within ~1e-2 of the fp32 policy).  Its weight
 * buffer is rmav_policy_weight_count_bf16() floats of pre-arranged MFMA fragments: per net
 *   A1 [2][64 lanes][8 bf16] | A2 [2][4][64][8] | A3 [4][64][8] | b1 [64] | b2 [64] | b3 [32] (fp32)
 * then logstd [4]; fragment (.., lane = (m = lane & 31, h = lane >> 5), j) holds
 *   layer 1: W1p[32 Mt + m][8 h + j]              (W1 zero-padded to 16 inputs)
 *   layer 2: W2 [32 Mt + m][rowmap(s, h, j)]
 *   layer 3: W3p[m][rowmap(s, h, j)]               (W3 zero-padded to 32 outputs)
 *   rowmap(s, h, j) = 32 (s >> 1) + (r & 3) + 8 (r >> 2) + 4 h,  r = 8 (s & 1) + j
 * (csrc/rmav_policy_mfma.hpp explains why; gym_reinmav_amd.ppo.pack_policy_weights_bf16 builds it). */
int64_t rmav_policy_weight_count(int kind);
int64_t rmav_policy_weight_count_bf16(void);
/* RMAV_POLICY_FP32_MFMA: rmav_policy_weight_count_f32_mfma() floats of pre-arranged A operands of
 * v_mfma_f32_32x32x2_f32, per net (policy, then value):
 *   A1 [2 T][2 sq][64 lanes][4]          lane (m, h), entry j: W1p[32 T + m][2 (4 sq + j) + h]   (W1 zero-padded to 16 inputs)
 *   A2 [2 To][2 Tin][4 rq][64 lanes][4]  lane (m, h), entry j: W2[32 To + m][32 Tin + row(4 rq + j, h)]
 *   W3 [2 h][4 outputs][32]              entry 16 Tin + r:      W3p[o][32 Tin + row(r, h)]        (W3 zero-padded to 4 outputs)
 *   b1 [64] | b2 [64] | b3 [4]
 * then logstd [4];  row(r, h) = (r & 3) + 8 (r >> 2) + 4 h  (csrc/rmav_policy_mfma32.hpp explains why;
 * gym_reinmav_amd.ppo.pack_policy_weights_f32_mfma builds it). */
int64_t rmav_policy_weight_count_f32_mfma(void);
int64_t rmav_policy_weight_count_shared(void);   /* RMAV_POLICY_F16_SHARED */
/* Builds such a weight buffer on the device in ONE launch on the handle's stream: with `flat` = the concatenation of the
 * n_params (<= 16) parameter tensors `params[k]` (DEVICE pointers in a HOST array; sizes[k] elements each) followed by zeros,
 * weights_out[i] = flat[idx_lo[i]] when idx_hi[i] < 0, else the two bf16 roundings of flat[idx_lo[i]] (low half) and
 * flat[idx_hi[i]] (high half) in one 32-bit word.  idx_lo / idx_hi: int32 [n_out] on the DEVICE - the fixed permutation of a
 * layout above (gym_reinmav_amd.ppo._PolicyPacker builds them once).  Replaces the chain of small tensor operations a
 * learner would otherwise run before every rollout (baselines: model.step reads the live variables; here the actor's copy
 * is re-derived from the learner's parameters). */
int rmav_pack_policy(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                     const int32_t *idx_hi, int64_t n_out, float *weights_out);
/* RMAV_POLICY_F16_MFMA: the bf16 layout above with f16 pairs in the fragment words (same idx_lo / idx_hi maps, n_out =
 * rmav_policy_weight_count_bf16()), and the fragments of layers 2 and 3 pre-multiplied (in fp32, before the one rounding to
 * f16) by -2 k and -2, k = 2 log2(e): the kernel hands r = 1 / (1 + e^(2z)) = (1 - tanh z) / 2 to the next layer instead of
 * tanh z and derives the matching biases b' = b + rowsum(W) from these rounded weights when it stages them
 * (gym_reinmav_amd.ppo.pack_policy_weights_f16 is the torch form of the same buffer). */
int rmav_pack_policy_f16(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                         const int32_t *idx_hi, int64_t n_out, float *weights_out);
int rmav_rollout_policy(rmav_handle h, int32_t n_steps, const float *weights, float *actions_out,
                        float *obs_out, float *rew_out, uint8_t *done_out, float *logp_out,
                        float *value_out, int precision);

/* ---- learner-side passes over a trajectory (DEVICE pointers, enqueued on the handle's stream) -------- */
/* Generalised advantage estimation, the backward pass of baselines ppo2 Runner.run():
 *   delta_t = reward_scale * r_t + gamma V_{t+1} (1 - done_t) - V_t,  A_t = delta_t + gamma lam (1 - done_t) A_{t+1}
 * rew [n_steps][N], done u8 [n_steps][N] (1 = the episode ended with step t), values [n_steps + 1][N]
 * (values[n_steps] = bootstrap value; exactly what rmav_rollout_policy writes); adv_out, ret_out [n_steps][N]
 * (ret = A + V).  sums_out (nullable): 2 doubles on the device <- (sum A, sum A^2) over all n_steps*N samples,
 * for the advantage normalisation (all-reduce them across ranks first when data parallel).  fp32 FMAs;
 * agrees with a float64 per-env recursion to ~1e-6 relative. */
int rmav_gae(rmav_handle h, int32_t n_steps, const float *rew, const uint8_t *done, const float *values,
             float gamma, float lam, float reward_scale, float *adv_out, float *ret_out, double *sums_out);
/* x[i] <- (x[i] - mean) * rstd for i < count (x 16-byte aligned): advantage normalisation in place. */
int rmav_normalize(rmav_handle h, float *x, int64_t count, float mean, float rstd);

#ifdef __cplusplus
}
#endif
#endif
