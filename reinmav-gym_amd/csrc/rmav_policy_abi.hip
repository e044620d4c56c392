// rmav_policy_abi.hip - launches of the policy-in-kernel rollouts (rmav_rollout_policy); the second translation unit of librmav.so.
//
// Compiled with -fno-slp-vectorize.  hipcc's SLP vectoriser packs adjacent scalar fp32 operations of the dynamics into
// v_pk_mul / v_pk_fma / v_pk_add_f32 with cross-register op_sel selects, among them `v_pk_fma_f32 D, P, Q, D op_sel:[0,1,0]`
// (quat_body_z).  On gfx950 the operand that op_sel[1] = 1 selects - the HIGH dword of src1 for the LOW result - reads as ZERO in
// lanes 48..63 while a v_mfma_f32_32x32x16_{f16,bf16} of any wavefront executes on the same SIMD: the x-axis thrust term of
// quadrotor3d's step was lost in 1 - 25 % of the wavefronts of a rollout, never with one wavefront per SIMD (three rounds of parity
// tests at BASELINE sizes were green).  Root cause, wait-state sweep (they do not help), flag A/B (the stock build fails as
// well) and the 148-line stand-alone reproducer: profiles/r05/packed_f32_hazard.md, tools/micro/pk_hazard.hip.  The Makefile
// disassembles every object and refuses to build one that contains the form (`check_isa`);
// tests/test_gpu_ppo.py::test_matrix_core_actors_are_deterministic and tests/test_resource_usage.py guard it as well.
#include "rmav_handle.hpp"
#include "rmav_policy_pair.hpp"

using namespace rmav;

namespace {

// one wavefront per 64 envs (32 for the fp32-MFMA actor: both half-waves work on the same 32 envs)
template <int K, int MODE> int launch_policy_1w(rmav_handle h, const RolloutArgs &a_in) {
    RolloutArgs a = a_in;
    take_armed_exchange(h, a, MODE == ACT_POLICY_F32M ? 32 : 64);
    const typename Env<K>::P p = derive_env<K>(h->params);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
    const size_t lds = sizeof(float) * (MODE == ACT_POLICY ? (size_t)PolicyLayout<Dims<K>::NS>::TOTAL
                                        : MODE == ACT_POLICY_BF16 ? (size_t)MfmaLayout::TOTAL : (size_t)Mfma32Layout::TOTAL);
    const int64_t per_wg = MODE == ACT_POLICY_F32M ? block_size(h) / 2 : block_size(h);
    hipLaunchKernelGGL((k_rollout<K, MODE, ST_DEFAULT>), dim3((unsigned)((h->n + per_wg - 1) / per_wg)), dim3(block_size(h)), lds, h->stream,
                       a, p, pc);
    return check_rollout_launch(h, a);
}

// The matrix-core actors as (actor, critic) wavefront pairs (rmav_policy_pair.hpp).  Pairs per workgroup: the pairs of a
// workgroup share one LDS copy of the weights (30 KB) but also one s_barrier; RMAV_TUNE_PAIR_GROUP = 1 .. 4 overrides.
template <int K, int FMT> int launch_rollout_pair(rmav_handle h, const RolloutArgs &a_in) {
    RolloutArgs a = a_in;
    take_armed_exchange(h, a, 64);
    const typename Env<K>::P p = derive_env<K>(h->params);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
    const int forced = h->tune[RMAV_TUNE_PAIR_GROUP];
    // measured (profiles/r04/actor_bench.txt, quadrotor3d x 32 steps): 65 536 envs 4 pairs 15.0 G env-steps/s, 2 pairs 14.1, 1 pair 13.7
    // (one workgroup per CU, weights staged once per CU); 131 072 envs 2 pairs 15.8 - 16.7, 4 pairs 15.5 - 16.3, 1 pair 11.1
    const int g = (forced >= 1 && forced <= kPairGroupMax) ? forced : (h->n <= 98304 ? 4 : 2);
    const int64_t per_wg = 64 * g;
    hipLaunchKernelGGL((k_rollout_pair<K, FMT>), dim3((unsigned)((h->n + per_wg - 1) / per_wg)), dim3(128 * g), pair_lds_bytes<K>(g),
                       h->stream, a, p, pc);
    return check_rollout_launch(h, a);
}

// RMAV_POLICY_F16_SHARED: one trunk, both wavefronts of a pair evaluate it for one 32-env column tile each (k_rollout_pair_shared)
template <int K> int launch_rollout_pair_shared(rmav_handle h, const RolloutArgs &a_in) {
    RolloutArgs a = a_in;
    take_armed_exchange(h, a, 64);
    const typename Env<K>::P p = derive_env<K>(h->params);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
    const int forced = h->tune[RMAV_TUNE_PAIR_GROUP];
    // measured (profiles/r04/actor_bench.txt): 65 536 envs 1 / 2 / 4 pairs per workgroup 20.8 / 21.5 / 20.4 G env-steps/s, 131 072: 19.9 / 25.9 / 26.1
    const int g = (forced >= 1 && forced <= kPairGroupMax) ? forced : 2;
    const int64_t per_wg = 64 * g;
    hipLaunchKernelGGL((k_rollout_pair_shared<K>), dim3((unsigned)((h->n + per_wg - 1) / per_wg)), dim3(128 * g), shared_lds_bytes<K>(g),
                       h->stream, a, p, pc);
    return check_rollout_launch(h, a);
}

template <int K> int launch_policy_k(rmav_handle h, int kmode, const RolloutArgs &a) {
    switch (kmode) {
    case RMAV_ACT_POLICY: return launch_policy_1w<K, ACT_POLICY>(h, a);
    case RMAV_ACT_POLICY_BF16:
        return h->tune[RMAV_TUNE_POLICY_PAIR] == 0 ? launch_policy_1w<K, ACT_POLICY_BF16>(h, a) : launch_rollout_pair<K, FMT_BF16>(h, a);
    case ACT_POLICY_F32M: return launch_policy_1w<K, ACT_POLICY_F32M>(h, a);
    case ACT_POLICY_F16: return launch_rollout_pair<K, FMT_F16>(h, a);
    case ACT_POLICY_F16_SHARED: return launch_rollout_pair_shared<K>(h, a);
    }
    return rmav_fail(RMAV_ERR_INVALID, "unknown policy mode %d", kmode);
}

}  // namespace

int rmav_launch_policy_rollout(rmav_handle h, int kmode, const RolloutArgs &a) {
    switch (h->kind) {
    case RMAV_QUAD2D: return launch_policy_k<QUAD2D>(h, kmode, a);
    case RMAV_QUAD2D_SL: return launch_policy_k<QUAD2D_SL>(h, kmode, a);
    case RMAV_QUAD3D: return launch_policy_k<QUAD3D>(h, kmode, a);
    case RMAV_QUAD3D_SL: return launch_policy_k<QUAD3D_SL>(h, kmode, a);
    case RMAV_REINMAV: return launch_policy_k<REINMAV>(h, kmode, a);
    }
    return rmav_fail(RMAV_ERR_INVALID, "bad kind");
}
