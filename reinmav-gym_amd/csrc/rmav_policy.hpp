// rmav_policy.hpp - in-kernel Gaussian MLP policy (the action source RMAV_ACT_POLICY of the fused rollout).
//
// The caller side of the hot path (baselines ppo2 Runner: model.step(obs) -> env.step(actions), reached from
// gym_reinmav/run.py:63-68) evaluates a 2 x 64 tanh MLP policy and a value net on every observation.  Done
// with framework GEMMs that costs ~180 us per env-step at 65 536 envs (six K<=64 GEMMs + elementwise
// launches) against ~1 us for the dynamics.  Here the policy is evaluated by the lane that owns the env,
// inside the rollout kernel, so observations never leave registers.
//
// * The ~10.5 k weights (42 KB) are the "per-step constants" of this mode: too many for scalar registers,
//   so they are staged ONCE per launch into LDS (160 KB per CU) and read back with wave-uniform
//   ds_read_b128 (a broadcast: no bank conflicts), 4 weights per LDS instruction.
// * Layer 1 and layer 2 are fused in one loop over the 64 hidden units of layer 1: unit i is produced
//   (nS FMAs + tanh) and immediately scattered into the 64 layer-2 accumulators, which stay in registers
//   with static indices; the loop index is dynamic, so code size stays ~150 instructions per net.
// * fp32 FMAs on the vector ALU (hipcc packs them into v_pk_fma_f32).  The same nets on the fp32-input MFMA
//   (same FLOP rate, ~40x fewer instructions) are rmav_policy_mfma32.hpp: 1.9x faster, same precision class, and
//   the default fp32 actor since round 2; the 16x faster bf16 MFMA (rmav_policy_mfma.hpp) gives up bitwise-class
//   agreement with the fp32 learner.  See DESIGN.md section 8.
//
// Weight buffer layout (fp32, H = 64, NSP = nS rounded up to a multiple of 4), per net, pi then vf:
//   W1 [H][NSP] (row j = hidden unit j, zero padded) | b1 [H] | W2T [H][H] (W2T[i][j] = W2[j][i]) | b2 [H] |
//   W3T [H][4] (W3T[j][k] = W3[k][j], zero padded to 4 outputs) | b3 [4]
// followed by logstd [4] (zero padded).  rmav_policy_weight_count(kind) returns the total.
#pragma once

#include "rmav_math.hpp"

namespace rmav {

constexpr int kHidden = 64;

template <int NS> struct PolicyLayout {
    static constexpr int NSP = (NS + 3) / 4 * 4;
    static constexpr int W1 = 0;
    static constexpr int B1 = W1 + kHidden * NSP;
    static constexpr int W2T = B1 + kHidden;
    static constexpr int B2 = W2T + kHidden * kHidden;
    static constexpr int W3T = B2 + kHidden;
    static constexpr int B3 = W3T + kHidden * 4;
    static constexpr int NET = B3 + 4;          // floats per net
    static constexpr int LOGSTD = 2 * NET;
    static constexpr int TOTAL = 2 * NET + 4;   // floats in the whole buffer
};

// tanh(x) = 1 - 2 / (1 + 2^(k x)),  k = 2 log2(e): v_mul, v_exp, v_add, v_rcp, v_fma.  No branches; saturates cleanly
// (2^(kx) = inf -> 1, 0 -> -1); |abs error| < 2e-7.
__device__ __forceinline__ float tanh_fast(float x) {
    const float t = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
    return __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + t), -2.0f, 1.0f);
}

// out[0..3] = W3 . tanh(W2 . tanh(W1 . x + b1) + b2) + b3   for one net whose weights start at `w` in LDS.
// A real function (not inlined): three call sites share one ~400-instruction body, arguments and result
// travel in VGPRs (structs by value), and the register allocator cannot smear the three evaluations over
// each other (inlined, hipcc needed 256 VGPRs + 200 AGPRs of spill space for this kernel).
template <int NSP> struct XVec { float v[NSP]; };

// The staged weights.  Declared here so that every access below is a known LDS (address space 3) access:
// through a generic `const float*` parameter hipcc emitted a flat->LDS cast (compare + select) and 64-bit
// address arithmetic per read, and serialised every ds_read behind its own s_waitcnt (117 us per env-step
// batch instead of ~15).
extern __shared__ __attribute__((aligned(16))) float lds_w[];

template <int NS>
__device__ __noinline__ float4 mlp_forward(XVec<PolicyLayout<NS>::NSP> xin, uint32_t net /*float offset of the net*/) {
    using L = PolicyLayout<NS>;
    constexpr int NSP = L::NSP;
    // The weights are loop-invariant across the env-steps of a launch; an opaque offset keeps LLVM from
    // hoisting ~10 k LDS loads out of the step loop (re-reading LDS is the point of staging them there).
    asm volatile("" : "+v"(net));
    const float *w = lds_w + net;
    const float (&x)[NSP] = xin.v;
    float acc[kHidden];
    {
        const float4 *b2 = reinterpret_cast<const float4 *>(w + L::B2);
#pragma unroll
        for (int q = 0; q < kHidden / 4; ++q) {
            const float4 b = b2[q];
            acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
        }
    }
#pragma unroll 1
    for (int i = 0; i < kHidden; ++i) {
        // fetch everything unit i needs first (19 independent LDS reads in flight), then compute
        float4 r1[NSP / 4], r2[kHidden / 4];
        const float4 *w1 = reinterpret_cast<const float4 *>(w + L::W1 + i * NSP);
        const float4 *w2 = reinterpret_cast<const float4 *>(w + L::W2T + i * kHidden);
        float a = w[L::B1 + i];
#pragma unroll
        for (int q = 0; q < NSP / 4; ++q) r1[q] = w1[q];
#pragma unroll
        for (int q = 0; q < kHidden / 4; ++q) r2[q] = w2[q];
        // keep all 19 reads in flight: without the barrier hipcc ping-pongs two register quads and waits
        // (lgkmcnt(1)) on every read - one LDS round trip (~64+ cycles) per 2 packed FMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NSP / 4; ++q) {
            a = rfma(r1[q].x, x[4 * q + 0], a);
            a = rfma(r1[q].y, x[4 * q + 1], a);
            a = rfma(r1[q].z, x[4 * q + 2], a);
            a = rfma(r1[q].w, x[4 * q + 3], a);
        }
        const float h = tanh_fast(a);
#pragma unroll
        for (int q = 0; q < kHidden / 4; ++q) {
            acc[4 * q + 0] = rfma(r2[q].x, h, acc[4 * q + 0]);
            acc[4 * q + 1] = rfma(r2[q].y, h, acc[4 * q + 1]);
            acc[4 * q + 2] = rfma(r2[q].z, h, acc[4 * q + 2]);
            acc[4 * q + 3] = rfma(r2[q].w, h, acc[4 * q + 3]);
        }
    }
    const float4 b3 = *reinterpret_cast<const float4 *>(w + L::B3);
    float o0 = b3.x, o1 = b3.y, o2 = b3.z, o3 = b3.w;
    const float4 *w3 = reinterpret_cast<const float4 *>(w + L::W3T);
#pragma unroll
    for (int j = 0; j < kHidden; ++j) {
        const float h = tanh_fast(acc[j]);
        const float4 ww = w3[j];
        o0 = rfma(ww.x, h, o0);
        o1 = rfma(ww.y, h, o1);
        o2 = rfma(ww.z, h, o2);
        o3 = rfma(ww.w, h, o3);
    }
    return make_float4(o0, o1, o2, o3);
}

// Four standard normals for (env, global step t): Philox stream tag 3, Box-Muller on (r0,r1) and (r2,r3):
//   u1 = ((r >> 8) + 1) * 2^-24 in (0,1],  u2 = (r' >> 8) * 2^-24 in [0,1)
//   z = sqrt(-2 ln u1) * (cos(2 pi u2), sin(2 pi u2))
__device__ __forceinline__ void gaussian4(uint64_t seed, uint64_t env_id, uint64_t t, float (&z)[4]) {
    uint32_t r[4];
    philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)t,
                  (3u << 24) | ((uint32_t)((t >> 32) & 0xFFFFu) << 8), (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = (float)((r[2 * p] >> 8) + 1u) * (1.0f / 16777216.0f);
        const float u2 = u01(r[2 * p + 1]);
        const float rad = __builtin_sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincospif(2.0f * u2, &sn, &cs);
        z[2 * p] = rad * cs;
        z[2 * p + 1] = rad * sn;
    }
}

}  // namespace rmav
