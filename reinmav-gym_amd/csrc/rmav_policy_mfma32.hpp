// rmav_policy_mfma32.hpp - the fp32 actor of the fused PPO rollout on the matrix cores (RMAV_POLICY_FP32_MFMA).
//
// The VALU policy (rmav_policy.hpp) issues ~10 k scalar FMAs per env-step from ONE wavefront per SIMD at the
// BASELINE batch (65 536 envs): 30 us per env-step batch, the vector ALU ~27 % busy.  gfx950 has fp32-input MFMAs
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, 4096 FLOP per instruction, 64 cycles per SIMD = the vector
// peak) that move those FMAs off the instruction stream: per net and 32 envs 10 + 64 matrix instructions
// instead of ~2 500 vector ones.  Same precision class as the VALU mode (fp32 operands, fp32 accumulate; only the
// summation order differs), so it stays interchangeable with the fp32 learner - unlike the bf16 mode.
// Measured (65 536 envs x 32 steps): 0.98 -> 0.51 ms per rollout (2.1 -> 4.1 G env-steps/s).  SQ counters of that
// kernel (tools/gpu_r02_q.sh): the matrix pipe is busy 52 % of the time (64.9 cycles per instruction, as documented),
// the wavefronts issue other instructions 25 % of the time and are stalled on matrix-instruction issue 58 % of it, and
// two wavefronts per SIMD run exactly as fast as one (the 64- and 32-env-per-wavefront versions of this file measure
// 0.52 and 0.51 ms): the fp32-input MFMA executes on the SIMD's fp32 lanes ("the f32 vector rate"), so neither the
// wavefront's own vector instructions nor the other wavefront's overlap with it - the ~1 200 remaining vector
// instructions per 32 envs and step (128 tanh, Box-Muller, the dynamics step, the output layer) add to, rather than
// hide under, the matrix time.  Pinning tanh between the matrix instructions with sched_group_barrier changed nothing.
//
// One wavefront = ONE 32-column tile = 32 envs, each env simulated by BOTH lanes (n, 0) and (n, 1) of its column (the
// dynamics step is ~3 % of this mode's work; the two copies execute the same instructions on the same inputs and stay
// bit-identical): that gives two wavefronts per SIMD at the 65 536-env batch, so one wavefront's activations / dynamics
// / stores issue under the other's 64-cycle matrix instructions, and the network input needs no lane exchange at all.
// OUT[64 x 32 envs] = W . IN with the weights as the A operand and the activations as B:
//   * C/D layout of the 32x32 MFMA: lane l = (n = l & 31, h = l >> 5), register r holds
//     D[row(r, h) = (r & 3) + 8 (r >> 2) + 4 h][col n].
//   * B operand of v_mfma_f32_32x32x2_f32: ONE float per lane, lane (n, h) supplies B[k = h][n].
//   So accumulator register r of row tile T, after tanh, IS a valid B operand of the next layer: it carries hidden
//   units k0 = 32 T + row(r, 0) (lanes h = 0) and k1 = 32 T + row(r, 1) (lanes h = 1) of column n.  The host
//   packs the A operand to match: slice (T, r), lane (m, h) holds W[m][32 T + row(r, h)].  32 slices per 64-unit
//   layer, no activation is moved, transposed or staged.
//   * The output layer (4 + 1 rows) would waste a 32-row tile per slice, so it runs on the vector ALU: every lane
//     holds 32 of the 64 hidden activations of its column, forms partial dot products and swaps them with lane
//     l ^ 32 (which holds the other 32 rows) - the only cross-lane traffic of the mode.
//
// LDS (44 KB per policy): per net  A1 [2 T][2 sq][64 lanes][4]  (layer-1 slices 4 sq .. 4 sq + 3 of the lane) |
// A2 [2 To][2 Tin][4 rq][64 lanes][4] | W3 [2 h][4 outputs][32] | b1 [64] | b2 [64] | b3 [4],  then logstd [4].
// One ds_read_b128 per lane fetches the A operands of four consecutive slices (16-byte lane stride: conflict-free).
#pragma once

#include "rmav_policy_mfma.hpp"

namespace rmav {

struct Mfma32Layout {
    static constexpr int A1 = 0;                      // 2 * 2 * 64 * 4 = 1024
    static constexpr int A2 = A1 + 1024;              // 2 * 2 * 4 * 64 * 4 = 4096
    static constexpr int W3 = A2 + 4096;              // 2 * 4 * 32 = 256
    static constexpr int B1 = W3 + 256;
    static constexpr int B2 = B1 + 64;
    static constexpr int B3 = B2 + 64;
    static constexpr int NET = B3 + 4;                // 5508 floats per net
    static constexpr int LOGSTD = 2 * NET;
    static constexpr int TOTAL = 2 * NET + 4;         // 11 020 floats
};

// NOUT = 4 (policy mean, zero padded) or 1 (value).  b_in[s]: layer-1 B operand of slice s (lane (n, h): component
// 2 s + h of the env of column n), NSL = ceil(nS / 2) slices.  Returns the outputs for column n in BOTH half-waves.
struct Mfma32In { float v[8]; };   // by value: a reference parameter of a real function would live on the stack
template <int NSL, int NOUT>
__device__ __noinline__ float4 mlp_mfma32(Mfma32In bin, uint32_t net) {
    using L = Mfma32Layout;
    const float (&b_in)[8] = bin.v;
    asm volatile("" : "+v"(net));   // keep LLVM from hoisting the weight reads out of the env-step loop
    const float *w = lds_w + net;
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    // ---- layer 1: [64 x 2 NSL] . [2 NSL x 32] -------------------------------------------------------------
    f32x16_t acc[2];   // [row tile T]
#pragma unroll
    for (int T = 0; T < 2; ++T) acc[T] = bias_frag(w + L::B1 + 32 * T, h);
#pragma unroll
    for (int sq = 0; sq < (NSL + 3) / 4; ++sq) {
        const float4 a40 = *reinterpret_cast<const float4 *>(w + L::A1 + ((0 * 2 + sq) * 64 + lane) * 4);
        const float4 a41 = *reinterpret_cast<const float4 *>(w + L::A1 + ((1 * 2 + sq) * 64 + lane) * 4);
        const float a0[4] = {a40.x, a40.y, a40.z, a40.w}, a1[4] = {a41.x, a41.y, a41.z, a41.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = 4 * sq + j;
            if (s < NSL) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b_in[s], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b_in[s], acc[1], 0, 0, 0);
            }
        }
    }
    // ---- layer 2: 32 slices of K = 2, fed straight from the accumulators ----------------------------------
    f32x16_t acc2[2];
#pragma unroll
    for (int To = 0; To < 2; ++To) acc2[To] = bias_frag(w + L::B2 + 32 * To, h);
#pragma unroll
    for (int Tin = 0; Tin < 2; ++Tin)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float4 a40 = *reinterpret_cast<const float4 *>(w + L::A2 + (((0 * 2 + Tin) * 4 + rq) * 64 + lane) * 4);
            const float4 a41 = *reinterpret_cast<const float4 *>(w + L::A2 + (((1 * 2 + Tin) * 4 + rq) * 64 + lane) * 4);
            const float a0[4] = {a40.x, a40.y, a40.z, a40.w}, a1[4] = {a41.x, a41.y, a41.z, a41.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float hb = tanh_fast(acc[Tin][4 * rq + j]);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], hb, acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], hb, acc2[1], 0, 0, 0);
            }
        }
    // ---- output layer on the vector ALU: partial dot products over the 32 hidden rows this lane holds --------
    float p[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) p[o] = 0.0f;
#pragma unroll
    for (int Tin = 0; Tin < 2; ++Tin)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float hv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = tanh_fast(acc2[Tin][4 * rq + j]);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const float4 w4 = *reinterpret_cast<const float4 *>(w + L::W3 + (h * 4 + o) * 32 + Tin * 16 + rq * 4);
                p[o] = rfma(w4.x, hv[0], p[o]);
                p[o] = rfma(w4.y, hv[1], p[o]);
                p[o] = rfma(w4.z, hv[2], p[o]);
                p[o] = rfma(w4.w, hv[3], p[o]);
            }
        }
    // the other 32 hidden rows of column n are in lane l ^ 32; the h = 0 half adds first in both lanes so that the two
    // copies of the env get bit-identical outputs
    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        const float other = xor32(p[o]);   // not ds_bpermute: rmav_policy_mfma.hpp
        const float lo = h ? other : p[o], hi = h ? p[o] : other;
        out[o] = (lo + hi) + w[L::B3 + o];
    }
    return make_float4(out[0], out[1], out[2], out[3]);
}

// Policy mean (4 padded outputs) and value for the env of this lane's column.  x: the env's state padded to 16
// (both lanes of a column hold the same state).
template <int NS>
__device__ __forceinline__ void policy_forward_mfma32(const float (&x)[16], float (&mean)[4], float &value) {
    constexpr int NSL = (NS + 1) / 2;
    const uint32_t h = (threadIdx.x & 63u) >> 5;
    // component 2 s + h by bit selection (v_bfi_b32): written as `h ? x[2 s + 1] : x[2 s]` LLVM folds the select into a
    // dynamically indexed load and the state array lands in scratch memory (80 bytes per lane, one round trip per step)
    const uint32_t hmask = 0u - h;   // h = 1: all ones
    Mfma32In b;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < NSL) {
            const uint32_t lo = __builtin_bit_cast(uint32_t, x[2 * s]), hi = __builtin_bit_cast(uint32_t, x[2 * s + 1]);
            b.v[s] = __builtin_bit_cast(float, (hi & hmask) | (lo & ~hmask));
        } else {
            b.v[s] = 0.0f;
        }
    }
    const float4 m = mlp_mfma32<NSL, 4>(b, 0u);
    const float4 v = mlp_mfma32<NSL, 1>(b, (uint32_t)Mfma32Layout::NET);
    mean[0] = m.x; mean[1] = m.y; mean[2] = m.z; mean[3] = m.w;
    value = v.x;
}

}  // namespace rmav
