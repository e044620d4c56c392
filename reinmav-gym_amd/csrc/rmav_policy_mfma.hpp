// rmav_policy_mfma.hpp - bf16-MFMA actor for the fused PPO rollout (RMAV_ACT_POLICY_BF16).
//
// The fp32 VALU policy (rmav_policy.hpp) spends ~10 k FMAs per env-step; this is the GEMM-shaped part of
// the caller loop, so it belongs on the matrix cores.  One wavefront owns 64 envs = two 32-column tiles;
// every layer is  OUT[64 x 64 envs] = W[64 x K] . IN[K x 64 envs]  done with v_mfma_f32_32x32x16_bf16
// (bf16 operands, fp32 accumulate), weights as the A operand, activations as the B operand.
//
// No activation ever touches LDS or gets transposed between layers:
//   * C/D layout of the 32x32 MFMA (MI355X guide): lane l = (n = l & 31, h = l >> 5), register r holds
//     D[row = (r & 3) + 8 (r >> 2) + 4 h][col = n].  So lane (n, h) ends a layer holding, for env column n,
//     the 16 rows {(r&3) + 8(r>>2) + 4h} of each 32-row tile.
//   * A B operand is 8 bf16 per lane: slot (h, j), j = 0..7, of K-slice s.  The hardware pairs A-slot (h, j)
//     with B-slot (h, j), whatever it calls that k.  We therefore feed K-slice s of the NEXT layer straight
//     from accumulator registers r = 8 (s & 1) + j of row tile T = s >> 1 (after tanh and bf16 rounding), and
//     the host packs W's columns in the matching order:  A-slot (h, j) of slice s, lane (m, h)  =
//     W[m][32 T + (r & 3) + 8 (r >> 2) + 4 h].   (rowmap() below; pack_policy_weights_bf16 in ppo.py.)
//   * Only the network input and output cross lanes: env e lives in lane e, but column tile Nt needs env
//     32 Nt + n in both half-waves, so the state is exchanged once between lanes l and l ^ 32 (8 shuffles),
//     and the action mean / value come back the same way (5 shuffles).
//
// LDS holds the pre-arranged bf16 weight fragments (one ds_read_b128 per lane per fragment, conflict-free:
// consecutive lanes read consecutive 16-byte slots) and the fp32 biases (read as the C operand's initial
// value, 4 consecutive rows per ds_read_b128).  ~30 KB per policy.
//
// Precision: bf16 operands (8 mantissa bits) with fp32 accumulation: means / values agree with the fp32
// policy to ~1e-2 (tested at 3e-2 * max(1, |y|)); log-probabilities are exact for the action actually taken
// under the bf16 mean.  That is an actor/learner precision split, so it is opt-in (the default ACT_POLICY
// mode is fp32 and agrees with torch to 2e-5).
#pragma once

#include "rmav_policy.hpp"

namespace rmav {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// float offsets inside the LDS buffer (one float = 4 bytes; a fragment = 64 lanes x 16 B = 256 floats)
struct MfmaLayout {
    static constexpr int FRAG = 256;
    static constexpr int A1 = 0;                    // [Mt=2]            layer 1 (K = 16: the padded state)
    static constexpr int A2 = A1 + 2 * FRAG;        // [Mt=2][s=4]       layer 2
    static constexpr int A3 = A2 + 8 * FRAG;        // [s=4]             layer 3 (output rows padded to 32)
    static constexpr int B1 = A3 + 4 * FRAG;        // fp32 bias tables: 64, 64, 32 rows
    static constexpr int B2 = B1 + 64;
    static constexpr int B3 = B2 + 64;
    static constexpr int NET = B3 + 32;             // floats per net
    static constexpr int LOGSTD = 2 * NET;
    static constexpr int TOTAL = 2 * NET + 4;
};

// C operand initialised with the bias of the rows this lane holds: rows 32 T + 8 q + 4 h + {0..3} for q = 0..3
__device__ __forceinline__ f32x16_t bias_frag(const float *tab, uint32_t h) {
    f32x16_t c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b = *reinterpret_cast<const float4 *>(tab + 8 * q + 4 * h);
        c[4 * q + 0] = b.x; c[4 * q + 1] = b.y; c[4 * q + 2] = b.z; c[4 * q + 3] = b.w;
    }
    return c;
}

__device__ __forceinline__ bf16x8_t ld_frag(const float *base, uint32_t lane) {
    return *reinterpret_cast<const bf16x8_t *>(base + lane * 4u);   // 16 bytes per lane
}

// tanh on the matrix-core path.  This kernel is bound by VALU throughput (256 activations per env per step), so
// the activation is cut to four instructions: with zk = k z, k = 2 log2(e),
//     tanh(z) = 1 - 2 / (1 + 2^zk)            v_exp_f32, v_add_f32, v_rcp_f32, v_fma_f32
// and the factor k never costs a multiply: the pre-activations arrive already scaled - layer 1 multiplies the state
// by k before it is rounded to bf16 and layer 2 receives k tanh() (the same fma with constants (-2k, k)), while the
// biases of both layers are scaled once when the weights are staged into LDS (kTanhScale, scale_biases_for_tanh).
// Saturates cleanly (2^zk = inf -> 1, 0 -> -1); absolute error ~1e-7, far below the bf16 rounding of the result.
// (A transcendental-free form - clamp + odd 6-term polynomial on packed FMAs - measured 9.7 against 12.0 G env-steps/s in round
// 3 and 3 % faster with 7 terms in round 2's slower kernel: v_exp / v_rcp cost this kernel no more than any other instruction,
// the instruction COUNT is what it is bound by.)
constexpr float kTanhScale = 2.8853900817779268f;   // 2 log2(e)

// registers [8 half .. 8 half + 8) of an accumulator holding k z -> (A tanh(z) ... ) as a bf16 B fragment:
// OUT_SCALE = k for the fragment that feeds layer 2, 1 for the one that feeds the output layer
template <bool SCALED_OUT>
__device__ __forceinline__ bf16x8_t act_frag(const f32x16_t &acc, int half) {
    constexpr float A = SCALED_OUT ? -2.0f * kTanhScale : -2.0f, B = SCALED_OUT ? kTanhScale : 1.0f;
    // explicit pairs: one v_cvt_pk_bf16_f32 per two activations, the four dwords are the fragment (left to itself
    // hipcc converted some values singly and re-paired them with v_alignbit / v_perm: +25 % VALU in this function)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    u32x4_t packed;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // 1 + 2^zk and the final fma on value PAIRS (v_pk_add_f32 / v_pk_fma_f32: one instruction per two activations)
        f32x2_t e;
        e[0] = __builtin_amdgcn_exp2f(acc[8 * half + 2 * j]);
        e[1] = __builtin_amdgcn_exp2f(acc[8 * half + 2 * j + 1]);
        e = e + 1.0f;
        f32x2_t rc;
        rc[0] = __builtin_amdgcn_rcpf(e[0]);
        rc[1] = __builtin_amdgcn_rcpf(e[1]);
        const f32x2_t v = __builtin_elementwise_fma(rc, (f32x2_t){A, A}, (f32x2_t){B, B});
        packed[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
    }
    return __builtin_bit_cast(bf16x8_t, packed);
}

// The fp32 bias tables of layers 1 and 2 of both nets, scaled by k in place (called once per launch by the
// threads of the block, between two barriers, right after the weights were copied into LDS).
__device__ __forceinline__ void scale_biases_for_tanh() {
    for (int q = threadIdx.x; q < 256; q += blockDim.x)
        lds_w[(q >> 7) * MfmaLayout::NET + MfmaLayout::B1 + (q & 127)] *= kTanhScale;
}

// Value of lane l ^ 32 - WITHOUT ds_bpermute.  v_permlane32_swap (gfx950) exchanges the upper half of one register with the
// lower half of another inside the vector ALU: two instructions per value and no trip through the LDS crossbar (~50+ cycles).
// (Round 4 also credited this change with removing the stale reads in lanes 48..63 of these kernels.  It only moved the
// instruction streams: the cause is a packed-fp32 instruction with op_sel on src1 beside a 32x32x16 MFMA - root-caused in round 5,
// profiles/r05/packed_f32_hazard.md, reproducer tools/micro/pk_hazard.hip - and the fix is that no object of this library may
// contain that instruction form: -fno-slp-vectorize + the `check_isa` step of the Makefile, which fails the build otherwise.)
__device__ __forceinline__ float xor32(float v) {
    const uint32_t x = __builtin_bit_cast(uint32_t, v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0]: the low half in both halves, r[1]: the high half
    return __builtin_bit_cast(float, (threadIdx.x & 32u) ? r[0] : r[1]);
}
// sum over the wavefront, valid in lane 0; same additions in the same order as wave_sum (rmav_kernels.hpp), with the 32-lane
// step on v_permlane32_swap and the rest on DPP row / bank shifts - no LDS instruction
template <typename T> __device__ __forceinline__ T wave_sum_x(T v) {
    v += __builtin_bit_cast(T, xor32(__builtin_bit_cast(float, v)));
    {   // lane l += lane l + 16: row_bcast / permlane16 are not what we need; swap the 16-lane rows inside each 32-lane half
        const uint32_t x = __builtin_bit_cast(uint32_t, v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);   // r[0]: even rows duplicated, r[1]: odd rows duplicated
        v += __builtin_bit_cast(T, (threadIdx.x & 16u) ? r[0] : r[1]);
    }
    // lane l += lane l + 8, 4, 2, 1 inside its 16-lane row: DPP row_shl (0x100 + n), in the vector ALU
#define RMAV_DPP_ADD(CTRL) v += __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true))
    RMAV_DPP_ADD(0x108);
    RMAV_DPP_ADD(0x104);
    RMAV_DPP_ADD(0x102);
    RMAV_DPP_ADD(0x101);
#undef RMAV_DPP_ADD
    return v;
}

// One net (weights at float offset `net` of lds_w) for the two column tiles of this wavefront.
// b_in0 / b_in1: layer-1 B fragments of the two column tiles.  Returns the first 4 output rows of each
// column tile (valid in lanes with h == 0).
struct MlpOut { float t0[4], t1[4]; };   // first 4 output rows of column tile 0 / 1 (returned in VGPRs)

__device__ __noinline__ MlpOut mlp_mfma(bf16x8_t b_in0, bf16x8_t b_in1, uint32_t net) {
    using L = MfmaLayout;
    asm volatile("" : "+v"(net));   // keep LLVM from hoisting the weight reads out of the env-step loop
    const float *w = lds_w + net;
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    // ---- layer 1: [64 x 16] . [16 x 32] per column tile -------------------------------------------------
    f32x16_t acc[2][2];   // [row tile T][column tile Nt]
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const bf16x8_t a = ld_frag(w + L::A1 + T * L::FRAG, lane);
        const f32x16_t c = bias_frag(w + L::B1 + 32 * T, h);
        acc[T][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b_in0, c, 0, 0, 0);
        acc[T][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b_in1, c, 0, 0, 0);
    }
    // ---- layer 2: K = 64 hidden units = 4 slices fed straight from the accumulators -----------------------
    bf16x8_t hb[2][4];    // [Nt][s]
#pragma unroll
    for (int Nt = 0; Nt < 2; ++Nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) hb[Nt][s] = act_frag<true>(acc[s >> 1][Nt], s & 1);
    f32x16_t acc2[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const f32x16_t c = bias_frag(w + L::B2 + 32 * T, h);
        acc2[T][0] = c;
        acc2[T][1] = c;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8_t a = ld_frag(w + L::A2 + (T * 4 + s) * L::FRAG, lane);
            acc2[T][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[0][s], acc2[T][0], 0, 0, 0);
            acc2[T][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[1][s], acc2[T][1], 0, 0, 0);
        }
    }
    // ---- layer 3: output rows padded to one 32-row tile ---------------------------------------------------
#pragma unroll
    for (int Nt = 0; Nt < 2; ++Nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) hb[Nt][s] = act_frag<false>(acc2[s >> 1][Nt], s & 1);
    f32x16_t o0 = bias_frag(w + L::B3, h), o1 = o0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8_t a = ld_frag(w + L::A3 + s * L::FRAG, lane);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[0][s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[1][s], o1, 0, 0, 0);
    }
    MlpOut out;
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // rows 0..3 = registers 0..3 of the lanes with h == 0
        out.t0[r] = o0[r];
        out.t1[r] = o1[r];
    }
    return out;
}

// Policy mean (4 padded outputs) and value for the env owned by this lane.  x: the env's state padded to 16.
__device__ __forceinline__ void policy_forward_mfma(const float (&x)[16], float (&mean)[4], float &value) {
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    // B fragment of column tile Nt, lane (n, h): components [8h, 8h+8) of env 32 Nt + n.
    // Own tile (Nt == h): own components.  Other tile: the partner lane l ^ 32 owns that env; it needs my
    // components [8 (1-h), ...) for its own fragment, I need its components [8h, ...).
    bf16x8_t own, other;
    const uint32_t hmask = 0u - h;   // h = 1: all ones
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // (the state is pre-multiplied by k = 2 log2 e so that layer 1's accumulators hold k z, see act_frag)
        // bit selection (v_bfi_b32) instead of `h ? x[8 + j] : x[j]`: LLVM may fold such a select into a dynamically
        // indexed load, which puts the state array into scratch memory (it did for the 13-component ReinmavEnv state)
        const uint32_t lo = __builtin_bit_cast(uint32_t, x[j]), hi = __builtin_bit_cast(uint32_t, x[8 + j]);
        const float mine = kTanhScale * __builtin_bit_cast(float, (hi & hmask) | (lo & ~hmask));        // x[8h + j]
        const float send = kTanhScale * __builtin_bit_cast(float, (lo & hmask) | (hi & ~hmask));        // x[8(1-h) + j]: what the partner's fragment needs
        const float recv = xor32(send);
        own[j] = (__bf16)mine;
        other[j] = (__bf16)recv;
    }
    const bf16x8_t b0 = h ? other : own;   // column tile 0 = envs of lanes 0..31
    const bf16x8_t b1 = h ? own : other;   // column tile 1 = envs of lanes 32..63
    const MlpOut m = mlp_mfma(b0, b1, 0u);
    const MlpOut v = mlp_mfma(b0, b1, (uint32_t)MfmaLayout::NET);
    // results sit in the h == 0 lanes: tile 0 is already home, tile 1 goes to the partner lane
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float from_partner = xor32(m.t1[r]);
        mean[r] = h ? from_partner : m.t0[r];
    }
    const float vp = xor32(v.t1[0]);
    value = h ? vp : v.t0[0];
}

}  // namespace rmav
