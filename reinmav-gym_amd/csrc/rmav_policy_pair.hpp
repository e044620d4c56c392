// rmav_policy_pair.hpp - the matrix-core actors of the fused PPO rollout as an (actor, critic) wavefront PAIR per 64 envs.
//
// Why.  tools/micro/issue_rate.hip (profiles/r04/issue_rate.md) measured what one instruction costs on a gfx950 SIMD:
//   * a LONE wavefront issues a vector instruction every ~5.0-5.6 cycles, whatever it is (plain, packed, conversion, AGPR move),
//     and a transcendental (v_exp_f32, v_rcp_f32) every 8.8;
//   * the pipe behind it is only half busy then: with TWO wavefronts on the SIMD plain vector instructions retire every 2.5-3.0
//     cycles, packed f32 / conversions every 4.5-5.1, transcendentals every 8.4 (unchanged: they are the pipe's quarter rate);
//   * a v_mfma_f32_32x32x16 takes its 32 cycles on the matrix pipe and hides ~5 independent vector instructions.
// The one-wavefront-per-64-envs actor of round 3 (k_rollout<K, ACT_POLICY_BF16>: ~1 730 instructions per env-step, 512 of them
// transcendental) therefore ran at the lone-wavefront ISSUE rate - 12.5 k cycles per env-step where its pipe time is ~8.5 k -
// at BASELINE's C5 shape (65 536 envs per GPU = one wavefront per SIMD).
//
// What.  Each 64 envs get two wavefronts, so every SIMD hosts two instruction streams at that batch:
//   actor  (wave 0): policy net -> action = mean + std * z -> dynamics / reward / termination / auto-reset / episode bookkeeping;
//                    hands (obs, reward, done, action) over in an LDS tile.  The state never leaves its registers.
//   critic (wave 1): value net of the state the actor hands over, the Gaussian noise z (Philox + Box-Muller: state-independent,
//                    drawn one step AHEAD into an LDS tile) and its log-probability, and EVERY trajectory store.
// One s_barrier per env-step swaps the halves of both double-buffered tiles:
//   actor :            B | read Z(0), pi, step 0 -> O(0) | B0 | read Z(1), pi, step 1 -> O(1) | B1 | ...            | B(T-1)
//   critic: draw Z(0)  B | V(s0), draw Z(1)              | B0 | drain O(0), V(s1), draw Z(2)  | B1 | ... drain O(T-2), V(s(T-1)) | B(T-1) | drain O(T-1), V(sT)
// Same Philox counters, same per-net arithmetic: FMT_BF16 produces the bits of the one-wavefront bf16 actor.
//
// FMT_F16 (RMAV_POLICY_F16_MFMA) is the faster AND more accurate variant: f16 operands (11-bit mantissa against bf16's 8) on
// v_mfma_f32_32x32x16_f16, and the activation handed to the next layer is the logistic term r = 1 / (1 + e^(2z)) itself,
// with tanh(z) = 1 - 2 r folded into the next layer's weights and bias (W' = -2 W, b' = b + rowsum(W): rmav_pack_policy_f16
// scales the weights, the kernel derives b' from the ROUNDED weights when it stages them, so the identity holds exactly for
// the weights the matrix cores see).  That removes the final multiply-add of every tanh (2 v_exp + 2 v_rcp + v_pk_add + cvt
// per value pair: 21.5 pipe cycles per value against 24).  Worst activation error 2^-12 absolute (bf16 tanh: 2^-9 relative).
#pragma once

#include "rmav_kernels.hpp"

namespace rmav {

enum : int { FMT_BF16 = 0, FMT_F16 = 1 };
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

constexpr int kPairGroupMax = 4;   // (actor, critic) pairs per workgroup: 1 .. 4 (512 threads), a launch parameter

template <int FMT> struct PairOps;
template <> struct PairOps<FMT_BF16> {
    using frag = bf16x8_t;
    static __device__ __forceinline__ f32x16_t mfma(frag a, frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct PairOps<FMT_F16> {
    using frag = f16x8_t;
    static __device__ __forceinline__ f32x16_t mfma(frag a, frag b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// LDS words of one pair's hand-over tiles (after the weights, which the pairs of a workgroup share)
template <int NS, int NA> struct PairTile {
    static constexpr int Z_HALF = 4 * 64, Z_WORDS = 2 * Z_HALF;            // noise [4][lane], double-buffered (critic -> actor)
    static constexpr int REW = NS * 64, DONE = REW + 64, ACT = DONE + 64;  // obs [c][lane], reward, done, action [c][lane]
    static constexpr int O_HALF = ACT + NA * 64, O_WORDS = 2 * O_HALF;     // (actor -> critic)
    static constexpr int WORDS = Z_WORDS + O_WORDS;
};
template <int K> constexpr size_t pair_lds_bytes(int g) {
    return sizeof(float) * ((size_t)MfmaLayout::TOTAL + (size_t)g * PairTile<Dims<K>::NS, Dims<K>::NA>::WORDS);
}

// registers [8 half, 8 half + 8) of an accumulator holding k z (k = 2 log2 e)  ->  r = 1 / (1 + 2^(k z)) = (1 - tanh z) / 2 as an
// f16 B fragment.  2^(kz) = inf -> r = 0, 0 -> r = 1: saturates cleanly.
__device__ __forceinline__ f16x8_t act_frag_f16(const f32x16_t &acc, int half) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    u32x4_t packed;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x2_t e;
        e[0] = __builtin_amdgcn_exp2f(acc[8 * half + 2 * j]);
        e[1] = __builtin_amdgcn_exp2f(acc[8 * half + 2 * j + 1]);
        e = e + 1.0f;
        f32x2_t r;
        r[0] = __builtin_amdgcn_rcpf(e[0]);
        r[1] = __builtin_amdgcn_rcpf(e[1]);
        packed[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2_t));   // round to nearest even
    }
    return __builtin_bit_cast(f16x8_t, packed);
}

template <int FMT> __device__ __forceinline__ typename PairOps<FMT>::frag ld_frag_t(const float *base, uint32_t lane) {
    return *reinterpret_cast<const typename PairOps<FMT>::frag *>(base + lane * 4u);   // 16 bytes per lane
}

// FMT_F16, once per launch, by the threads of the block between two barriers (the weights are in LDS): the bias tables
// become those of the folded network (header comment):  b1 <- k b1,  b2 <- k b2 - sum_j A2'[i][j] / 2  (A2' = -2k W2 rounded
// to f16),  b3 <- b3 - sum_j A3'[i][j] / 2  (A3' = -2 W3).
__device__ __forceinline__ void fold_biases_f16() {
    using L = MfmaLayout;
    for (int q = threadIdx.x; q < 2 * (64 + 64 + 32); q += blockDim.x) {
        const int net = q / 160, o = q % 160;
        float *w = lds_w + net * L::NET;
        if (o < 64) {
            w[L::B1 + o] *= kTanhScale;
        } else {
            const bool l2 = o < 128;
            const int i = l2 ? o - 64 : o - 128;            // row of layer 2 (64 rows) / layer 3 (32 rows, padded)
            const int T = l2 ? (i >> 5) : 0, m = i & 31;
            const float *frag0 = w + (l2 ? L::A2 + T * 4 * L::FRAG : L::A3);
            float sum = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f16x8_t v = *reinterpret_cast<const f16x8_t *>(frag0 + s * L::FRAG + (m + 32 * h) * 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += (float)v[j];
                }
            float *b = w + (l2 ? L::B2 : L::B3) + i;
            *b = (l2 ? kTanhScale * *b : *b) - 0.5f * sum;
        }
    }
}

// One net (weights at float offset `net` of lds_w) for the two 32-env column tiles of this wavefront: rows 0..3 of the output
// layer for column tile 0 / 1 (valid in the lanes with h == 0).  Same instruction sequence per net as mlp_mfma (rmav_policy_mfma.hpp).
template <int FMT>
__device__ __forceinline__ void mlp_pair(typename PairOps<FMT>::frag b_in0, typename PairOps<FMT>::frag b_in1, uint32_t net,
                                         float (&t0)[4], float (&t1)[4]) {
    using L = MfmaLayout;
    using O = PairOps<FMT>;
    using frag = typename O::frag;
    asm volatile("" : "+v"(net));   // keep LLVM from hoisting the weight reads out of the env-step loop (136 registers)
    const float *w = lds_w + net;
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    f32x16_t acc[2][2];   // [row tile T][column tile Nt]
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const frag a = ld_frag_t<FMT>(w + L::A1 + T * L::FRAG, lane);
        const f32x16_t c = bias_frag(w + L::B1 + 32 * T, h);
        acc[T][0] = O::mfma(a, b_in0, c);
        acc[T][1] = O::mfma(a, b_in1, c);
    }
    frag hb[2][4];        // [Nt][s]
#pragma unroll
    for (int Nt = 0; Nt < 2; ++Nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (FMT == FMT_F16) hb[Nt][s] = act_frag_f16(acc[s >> 1][Nt], s & 1);
            else hb[Nt][s] = act_frag<true>(acc[s >> 1][Nt], s & 1);
        }
    f32x16_t acc2[2][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const f32x16_t c = bias_frag(w + L::B2 + 32 * T, h);
        acc2[T][0] = c;
        acc2[T][1] = c;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const frag a = ld_frag_t<FMT>(w + L::A2 + (T * 4 + s) * L::FRAG, lane);
            acc2[T][0] = O::mfma(a, hb[0][s], acc2[T][0]);
            acc2[T][1] = O::mfma(a, hb[1][s], acc2[T][1]);
        }
    }
#pragma unroll
    for (int Nt = 0; Nt < 2; ++Nt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (FMT == FMT_F16) hb[Nt][s] = act_frag_f16(acc2[s >> 1][Nt], s & 1);
            else hb[Nt][s] = act_frag<false>(acc2[s >> 1][Nt], s & 1);
        }
    f32x16_t o0 = bias_frag(w + L::B3, h), o1 = o0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const frag a = ld_frag_t<FMT>(w + L::A3 + s * L::FRAG, lane);
        o0 = O::mfma(a, hb[0][s], o0);
        o1 = O::mfma(a, hb[1][s], o1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        t0[r] = o0[r];
        t1[r] = o1[r];
    }
}

// Layer-1 B fragments of the wavefront's two column tiles from the env state this lane holds (x: padded to 16).  Lane (n, h)
// of column tile Nt carries components [8h, 8h + 8) of env 32 Nt + n, pre-multiplied by k = 2 log2 e (act_frag): the lane's own
// for its own tile, the partner lane's (l ^ 32) for the other (see policy_forward_mfma).
template <int FMT>
__device__ __forceinline__ void state_frags(const float (&x)[16], typename PairOps<FMT>::frag &b0, typename PairOps<FMT>::frag &b1) {
    using frag = typename PairOps<FMT>::frag;
    const uint32_t h = (threadIdx.x & 63u) >> 5, hmask = 0u - h;
    float mine[8], recv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t lo = __builtin_bit_cast(uint32_t, x[j]), hi = __builtin_bit_cast(uint32_t, x[8 + j]);
        mine[j] = kTanhScale * __builtin_bit_cast(float, (hi & hmask) | (lo & ~hmask));              // x[8h + j]
        const float send = kTanhScale * __builtin_bit_cast(float, (lo & hmask) | (hi & ~hmask));     // x[8(1-h) + j]
        recv[j] = xor32(send);
    }
    frag own, other;
    if constexpr (FMT == FMT_F16) {
        // round-toward-zero pack: one instruction per pair, and a diverged env's huge state saturates at 65504 instead of
        // becoming inf (an inf operand would make the whole tile's outputs NaN)
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
        u32x4_t po, pr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            po[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(mine[2 * j], mine[2 * j + 1]));
            pr[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(recv[2 * j], recv[2 * j + 1]));
        }
        own = __builtin_bit_cast(frag, po);
        other = __builtin_bit_cast(frag, pr);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            own[j] = (__bf16)mine[j];
            other[j] = (__bf16)recv[j];
        }
    }
    b0 = h ? other : own;   // column tile 0 = envs of lanes 0..31
    b1 = h ? own : other;   // column tile 1 = envs of lanes 32..63
}

// G pairs per workgroup: threads [0, 64 G) are the actors, [64 G, 128 G) their critics; pair g owns envs 64 (G blockIdx + g) ..
// Lanes past the end of the batch are clones of env N-1 (every lane of both wavefronts reaches every barrier and every MFMA).
template <int K, int FMT>
__global__ __launch_bounds__(128 * kPairGroupMax, 2) void k_rollout_pair(const RolloutArgs a, const typename Env<K>::P p_shared,
                                                                          const ParamsT<double> pc_shared) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    using L = MfmaLayout;
    using PT = PairTile<NS, NA>;
    using frag = typename PairOps<FMT>::frag;
    const uint32_t G = blockDim.x >> 7;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool critic = wave >= G;
    const uint32_t pair = critic ? wave - G : wave, lane = threadIdx.x & 63u;
    const uint32_t gi = (blockIdx.x * G + pair) * 64u + lane;
    const int64_t n = a.n;
    const bool valid = gi < (uint64_t)n;
    const uint32_t li = valid ? gi : (uint32_t)n - 1u;
    const uint32_t col = (uint32_t)n * 4u, off = li * 4u;
    const int32_t T = a.n_steps;
    const bool track = (a.flags & F_TRACK) != 0, auto_reset = (a.flags & F_AUTO_RESET) != 0;
    float *tile = lds_w + L::TOTAL + pair * PT::WORDS;   // this pair's hand-over tiles
    float *ztile = tile + lane, *otile = tile + PT::Z_WORDS + lane;

    if (a.xsend && blockIdx.x == 0 && threadIdx.x == 0)   // armed statistics exchange: this launch has begun (see k_rollout)
        __hip_atomic_store(a.xstarted, a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // stage the weights of both nets (every thread helps), then derive the bias tables the activations need
        const float4 *src = reinterpret_cast<const float4 *>(a.policy_w);
        float4 *dst = reinterpret_cast<float4 *>(lds_w);
        for (int q = threadIdx.x; q < L::TOTAL / 4; q += blockDim.x) dst[q] = src[q];
        __syncthreads();
        if constexpr (FMT == FMT_F16) fold_biases_f16();
        else scale_biases_for_tanh();
        __syncthreads();
    }

    const rsrc_t r_state = make_rsrc(a.state);
    float s[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] = buf_ld(r_state, off, (uint32_t)c * col);
    const uint64_t env_id = a.env_base + (uint64_t)li;

    if (critic) {
        // ---- critic: noise one step ahead, value net, every trajectory store ---------------------------------------
        float sl = 0.0f;
#pragma unroll
        for (int c = 0; c < NA; ++c) sl += lds_w[L::LOGSTD + c];
        const float logp0 = -sl - 0.5f * (float)NA * 1.8378770664093453f;   // - sum(logstd) - NA/2 ln(2 pi)
        float *logp_out = a.logp_out, *val_out = a.val_out;
        auto draw = [&](int32_t k) {   // z of step k -> its tile half; log-probability of the action it will make
            float z[4];
            gaussian4(a.seed, env_id, a.t0 + (uint64_t)k, z);
            float *zt = ztile + (k & 1) * PT::Z_HALF;
            float q = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) zt[c * 64] = z[c];
#pragma unroll
            for (int c = 0; c < NA; ++c) q = rfma(z[c], z[c], q);
            buf_st(make_rsrc(logp_out), off, 0, rfma(-0.5f, q, logp0));
            logp_out += n;
        };
        draw(0);
        __syncthreads();                                              // B: Z(0) is in the tile
        float *act_out = a.act_out, *obs_out = a.obs_out, *rew_out = a.rew_out;
        uint8_t *done_out = a.done_out;
        for (int32_t k = 0; k <= T; ++k) {
            if (k > 0) {   // outputs of step k - 1: LDS -> trajectory; the obs is the state whose value is due now
                const float *row = otile + ((k - 1) & 1) * PT::O_HALF;
#pragma unroll
                for (int c = 0; c < NS; ++c) s[c] = row[c * 64];
                const float rw = row[PT::REW], dn = row[PT::DONE];
                float av[NA];
#pragma unroll
                for (int c = 0; c < NA; ++c) av[c] = row[PT::ACT + c * 64];
                // a missing output gets a descriptor with num_records = 0: the hardware range check drops its stores (no branch)
                const rsrc_t rA = act_out ? make_rsrc(act_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rO = obs_out ? make_rsrc(obs_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rR = rew_out ? make_rsrc(rew_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rD = done_out ? make_rsrc(done_out) : make_rsrc_bounded(a.state, 0u);
#pragma unroll
                for (int c = 0; c < NA; ++c) buf_st(rA, off, (uint32_t)c * col, av[c]);
#pragma unroll
                for (int c = 0; c < NS; ++c) buf_st(rO, off, (uint32_t)c * col, s[c]);
                buf_st(rR, off, 0, rw);
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(dn != 0.0f ? 1 : 0), rD, li, 0, 0);
                if (act_out) act_out += (int64_t)NA * n;
                if (obs_out) obs_out += (int64_t)NS * n;
                if (rew_out) rew_out += n;
                if (done_out) done_out += n;
            }
            float x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = (c < NS) ? s[c] : 0.0f;
            frag b0, b1;
            state_frags<FMT>(x, b0, b1);
            float t0[4], t1[4];
            mlp_pair<FMT>(b0, b1, (uint32_t)L::NET, t0, t1);
            const float vp = xor32(t1[0]);
            buf_st(make_rsrc(val_out), off, 0, (lane >> 5) ? vp : t0[0]);
            val_out += n;
            if (k + 1 < T) draw(k + 1);
            if (k < T) __syncthreads();                               // B(k): O(k) handed over, Z(k + 1) in the tile
        }
        return;
    }

    // ---- actor: policy net, action, dynamics, bookkeeping --------------------------------------------------------------
    unsigned int fin_n = 0, fin_len = 0;
    float fin_ret = 0.0f;
    float er = 0.0f;
    int32_t el = 0;
    int32_t sb;   // the env's record (EnvRec): steps_beyond_done, reset counter and - when tracking - the episode's start in ONE access
    uint32_t rc;
    if (track) {
        er = buf_ld(make_rsrc(a.ep_ret), off, 0);
        const u32x3_t q = rec_ld3(make_rsrc(a.rec), li);
        sb = (int32_t)q.x;
        rc = q.y;
        el = (int32_t)(ep_clock0(a) - q.z);
    } else {
        const u32x2_t q = rec_ld2(make_rsrc(a.rec), li);
        sb = (int32_t)q.x;
        rc = q.y;
    }
    typename Env<K>::P pl = p_shared;
    if constexpr (K != REINMAV) {
        if (a.pe[0] || a.pe[1] || a.pe[2]) {
            const double m = a.pe[0] ? (double)a.pe[0][li] : (double)pc_shared.mass;
            const double ml = a.pe[1] ? (double)a.pe[1][li] : (double)pc_shared.load_mass;
            const double Lt = a.pe[2] ? (double)a.pe[2][li] : (double)pc_shared.L;
            override_params(pl, m, ml, Lt);
        }
    }
    const typename Env<K>::P &p = pl;
    double tenv = 0.0;
    if constexpr (K == REINMAV) tenv = a.env_time[li];
    // spare reset state, drawn once per launch (see k_rollout)
    float spare[NS];
    bool have_spare = false;
    if (K != REINMAV && auto_reset && T >= 8) {
        reset_state<K>(a.seed, env_id, rc, spare);
        have_spare = true;
    }
    float pol_std[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NA; ++c) pol_std[c] = expf(lds_w[L::LOGSTD + c]);
    __syncthreads();                                                  // B: Z(0) is in the tile
    for (int32_t k = 0; k < T; ++k) {
        float z[NA];
        {
            const float *zt = ztile + (k & 1) * PT::Z_HALF;
#pragma unroll
            for (int c = 0; c < NA; ++c) z[c] = zt[c * 64];
        }
        float x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = (c < NS) ? s[c] : 0.0f;
        frag b0, b1;
        state_frags<FMT>(x, b0, b1);
        float t0[4], t1[4], act[NA];
        mlp_pair<FMT>(b0, b1, 0u, t0, t1);
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const float from_partner = xor32(t1[c]);
            act[c] = rfma(pol_std[c], z[c], (lane >> 5) ? from_partner : t0[c]);
        }
        float dist = 0.0f, r;
        bool done;
        if constexpr (K == REINMAV) {
            float fm0[4];
            Env<K>::step(s, act, false, tenv, p, fm0);
            done = true;   // reinmav_env.py:110
            r = 90.0f;     // reinmav_env.py:111-116
        } else {
            Env<K>::step(s, act, p, dist, done);
            r = -dist;     // reward / steps_beyond_done machine  (quadrotor3d.py:112-122 and siblings)
            if (done) {
                r = (sb < 0) ? 1.0f : 0.0f;
                sb = (sb < 0) ? 0 : sb + 1;
            }
        }
        if (track) {
            er += r;
            el += 1;
            if (done) {
                buf_st(make_rsrc(a.last_ret), off, 0, er);
                rec_st_last_len(make_rsrc(a.rec), li, el);
                if (valid) {
                    fin_n += 1;
                    fin_len += (unsigned int)el;
                    fin_ret += er;
                }
                er = 0.0f;
                el = 0;
            }
        }
        if (K != REINMAV && auto_reset) {
            if (__ballot(done && !have_spare) != 0) {
                if (!have_spare) {   // every lane that has used its spare up (see k_rollout)
                    reset_state<K>(a.seed, env_id, rc, spare);
                    have_spare = true;
                }
            }
            if (done) {
#pragma unroll
                for (int c = 0; c < NS; ++c) s[c] = spare[c];
                have_spare = false;
                rc += 1;
            }
        }
        float *row = otile + (k & 1) * PT::O_HALF;
#pragma unroll
        for (int c = 0; c < NS; ++c) row[c * 64] = s[c];
        row[PT::REW] = r;
        row[PT::DONE] = done ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < NA; ++c) row[PT::ACT + c * 64] = act[c];
        __syncthreads();                                              // B(k)
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) buf_st(r_state, off, (uint32_t)c * col, s[c]);
    if constexpr (K == REINMAV) a.env_time[li] = tenv;
    if (track) {
        buf_st(make_rsrc(a.ep_ret), off, 0, er);
        rec_st3(make_rsrc(a.rec), li, u32x3_t{(uint32_t)sb, rc, ep_clock0(a) + (uint32_t)a.n_steps - (uint32_t)el});
    } else {
        rec_st2(make_rsrc(a.rec), li, u32x2_t{(uint32_t)sb, rc});
    }
    if (track && __ballot(fin_n != 0) != 0) {   // episode totals: this wavefront's slot (see k_rollout)
        Totals *slot = a.totals + (gi >> 6);
        const unsigned int wn = wave_sum_x(fin_n);
        const unsigned int wl = wave_sum_x(fin_len);
        const float wr = wave_sum_x(fin_ret);
        if (lane == 0) {
            atomicAdd(&slot->episodes, (unsigned long long)wn);
            atomicAdd(&slot->length_sum, (unsigned long long)wl);
            atomicAdd(&slot->return_sum, (double)wr);
        }
    }
    if (a.xsend) {   // snapshot for the armed statistics exchange, then this wavefront's arrival word (see k_rollout)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (valid) {
            const float lr = a.last_ret[li];
            const int32_t ll = a.rec[li].last_len;
            __hip_atomic_store(a.xsend + li, __builtin_bit_cast(int32_t, lr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.xsend + a.xcmax + li, ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && valid) __hip_atomic_store(a.xarrive + (gi >> 6), a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- one shared 2 x 64 trunk with a mean head and a value head (RMAV_POLICY_F16_SHARED) -------------------------------------------
//
// baselines' build_policy(value_network=None) - what `python -m gym_reinmav.run --alg=ppo2 --network=mlp` builds for the NATIVE envs:
// their env_type is 'native' (entry point gym_reinmav.envs.native:...), for which baselines' ppo2 has no defaults entry, so
// value_network stays None = 'shared': the value function is a linear head on the policy's latent (run.py:63-68; third-party
// behaviour restated from memory).  The MuJoCo defaults (value_network='copy': two separate nets) are what
// k_rollout_pair above evaluates.  One net = half the activations (128 tanh per env and step instead of 256), and the pair splits
// differently: both wavefronts evaluate the net, each for ONE 32-env column tile -
//   A (wave 0): tile 0 = envs 0..31 from its registers;  then action = mean + std z, dynamics, bookkeeping for all 64 envs
//   B (wave 1): tile 1 = envs 32..63 from the hand-over tile; hands the 32 means to A; noise one step ahead, log-prob, all stores
// with two barriers per env-step (X: the means are there; Y: the step's outputs are there).  Output rows 0..3 of the padded
// 32-row output tile are the action mean, row 4 the value: after the last MFMA the mean of env column n sits in lane n
// (h = 0) and its value in lane 32 + n (h = 1), register 0 - for tile 0 that is already the lane that owns the env.
template <int NS, int NA> struct SharedTile {
    using PT = PairTile<NS, NA>;
    static constexpr int MEAN = PT::WORDS;            // means of tile 1's envs [4][32], B -> A
    static constexpr int WORDS = PT::WORDS + 4 * 32;
};
constexpr int kSharedWeights = MfmaLayout::NET + 4;   // one net + logstd[4]
template <int K> constexpr size_t shared_lds_bytes(int g) {
    return sizeof(float) * ((size_t)kSharedWeights + (size_t)g * SharedTile<Dims<K>::NS, Dims<K>::NA>::WORDS);
}

// the net for ONE column tile: o4 = registers 0..3 of the output accumulator (lanes h = 0: rows 0..3 = mean; h = 1: row 4 = value)
__device__ __forceinline__ void mlp_half_f16(f16x8_t b_in, float (&o4)[4]) {
    using L = MfmaLayout;
    using O = PairOps<FMT_F16>;
    uint32_t net = 0u;
    asm volatile("" : "+v"(net));   // keep the weight reads inside the env-step loop
    const float *w = lds_w + net;
    const uint32_t lane = threadIdx.x & 63u, h = lane >> 5;
    f32x16_t acc[2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
        acc[T] = O::mfma(ld_frag_t<FMT_F16>(w + L::A1 + T * L::FRAG, lane), b_in, bias_frag(w + L::B1 + 32 * T, h));
    f16x8_t hb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) hb[s] = act_frag_f16(acc[s >> 1], s & 1);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        acc[T] = bias_frag(w + L::B2 + 32 * T, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[T] = O::mfma(ld_frag_t<FMT_F16>(w + L::A2 + (T * 4 + s) * L::FRAG, lane), hb[s], acc[T]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) hb[s] = act_frag_f16(acc[s >> 1], s & 1);
    f32x16_t o = bias_frag(w + L::B3, h);
#pragma unroll
    for (int s = 0; s < 4; ++s) o = O::mfma(ld_frag_t<FMT_F16>(w + L::A3 + s * L::FRAG, lane), hb[s], o);
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = o[r];
}

__device__ __forceinline__ f16x8_t pack_frag_f16(const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
    u32x4_t pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk[j] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(kTanhScale * v[2 * j], kTanhScale * v[2 * j + 1]));
    return __builtin_bit_cast(f16x8_t, pk);
}

template <int K>
__global__ __launch_bounds__(128 * kPairGroupMax, 2) void k_rollout_pair_shared(const RolloutArgs a, const typename Env<K>::P p_shared,
                                                                                 const ParamsT<double> pc_shared) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    using L = MfmaLayout;
    using PT = PairTile<NS, NA>;
    using ST_ = SharedTile<NS, NA>;
    const uint32_t G = blockDim.x >> 7;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool helper = wave >= G;
    const uint32_t pair = helper ? wave - G : wave, lane = threadIdx.x & 63u, h = lane >> 5;
    const uint32_t gi = (blockIdx.x * G + pair) * 64u + lane;
    const int64_t n = a.n;
    const bool valid = gi < (uint64_t)n;
    const uint32_t li = valid ? gi : (uint32_t)n - 1u;
    const uint32_t col = (uint32_t)n * 4u, off = li * 4u;
    const int32_t T = a.n_steps;
    const bool track = (a.flags & F_TRACK) != 0, auto_reset = (a.flags & F_AUTO_RESET) != 0;
    float *tile = lds_w + kSharedWeights + pair * ST_::WORDS;
    float *ztile = tile + lane, *otile = tile + PT::Z_WORDS + lane, *mtile = tile + ST_::MEAN;

    if (a.xsend && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.xstarted, a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {   // stage the one net (+ logstd: the buffer's last four words land where LOGSTD of a two-net buffer would not be: keep them apart)
        const float4 *src = reinterpret_cast<const float4 *>(a.policy_w);
        float4 *dst = reinterpret_cast<float4 *>(lds_w);
        for (int q = threadIdx.x; q < kSharedWeights / 4; q += blockDim.x) dst[q] = src[q];
        __syncthreads();
        for (int q = threadIdx.x; q < 160; q += blockDim.x) {   // fold_biases_f16 for net 0 only
            float *w = lds_w;
            if (q < 64) {
                w[L::B1 + q] *= kTanhScale;
            } else {
                const bool l2 = q < 128;
                const int i = l2 ? q - 64 : q - 128, Tt = l2 ? (i >> 5) : 0, m = i & 31;
                const float *frag0 = w + (l2 ? L::A2 + Tt * 4 * L::FRAG : L::A3);
                float sum = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const f16x8_t v = *reinterpret_cast<const f16x8_t *>(frag0 + s * L::FRAG + (m + 32 * hh) * 4);
#pragma unroll
                        for (int j = 0; j < 8; ++j) sum += (float)v[j];
                    }
                float *b = w + (l2 ? L::B2 : L::B3) + i;
                *b = (l2 ? kTanhScale * *b : *b) - 0.5f * sum;
            }
        }
        __syncthreads();
    }
    const float *logstd = lds_w + L::NET;
    const uint64_t env_id = a.env_base + (uint64_t)li;

    if (helper) {
        // ---- B: tile 1 of the net, noise one step ahead, log-prob, every trajectory store ---------------------------------------
        float sl = 0.0f;
#pragma unroll
        for (int c = 0; c < NA; ++c) sl += logstd[c];
        const float logp0 = -sl - 0.5f * (float)NA * 1.8378770664093453f;
        float *logp_out = a.logp_out, *val_out = a.val_out;
        auto draw = [&](int32_t k) {
            float z[4];
            gaussian4(a.seed, env_id, a.t0 + (uint64_t)k, z);
            float *zt = ztile + (k & 1) * PT::Z_HALF;
            float q = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) zt[c * 64] = z[c];
#pragma unroll
            for (int c = 0; c < NA; ++c) q = rfma(z[c], z[c], q);
            buf_st(make_rsrc(logp_out), off, 0, rfma(-0.5f, q, logp0));
            logp_out += n;
        };
        // the net for envs 32..63 of the pair from the obs tile half `half`: lane (n, h) takes components [8h, 8h + 8) of env 32 + n
        auto eval_tile1 = [&](int half) {
            const float *obs = tile + PT::Z_WORDS + half * PT::O_HALF + 32u + (lane & 31u);
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float lo = (j < NS) ? obs[j * 64] : 0.0f, hi = (8 + j < NS) ? obs[(8 + j) * 64] : 0.0f;
                x[j] = h ? hi : lo;
            }
            float o4[4];
            mlp_half_f16(pack_frag_f16(x), o4);
            if (!h) {
#pragma unroll
                for (int c = 0; c < 4; ++c) mtile[c * 32 + lane] = o4[c];    // means of envs 32..63 -> A
            } else {
                buf_st(make_rsrc(val_out), off, 0, o4[0]);                    // lane 32 + n IS env 32 + n of the pair
            }
            val_out += n;
        };
        draw(0);
        __syncthreads();                                                  // P: Z(0) and the initial obs are in the tiles
        float *act_out = a.act_out, *obs_out = a.obs_out, *rew_out = a.rew_out;
        uint8_t *done_out = a.done_out;
        auto drain = [&](int half) {
            const float *row = otile + half * PT::O_HALF;
            float o[NS], av[NA];
#pragma unroll
            for (int c = 0; c < NS; ++c) o[c] = row[c * 64];
            const float rw = row[PT::REW], dn = row[PT::DONE];
#pragma unroll
            for (int c = 0; c < NA; ++c) av[c] = row[PT::ACT + c * 64];
            const rsrc_t rA = act_out ? make_rsrc(act_out) : make_rsrc_bounded(a.state, 0u);
            const rsrc_t rO = obs_out ? make_rsrc(obs_out) : make_rsrc_bounded(a.state, 0u);
            const rsrc_t rR = rew_out ? make_rsrc(rew_out) : make_rsrc_bounded(a.state, 0u);
            const rsrc_t rD = done_out ? make_rsrc(done_out) : make_rsrc_bounded(a.state, 0u);
#pragma unroll
            for (int c = 0; c < NA; ++c) buf_st(rA, off, (uint32_t)c * col, av[c]);
#pragma unroll
            for (int c = 0; c < NS; ++c) buf_st(rO, off, (uint32_t)c * col, o[c]);
            buf_st(rR, off, 0, rw);
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(dn != 0.0f ? 1 : 0), rD, li, 0, 0);
            if (act_out) act_out += (int64_t)NA * n;
            if (obs_out) obs_out += (int64_t)NS * n;
            if (rew_out) rew_out += n;
            if (done_out) done_out += n;
        };
        for (int32_t k = 0; k < T; ++k) {
            eval_tile1((k - 1) & 1);                                      // obs before step k
            __syncthreads();                                              // X(k): the means of envs 32..63 are in the tile
            if (k > 0) drain((k - 1) & 1);
            if (k + 1 < T) draw(k + 1);
            __syncthreads();                                              // Y(k): step k's outputs are in the tile
        }
        eval_tile1((T - 1) & 1);                                          // bootstrap values of envs 32..63
        drain((T - 1) & 1);
        return;
    }

    // ---- A: tile 0 of the net from its registers, then action, dynamics, bookkeeping for all 64 envs -----------------------------
    const rsrc_t r_state = make_rsrc(a.state);
    float s[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] = buf_ld(r_state, off, (uint32_t)c * col);
    unsigned int fin_n = 0, fin_len = 0;
    float fin_ret = 0.0f;
    float er = 0.0f;
    int32_t el = 0;
    int32_t sb;   // the env's record (EnvRec): steps_beyond_done, reset counter and - when tracking - the episode's start in ONE access
    uint32_t rc;
    if (track) {
        er = buf_ld(make_rsrc(a.ep_ret), off, 0);
        const u32x3_t q = rec_ld3(make_rsrc(a.rec), li);
        sb = (int32_t)q.x;
        rc = q.y;
        el = (int32_t)(ep_clock0(a) - q.z);
    } else {
        const u32x2_t q = rec_ld2(make_rsrc(a.rec), li);
        sb = (int32_t)q.x;
        rc = q.y;
    }
    typename Env<K>::P pl = p_shared;
    if constexpr (K != REINMAV) {
        if (a.pe[0] || a.pe[1] || a.pe[2]) {
            const double m = a.pe[0] ? (double)a.pe[0][li] : (double)pc_shared.mass;
            const double ml = a.pe[1] ? (double)a.pe[1][li] : (double)pc_shared.load_mass;
            const double Lt = a.pe[2] ? (double)a.pe[2][li] : (double)pc_shared.L;
            override_params(pl, m, ml, Lt);
        }
    }
    const typename Env<K>::P &p = pl;
    double tenv = 0.0;
    if constexpr (K == REINMAV) tenv = a.env_time[li];
    float spare[NS];
    bool have_spare = false;
    if (K != REINMAV && auto_reset && T >= 8) {
        reset_state<K>(a.seed, env_id, rc, spare);
        have_spare = true;
    }
    float pol_std[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NA; ++c) pol_std[c] = expf(logstd[c]);
    // value of env gi - 32 (tile 0's column lane - 32) leaves through this lane
    const bool vvalid = h && (uint64_t)(gi - 32u) < (uint64_t)n;
    const uint32_t voff = (gi - 32u) * 4u;
    float *val_out = a.val_out;
    auto eval_tile0 = [&](float (&o4)[4]) {   // lane (n, h): components [8h, 8h + 8) of env n - its own for h = 0, lane n's for h = 1
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lo = (j < NS) ? s[j] : 0.0f;
            float hi = 0.0f;
            if (8 + j < NS) hi = xor32(s[8 + j]);   // lanes 32..63 receive lane - 32's component 8 + j  (folded: NS is a constant)
            x[j] = h ? hi : lo;
        }
        mlp_half_f16(pack_frag_f16(x), o4);
        if (vvalid) buf_st(make_rsrc(val_out), voff, 0, o4[0]);
        val_out += n;
    };
    {   // the initial obs of the pair's envs, for B's first evaluation: the obs half step "-1" would have written
        float *row = otile + PT::O_HALF;
#pragma unroll
        for (int c = 0; c < NS; ++c) row[c * 64] = s[c];
    }
    __syncthreads();                                                      // P
    for (int32_t k = 0; k < T; ++k) {
        float o4[4];
        eval_tile0(o4);
        __syncthreads();                                                  // X(k)
        float act[NA];
        {
            const float *zt = ztile + (k & 1) * PT::Z_HALF;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const float mean = h ? mtile[c * 32 + (lane & 31u)] : o4[c];
                act[c] = rfma(pol_std[c], zt[c * 64], mean);
            }
        }
        float dist = 0.0f, r;
        bool done;
        if constexpr (K == REINMAV) {
            float fm0[4];
            Env<K>::step(s, act, false, tenv, p, fm0);
            done = true;
            r = 90.0f;
        } else {
            Env<K>::step(s, act, p, dist, done);
            r = -dist;
            if (done) {
                r = (sb < 0) ? 1.0f : 0.0f;
                sb = (sb < 0) ? 0 : sb + 1;
            }
        }
        if (track) {
            er += r;
            el += 1;
            if (done) {
                buf_st(make_rsrc(a.last_ret), off, 0, er);
                rec_st_last_len(make_rsrc(a.rec), li, el);
                if (valid) {
                    fin_n += 1;
                    fin_len += (unsigned int)el;
                    fin_ret += er;
                }
                er = 0.0f;
                el = 0;
            }
        }
        if (K != REINMAV && auto_reset) {
            if (__ballot(done && !have_spare) != 0) {
                if (!have_spare) {   // every lane that has used its spare up (see k_rollout)
                    reset_state<K>(a.seed, env_id, rc, spare);
                    have_spare = true;
                }
            }
            if (done) {
#pragma unroll
                for (int c = 0; c < NS; ++c) s[c] = spare[c];
                have_spare = false;
                rc += 1;
            }
        }
        float *row = otile + (k & 1) * PT::O_HALF;
#pragma unroll
        for (int c = 0; c < NS; ++c) row[c * 64] = s[c];
        row[PT::REW] = r;
        row[PT::DONE] = done ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < NA; ++c) row[PT::ACT + c * 64] = act[c];
        __syncthreads();                                                  // Y(k)
    }
    {
        float o4[4];
        eval_tile0(o4);                                                   // bootstrap values of envs 0..31
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) buf_st(r_state, off, (uint32_t)c * col, s[c]);
    if constexpr (K == REINMAV) a.env_time[li] = tenv;
    if (track) {
        buf_st(make_rsrc(a.ep_ret), off, 0, er);
        rec_st3(make_rsrc(a.rec), li, u32x3_t{(uint32_t)sb, rc, ep_clock0(a) + (uint32_t)a.n_steps - (uint32_t)el});
    } else {
        rec_st2(make_rsrc(a.rec), li, u32x2_t{(uint32_t)sb, rc});
    }
    if (track && __ballot(fin_n != 0) != 0) {
        Totals *slot = a.totals + (gi >> 6);
        const unsigned int wn = wave_sum_x(fin_n);
        const unsigned int wl = wave_sum_x(fin_len);
        const float wr = wave_sum_x(fin_ret);
        if (lane == 0) {
            atomicAdd(&slot->episodes, (unsigned long long)wn);
            atomicAdd(&slot->length_sum, (unsigned long long)wl);
            atomicAdd(&slot->return_sum, (double)wr);
        }
    }
    if (a.xsend) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (valid) {
            const float lr = a.last_ret[li];
            const int32_t ll = a.rec[li].last_len;
            __hip_atomic_store(a.xsend + li, __builtin_bit_cast(int32_t, lr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.xsend + a.xcmax + li, ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && valid) __hip_atomic_store(a.xarrive + (gi >> 6), a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


}  // namespace rmav
