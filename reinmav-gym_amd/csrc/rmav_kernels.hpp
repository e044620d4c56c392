// rmav_kernels.hpp - HIP kernels of the batched quadrotor path (gfx950 / CDNA4, wave64).
//
// Data layout in HBM (per handle, N envs):
//   state      f32 [nS][N]   struct-of-arrays, updated in place; lane i of a wavefront owns env i,
//                            so every load/store of a component is one fully coalesced 256-byte
//                            wave transaction
//   sbd        i32 [N]       steps_beyond_done (-1 = None); touched only by lanes whose env is done
//   reset_cnt  u32 [N]       resets drawn so far (RNG counter); touched only on reset
//   ep_ret/ep_len, last_ret/last_len   optional Monitor-style episode accumulators
//   totals     {u64,f64,u64} [ceil(N/64)]  per-wavefront partial sums of finished episodes
// Caller buffers: actions [T][nA][N] | [T][N][nA], obs [T][nS][N] | [T][N][nS], rew f32 [T][N],
// done u8 [T][N].
//
// One kernel template covers step (n_steps = 1) and the fused rollout (n_steps = T, state held in
// registers between steps, so per-step HBM traffic shrinks to actions-in + trajectory-out).
// Per-step constants arrive as kernel arguments (scalar registers via s_load), not LDS: they are
// wave-uniform, ~200 bytes, and an LDS copy would cost a barrier per launch for nothing.
#pragma once

#include "rmav_math.hpp"

namespace rmav {

enum : int { ACT_BUFFER = 0, ACT_RANDOM = 1, ACT_CONTROLLER = 2 };
enum : uint32_t { F_AUTO_RESET = 1u, F_TRACK = 2u, F_AOS = 4u };

constexpr int kBlock = 256;  // upper bound (launch bounds); the launch may use 64/128/256

struct Totals {
    unsigned long long episodes;
    double return_sum;
    unsigned long long length_sum;
};

struct RolloutArgs {
    float *state;
    int64_t n;
    const float *act_in;
    float *act_out;
    float *obs_out;
    float *rew_out;
    uint8_t *done_out;
    int32_t *sbd;
    uint32_t *reset_cnt;
    float *ep_ret;
    int32_t *ep_len;
    float *last_ret;
    int32_t *last_len;
    Totals *totals;
    uint64_t seed;
    uint64_t env_base;
    uint64_t t0;
    int32_t n_steps;
    uint32_t flags;
    float act_lo, act_hi;
};

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int K, int MODE>
__global__ __launch_bounds__(kBlock) void k_rollout(const RolloutArgs a,
                                                    const ParamsT<typename Env<K>::R> p,
                                                    const ParamsT<double> pc) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = a.n;
    const bool aos = (a.flags & F_AOS) != 0;
    const bool track = (a.flags & F_TRACK) != 0;
    const bool auto_reset = (a.flags & F_AUTO_RESET) != 0;

    unsigned int fin_n = 0, fin_len = 0;
    float fin_ret = 0.0f;

    if (i < n) {
        float s[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) s[c] = a.state[(int64_t)c * n + i];
        float er = 0.0f;
        int32_t el = 0;
        if (track) {
            er = a.ep_ret[i];
            el = a.ep_len[i];
        }
        // steps_beyond_done and the reset counter ride in registers for the whole launch: loading them
        // on demand (only lanes that terminate need them) would put one or two dependent HBM round
        // trips into every step of every wavefront that has a finishing lane (~57 % of them at the
        // 1.3 %/step termination rate of random actions).
        int32_t sb = a.sbd[i];
        uint32_t rc = a.reset_cnt[i];
        const int32_t sb0 = sb;
        const uint32_t rc0 = rc;
        const uint64_t env_id = a.env_base + (uint64_t)i;

        for (int32_t k = 0; k < a.n_steps; ++k) {
            float act[NA];
            if constexpr (MODE == ACT_BUFFER) {
                if (aos) {
                    const float *src = a.act_in + ((int64_t)k * n + i) * NA;
#pragma unroll
                    for (int c = 0; c < NA; ++c) act[c] = src[c];
                } else {
                    const float *src = a.act_in + (int64_t)k * NA * n + i;
#pragma unroll
                    for (int c = 0; c < NA; ++c) act[c] = src[(int64_t)c * n];
                }
            } else if constexpr (MODE == ACT_RANDOM) {
                random_action<K>(a.seed, env_id, a.t0 + (uint64_t)k, a.act_lo, a.act_hi, act);
            } else {
                env_control<K>(s, pc, act);
            }
            if (MODE != ACT_BUFFER && a.act_out) {
                if (aos) {
                    float *dst = a.act_out + ((int64_t)k * n + i) * NA;
#pragma unroll
                    for (int c = 0; c < NA; ++c) dst[c] = act[c];
                } else {
                    float *dst = a.act_out + (int64_t)k * NA * n + i;
#pragma unroll
                    for (int c = 0; c < NA; ++c) dst[(int64_t)c * n] = act[c];
                }
            }

            float dist;
            bool done;
            Env<K>::step(s, act, p, dist, done);

            // reward / steps_beyond_done machine  (quadrotor3d.py:112-122 and siblings)
            float r = -dist;
            if (done) {
                r = (sb < 0) ? 1.0f : 0.0f;
                sb = (sb < 0) ? 0 : sb + 1;
            }
            if (track) {
                er += r;
                el += 1;
                if (done) {
                    a.last_ret[i] = er;
                    a.last_len[i] = el;
                    fin_n += 1;
                    fin_len += (unsigned int)el;
                    fin_ret += er;
                    er = 0.0f;
                    el = 0;
                }
            }
            if (done && auto_reset) {
                reset_state<K>(a.seed, env_id, rc, s);
                rc += 1;
            }
            if (a.obs_out) {
                if (aos) {
                    float *dst = a.obs_out + ((int64_t)k * n + i) * NS;
#pragma unroll
                    for (int c = 0; c < NS; ++c) dst[c] = s[c];
                } else {
                    float *dst = a.obs_out + (int64_t)k * NS * n + i;
#pragma unroll
                    for (int c = 0; c < NS; ++c) dst[(int64_t)c * n] = s[c];
                }
            }
            if (a.rew_out) a.rew_out[(int64_t)k * n + i] = r;
            if (a.done_out) a.done_out[(int64_t)k * n + i] = done ? 1 : 0;
        }

#pragma unroll
        for (int c = 0; c < NS; ++c) a.state[(int64_t)c * n + i] = s[c];
        if (track) {
            a.ep_ret[i] = er;
            a.ep_len[i] = el;
        }
        if (sb != sb0) a.sbd[i] = sb;
        if (rc != rc0) a.reset_cnt[i] = rc;
    }

    if (track) {
        // Episode totals: each wavefront owns one slot of a [ceil(N/64)] partials array, so the adds never
        // contend (same-address device atomics cost ~12 ns each: ~600 finishing waves per step made the
        // single-step kernel 20 us slower than its memory time).  The adds are result-less atomics:
        // fire-and-forget at the L2, no load -> add -> store round trip at the tail of the kernel.
        // rmav_episode_totals sums the slots.
        const unsigned int wn = wave_sum(fin_n);
        const unsigned int wl = wave_sum(fin_len);
        const float wr = wave_sum(fin_ret);
        if ((threadIdx.x & 63) == 0 && wn != 0) {
            Totals *slot = a.totals + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
            atomicAdd(&slot->episodes, (unsigned long long)wn);
            atomicAdd(&slot->length_sum, (unsigned long long)wl);
            atomicAdd(&slot->return_sum, (double)wr);
        }
    }
}

// reset() of every env
template <int K>
__global__ __launch_bounds__(kBlock) void k_reset(float *state, int64_t n, uint32_t *reset_cnt,
                                                  float *ep_ret, int32_t *ep_len, float *obs_out,
                                                  uint64_t seed, uint64_t env_base, uint32_t flags) {
    constexpr int NS = Dims<K>::NS;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[NS];
    const uint32_t rc = reset_cnt[i];
    reset_state<K>(seed, env_base + (uint64_t)i, rc, s);
    reset_cnt[i] = rc + 1;
#pragma unroll
    for (int c = 0; c < NS; ++c) state[(int64_t)c * n + i] = s[c];
    if (flags & F_TRACK) {
        ep_ret[i] = 0.0f;
        ep_len[i] = 0;
    }
    if (obs_out) {
        if (flags & F_AOS) {
#pragma unroll
            for (int c = 0; c < NS; ++c) obs_out[i * NS + c] = s[c];
        } else {
#pragma unroll
            for (int c = 0; c < NS; ++c) obs_out[(int64_t)c * n + i] = s[c];
        }
    }
}

// control(): state -> action
template <int K>
__global__ __launch_bounds__(kBlock) void k_control(const float *state, int64_t n, float *act_out,
                                                    uint32_t flags, const ParamsT<double> pc) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[NS], act[NA];
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] = state[(int64_t)c * n + i];
    env_control<K>(s, pc, act);
    if (flags & F_AOS) {
#pragma unroll
        for (int c = 0; c < NA; ++c) act_out[i * NA + c] = act[c];
    } else {
#pragma unroll
        for (int c = 0; c < NA; ++c) act_out[(int64_t)c * n + i] = act[c];
    }
}

// [dim][n] <-> [n][dim]
__global__ __launch_bounds__(kBlock) void k_soa_to_aos(const float *src, float *dst, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < dim; ++c) dst[i * dim + c] = src[(int64_t)c * n + i];
}
__global__ __launch_bounds__(kBlock) void k_aos_to_soa(const float *src, float *dst, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < dim; ++c) dst[(int64_t)c * n + i] = src[i * dim + c];
}

}  // namespace rmav
