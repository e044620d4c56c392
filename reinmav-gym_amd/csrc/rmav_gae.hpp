// rmav_gae.hpp - generalised advantage estimation over the time-major [T][N] trajectory a fused rollout leaves
// in HBM (the PPO2 caller loop of gym_reinmav/run.py:63-68; baselines ppo2 Runner.run() computes
//   delta_t = r_t + gamma V_{t+1} (1 - done_t) - V_t ,   A_t = delta_t + gamma lambda (1 - done_t) A_{t+1}
// backwards over the nsteps it collected, returns = A + V).
//
// One env per lane, like the dynamics kernels: lane i walks its own column backwards, so every access of a
// time step is one coalesced 256-byte (64-byte for `done`) wave transaction and the recurrence lives in two
// registers.  9 bytes read + 8 written per sample: a pure HBM stream.  The loads of a chunk of kGaeUnroll
// steps do not depend on the recurrence, so they are all issued before the first one is consumed.
// The kernel also leaves sum / sum-of-squares of the advantages (per-block partials in fp64, folded by
// k_gae_fold) for the advantage normalisation of the learner.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace rmav {

constexpr int kGaeUnroll = 8;

__global__ __launch_bounds__(256) void k_gae(const float *__restrict__ rew, const uint8_t *__restrict__ done,
                                             const float *__restrict__ val, float *__restrict__ adv,
                                             float *__restrict__ ret, int64_t n, int32_t T, float gamma, float lam,
                                             float rew_scale, double *__restrict__ partial) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float s1 = 0.0f, s2 = 0.0f;
    if (i < n) {
        float v_next = val[(int64_t)T * n + i];
        float last = 0.0f;
        const float gl = gamma * lam;
        int32_t t = T - 1;
        // head: bring t + 1 to a multiple of the unroll factor
        for (; t >= 0 && ((t + 1) % kGaeUnroll) != 0; --t) {
            const int64_t o = (int64_t)t * n + i;
            const float nt = done[o] ? 0.0f : 1.0f, v = val[o];
            const float delta = fmaf(gamma * nt, v_next, fmaf(rew[o], rew_scale, -v));
            last = fmaf(gl * nt, last, delta);
            adv[o] = last;
            ret[o] = last + v;
            s1 += last;
            s2 = fmaf(last, last, s2);
            v_next = v;
        }
        for (; t >= 0; t -= kGaeUnroll) {
            float r[kGaeUnroll], v[kGaeUnroll], nt[kGaeUnroll];
#pragma unroll
            for (int j = 0; j < kGaeUnroll; ++j) {
                const int64_t o = (int64_t)(t - j) * n + i;
                r[j] = rew[o];
                v[j] = val[o];
                nt[j] = done[o] ? 0.0f : 1.0f;
            }
#pragma unroll
            for (int j = 0; j < kGaeUnroll; ++j) {
                const int64_t o = (int64_t)(t - j) * n + i;
                const float delta = fmaf(gamma * nt[j], v_next, fmaf(r[j], rew_scale, -v[j]));
                last = fmaf(gl * nt[j], last, delta);
                adv[o] = last;
                ret[o] = last + v[j];
                s1 += last;
                s2 = fmaf(last, last, s2);
                v_next = v[j];
            }
        }
    }
    if (partial) {   // block partial of (sum A, sum A^2); uniform branch
        __shared__ double sh[2][4];
        double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            d1 += __shfl_down(d1, off, 64);
            d2 += __shfl_down(d2, off, 64);
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            sh[0][w] = d1;
            sh[1][w] = d2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a1 = 0.0, a2 = 0.0;
            for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
                a1 += sh[0][k];
                a2 += sh[1][k];
            }
            partial[2 * blockIdx.x] = a1;
            partial[2 * blockIdx.x + 1] = a2;
        }
    }
}

// one block: sums_out[0..1] = (sum A, sum A^2) over all blocks' partials
__global__ __launch_bounds__(256) void k_gae_fold(const double *__restrict__ partial, int nblocks, double *__restrict__ sums_out) {
    double a1 = 0.0, a2 = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
        a1 += partial[2 * b];
        a2 += partial[2 * b + 1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a1 += __shfl_down(a1, off, 64);
        a2 += __shfl_down(a2, off, 64);
    }
    __shared__ double sh[2][4];
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = a1;
        sh[1][threadIdx.x >> 6] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums_out[0] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        sums_out[1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// x <- (x - mean) * rstd, 16 bytes per lane, grid-stride; count4 = count / 4 full quads, the tail by scalar lanes
__global__ __launch_bounds__(256) void k_affine(float *__restrict__ x, int64_t count, float mean, float rstd) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = count >> 2;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    for (int64_t q = tid; q < n4; q += stride) {
        float4 v = x4[q];
        v.x = (v.x - mean) * rstd;
        v.y = (v.y - mean) * rstd;
        v.z = (v.z - mean) * rstd;
        v.w = (v.w - mean) * rstd;
        x4[q] = v;
    }
    for (int64_t q = (n4 << 2) + tid; q < count; q += stride) x[q] = (x[q] - mean) * rstd;
}

// ---- policy weights -> the buffer rmav_rollout_policy reads ---------------------------------------------------
// out word i = flat[lo[i]]  (hi[i] < 0), or the bf16 pair (flat[lo[i]], flat[hi[i]]) in one word (low half first), where `flat`
// is the concatenation of the caller's parameter tensors (<= kPackMaxParams of them) followed by zeros.  The layouts of
// include/rmav.h are fixed permutations + zero padding (+ bf16 rounding) of the parameters, so one gather launch replaces
// the ~8 dependent torch launches (cat, index, convert, cat, copy: ~35 us) a repack used to cost before every rollout.
constexpr int kPackMaxParams = 16;
struct PackSrc {
    const float *p[kPackMaxParams];
    int32_t end[kPackMaxParams];   // exclusive prefix ends of the parameters inside `flat`
    int32_t n;
};
__device__ __forceinline__ float pack_fetch(const PackSrc &src, int32_t j) {
    int32_t begin = 0;
#pragma unroll
    for (int k = 0; k < kPackMaxParams; ++k) {
        if (k < src.n && j >= begin && j < src.end[k]) return src.p[k][j - begin];
        if (k < src.n) begin = src.end[k];
    }
    return 0.0f;   // the appended zero (padding)
}
// F16: the pair words are f16 (round to nearest even) and word i of the MfmaLayout buffer is pre-scaled by `scale2` inside
// the layer-2 fragments, by `scale3` inside the layer-3 fragments (rmav_pack_policy_f16: tanh folded into the next layer).
template <bool F16>
__global__ __launch_bounds__(256) void k_pack_policy(const PackSrc src, const int32_t *__restrict__ lo, const int32_t *__restrict__ hi,
                                                     int64_t n_out, float *__restrict__ out, int32_t net_words, int32_t a2_begin,
                                                     int32_t a3_begin, int32_t a3_end, float scale2, float scale3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const float a = pack_fetch(src, lo[i]);
    const int32_t h = hi[i];
    if (h < 0) {
        out[i] = a;
    } else {
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        f32x2_t v = {a, pack_fetch(src, h)};
        if constexpr (F16) {
            typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
            const int32_t o = (int32_t)(i % net_words);
            const float sc = (i < 2 * (int64_t)net_words && o >= a2_begin && o < a3_end) ? (o < a3_begin ? scale2 : scale3) : 1.0f;
            v = v * sc;
            out[i] = __builtin_bit_cast(float, __builtin_convertvector(v, f16x2_t));
        } else {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            out[i] = __builtin_bit_cast(float, __builtin_convertvector(v, bf16x2_t));   // round to nearest even, as torch's .to(bfloat16)
        }
    }
}

// running episode lengths for rmav_episode_buffers: clock - ep_start
__global__ __launch_bounds__(256) void k_cur_length(int32_t *out, const EnvRec *rec, uint32_t clock, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)(clock - rec[i].ep_start);
}

// the episode clock moved by `delta` (rmav_seed, rmav_set_step_count): every running episode's start moves with it
__global__ __launch_bounds__(256) void k_shift_ep_start(EnvRec *rec, uint32_t delta, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rec[i].ep_start += delta;
}

// One 32-bit field of the per-env records <-> a dense array (rmav_get_sbd / rmav_set_sbd, the reset counters, last lengths: the
// accessors of the C ABI; not on any hot path).  field = word index in EnvRec: 0 sbd, 1 reset_cnt, 2 ep_start, 3 last_len.
__global__ __launch_bounds__(256) void k_rec_get(uint32_t *out, const EnvRec *rec, int field, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = reinterpret_cast<const uint32_t *>(rec)[4 * i + field];
}
__global__ __launch_bounds__(256) void k_rec_set(EnvRec *rec, const uint32_t *in, int field, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<uint32_t *>(rec)[4 * i + field] = in[i];
}
// field < 0: every record = `value`; otherwise that field of every record = value's
__global__ __launch_bounds__(256) void k_rec_fill(EnvRec *rec, EnvRec value, int field, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (field < 0) rec[i] = value;
    else reinterpret_cast<uint32_t *>(rec)[4 * i + field] = reinterpret_cast<const uint32_t *>(&value)[field];
}

// ---- episode statistics exchange (the path's one collective) ------------------------------------------------
// send = [2][cmax] int32: returns (bit pattern) then lengths of this rank's `count` envs, zero padded to cmax
__global__ __launch_bounds__(256) void k_pack_stats(const float *__restrict__ last_ret, const EnvRec *__restrict__ rec,
                                                    int64_t count, int64_t cmax, int32_t *__restrict__ send) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cmax) return;
    send[i] = i < count ? __float_as_int(last_ret[i]) : 0;
    send[cmax + i] = i < count ? rec[i].last_len : 0;
}
// one thread: publish `seq` in a signal word another HIP stream waits on with hipStreamWaitValue32 (the kernel boundary
// in front of this launch has released the payload).  Folding this into k_pack_stats - every workgroup releases and takes
// a ticket, the last one signals - measured SLOWER (exchange cost per 131 072-env rollout +15..19 us instead of +9..15:
// 512 agent-scope releases each write the L2 back) - profiles/r02/handover_chunk.md.
__global__ void k_signal(uint32_t *flag, uint32_t seq) {
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The communicator's stream waits here until every wavefront of the armed rollout launch has published `seq` (or a later
// post's) in its arrival word: one workgroup polling `count` words with agent-scope loads.  hipStreamWaitValue32 is a
// spinning one-wavefront kernel of the runtime as well (__amd_rocclr_streamOpsWait in the kernel trace); this one waits
// for the rollout kernel's own wavefronts, so the compute stream needs no pack and no signal kernel.
// Bounded, so that a communicator stream can never spin for ever (it would block every other rank's collective too):
//   * `after_start_ticks` (2 s of the constant 100 MHz clock) from the moment the armed launch's first workgroup published
//     `seq` in *started - NOT from when this waiter began: an armed rollout may sit behind seconds of queued compute-stream work
//     (a PPO update, another job on the GPU), and a clock that started here gave up on launches that had not begun yet
//     (ADVICE r03);
//   * `total_ticks` (10 min) overall, against a launch that never runs at all (device fault after a successful enqueue).
// On giving up the waiter POISONS this rank's payload - return = NaN, length = -1 for every env - writes `seq` into
// *timeout_seq (pinned host memory, one word per buffer pair: rmav_allgather_stats_wait / _result report RMAV_ERR_TIMEOUT for
// THAT post only) and lets the stream go on: the collective is still issued, so the peers neither hang nor mistake the
// half-written snapshot for statistics.
__global__ __launch_bounds__(256) void k_wait_arrivals(const uint32_t *arrive, uint32_t count, uint32_t seq, const uint32_t *started,
                                                       unsigned long long after_start_ticks, unsigned long long total_ticks,
                                                       uint32_t *timeout_seq, int32_t *send, int64_t cmax) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long t_start = 0;   // per thread: when THIS thread first saw the launch begun (the give-up decision is made uniform below)
    for (;;) {
        int ok = 1;
        for (uint32_t i = threadIdx.x; i < count; i += 256u)
            ok &= (int32_t)(__hip_atomic_load(arrive + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0;
        if (__syncthreads_and(ok)) break;
        const unsigned long long now = wall_clock64();
        // (started == nullptr: a kernel family that publishes no start word - only the overall bound applies)
        const int begun = started ? __syncthreads_or((int32_t)(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0) : 0;
        // (signed differences: the threads read the clock a few ticks apart, and an unsigned `now - t_start` of a thread that read
        // it just BEFORE the one whose reading became t_start would wrap around to "expired")
        int expired = (long long)(now - t0) > (long long)total_ticks;
        if (begun) {
            if (t_start == 0) t_start = now | 1ull;
            expired |= (long long)(now - t_start) > (long long)after_start_ticks;
        }
        if (__syncthreads_or(expired)) {
            for (int64_t i = threadIdx.x; i < cmax; i += 256) {
                send[i] = 0x7fc00000;      // NaN
                send[cmax + i] = -1;
            }
            if (threadIdx.x == 0) __hip_atomic_store(timeout_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}
// recv = [world][2][cmax] -> returns_out / lengths_out [n_total] in global env order (rank r owns
// base + (r < rem) envs starting at r * base + min(r, rem))
__global__ __launch_bounds__(256) void k_unpack_stats(const int32_t *__restrict__ recv, int64_t n_total, int32_t world,
                                                      int64_t cmax, float *__restrict__ ret_out, int32_t *__restrict__ len_out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_total) return;
    const int64_t base = n_total / world, rem = n_total % world;
    // owner of global env g: the first `rem` ranks hold base + 1 envs
    const int64_t cut = rem * (base + 1);
    int64_t r, local;
    if (g < cut) {
        r = g / (base + 1);
        local = g - r * (base + 1);
    } else {
        r = rem + (g - cut) / base;
        local = (g - cut) - (r - rem) * base;
    }
    const int32_t *src = recv + r * 2 * cmax;
    ret_out[g] = __int_as_float(src[local]);
    len_out[g] = src[cmax + local];
}

}  // namespace rmav
