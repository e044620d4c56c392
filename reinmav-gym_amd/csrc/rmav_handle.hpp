// rmav_handle.hpp - what the translation units behind the C ABI share: the handle / communicator structs, the error helper, and
// the one entry point of the second translation unit.  librmav.so is built from two of them because the policy-in-kernel
// rollouts must be compiled WITHOUT the SLP vectoriser (rmav_policy_abi.hip explains why); everything else keeps it.
#pragma once

#include "../../include/rmav.h"

#include <hip/hip_runtime.h>

#include <cstdint>

#include <rccl/rccl.h>   // types only: the RCCL entry points are resolved with dlopen/dlsym on first use

#include "rmav_derive.hpp"
#include "rmav_kernels.hpp"

#define RMAV_INTERNAL __attribute__((visibility("hidden")))

// sets the thread-local message rmav_last_error() returns and hands `code` back (defined in rmav_abi.hip)
RMAV_INTERNAL int rmav_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return rmav_fail(RMAV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                             __FILE__, __LINE__);                                                  \
    } while (0)

constexpr uint32_t kMagic = 0x524d4156u;  // 'RMAV'
constexpr int kNumKinds = 5;
constexpr int kStateDim[kNumKinds] = {5, 9, 10, 16, 13};
constexpr int kActionDim[kNumKinds] = {2, 2, 4, 4, 4};

struct rmav_env_s {
    uint32_t magic;
    int kind;
    int64_t n;
    int device;
    uint64_t seed;
    uint64_t env_base;
    uint32_t flags;
    rmav_params params;
    hipStream_t stream;
    bool own_stream;
    uint64_t t;  // global step counter
    int64_t chunk;  // > 0 only inside rmav_rollout_chunked: the call's trajectory arrays are chunk-major [n_chunks][T][dim][chunk]
    // device-resident env data
    float *state;
    rmav::EnvRec *rec;   // per-env termination record {sbd, reset_cnt, ep_start, last_len} (rmav_kernels.hpp: EnvRec, ep_clock0)
    float *ep_ret, *last_ret;
    rmav::Totals *totals;
    double *env_time;  // RMAV_REINMAV only
    void *arena;       // ONE allocation behind all of the arrays above (see rmav_create)
    float *pe[3];      // per-env constants (rmav_set_env_param), nullptr = shared
    // scratch for host-pointer calls and layout conversion (grown on demand)
    void *scratch;
    size_t scratch_bytes;
    // small host-pointer calls (the gym-shaped single env, batch <= a few thousand): one block of pinned,
    // device-mapped host memory.  The kernel reads the actions from it and writes obs / reward / done into it
    // over PCIe, so such a call is one launch + one stream synchronise - no staging copies at all.
    void *pinned;
    void *pinned_dev;
    size_t pinned_bytes;
    // completion word of single-wavefront k_step launches through the pinned block (RolloutArgs::done_flag)
    uint32_t *done_flag, *done_flag_dev;
    uint32_t done_seq;
    // statistics exchange armed for the next fused rollout launch (rmav_allgather_stats_arm): where that launch's wavefronts
    // snapshot their envs' statistics and publish their arrival; `fired` once a launch has taken it
    struct {
        bool armed, fired;
        bool allow;   // the call in progress is ONE fused launch over all envs (set by rollout_impl / rmav_rollout_policy)
        bool stale;   // another stepping launch followed the one that took the snapshot: _post must pack again
        bool no_start;   // the launch that took it does not publish a start word (rmav::publishes_start): bounded by the overall limit only
        struct rmav_comm_s *comm;
        int slot;
        int64_t cmax;
        uint32_t seq, expected;
    } xchg;
    // explicit per-handle overrides of the launch heuristics (rmav_set_tuning); -1 / 0 = automatic
    int tune[RMAV_TUNE_COUNT];
};

constexpr int kExchangeDepth = 8;   // buffer pairs of the overlapped statistics exchange
// bounds of k_wait_arrivals, in ticks of the 100 MHz wall clock: 2 s once the armed launch has begun, 10 min overall
constexpr unsigned long long kArrivalWaitTicks = 200000000ull, kArrivalTotalTicks = 60000000000ull;
struct rmav_comm_s {
    uint32_t magic;
    int rank, world, device;
    ncclComm_t comm;
    // overlapped exchange: the collective runs on the communicator's own stream, double-buffered
    hipStream_t stream;
    hipEvent_t ready[kExchangeDepth], done[kExchangeDepth];
    bool used[kExchangeDepth];
    int32_t *send[kExchangeDepth], *recv[kExchangeDepth];
    int depth;         // buffer pairs in use (RMAV_EXCHANGE_DEPTH, 2 .. kExchangeDepth)
    uint32_t *arrive;  // arrival words of armed launches, one per wavefront: ceil(cmax / 32) of them
    uint32_t *flag;    // signal word (hipMallocSignalMemory): the compute stream publishes post numbers, the comm stream waits
    int64_t cmax;      // capacity of the buffers (per-rank slots of 2 * cmax int32)
    int posts;         // number of posts so far (buffer pair of post i is i % depth)
    struct rmav_env_s *armed_by;   // the handle whose armed exchange points at this communicator (cleared by _post)
    uint32_t *started;             // device word: the armed launch's first workgroup publishes its post number here
    // pinned host words (device-mapped), one per buffer pair: k_wait_arrivals writes the post number it gave up on
    uint32_t *timeout_seq, *timeout_seq_dev;
    uint32_t slot_seq[kExchangeDepth];   // post number that last used each buffer pair
    bool armed_slot[kExchangeDepth];     // ... and whether that post went through the waiter (an armed launch)
};


// Workgroup size of the one-wavefront-per-64-envs kernels: 256, or rmav_set_tuning(RMAV_TUNE_BLOCK, 64 | 128 | 256).
inline int block_size(rmav_handle h) {
    const int v = h->tune[RMAV_TUNE_BLOCK];
    return (v == 64 || v == 128 || v == 256) ? v : 256;
}
inline dim3 grid_for(rmav_handle h) { return dim3((unsigned)((h->n + block_size(h) - 1) / block_size(h))); }

// An armed statistics exchange rides on the first call after rmav_allgather_stats_arm that is ONE fused launch over all
// envs (xchg.allow: rollout_impl with fused != 0 / rmav_rollout_policy; not the fused = 0 loop of single-step launches,
// whose first launch would snapshot the statistics T - 1 steps early, and not a sliced launch).  Any later stepping
// launch makes that snapshot stale, and _post then packs afresh.  envs_per_word: envs behind one arrival word (64; 32 for
// the fp32-MFMA actor's half-wavefront layout).
inline void take_armed_exchange(rmav_handle h, rmav::RolloutArgs &a, int envs_per_word, bool publishes_start = true) {
    if (h->xchg.armed && h->xchg.fired) h->xchg.stale = true;
    if (h->xchg.armed && !h->xchg.fired && h->xchg.allow && a.slice_count == 0 && (a.flags & rmav::F_TRACK)) {
        rmav_comm_s *c = h->xchg.comm;
        a.xsend = c->send[h->xchg.slot];
        a.xcmax = h->xchg.cmax;
        a.xarrive = c->arrive;
        a.xseq = h->xchg.seq;
        a.xstarted = c->started;
        h->xchg.expected = (uint32_t)((h->n + envs_per_word - 1) / envs_per_word);
        h->xchg.fired = true;
        h->xchg.no_start = !publishes_start;
    }
}

// Right after the hipLaunchKernelGGL of a rollout that may have taken the armed exchange: a launch that failed will never publish
// its arrival words, so the exchange goes back to "armed, not fired" and rmav_allgather_stats_post packs as usual.
inline int check_rollout_launch(rmav_handle h, const rmav::RolloutArgs &a) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return RMAV_OK;
    if (a.xsend) h->xchg.fired = false;
    return rmav_fail(RMAV_ERR_HIP, "rollout kernel launch failed: %s", hipGetErrorString(e));
}

// rmav_policy_abi.hip: launches rmav_rollout_policy's kernel for kmode = RMAV_ACT_POLICY | RMAV_ACT_POLICY_BF16 | ACT_POLICY_F32M |
// ACT_POLICY_F16 | ACT_POLICY_F16_SHARED on the handle's stream
RMAV_INTERNAL int rmav_launch_policy_rollout(rmav_handle h, int kmode, const rmav::RolloutArgs &a);
