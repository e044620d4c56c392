// rmav_abi.hip - the C ABI of include/rmav.h: handle management, launches, host/device staging.
// There is deliberately no CPU implementation in this library: without a GPU rmav_create fails.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>

#include <dlfcn.h>

#include "rmav_handle.hpp"
#include "rmav_gae.hpp"

using namespace rmav;

namespace {

thread_local char g_err[768] = "";

}  // namespace

int rmav_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

bool valid(rmav_handle h) { return h && h->magic == kMagic; }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

#define CHECK_HANDLE(h)                                                                            \
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");                           \
    DeviceGuard guard_(h->device);                                                                 \
    if (!guard_.ok) return rmav_fail(RMAV_ERR_HIP, "hipSetDevice(%d) failed", h->device)

int check_params(const rmav_params &q) {
    if (q.integrator != RMAV_INT_EULER && q.integrator != RMAV_INT_RK4)
        return rmav_fail(RMAV_ERR_INVALID, "rmav_params.integrator must be RMAV_INT_EULER or RMAV_INT_RK4");
    if (!(q.mass > 0) || !(q.dt > 0) || !(q.tau != 0) || !(q.mass + q.load_mass > 0))
        return rmav_fail(RMAV_ERR_INVALID, "rmav_params: mass, dt must be > 0 and tau != 0");
    for (int i = 0; i < 3; ++i)
        if (!(q.g_vec[i] == q.g_vec[i]) || q.g_vec[i] - q.g_vec[i] != 0.0)
            return rmav_fail(RMAV_ERR_INVALID, "rmav_params.g_vec must be finite");
    return RMAV_OK;
}

// slots of the per-wavefront episode totals: one per 32 envs (the fp32-MFMA policy mode runs 32 envs per wavefront)
// + 4: k_step reads its wavefront's slot in EVERY wavefront of the launch grid, also in those of the last 256-thread workgroup that lie
// wholly past N (they add nothing and write nothing); the padding keeps those reads inside the array
inline size_t n_total_slots(int64_t n) { return (size_t)((n + 31) / 32) + 4; }


int ensure_scratch(rmav_handle h, size_t bytes) {
    if (bytes <= h->scratch_bytes) return RMAV_OK;
    if (h->scratch) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipFree(h->scratch));
        h->scratch = nullptr;
        h->scratch_bytes = 0;
    }
    size_t want = bytes + (bytes >> 2);
    if (hipMalloc(&h->scratch, want) != hipSuccess) {
        (void)hipGetLastError();
        return rmav_fail(RMAV_ERR_ALLOC, "hipMalloc(%zu) for scratch failed", want);
    }
    h->scratch_bytes = want;
    return RMAV_OK;
}

// Host-pointer calls that move at most this many bytes go through the pinned block (zero-copy); bigger ones
// stage through device scratch with hipMemcpyAsync, which is the faster route for bulk data.
constexpr size_t kPinnedMax = 256u << 10;

int ensure_pinned(rmav_handle h, size_t bytes) {
    if (bytes <= h->pinned_bytes) return RMAV_OK;
    if (h->pinned) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipHostFree(h->pinned));
        h->pinned = h->pinned_dev = nullptr;
        h->pinned_bytes = 0;
    }
    size_t want = bytes < 4096 ? 4096 : bytes;
    if (hipHostMalloc(&h->pinned, want, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        h->pinned = nullptr;
        return rmav_fail(RMAV_ERR_ALLOC, "hipHostMalloc(%zu) for the pinned staging block failed", want);
    }
    if (hipHostGetDevicePointer(&h->pinned_dev, h->pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipHostFree(h->pinned);
        h->pinned = nullptr;
        return rmav_fail(RMAV_ERR_HIP, "hipHostGetDevicePointer failed");
    }
    h->pinned_bytes = want;
    return RMAV_OK;
}

// Cache policy of the trajectory stores for one launch.  Measured with COLD trajectory buffers (bench.py's ring of
// buffer sets; profiles/r02/policy_split_sweep.md), 64-step launches, every kind, 65 536 .. 1 048 576 envs:
//   two-wavefront kernel : write-through (sc0 sc1) is best or within 1 % of best everywhere
//   one-wavefront kernel : non-temporal (nt) is best from 131 072 envs up (+3..7 % over the default policy, which is
//                          never the best choice for a fused launch); small trajectories that stay in the Infinity
//                          Cache for their consumer keep write-through
// Single-step and very short launches use the default policy.  rmav_set_tuning(RMAV_TUNE_STORE_POLICY, 0|1|2|3) overrides.
int pick_store_policy(rmav_handle h, const RolloutArgs &a, bool split) {
    const int forced = h->tune[RMAV_TUNE_STORE_POLICY];
    if (forced >= 0 && forced <= 2 && !(a.flags & F_AOS)) return forced;
    if (a.n_steps < 8 && !split) return ST_DEFAULT;
    double per_step = 0.0;
    if (a.act_out) per_step += 4.0 * kActionDim[h->kind];
    if (a.obs_out) per_step += 4.0 * kStateDim[h->kind];
    if (a.rew_out) per_step += 4.0;
    if (a.done_out) per_step += 1.0;
    const double bytes = per_step * (double)h->n * (double)a.n_steps;
    if (a.flags & F_AOS) {   // batch-major trajectories: LDS-transposed obs stores once the launch is big (RMAV_TUNE_STORE_POLICY 3: always, 0: never)
        if (forced == 0) return ST_DEFAULT;
        if (split) return ST_WRITE_THROUGH;   // the memory wavefront drains batch-major tiles itself
        // measured (profiles/r01/layout_sweep.md): pays from 131 072 envs x 64 steps (512 MB), costs 10-15 % at 65 536 (256 MB)
        return (a.obs_out && (bytes >= 448.0e6 || forced == ST_AOS_LDS)) ? ST_AOS_LDS : ST_WRITE_THROUGH;
    }
    // Feature-major columns are 4 N bytes apart: unless N is a multiple of 16 they start off a 64-byte line, every wavefront's
    // 256-byte store ends in partial lines, and only the write-back L2 can merge them with the neighbouring wavefront's part -
    // write-through / non-temporal stores send the fragments on (same box: 65 599 envs 100.4 us per launch, write-back 67.7;
    // 131 071: 200.9 -> 116.5; 1 048 575: 1700 -> 1564; the aligned sizes next to them: 48.8, 90.8, 730).
    if ((a.pitch & 15) != 0) return ST_DEFAULT;   // (rmav_rollout_pitched: the caller padded the columns, the batch size is free)
    if (split) return ST_WRITE_THROUGH;
    return bytes <= 192.0e6 ? ST_WRITE_THROUGH : ST_STREAM;
}

// Two-wavefront (integrator + memory wavefront) kernel or one wavefront per 64 envs?  Measured on cold trajectory
// buffers (profiles/r02/split_autog.md: every kind, both in-kernel action sources, 16 384 .. 163 840 envs): the
// two-wavefront kernel wins or ties whenever the whole batch fits ONE workgroup per CU - ceil(N / 16 384) pairs per
// workgroup, 256 workgroups - and loses as soon as it does not (a second round of workgroups, or pairs capped by the
// 1024-thread / 160 KiB-LDS limits).  So the rule is capacity, not a tuned constant: two wavefronts iff
// N <= 16 384 x (pairs that fit one workgroup for this kind and action source).
template <int K, int SMODE> constexpr int split_pairs_max() {   // SMODE: ACT_RANDOM_SPLIT | ACT_CONTROLLER_SPLIT | ACT_BUFFER_SPLIT
    constexpr int by_lds = (int)((160u << 10) / (sizeof(float) * split_words_per_pair<K, SMODE>()));
    constexpr int cap = split_group_cap<K, split_feeds_actions(SMODE)>();   // threads / registers (rmav_kernels.hpp)
    return by_lds < cap ? by_lds : cap;
}
#define RMAV_PAIRS(SMODE) {split_pairs_max<QUAD2D, SMODE>(), split_pairs_max<QUAD2D_SL, SMODE>(), split_pairs_max<QUAD3D, SMODE>(), split_pairs_max<QUAD3D_SL, SMODE>()}
constexpr int kSplitPairsRandom[4] = RMAV_PAIRS(ACT_RANDOM_SPLIT);
constexpr int kSplitPairsController[4] = RMAV_PAIRS(ACT_CONTROLLER_SPLIT);
constexpr int kSplitPairsBuffer[4] = RMAV_PAIRS(ACT_BUFFER_SPLIT);
#undef RMAV_PAIRS
// pairs of one workgroup by the public action mode
inline const int *split_pairs_of(int action_mode) {
    return action_mode == RMAV_ACT_CONTROLLER ? kSplitPairsController : action_mode == RMAV_ACT_BUFFER ? kSplitPairsBuffer : kSplitPairsRandom;
}
constexpr int64_t kEnvsPerCuSlot = 16384;   // 256 CUs x 64 envs: one pair per CU
constexpr bool kSliceByDefault = false;     // sliced two-wavefront launches beyond the capacity: measured, see DESIGN.md

template <int K, int MODE, int ST>
int launch_rollout_kms(rmav_handle h, const RolloutArgs &a_in) {
    RolloutArgs a = a_in;
    take_armed_exchange(h, a, 64, publishes_start(MODE));
    const typename Env<K>::P p = derive_env<K>(h->params);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
    static_assert(!is_policy(MODE), "the policy-in-kernel rollouts are launched from rmav_policy_abi.hip");
    const size_t lds = (ST == ST_AOS_LDS) ? sizeof(float) * AosTile<Dims<K>::NS>::WORDS * (block_size(h) / 64) : 0;
    if constexpr (is_split(MODE)) {
        // (integrator, memory wavefront) pairs: as many per workgroup as make ONE workgroup per CU (256 workgroups),
        // within 1024 threads and the CU's 160 KiB of LDS.  RMAV_TUNE_SPLIT_GROUP = 1..8 overrides.
        constexpr size_t lds_per_pair = sizeof(float) * split_words_per_pair<K, MODE>();
        const int forced = h->tune[RMAV_TUNE_SPLIT_GROUP];
        constexpr int g_max = split_pairs_max<K, MODE>();
        const int64_t count = a.slice_count ? (int64_t)a.slice_count : h->n;   // envs of this launch
        // Round 3 (profiles/r03/pairs_per_workgroup.md): one s_barrier synchronises ALL pairs of a workgroup, so every pair
        // pays for the slowest one's reset path each step.  Where the step time is the integrator's latency (the 2-D kinds,
        // whose steps move half the bytes, and every controller-driven rollout: fp64 controller in the integrator) rather than
        // the store stream, ONE pair per workgroup is faster: 65 536 envs quadrotor2d-slungload 47.2 -> 43.1 us per 64-step launch,
        // controller-driven quadrotor3d 56.7 -> 52.3.  The store-bound combinations (3-D kinds with random / caller actions) keep
        // one workgroup per CU (adjacent pairs store adjacent 256-byte segments: 65 536 envs quadrotor3d 41.6 vs 50.7 us), and so
        // do the others at 131 072 envs, where two pairs share every SIMD anyway (quadrotor2d excepted: 51.6 vs 54.5).
        constexpr bool latency_bound = (K == QUAD2D || K == QUAD2D_SL) || MODE == ACT_CONTROLLER_SPLIT;
        const bool one_pair = latency_bound && (count <= 98304 || K == QUAD2D);
        int g = (forced >= 1 && forced <= g_max) ? forced : one_pair ? 1 : (int)((count + kEnvsPerCuSlot - 1) / kEnvsPerCuSlot);
        if (g < 1) g = 1;
        if (g > g_max) g = g_max;
        const int64_t per_wg = 64 * g;
        // lean addressing in the memory wavefront: feature-major, every trajectory array below 4 GiB (32-bit scalar step offsets)
        if (!(a.flags & F_AOS) && (int64_t)a.n_steps * Dims<K>::NS * a.pitch < ((int64_t)1 << 30)) a.flags |= F_LEAN;
        const dim3 grid((unsigned)((count + per_wg - 1) / per_wg)), block(128 * g);
        bool launched = false;
        if constexpr (ST == ST_WRITE_THROUGH) {   // the usual options, compiled in (k_rollout's FIXED)
            if ((a.flags & (F_AOS | F_TRACK | F_AUTO_RESET)) == (F_TRACK | F_AUTO_RESET)) {
                hipLaunchKernelGGL((k_rollout<K, MODE, ST, true>), grid, block, lds_per_pair * g, h->stream, a, p, pc);
                launched = true;
            }
        }
        if (!launched) hipLaunchKernelGGL((k_rollout<K, MODE, ST>), grid, block, lds_per_pair * g, h->stream, a, p, pc);
    } else {
        hipLaunchKernelGGL((k_rollout<K, MODE, ST>), grid_for(h), dim3(block_size(h)), lds, h->stream, a, p, pc);
    }
    return check_rollout_launch(h, a);
}

// RMAV_TUNE_SPLIT = 0 | 1 overrides the rule.
// Batches beyond that capacity can still run on the two-wavefront kernel as a sequence of launches over balanced
// slices of the env range, each one workgroup per CU (`slices` > 1): see launch_rollout_km.  RMAV_TUNE_SLICE = 0 | 1 overrides.
bool use_split(rmav_handle h, const RolloutArgs &a, int action_mode, int *slices, bool random_actions) {
    const int forced = h->tune[RMAV_TUNE_SPLIT], slice_forced = h->tune[RMAV_TUNE_SLICE];
    *slices = 1;
    if (h->chunk > 0) {   // rmav_rollout_chunked: one two-wavefront launch per chunk, whatever the other rules say
        *slices = (int)((h->n + h->chunk - 1) / h->chunk);
        return true;
    }
    // the two-wavefront kernel also wins for short launches (2 .. 7 steps: -15 .. -30 %, measured)
    if (a.n_steps < 2 || h->kind > RMAV_QUAD3D_SL) return false;
    const int64_t cap = kEnvsPerCuSlot * split_pairs_of(action_mode)[h->kind];
    if (forced == 0) return false;
    if (h->n <= cap) return true;
    // Two rounds of the two-wavefront kernel - two launches over balanced halves of the env range - beat one launch of the
    // one-wavefront kernel for the slung-load kinds (fp64 integrator: the one-wavefront kernel holds only 3-4 of them per SIMD)
    // when both halves (nearly) fill the machine, 1.75 .. 2 x the capacity, random actions (profiles/r02/slice_two_rounds.md:
    // BASELINE C4 = quadrotor3d-slungload at 262 144 envs 269-292 -> 249-252 us, quadrotor2d-slungload 172 -> 158).  Smaller
    // second halves, more than two rounds, the plain kinds (quadrotor3d: +-4 %, quadrotor2d: slower) and the
    // controller-driven rollouts measured equal or slower, so they stay on one launch.
    const bool two_rounds = random_actions && (h->kind == RMAV_QUAD3D_SL || h->kind == RMAV_QUAD2D_SL) && h->n <= 2 * cap &&
                            4 * h->n >= 7 * cap;
    if (slice_forced == 1 || (slice_forced != 0 && (kSliceByDefault || two_rounds))) {
        *slices = (int)((h->n + cap - 1) / cap);
        return true;
    }
    return forced == 1;   // one launch beyond the capacity: only when asked for
}

template <int K, int MODE>
int launch_rollout_km(rmav_handle h, const RolloutArgs &a) {
    {
        if constexpr ((MODE == ACT_RANDOM || MODE == ACT_CONTROLLER || MODE == ACT_BUFFER) && K != REINMAV) {
            constexpr int SMODE = (MODE == ACT_RANDOM) ? ACT_RANDOM_SPLIT : (MODE == ACT_BUFFER) ? ACT_BUFFER_SPLIT : ACT_CONTROLLER_SPLIT;
            int slices = 1;
            if (use_split(h, a, MODE == ACT_CONTROLLER ? RMAV_ACT_CONTROLLER : MODE == ACT_BUFFER ? RMAV_ACT_BUFFER : RMAV_ACT_RANDOM, &slices, MODE == ACT_RANDOM)) {
                // balanced slices, each a multiple of 64 envs
                const int64_t per = h->chunk > 0 ? h->chunk : slices > 1 ? (((h->n + slices - 1) / slices + 63) / 64) * 64 : h->n;
                RolloutArgs ap = a;
                if (h->chunk > 0) ap.pitch = per;   // chunk-major columns are `chunk` (a multiple of 64) apart whatever N is
                const int st = pick_store_policy(h, ap, true);
                for (int64_t first = 0; first < h->n; first += per) {
                    RolloutArgs b = a;
                    if (slices > 1 || h->chunk > 0) {
                        b.slice_first = (uint32_t)first;
                        b.slice_count = (uint32_t)((h->n - first < per) ? h->n - first : per);
                    }
                    if (h->chunk > 0) {
                        // chunk-major trajectory arrays [n_chunks][T][dim][chunk]: this launch's chunk is a dense region of its own with
                        // column pitch `chunk`; the kernels index columns by the env's index in the handle, so the base pointers are moved
                        // back by `first` columns (a multiple of 64 elements: alignment is kept)
                        const int64_t c = first / per, T = a.n_steps;
                        constexpr int64_t NS = Dims<K>::NS, NA = Dims<K>::NA;
                        b.pitch = per;
                        if (a.act_in) b.act_in = a.act_in + c * T * NA * per - first;
                        if (a.act_out) b.act_out = a.act_out + c * T * NA * per - first;
                        if (a.obs_out) b.obs_out = a.obs_out + c * T * NS * per - first;
                        if (a.rew_out) b.rew_out = a.rew_out + c * T * per - first;
                        if (a.done_out) b.done_out = a.done_out + c * T * per - first;
                    }
                    int rc;
                    switch (st) {
                    case ST_STREAM: rc = launch_rollout_kms<K, SMODE, ST_STREAM>(h, b); break;
                    case ST_DEFAULT: rc = launch_rollout_kms<K, SMODE, ST_DEFAULT>(h, b); break;
                    default: rc = launch_rollout_kms<K, SMODE, ST_WRITE_THROUGH>(h, b); break;
                    }
                    if (rc) return rc;
                }
                return RMAV_OK;
            }
        }
        switch (pick_store_policy(h, a, false)) {
        case ST_WRITE_THROUGH: return launch_rollout_kms<K, MODE, ST_WRITE_THROUGH>(h, a);
        case ST_STREAM: return launch_rollout_kms<K, MODE, ST_STREAM>(h, a);
        case ST_AOS_LDS: return launch_rollout_kms<K, MODE, ST_AOS_LDS>(h, a);
        default: return launch_rollout_kms<K, MODE, ST_DEFAULT>(h, a);
        }
    }
}

template <int K> int launch_rollout_k(rmav_handle h, int mode, const RolloutArgs &a) {
    switch (mode) {
    case RMAV_ACT_BUFFER: return launch_rollout_km<K, ACT_BUFFER>(h, a);
    case RMAV_ACT_RANDOM: return launch_rollout_km<K, ACT_RANDOM>(h, a);
    case RMAV_ACT_CONTROLLER: return launch_rollout_km<K, ACT_CONTROLLER>(h, a);
    case ACT_BUFFER_CTRL: return launch_rollout_kms<K, ACT_BUFFER_CTRL, ST_DEFAULT>(h, a);   // internal (rmav_step_control)
    }
    return rmav_fail(RMAV_ERR_INVALID, "unknown action_mode %d", mode);
}

// k_step's launch rules by batch size (profiles/r05/step_sweep.md; every variant writes the same bits).  Up to ~196 608 envs a
// launch is mostly launch latency: eager bookkeeping loads (a dependent round trip costs more than the 12 B it saves), 256-thread
// workgroups, write-back stores.  Up to ~786 432 envs: non-temporal stores (7.0 -> 6.7-6.9 us at 262 144).  Beyond: bookkeeping
// loaded only in lanes whose env terminates and 128-thread workgroups (1 048 576 envs 26.4 -> 23.5 us, 4 194 304: 92.9 -> 78.9).
// rmav_set_tuning(RMAV_TUNE_STEP_LAZY | _BLOCK | _STEP_STORE) overrides each.
inline bool step_lazy(rmav_handle h) {
    const int t = h->tune[RMAV_TUNE_STEP_LAZY];
    return t >= 0 ? t == 1 : h->n >= 786432;
}
inline int step_block(rmav_handle h) {
    const int v = h->tune[RMAV_TUNE_BLOCK];
    return (v == 64 || v == 128 || v == 256) ? v : (h->n >= 786432 ? 128 : 256);
}
inline int step_store(rmav_handle h) {
    const int v = h->tune[RMAV_TUNE_STEP_STORE];
    return v >= 0 ? v : ((h->n >= 196608 && h->n < 786432) ? (int)ST_STREAM : (int)ST_DEFAULT);
}

// n_steps == 1 with caller actions: the latency-cut single-step kernel
template <int K> int launch_step_k(rmav_handle h, const RolloutArgs &a, bool ctrl) {
    if (h->xchg.armed && h->xchg.fired) h->xchg.stale = true;   // the armed launch's snapshot is no longer the latest
    const typename Env<K>::P p = derive_env<K>(h->params);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
    const int st = step_store(h), bs = step_block(h);
    const bool lazy = step_lazy(h);
    const dim3 grid((unsigned)((h->n + bs - 1) / bs));
    const bool big = h->n >= 786432;   // no argument preloading for the big batches (k_step_big in rmav_kernels.hpp says why)
    // (the leading scalars are what the kernel's first loads need: preloaded into scalar registers, see StepHot in rmav_kernels.hpp)
#define RMAV_STEP_ARGS a.state, a.n, a.act_in, a.pitch, (uint32_t)bs, a.flags, a.ep_ret, a.rec, a, p, pc
#define RMAV_STEP(LAZY, ST)                                                                                           \
    do {                                                                                                              \
        if (big) hipLaunchKernelGGL((k_step_big<K, LAZY, ST>), grid, dim3(bs), 0, h->stream, a, p, pc);               \
        else hipLaunchKernelGGL((k_step<K, false, LAZY, ST>), grid, dim3(bs), 0, h->stream, RMAV_STEP_ARGS);          \
    } while (0)
    if (ctrl) hipLaunchKernelGGL((k_step<K, true>), grid, dim3(bs), 0, h->stream, RMAV_STEP_ARGS);
    else if (lazy && st == ST_STREAM) RMAV_STEP(true, ST_STREAM);
    else if (lazy && st == ST_WRITE_THROUGH) RMAV_STEP(true, ST_WRITE_THROUGH);
    else if (lazy) RMAV_STEP(true, ST_DEFAULT);
    else if (st == ST_WRITE_THROUGH) RMAV_STEP(false, ST_WRITE_THROUGH);
    else if (st == ST_STREAM) RMAV_STEP(false, ST_STREAM);
    else RMAV_STEP(false, ST_DEFAULT);
#undef RMAV_STEP
#undef RMAV_STEP_ARGS
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int launch_rollout(rmav_handle h, int mode, const RolloutArgs &a) {
    if (a.n_steps == 1 && (mode == RMAV_ACT_BUFFER || mode == ACT_BUFFER_CTRL) && h->kind != RMAV_REINMAV) {
        const bool ctrl = mode == ACT_BUFFER_CTRL;
        switch (h->kind) {
        case RMAV_QUAD2D: return launch_step_k<QUAD2D>(h, a, ctrl);
        case RMAV_QUAD2D_SL: return launch_step_k<QUAD2D_SL>(h, a, ctrl);
        case RMAV_QUAD3D: return launch_step_k<QUAD3D>(h, a, ctrl);
        case RMAV_QUAD3D_SL: return launch_step_k<QUAD3D_SL>(h, a, ctrl);
        }
    }
    switch (h->kind) {
    case RMAV_QUAD2D: return launch_rollout_k<QUAD2D>(h, mode, a);
    case RMAV_QUAD2D_SL: return launch_rollout_k<QUAD2D_SL>(h, mode, a);
    case RMAV_QUAD3D: return launch_rollout_k<QUAD3D>(h, mode, a);
    case RMAV_QUAD3D_SL: return launch_rollout_k<QUAD3D_SL>(h, mode, a);
    case RMAV_REINMAV: return launch_rollout_k<REINMAV>(h, mode, a);
    }
    return rmav_fail(RMAV_ERR_INVALID, "bad kind");
}

RolloutArgs base_args(rmav_handle h) {
    RolloutArgs a;
    memset(&a, 0, sizeof(a));
    a.state = h->state;
    a.n = h->n;
    a.pitch = h->n;
    a.rec = h->rec;
    a.ep_ret = h->ep_ret;
    a.last_ret = h->last_ret;
    a.totals = h->totals;
    a.env_time = h->env_time;
    for (int i = 0; i < 3; ++i) a.pe[i] = h->pe[i];
    a.seed = h->seed;
    a.env_base = h->env_base;
    a.t0 = h->t;
    a.n_steps = 1;
    a.flags = h->flags & (F_AUTO_RESET | F_TRACK);
    a.act_lo = (float)h->params.act_lo;
    a.act_hi = (float)h->params.act_hi;
    return a;
}

int launch_reset(rmav_handle h, float *obs_dev, int layout) {
    const uint32_t fl = (h->flags & F_TRACK) | (layout == RMAV_AOS ? F_AOS : 0u);
#define RMAV_RESET_CASE(KIND)                                                                      \
    hipLaunchKernelGGL((k_reset<KIND>), grid_for(h), dim3(block_size(h)), 0, h->stream, h->state,      \
                       h->n, h->rec, h->ep_ret, (uint32_t)h->t, obs_dev, h->seed, h->env_base, fl)
    switch (h->kind) {
    case RMAV_QUAD2D: RMAV_RESET_CASE(QUAD2D); break;
    case RMAV_QUAD2D_SL: RMAV_RESET_CASE(QUAD2D_SL); break;
    case RMAV_QUAD3D: RMAV_RESET_CASE(QUAD3D); break;
    case RMAV_QUAD3D_SL: RMAV_RESET_CASE(QUAD3D_SL); break;
    case RMAV_REINMAV: RMAV_RESET_CASE(REINMAV); break;
    }
#undef RMAV_RESET_CASE
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int launch_control(rmav_handle h, float *act_dev, int layout) {
    const uint32_t fl = (layout == RMAV_AOS ? F_AOS : 0u);
    const ParamsT<double> pc = derive<double>(h->params, h->kind == RMAV_QUAD2D || h->kind == RMAV_QUAD2D_SL);
#define RMAV_CTRL_CASE(KIND)                                                                       \
    hipLaunchKernelGGL((k_control<KIND>), grid_for(h), dim3(block_size(h)), 0, h->stream, h->state,    \
                       h->n, act_dev, fl, pc, h->pe[0], h->pe[1], h->pe[2])
    switch (h->kind) {
    case RMAV_QUAD2D: RMAV_CTRL_CASE(QUAD2D); break;
    case RMAV_QUAD2D_SL: RMAV_CTRL_CASE(QUAD2D_SL); break;
    case RMAV_QUAD3D: RMAV_CTRL_CASE(QUAD3D); break;
    case RMAV_QUAD3D_SL: RMAV_CTRL_CASE(QUAD3D_SL); break;
    case RMAV_REINMAV:
        hipLaunchKernelGGL(k_control_reinmav, grid_for(h), dim3(block_size(h)), 0, h->stream, h->state,
                           h->env_time, h->n, act_dev, fl, derive_reinmav(h->params));
        break;
    }
#undef RMAV_CTRL_CASE
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int check_mem_layout(int mem, int layout) {
    if (mem != RMAV_HOST && mem != RMAV_DEVICE) return rmav_fail(RMAV_ERR_INVALID, "mem must be RMAV_HOST or RMAV_DEVICE");
    if (layout != RMAV_SOA && layout != RMAV_AOS) return rmav_fail(RMAV_ERR_INVALID, "layout must be RMAV_SOA or RMAV_AOS");
    return RMAV_OK;
}

// Generic "copy a per-env array out of / into the handle".
template <typename T> int copy_out(rmav_handle h, const T *dev, T *out, size_t count, int mem) {
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "output pointer is NULL");
    if (mem == RMAV_DEVICE) {
        HIP_TRY(hipMemcpyAsync(out, dev, count * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    } else if (count * sizeof(T) <= kPinnedMax && ensure_pinned(h, count * sizeof(T)) == RMAV_OK) {
        // pinned target: a true asynchronous DMA, then one synchronise (a pageable target makes the runtime stage and block)
        HIP_TRY(hipMemcpyAsync(h->pinned, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out, h->pinned, count * sizeof(T));
    } else {
        HIP_TRY(hipMemcpyAsync(out, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return RMAV_OK;
}
template <typename T> int copy_in(rmav_handle h, T *dev, const T *in, size_t count, int mem) {
    if (!in) return rmav_fail(RMAV_ERR_INVALID, "input pointer is NULL");
    if (mem == RMAV_DEVICE) {
        HIP_TRY(hipMemcpyAsync(dev, in, count * sizeof(T), hipMemcpyDeviceToDevice, h->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(dev, in, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return RMAV_OK;
}

// One 32-bit field of the per-env records (EnvRec: 0 sbd, 1 reset_cnt, 2 ep_start, 3 last_len) as a dense array: the state accessors
// of the ABI gather / scatter it with one small kernel (host pointers: through device scratch)
int rec_field_get(rmav_handle h, int field, uint32_t *out, int mem) {
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "output pointer is NULL");
    const size_t n = (size_t)h->n;
    uint32_t *dst = out;
    if (mem != RMAV_DEVICE) {
        if (int rc = ensure_scratch(h, n * sizeof(uint32_t))) return rc;
        dst = (uint32_t *)h->scratch;
    }
    hipLaunchKernelGGL(k_rec_get, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dst, (const EnvRec *)h->rec, field, (int64_t)n);
    HIP_TRY(hipGetLastError());
    return mem == RMAV_DEVICE ? RMAV_OK : copy_out(h, (const uint32_t *)dst, out, n, RMAV_HOST);
}
int rec_field_set(rmav_handle h, int field, const uint32_t *in, int mem) {
    if (!in) return rmav_fail(RMAV_ERR_INVALID, "input pointer is NULL");
    const size_t n = (size_t)h->n;
    const uint32_t *src = in;
    if (mem != RMAV_DEVICE) {
        if (int rc = ensure_scratch(h, n * sizeof(uint32_t))) return rc;
        if (int rc = copy_in(h, (uint32_t *)h->scratch, in, n, RMAV_HOST)) return rc;
        src = (const uint32_t *)h->scratch;
    }
    hipLaunchKernelGGL(k_rec_set, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->rec, src, field, (int64_t)n);
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

void free_all(rmav_handle h) {
    void *ptrs[] = {h->arena, h->pe[0], h->pe[1], h->pe[2], h->scratch};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->done_flag) (void)hipHostFree(h->done_flag);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->magic = 0;
    delete h;
}

}  // namespace

// =================================================================================================
extern "C" {

int rmav_version(void) { return RMAV_VERSION; }

const char *rmav_last_error(void) { return g_err; }

int rmav_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n < 0 ? 0 : n;
}

int rmav_state_dim(int kind) { return (kind < 0 || kind >= kNumKinds) ? -1 : kStateDim[kind]; }
int rmav_action_dim(int kind) { return (kind < 0 || kind >= kNumKinds) ? -1 : kActionDim[kind]; }

int rmav_algorithmic_bytes(int kind) {
    if (kind < 0 || kind >= kNumKinds) return -1;
    // read state + read action + write state + write reward (f32) + write done (u8)
    return 4 * (2 * kStateDim[kind] + kActionDim[kind] + 1) + 1;
}

int rmav_default_params(int kind, int reading_2d, rmav_params *p) {
    if (kind < 0 || kind >= kNumKinds) return rmav_fail(RMAV_ERR_INVALID, "bad kind %d", kind);
    if (!p) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    if (reading_2d != 0 && reading_2d != 'A' && reading_2d != 'B')
        return rmav_fail(RMAV_ERR_INVALID, "reading_2d must be 0, 'A' or 'B'");
    memset(p, 0, sizeof(*p));
    p->mass = 1.0;
    p->load_mass = 0.1;
    p->dt = 0.01;
    p->g = 9.8;
    // self.g: (0, 0, -9.8) quadrotor3d.py:47, quadrotor3d_slungload.py:48; (0, -9.8) quadrotor2d.py:46, quadrotor2d_slungload.py:47
    p->g_vec[(kind == RMAV_QUAD2D || kind == RMAV_QUAD2D_SL) ? 1 : 2] = -9.8;
    p->thrust_scale = 1.0;
    p->kp = -5.0;
    p->kv = -4.0;
    p->act_lo = -10.0;
    p->act_hi = 10.0;
    switch (kind) {
    case RMAV_QUAD2D:
        p->pos_limit = 3.0;
        p->vel_limit = (reading_2d == 'A') ? 10.0 : 2.0;
        p->thrust_scale = 10.0;
        p->clamp_thrust = 1;
        p->tau = 0.1;
        break;
    case RMAV_QUAD2D_SL:
        p->tether_length = 0.5;
        p->pos_limit = 2.0;
        p->vel_limit = 10.0;
        p->tau = 0.1;
        break;
    case RMAV_QUAD3D:
        p->pos_limit = 3.0;
        p->vel_limit = 10.0;
        p->ref_pos[2] = 2.0;
        p->tau = 0.3;
        p->act_lo = 0.0;  // quadrotor3d.py:70 Box(low=0, high=10)
        break;
    case RMAV_QUAD3D_SL:
        p->tether_length = 1.5;
        p->pos_limit = 3.0;
        p->vel_limit = 10.0;
        p->ref_pos[2] = 1.0;
        p->tau = 0.3;
        break;
    case RMAV_REINMAV:       // reinmav_env.py:55-73; only mass, g and dt are read from this struct
        p->mass = 0.1800;
        p->load_mass = 0.0;
        p->g = 9.8100;
        p->g_vec[2] = 0.0;   // not read: ReinmavEnv's gravity is the scalar above
        p->dt = 1.0 / 100;
        p->tau = 1.0;        // unused (keeps check_params happy)
        p->act_lo = 0.0;     // RMAV_ACT_RANDOM range for (F, Mx, My, Mz): [0, max_force)
        p->act_hi = 3.5316;
        break;
    }
    return RMAV_OK;
}

int rmav_create(rmav_handle *out, int kind, int64_t n_envs, int device, uint64_t seed,
                uint64_t env_id_base, uint32_t flags, const rmav_params *params, void *hip_stream) {
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (kind < 0 || kind >= kNumKinds) return rmav_fail(RMAV_ERR_INVALID, "bad kind %d", kind);
    if (n_envs <= 0 || n_envs > ((int64_t)1 << 25))  // 32-bit buffer offsets, see rmav_kernels.hpp
        return rmav_fail(RMAV_ERR_INVALID, "n_envs out of range: %lld", (long long)n_envs);
    if (flags & ~(RMAV_F_AUTO_RESET | RMAV_F_TRACK_EPISODES))
        return rmav_fail(RMAV_ERR_INVALID, "unknown flag bits 0x%x", flags);
    const int ndev = rmav_device_count();
    if (ndev <= 0) return rmav_fail(RMAV_ERR_NO_DEVICE, "no HIP device visible; librmav has no CPU path");
    if (device < 0 || device >= ndev) return rmav_fail(RMAV_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    rmav_params pr;
    if (params) pr = *params;
    else rmav_default_params(kind, 0, &pr);
    if (int rc = check_params(pr)) return rc;

    DeviceGuard guard(device);
    if (!guard.ok) return rmav_fail(RMAV_ERR_HIP, "hipSetDevice(%d) failed", device);

    rmav_handle h = new (std::nothrow) rmav_env_s();
    if (!h) return rmav_fail(RMAV_ERR_ALLOC, "host allocation failed");
    memset(h, 0, sizeof(*h));
    h->magic = kMagic;
    h->kind = kind;
    h->n = n_envs;
    h->device = device;
    h->seed = seed;
    h->env_base = env_id_base;
    h->flags = flags;
    h->params = pr;
    for (int i = 0; i < RMAV_TUNE_COUNT; ++i) h->tune[i] = -1;
    if (hip_stream) {
        // (void*)1 names the legacy default stream, whose real handle is 0: use that (some runtime entry points -
        // hipEventRecord - do not accept the hipStreamLegacy constant)
        h->stream = (hip_stream == (void *)1) ? nullptr : (hipStream_t)hip_stream;
        h->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            free_all(h);
            return rmav_fail(RMAV_ERR_HIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    const size_t n = (size_t)n_envs;
    const int nS = kStateDim[kind];
    // One arena for every per-env array.  A single-step launch at 65 536 envs is latency-bound and touches ten
    // small arrays (256 KB each); as separate hipMalloc blocks each of them sits in its own small-page mapping and
    // every launch pays their address translations, while one 2 MiB-aligned block is covered by a few large
    // fragments.  Sub-arrays start on 4 KiB boundaries.
    {
        auto up = [](size_t b) { return (b + 4095) & ~(size_t)4095; };
        const bool tr = (flags & RMAV_F_TRACK_EPISODES) != 0;
        size_t off = 0;
        const size_t o_state = off; off += up(n * nS * sizeof(float));
        const size_t o_rec = off; off += up(n * sizeof(EnvRec));
        const size_t o_tot = off; off += up(n_total_slots(n_envs) * sizeof(Totals));
        const size_t o_time = off; off += (kind == RMAV_REINMAV) ? up(n * sizeof(double)) : 0;
        const size_t o_er = off; off += tr ? up(n * sizeof(float)) : 0;
        const size_t o_lr = off; off += tr ? up(n * sizeof(float)) : 0;
        if (hipMalloc(&h->arena, off) != hipSuccess) {
            (void)hipGetLastError();
            h->arena = nullptr;
            free_all(h);
            return rmav_fail(RMAV_ERR_ALLOC, "device allocation of %zu bytes failed for %lld envs", off, (long long)n_envs);
        }
        char *b = (char *)h->arena;
        h->state = (float *)(b + o_state);
        h->rec = (EnvRec *)(b + o_rec);
        h->totals = (Totals *)(b + o_tot);
        if (kind == RMAV_REINMAV) h->env_time = (double *)(b + o_time);
        if (tr) {
            h->ep_ret = (float *)(b + o_er);
            h->last_ret = (float *)(b + o_lr);
        }
    }
    // every record: steps_beyond_done = None (-1), no reset drawn yet, the episode clock starts at 0, no finished episode
    hipLaunchKernelGGL(k_rec_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->rec, EnvRec{-1, 0u, 0u, 0}, -1, (int64_t)n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemsetAsync(h->totals, 0, n_total_slots(n_envs) * sizeof(Totals), h->stream);
    if (e == hipSuccess && (flags & RMAV_F_TRACK_EPISODES)) {
        e = hipMemsetAsync(h->ep_ret, 0, n * sizeof(float), h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(h->last_ret, 0, n * sizeof(float), h->stream);
    }
    if (e != hipSuccess) {
        free_all(h);
        return rmav_fail(RMAV_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    if (kind == RMAV_REINMAV) {
        // ReinmavEnv.__init__ (reinmav_env.py:79-81): state = (0,0,0, 0,0,0, 1,0,0,0, 0,0,0), t = 0; no RNG
        const float one = 1.0f;
        uint32_t one_bits;
        memcpy(&one_bits, &one, 4);
        e = hipMemsetAsync(h->state, 0, n * nS * sizeof(float), h->stream);
        if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(h->state + 6 * n), (int)one_bits, n, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(h->env_time, 0, n * sizeof(double), h->stream);
        if (e != hipSuccess) {
            free_all(h);
            return rmav_fail(RMAV_ERR_HIP, "initial state failed: %s", hipGetErrorString(e));
        }
    } else if (int rc = launch_reset(h, nullptr, RMAV_SOA)) {
        // the reference constructors call seed() then reset()  (quadrotor3d.py:73-74)
        free_all(h);
        return rc;
    }
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        free_all(h);
        return rmav_fail(RMAV_ERR_HIP, "initial reset failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return RMAV_OK;
}

int rmav_destroy(rmav_handle h) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (h->xchg.comm && h->xchg.comm->armed_by == h) h->xchg.comm->armed_by = nullptr;
    DeviceGuard guard(h->device);
    (void)hipStreamSynchronize(h->stream);
    free_all(h);
    return RMAV_OK;
}

// t is also the episode clock (rmav_kernels.hpp: ep_clock0): when a caller moves it, every env's episode start moves along,
// so running episode lengths carry over the jump
static int move_step_counter(rmav_handle h, uint64_t t) {
    const uint32_t delta = (uint32_t)t - (uint32_t)h->t;
    if (delta != 0u && (h->flags & RMAV_F_TRACK_EPISODES)) {
        hipLaunchKernelGGL(k_shift_ep_start, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->rec, delta, (int64_t)h->n);
        HIP_TRY(hipGetLastError());
    }
    h->t = t;
    return RMAV_OK;
}

int rmav_seed(rmav_handle h, uint64_t seed) {
    CHECK_HANDLE(h);
    h->seed = seed;
    if (int rc = move_step_counter(h, 0)) return rc;
    hipLaunchKernelGGL(k_rec_fill, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->stream, h->rec, EnvRec{0, 0u, 0u, 0}, 1, (int64_t)h->n);   // reset_cnt = 0
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int rmav_get_params(rmav_handle h, rmav_params *out) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    *out = h->params;
    return RMAV_OK;
}

int rmav_set_params(rmav_handle h, const rmav_params *in) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (!in) return rmav_fail(RMAV_ERR_INVALID, "in is NULL");
    if (int rc = check_params(*in)) return rc;
    h->params = *in;
    return RMAV_OK;
}

int rmav_set_stream(rmav_handle h, void *hip_stream) {
    CHECK_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) {
        (void)hipStreamDestroy(h->stream);
        h->own_stream = false;
    }
    if (hip_stream) {
        // (void*)1 names the legacy default stream, whose real handle is 0: use that (some runtime entry points -
        // hipEventRecord - do not accept the hipStreamLegacy constant)
        h->stream = (hip_stream == (void *)1) ? nullptr : (hipStream_t)hip_stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    return RMAV_OK;
}

int rmav_set_env_param(rmav_handle h, int which, const float *values, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    if (which < 0 || which > 2) return rmav_fail(RMAV_ERR_INVALID, "unknown env param %d", which);
    if (h->kind == RMAV_REINMAV) return rmav_fail(RMAV_ERR_INVALID, "per-env constants are for the quadrotor kinds");
    if (!values) {  // back to the shared value
        if (h->pe[which]) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            HIP_TRY(hipFree(h->pe[which]));
            h->pe[which] = nullptr;
        }
        return RMAV_OK;
    }
    if (!h->pe[which] && hipMalloc((void **)&h->pe[which], (size_t)h->n * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        h->pe[which] = nullptr;
        return rmav_fail(RMAV_ERR_ALLOC, "device allocation failed");
    }
    return copy_in(h, h->pe[which], values, (size_t)h->n, mem);
}

int rmav_set_tuning(rmav_handle h, int key, int value) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (key < 0 || key >= RMAV_TUNE_COUNT) return rmav_fail(RMAV_ERR_INVALID, "unknown tuning key %d", key);
    h->tune[key] = value;
    return RMAV_OK;
}
int rmav_get_tuning(rmav_handle h, int key, int *value_out) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (key < 0 || key >= RMAV_TUNE_COUNT || !value_out) return rmav_fail(RMAV_ERR_INVALID, "unknown tuning key %d or NULL out", key);
    *value_out = h->tune[key];
    return RMAV_OK;
}

int64_t rmav_num_envs(rmav_handle h) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    return h->n;
}

int rmav_sync(rmav_handle h) {
    CHECK_HANDLE(h);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RMAV_OK;
}

int rmav_reset(rmav_handle h, float *obs_out, int mem, int layout) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, layout)) return rc;
    const size_t nobs = (size_t)h->n * kStateDim[h->kind];
    if (mem == RMAV_DEVICE || !obs_out) return launch_reset(h, obs_out, layout);
    if (nobs * sizeof(float) <= kPinnedMax && ensure_pinned(h, nobs * sizeof(float)) == RMAV_OK) {   // zero-copy
        if (int rc = launch_reset(h, (float *)h->pinned_dev, layout)) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(obs_out, h->pinned, nobs * sizeof(float));
        return RMAV_OK;
    }
    if (int rc = ensure_scratch(h, nobs * sizeof(float))) return rc;
    if (int rc = launch_reset(h, (float *)h->scratch, layout)) return rc;
    return copy_out(h, (const float *)h->scratch, obs_out, nobs, RMAV_HOST);
}

// rmav_rollout / rmav_step / rmav_step_control / rmav_control_step.  ctrl_out (nullable): control() of the state the
// call leaves behind, nA*N floats in `layout` (action_mode must be RMAV_ACT_BUFFER, n_steps 1).
static int rollout_impl(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                        float *obs_out, float *rew_out, uint8_t *done_out, float *ctrl_out, int mem, int layout,
                        int fused, int64_t pitch = 0) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, layout)) return rc;
    if (n_steps <= 0) return rmav_fail(RMAV_ERR_INVALID, "n_steps must be > 0");
    if (pitch != 0 && (pitch < h->n || pitch > (int64_t)0x3fffffff || mem != RMAV_DEVICE || layout != RMAV_SOA))
        return rmav_fail(RMAV_ERR_INVALID, "a column pitch must be in [num_envs, 2^30) and needs device pointers and the feature-major layout");
    if (action_mode < RMAV_ACT_BUFFER || action_mode > RMAV_ACT_CONTROLLER)
        return rmav_fail(RMAV_ERR_INVALID, "unknown action_mode %d", action_mode);
    if (action_mode == RMAV_ACT_BUFFER && !actions_in)
        return rmav_fail(RMAV_ERR_INVALID, "RMAV_ACT_BUFFER needs actions_in");
    if (ctrl_out && h->kind == RMAV_REINMAV)
        return rmav_fail(RMAV_ERR_INVALID, "ReinmavEnv's controller runs inside its step (the controller action mode); there is no separate control()");
    const size_t n = pitch ? (size_t)pitch : (size_t)h->n, T = (size_t)n_steps;   // (the trajectory arrays' column pitch)
    const size_t nS = kStateDim[h->kind], nA = kActionDim[h->kind];
    const size_t b_act = T * nA * n * sizeof(float), b_obs = T * nS * n * sizeof(float);
    const size_t b_rew = T * n * sizeof(float), b_done = T * n, b_ctrl = nA * n * sizeof(float);
    const bool want_aout = actions_out && action_mode != RMAV_ACT_BUFFER;

    const float *d_act_in = actions_in;
    float *d_act_out = actions_out, *d_obs = obs_out, *d_rew = rew_out, *d_ctrl = ctrl_out;
    uint8_t *d_done = done_out;
    bool pinned = false;
    char *hbase = nullptr;   // host view of the staging block (pinned path only)
    size_t o_ain = 0, o_aout = 0, o_obs = 0, o_rew = 0, o_done = 0, o_ctrl = 0;
    if (mem == RMAV_HOST) {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        size_t off = 0;
        if (action_mode == RMAV_ACT_BUFFER) { o_ain = off; off += up(b_act); }
        if (want_aout) { o_aout = off; off += up(b_act); }
        if (obs_out) { o_obs = off; off += up(b_obs); }
        if (rew_out) { o_rew = off; off += up(b_rew); }
        if (done_out) { o_done = off; off += up(b_done); }
        if (ctrl_out) { o_ctrl = off; off += up(b_ctrl); }
        if (!off) off = 256;
        char *base;
        if (off <= kPinnedMax && ensure_pinned(h, off) == RMAV_OK) {
            // zero-copy: the kernel reads / writes the pinned host block directly
            pinned = true;
            hbase = (char *)h->pinned;
            base = (char *)h->pinned_dev;
            if (action_mode == RMAV_ACT_BUFFER) memcpy(hbase + o_ain, actions_in, b_act);
        } else {   // bulk: stage through device scratch
            if (int rc = ensure_scratch(h, off)) return rc;
            base = (char *)h->scratch;
            if (action_mode == RMAV_ACT_BUFFER)
                HIP_TRY(hipMemcpyAsync(base + o_ain, actions_in, b_act, hipMemcpyHostToDevice, h->stream));
        }
        d_act_in = action_mode == RMAV_ACT_BUFFER ? (const float *)(base + o_ain) : nullptr;
        d_act_out = want_aout ? (float *)(base + o_aout) : nullptr;
        d_obs = obs_out ? (float *)(base + o_obs) : nullptr;
        d_rew = rew_out ? (float *)(base + o_rew) : nullptr;
        d_done = done_out ? (uint8_t *)(base + o_done) : nullptr;
        d_ctrl = ctrl_out ? (float *)(base + o_ctrl) : nullptr;
    }

    RolloutArgs a = base_args(h);
    if (layout == RMAV_AOS) a.flags |= F_AOS;
    if (pitch) a.pitch = pitch;
    a.ctrl_out = d_ctrl;
    const int kmode = d_ctrl ? (int)ACT_BUFFER_CTRL : action_mode;
    // One wavefront, one k_step launch, outputs in the pinned block: the kernel publishes its completion in a pinned word and
    // the host spins on that (bounded) instead of hipStreamSynchronize.  What the gym-shaped single env runs.
    bool flag_wait = false;
    if (pinned && n_steps == 1 && h->n <= 64 && action_mode == RMAV_ACT_BUFFER && h->kind != RMAV_REINMAV) {
        if (!h->done_flag && hipHostMalloc((void **)&h->done_flag, 64, hipHostMallocMapped) == hipSuccess) {
            *h->done_flag = 0;
            if (hipHostGetDevicePointer((void **)&h->done_flag_dev, h->done_flag, 0) != hipSuccess) {
                (void)hipHostFree(h->done_flag);
                h->done_flag = nullptr;
            }
        }
        (void)hipGetLastError();
        if (h->done_flag) {
            a.done_flag = h->done_flag_dev;
            a.done_seq = ++h->done_seq;
            flag_wait = true;
        }
    }
    h->xchg.allow = fused != 0;   // one fused launch may carry an armed exchange's snapshot; the fused = 0 loop may not
    if (fused) {
        a.n_steps = n_steps;
        a.act_in = d_act_in;
        a.act_out = d_act_out;
        a.obs_out = d_obs;
        a.rew_out = d_rew;
        a.done_out = d_done;
        if (int rc = launch_rollout(h, kmode, a)) return rc;
    } else {
        for (size_t k = 0; k < T; ++k) {
            a.n_steps = 1;
            a.t0 = h->t + k;
            a.act_in = d_act_in ? d_act_in + k * nA * n : nullptr;
            a.act_out = d_act_out ? d_act_out + k * nA * n : nullptr;
            a.obs_out = d_obs ? d_obs + k * nS * n : nullptr;
            a.rew_out = d_rew ? d_rew + k * n : nullptr;
            a.done_out = d_done ? d_done + k * n : nullptr;
            if (int rc = launch_rollout(h, (k + 1 == T) ? kmode : action_mode, a)) return rc;
        }
    }
    h->t += T;

    if (mem == RMAV_HOST) {
        if (pinned) {
            bool seen = false;
            if (flag_wait) {   // ~10 us is the whole launch; 2 ms covers a cold first launch, then fall back to the stream
                timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                const volatile uint32_t *f = h->done_flag;
                for (uint32_t spin = 0;; ++spin) {
                    if (*f == h->done_seq) { seen = true; break; }
                    if ((spin & 255u) == 255u) {
                        clock_gettime(CLOCK_MONOTONIC, &t1);
                        if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 2000000ll) break;
                    }
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
            }
            if (!seen) HIP_TRY(hipStreamSynchronize(h->stream));
            if (want_aout) memcpy(actions_out, hbase + o_aout, b_act);
            if (obs_out) memcpy(obs_out, hbase + o_obs, b_obs);
            if (rew_out) memcpy(rew_out, hbase + o_rew, b_rew);
            if (done_out) memcpy(done_out, hbase + o_done, b_done);
            if (ctrl_out) memcpy(ctrl_out, hbase + o_ctrl, b_ctrl);
        } else {
            if (want_aout) HIP_TRY(hipMemcpyAsync(actions_out, d_act_out, b_act, hipMemcpyDeviceToHost, h->stream));
            if (obs_out) HIP_TRY(hipMemcpyAsync(obs_out, d_obs, b_obs, hipMemcpyDeviceToHost, h->stream));
            if (rew_out) HIP_TRY(hipMemcpyAsync(rew_out, d_rew, b_rew, hipMemcpyDeviceToHost, h->stream));
            if (done_out) HIP_TRY(hipMemcpyAsync(done_out, d_done, b_done, hipMemcpyDeviceToHost, h->stream));
            if (ctrl_out) HIP_TRY(hipMemcpyAsync(ctrl_out, d_ctrl, b_ctrl, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
        }
        if (actions_out && action_mode == RMAV_ACT_BUFFER && actions_out != actions_in)
            memcpy(actions_out, actions_in, b_act);
    } else if (actions_out && action_mode == RMAV_ACT_BUFFER && actions_out != actions_in) {
        HIP_TRY(hipMemcpyAsync(actions_out, actions_in, b_act, hipMemcpyDeviceToDevice, h->stream));
    }
    return RMAV_OK;
}

int rmav_rollout(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in,
                 float *actions_out, float *obs_out, float *rew_out, uint8_t *done_out, int mem,
                 int layout, int fused) {
    return rollout_impl(h, n_steps, action_mode, actions_in, actions_out, obs_out, rew_out, done_out, nullptr, mem,
                        layout, fused);
}

int64_t rmav_trajectory_pitch(rmav_handle h) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid handle");
    return (h->n + 63) & ~(int64_t)63;
}

int rmav_rollout_pitched(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t pitch, int fused) {
    if (pitch <= 0) return rmav_fail(RMAV_ERR_INVALID, "pitch must be > 0 (rmav_trajectory_pitch)");
    return rollout_impl(h, n_steps, action_mode, actions_in, actions_out, obs_out, rew_out, done_out, nullptr, RMAV_DEVICE,
                        RMAV_SOA, fused, pitch);
}

// Chunk-major trajectories: see include/rmav.h.  The recommended chunk is the batch the two-wavefront kernel runs best at for the
// kind whose launches are bound by the trajectory stores (65 536 envs = one (integrator, memory) pair per SIMD): profiles/r05/chunk_probe.md.
int64_t rmav_chunk_envs(rmav_handle h) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid handle");
    // measured: quadrotor3d with random / caller actions gains 5 - 9 % at 131 072 - 262 144 envs; every other kind loses 3 - 10 %
    // no chunking: ONE chunk whose width is N rounded up to the 64-env granule every chunk width must have (rmav_rollout_chunked
    // turns a chunk wider than N into the column pitch of the plain layout)
    return (h->kind == RMAV_QUAD3D && h->n > 65536) ? 65536 : ((h->n + 63) & ~(int64_t)63);
}

int rmav_rollout_chunked(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t chunk_envs) {
    CHECK_HANDLE(h);
    if (chunk_envs <= 0 || (chunk_envs & 63)) return rmav_fail(RMAV_ERR_INVALID, "chunk_envs must be a positive multiple of 64 (rmav_chunk_envs)");
    if (chunk_envs >= h->n)   // one chunk: the plain feature-major layout
        return rollout_impl(h, n_steps, action_mode, actions_in, actions_out, obs_out, rew_out, done_out, nullptr, RMAV_DEVICE, RMAV_SOA, 1,
                            chunk_envs > h->n ? chunk_envs : 0);
    if (h->kind > RMAV_QUAD3D_SL) return rmav_fail(RMAV_ERR_INVALID, "chunk-major rollouts are for the quadrotor kinds");
    const int64_t cap = kEnvsPerCuSlot * split_pairs_of(action_mode)[h->kind];
    if (n_steps < 2 || chunk_envs > cap)
        return rmav_fail(RMAV_ERR_INVALID, "chunk-major rollouts need n_steps >= 2 and chunk_envs <= %lld for this kind and action source", (long long)cap);
    if (actions_out && action_mode == RMAV_ACT_BUFFER)   // (rollout_impl's echo copy is sized for the plain layout)
        return rmav_fail(RMAV_ERR_INVALID, "chunk-major rollouts do not echo caller actions: pass actions_out = NULL with RMAV_ACT_BUFFER");
    h->chunk = chunk_envs;
    const int rc = rollout_impl(h, n_steps, action_mode, actions_in, actions_out, obs_out, rew_out, done_out, nullptr, RMAV_DEVICE, RMAV_SOA, 1);
    h->chunk = 0;
    return rc;
}

int rmav_step_control(rmav_handle h, const float *actions, float *obs_out, float *rew_out, uint8_t *done_out,
                      float *next_actions_out, int mem, int layout) {
    if (!actions) return rmav_fail(RMAV_ERR_INVALID, "actions is NULL");
    if (!next_actions_out) return rmav_fail(RMAV_ERR_INVALID, "next_actions_out is NULL");
    return rollout_impl(h, 1, RMAV_ACT_BUFFER, actions, nullptr, obs_out, rew_out, done_out, next_actions_out, mem,
                        layout, 1);
}

int rmav_control_step(rmav_handle h, float *actions_out, float *obs_out, float *rew_out, uint8_t *done_out, int mem,
                      int layout) {
    return rollout_impl(h, 1, RMAV_ACT_CONTROLLER, nullptr, actions_out, obs_out, rew_out, done_out, nullptr, mem,
                        layout, 1);
}

int64_t rmav_policy_weight_count(int kind) {
    switch (kind) {
    case RMAV_QUAD2D: return PolicyLayout<5>::TOTAL;
    case RMAV_QUAD2D_SL: return PolicyLayout<9>::TOTAL;
    case RMAV_QUAD3D: return PolicyLayout<10>::TOTAL;
    case RMAV_QUAD3D_SL: return PolicyLayout<16>::TOTAL;
    case RMAV_REINMAV: return PolicyLayout<13>::TOTAL;
    }
    return rmav_fail(RMAV_ERR_INVALID, "bad kind %d", kind);
}

int64_t rmav_policy_weight_count_bf16(void) { return MfmaLayout::TOTAL; }
int64_t rmav_policy_weight_count_f32_mfma(void) { return Mfma32Layout::TOTAL; }
int64_t rmav_policy_weight_count_shared(void) { return MfmaLayout::NET + 4; }

static int pack_policy_impl(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                            const int32_t *idx_hi, int64_t n_out, float *weights_out, bool f16) {
    CHECK_HANDLE(h);
    if (n_params <= 0 || n_params > kPackMaxParams) return rmav_fail(RMAV_ERR_INVALID, "n_params must be in [1, %d]", kPackMaxParams);
    if (!params || !sizes || !idx_lo || !idx_hi || !weights_out || n_out <= 0)
        return rmav_fail(RMAV_ERR_INVALID, "params, sizes, idx_lo, idx_hi, weights_out are required and n_out > 0");
    PackSrc src;
    memset(&src, 0, sizeof(src));
    int64_t end = 0;
    for (int k = 0; k < n_params; ++k) {
        if (!params[k] || sizes[k] < 0) return rmav_fail(RMAV_ERR_INVALID, "parameter %d is NULL or has a negative size", k);
        end += sizes[k];
        if (end > 0x7fffffff) return rmav_fail(RMAV_ERR_INVALID, "too many parameter elements");
        src.p[k] = params[k];
        src.end[k] = (int32_t)end;
    }
    src.n = n_params;
    const dim3 grid((unsigned)((n_out + 255) / 256));
    if (f16) {
        if (n_out != MfmaLayout::TOTAL && n_out != MfmaLayout::NET + 4)
            return rmav_fail(RMAV_ERR_INVALID, "n_out must be rmav_policy_weight_count_bf16() = %d or rmav_policy_weight_count_shared() = %d",
                             (int)MfmaLayout::TOTAL, (int)MfmaLayout::NET + 4);
        hipLaunchKernelGGL(k_pack_policy<true>, grid, dim3(256), 0, h->stream, src, idx_lo, idx_hi, n_out, weights_out, (int32_t)MfmaLayout::NET,
                           (int32_t)MfmaLayout::A2, (int32_t)MfmaLayout::A3, (int32_t)MfmaLayout::B1, -2.0f * kTanhScale, -2.0f);
    } else {
        hipLaunchKernelGGL(k_pack_policy<false>, grid, dim3(256), 0, h->stream, src, idx_lo, idx_hi, n_out, weights_out, 1, 0, 0, 0, 1.0f, 1.0f);
    }
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int rmav_pack_policy(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                     const int32_t *idx_hi, int64_t n_out, float *weights_out) {
    return pack_policy_impl(h, n_params, params, sizes, idx_lo, idx_hi, n_out, weights_out, false);
}
int rmav_pack_policy_f16(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                         const int32_t *idx_hi, int64_t n_out, float *weights_out) {
    return pack_policy_impl(h, n_params, params, sizes, idx_lo, idx_hi, n_out, weights_out, true);
}

int rmav_rollout_policy(rmav_handle h, int32_t n_steps, const float *weights, float *actions_out,
                        float *obs_out, float *rew_out, uint8_t *done_out, float *logp_out,
                        float *value_out, int precision) {
    CHECK_HANDLE(h);
    if (precision < RMAV_POLICY_FP32 || precision > RMAV_POLICY_F16_SHARED)
        return rmav_fail(RMAV_ERR_INVALID, "precision must be one of RMAV_POLICY_FP32 ... RMAV_POLICY_F16_SHARED (rmav_policy_precision)");
    if (n_steps <= 0) return rmav_fail(RMAV_ERR_INVALID, "n_steps must be > 0");
    if (!weights || !logp_out || !value_out)
        return rmav_fail(RMAV_ERR_INVALID, "weights, logp_out and value_out are required (device pointers)");
    if ((reinterpret_cast<uintptr_t>(weights) & 15u) != 0)
        return rmav_fail(RMAV_ERR_INVALID, "weights must be 16-byte aligned");
    RolloutArgs a = base_args(h);
    a.n_steps = n_steps;
    a.act_out = actions_out;
    a.obs_out = obs_out;
    a.rew_out = rew_out;
    a.done_out = done_out;
    a.policy_w = weights;
    a.logp_out = logp_out;
    a.val_out = value_out;
    const int kmode = precision == RMAV_POLICY_FP32        ? (int)RMAV_ACT_POLICY
                      : precision == RMAV_POLICY_BF16_MFMA ? (int)RMAV_ACT_POLICY_BF16
                      : precision == RMAV_POLICY_F16_MFMA  ? (int)ACT_POLICY_F16
                      : precision == RMAV_POLICY_F16_SHARED ? (int)ACT_POLICY_F16_SHARED
                                                           : (int)ACT_POLICY_F32M;
    h->xchg.allow = true;
    if (int rc = rmav_launch_policy_rollout(h, kmode, a)) return rc;
    h->t += (uint64_t)n_steps;
    return RMAV_OK;
}

int rmav_step(rmav_handle h, const float *actions, float *obs_out, float *rew_out,
              uint8_t *done_out, int mem, int layout) {
    if (!actions) return rmav_fail(RMAV_ERR_INVALID, "actions is NULL");
    return rmav_rollout(h, 1, RMAV_ACT_BUFFER, actions, nullptr, obs_out, rew_out, done_out, mem,
                        layout, 1);
}

int rmav_control(rmav_handle h, float *actions_out, int mem, int layout) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, layout)) return rc;
    if (!actions_out) return rmav_fail(RMAV_ERR_INVALID, "actions_out is NULL");
    const size_t nact = (size_t)h->n * kActionDim[h->kind];
    if (mem == RMAV_DEVICE) return launch_control(h, actions_out, layout);
    if (nact * sizeof(float) <= kPinnedMax && ensure_pinned(h, nact * sizeof(float)) == RMAV_OK) {   // zero-copy
        if (int rc = launch_control(h, (float *)h->pinned_dev, layout)) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(actions_out, h->pinned, nact * sizeof(float));
        return RMAV_OK;
    }
    if (int rc = ensure_scratch(h, nact * sizeof(float))) return rc;
    if (int rc = launch_control(h, (float *)h->scratch, layout)) return rc;
    return copy_out(h, (const float *)h->scratch, actions_out, nact, RMAV_HOST);
}

int rmav_get_state(rmav_handle h, float *out, int mem, int layout) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, layout)) return rc;
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    const int nS = kStateDim[h->kind];
    const size_t cnt = (size_t)h->n * nS;
    if (layout == RMAV_SOA) return copy_out(h, (const float *)h->state, out, cnt, mem);
    if (mem == RMAV_DEVICE) {
        hipLaunchKernelGGL(k_soa_to_aos, grid_for(h), dim3(block_size(h)), 0, h->stream, h->state, out, h->n, nS);
        HIP_TRY(hipGetLastError());
        return RMAV_OK;
    }
    if (int rc = ensure_scratch(h, cnt * sizeof(float))) return rc;
    hipLaunchKernelGGL(k_soa_to_aos, grid_for(h), dim3(block_size(h)), 0, h->stream, h->state,
                       (float *)h->scratch, h->n, nS);
    HIP_TRY(hipGetLastError());
    return copy_out(h, (const float *)h->scratch, out, cnt, RMAV_HOST);
}

int rmav_set_state(rmav_handle h, const float *in, int mem, int layout) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, layout)) return rc;
    if (!in) return rmav_fail(RMAV_ERR_INVALID, "in is NULL");
    const int nS = kStateDim[h->kind];
    const size_t cnt = (size_t)h->n * nS;
    if (layout == RMAV_SOA) return copy_in(h, h->state, in, cnt, mem);
    const float *src = in;
    if (mem == RMAV_HOST) {
        if (int rc = ensure_scratch(h, cnt * sizeof(float))) return rc;
        HIP_TRY(hipMemcpyAsync(h->scratch, in, cnt * sizeof(float), hipMemcpyHostToDevice, h->stream));
        src = (const float *)h->scratch;
    }
    hipLaunchKernelGGL(k_aos_to_soa, grid_for(h), dim3(block_size(h)), 0, h->stream, src, h->state, h->n, nS);
    HIP_TRY(hipGetLastError());
    if (mem == RMAV_HOST) HIP_TRY(hipStreamSynchronize(h->stream));
    return RMAV_OK;
}

int rmav_get_sbd(rmav_handle h, int32_t *out, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    return rec_field_get(h, 0, reinterpret_cast<uint32_t *>(out), mem);
}
int rmav_set_sbd(rmav_handle h, const int32_t *in, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    return rec_field_set(h, 0, reinterpret_cast<const uint32_t *>(in), mem);
}
int rmav_get_reset_counts(rmav_handle h, uint32_t *out, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    return rec_field_get(h, 1, out, mem);
}
int rmav_set_reset_counts(rmav_handle h, const uint32_t *in, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    return rec_field_set(h, 1, in, mem);
}

int rmav_get_time(rmav_handle h, double *out, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    if (!h->env_time) return rmav_fail(RMAV_ERR_INVALID, "only reinmav envs carry their own clock");
    return copy_out(h, (const double *)h->env_time, out, (size_t)h->n, mem);
}
int rmav_set_time(rmav_handle h, const double *in, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    if (!h->env_time) return rmav_fail(RMAV_ERR_INVALID, "only reinmav envs carry their own clock");
    return copy_in(h, h->env_time, in, (size_t)h->n, mem);
}

int rmav_get_step_count(rmav_handle h, uint64_t *out) {
    if (!valid(h)) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_handle");
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    *out = h->t;
    return RMAV_OK;
}
int rmav_set_step_count(rmav_handle h, uint64_t t) {
    CHECK_HANDLE(h);
    return move_step_counter(h, t);
}

int rmav_episode_totals(rmav_handle h, rmav_ep_totals *out, int clear) {
    CHECK_HANDLE(h);
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    if (!(h->flags & RMAV_F_TRACK_EPISODES))
        return rmav_fail(RMAV_ERR_INVALID, "handle was created without RMAV_F_TRACK_EPISODES");
    const size_t nw = n_total_slots(h->n);
    Totals *host = new (std::nothrow) Totals[nw];
    if (!host) return rmav_fail(RMAV_ERR_ALLOC, "host allocation failed");
    hipError_t e = hipMemcpyAsync(host, h->totals, nw * sizeof(Totals), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        delete[] host;
        return rmav_fail(RMAV_ERR_HIP, "reading episode totals failed: %s", hipGetErrorString(e));
    }
    out->episodes = 0;
    out->return_sum = 0.0;
    out->length_sum = 0;
    for (size_t i = 0; i < nw; ++i) {
        out->episodes += host[i].episodes;
        out->return_sum += host[i].return_sum;
        out->length_sum += host[i].length_sum;
    }
    delete[] host;
    if (clear) HIP_TRY(hipMemsetAsync(h->totals, 0, nw * sizeof(Totals), h->stream));
    return RMAV_OK;
}

int rmav_episode_buffers(rmav_handle h, float *last_return, int32_t *last_length, float *cur_return,
                         int32_t *cur_length, int mem) {
    CHECK_HANDLE(h);
    if (int rc = check_mem_layout(mem, RMAV_SOA)) return rc;
    if (!(h->flags & RMAV_F_TRACK_EPISODES))
        return rmav_fail(RMAV_ERR_INVALID, "handle was created without RMAV_F_TRACK_EPISODES");
    const size_t n = (size_t)h->n;
    const hipMemcpyKind kind = (mem == RMAV_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (last_return) HIP_TRY(hipMemcpyAsync(last_return, h->last_ret, n * sizeof(float), kind, h->stream));
    if (cur_return) HIP_TRY(hipMemcpyAsync(cur_return, h->ep_ret, n * sizeof(float), kind, h->stream));
    // lengths live in the per-env records (EnvRec): last_len as it is, the running length = episode clock - the episode's start
    if (last_length)
        if (int rc = rec_field_get(h, 3, reinterpret_cast<uint32_t *>(last_length), mem)) return rc;
    if (cur_length) {
        const uint32_t clock = (uint32_t)h->t;
        int32_t *dst = cur_length;
        if (mem != RMAV_DEVICE) {
            if (int rc = ensure_scratch(h, n * sizeof(int32_t))) return rc;
            dst = (int32_t *)h->scratch;
        }
        hipLaunchKernelGGL(k_cur_length, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dst, (const EnvRec *)h->rec, clock, (int64_t)n);
        HIP_TRY(hipGetLastError());
        if (mem != RMAV_DEVICE)
            if (int rc = copy_out(h, (const int32_t *)dst, cur_length, n, RMAV_HOST)) return rc;
    }
    if (mem == RMAV_HOST) HIP_TRY(hipStreamSynchronize(h->stream));
    return RMAV_OK;
}

// ---- learner-side helpers on the trajectory ----------------------------------------------------
int rmav_gae(rmav_handle h, int32_t n_steps, const float *rew, const uint8_t *done, const float *values,
             float gamma, float lam, float reward_scale, float *adv_out, float *ret_out, double *sums_out) {
    CHECK_HANDLE(h);
    if (n_steps <= 0) return rmav_fail(RMAV_ERR_INVALID, "n_steps must be > 0");
    if (!rew || !done || !values || !adv_out || !ret_out)
        return rmav_fail(RMAV_ERR_INVALID, "rew, done, values, adv_out and ret_out are required (device pointers)");
    const unsigned nblk = (unsigned)((h->n + 255) / 256);
    double *partial = nullptr;
    if (sums_out) {
        if (int rc = ensure_scratch(h, (size_t)nblk * 2 * sizeof(double))) return rc;
        partial = (double *)h->scratch;
    }
    hipLaunchKernelGGL(k_gae, dim3(nblk), dim3(256), 0, h->stream, rew, done, values, adv_out, ret_out, h->n, n_steps,
                       gamma, lam, reward_scale, partial);
    HIP_TRY(hipGetLastError());
    if (sums_out) {
        hipLaunchKernelGGL(k_gae_fold, dim3(1), dim3(256), 0, h->stream, (const double *)partial, (int)nblk, sums_out);
        HIP_TRY(hipGetLastError());
    }
    return RMAV_OK;
}

int rmav_normalize(rmav_handle h, float *x, int64_t count, float mean, float rstd) {
    CHECK_HANDLE(h);
    if (!x || count < 0) return rmav_fail(RMAV_ERR_INVALID, "x is NULL or count < 0");
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return rmav_fail(RMAV_ERR_INVALID, "x must be 16-byte aligned");
    if (count == 0) return RMAV_OK;
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;   // grid-stride: 16 blocks per CU keep the memory system full
    hipLaunchKernelGGL(k_affine, dim3((unsigned)blocks), dim3(256), 0, h->stream, x, count, mean, rstd);
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

// ---- the path's one collective, behind the C ABI: RCCL all-gather of per-env episode statistics ---------------
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;      // optional (rmav_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;   // optional
};
// resolved on first use: librmav.so has no link-time dependency on RCCL, and a process that already loaded
// librccl.so.1 (torch does) shares that copy
char g_rccl_path[1024] = "";   // rmav_comm_use_library
bool g_rccl_tried = false;
RcclApi *rccl() {
    static RcclApi api;
    if (!g_rccl_tried) {
        g_rccl_tried = true;
        if (g_rccl_path[0]) {
            api.lib = dlopen(g_rccl_path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (api.lib) break;
            }
        }
        if (api.lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
            api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
            api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) api.lib = nullptr;
        }
    }
    return api.lib ? &api : nullptr;
}
constexpr uint32_t kCommMagic = 0x524d4143u;  // 'RMAC'
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return rmav_fail(RMAV_ERR_HIP, "%s failed: %s", #expr, R->GetErrorString ? R->GetErrorString(r_) : "RCCL error"); \
    } while (0)
}  // namespace

int rmav_comm_use_library(const char *path) {
    if (g_rccl_tried) return rmav_fail(RMAV_ERR_INVALID, "the collective library has already been loaded: call this before any other rmav_comm_* function");
    if (!path || !path[0] || strlen(path) >= sizeof(g_rccl_path)) return rmav_fail(RMAV_ERR_INVALID, "path is NULL, empty or too long");
    snprintf(g_rccl_path, sizeof(g_rccl_path), "%s", path);
    return RMAV_OK;
}

int rmav_comm_unique_id(void *id_out) {
    if (!id_out) return rmav_fail(RMAV_ERR_INVALID, "id_out is NULL");
    RcclApi *R = rccl();
    if (!R) return rmav_fail(RMAV_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == RMAV_COMM_ID_BYTES, "RCCL unique id size");
    memcpy(id_out, &id, sizeof(id));
    return RMAV_OK;
}

int rmav_comm_info(rmav_comm c, int *rank_out, int *world_out, int *lib_rank_out, int *lib_world_out) {
    if (!c || c->magic != kCommMagic) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_comm");
    if (rank_out) *rank_out = c->rank;
    if (world_out) *world_out = c->world;
    RcclApi *R = rccl();
    int lr = -1, lw = -1;
    if (R && R->CommUserRank && R->CommUserRank(c->comm, &lr) != ncclSuccess) lr = -1;
    if (R && R->CommCount && R->CommCount(c->comm, &lw) != ncclSuccess) lw = -1;
    if (lib_rank_out) *lib_rank_out = lr;
    if (lib_world_out) *lib_world_out = lw;
    return RMAV_OK;
}

int rmav_comm_create(rmav_comm *out, const void *id, int rank, int world, int device) {
    if (!out) return rmav_fail(RMAV_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id || world <= 0 || rank < 0 || rank >= world) return rmav_fail(RMAV_ERR_INVALID, "need id and 0 <= rank < world");
    const int ndev = rmav_device_count();
    if (ndev <= 0) return rmav_fail(RMAV_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return rmav_fail(RMAV_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    RcclApi *R = rccl();
    if (!R) return rmav_fail(RMAV_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
    DeviceGuard guard(device);
    if (!guard.ok) return rmav_fail(RMAV_ERR_HIP, "hipSetDevice(%d) failed", device);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    rmav_comm c = new (std::nothrow) rmav_comm_s();
    if (!c) return rmav_fail(RMAV_ERR_ALLOC, "host allocation failed");
    memset(c, 0, sizeof(*c));
    c->magic = kCommMagic;
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclResult_t r = R->CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return rmav_fail(RMAV_ERR_HIP, "ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
    }
    // A high-priority stream: HIP multiplexes all streams of one priority onto a few hardware queues (GPU_MAX_HW_QUEUES,
    // 4 by default) round-robin, and a process that also runs torch has dozens - when the communicator's stream lands on
    // the compute stream's hardware queue their packets serialise (measured: a 131 072-env rollout 85 -> 108 us with the
    // default mapping, 189 us with GPU_MAX_HW_QUEUES=8, 93 us with 2).  Priority levels have queues of their own.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
    c->depth = kExchangeDepth;
    for (int k = 0; k < kExchangeDepth && e == hipSuccess; ++k) {
        // device-scope release: these events only order streams of this GPU
        const unsigned evf = hipEventDisableTiming | hipEventReleaseToDevice;
        e = hipEventCreateWithFlags(&c->ready[k], evf);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[k], evf);
    }
    // the words k_wait_arrivals writes when it gives up on an armed launch (pinned host memory: the host reads them for free)
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->timeout_seq, kExchangeDepth * sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        memset(c->timeout_seq, 0, kExchangeDepth * sizeof(uint32_t));
        e = hipHostGetDevicePointer((void **)&c->timeout_seq_dev, c->timeout_seq, 0);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&c->started, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->started, 0, sizeof(uint32_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)rmav_comm_destroy(c);
        return rmav_fail(RMAV_ERR_HIP, "stream / event creation for the communicator failed: %s", hipGetErrorString(e));
    }
    // Hand-over from the compute stream to the communicator's stream without an event: hipEventRecord puts a barrier
    // packet into the COMPUTE stream (~8 us in front of the next rollout launch, measured); a one-thread kernel that
    // publishes the post number in a signal word, and hipStreamWaitValue32 on the communicator's stream, cost the
    // compute stream one tiny launch (hipStreamWriteValue32 in its place: +3 us per post, measured).  Falls back to the event when the device cannot wait on memory.
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, device);
    if (can_wait) {
        if (hipExtMallocWithFlags((void **)&c->flag, 8, hipMallocSignalMemory) != hipSuccess) {
            (void)hipGetLastError();
            c->flag = nullptr;
        } else {
            (void)hipMemset(c->flag, 0, 8);
        }
    }
    *out = c;
    return RMAV_OK;
}

int rmav_comm_destroy(rmav_comm c) {
    if (!c || c->magic != kCommMagic) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_comm");
    DeviceGuard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    RcclApi *R = rccl();
    if (R && c->comm) (void)R->CommDestroy(c->comm);
    for (int k = 0; k < kExchangeDepth; ++k) {
        if (c->ready[k]) (void)hipEventDestroy(c->ready[k]);
        if (c->done[k]) (void)hipEventDestroy(c->done[k]);
        if (c->send[k]) (void)hipFree(c->send[k]);
        if (c->recv[k]) (void)hipFree(c->recv[k]);
    }
    if (c->flag) (void)hipFree(c->flag);
    if (c->arrive) (void)hipFree(c->arrive);
    if (c->timeout_seq) (void)hipHostFree(c->timeout_seq);
    if (c->started) (void)hipFree(c->started);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->armed_by && c->armed_by->magic == kMagic && c->armed_by->xchg.comm == c) {
        // a handle still points at this communicator: disarm it, or its next rollout would dereference freed memory
        c->armed_by->xchg.armed = c->armed_by->xchg.fired = false;
        c->armed_by->xchg.comm = nullptr;
    }
    c->magic = 0;
    delete c;
    return RMAV_OK;
}

int rmav_comm_warmup(rmav_comm c, double timeout_s) {
    if (!c || c->magic != kCommMagic) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_comm");
    RcclApi *R = rccl();
    if (!R) return rmav_fail(RMAV_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
    DeviceGuard guard(c->device);
    int32_t *buf = nullptr;   // [1 + world]: this rank's word, then the gathered words
    HIP_TRY(hipMalloc((void **)&buf, sizeof(int32_t) * (size_t)(1 + c->world)));
    hipEvent_t ev = nullptr;
    hipError_t e = hipMemsetAsync(buf, 0, sizeof(int32_t) * (size_t)(1 + c->world), c->stream);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    int rc = RMAV_OK;
    if (e != hipSuccess) {
        rc = rmav_fail(RMAV_ERR_HIP, "rmav_comm_warmup: %s", hipGetErrorString(e));
    } else {
        // RCCL connects its transports inside the first collective's enqueue (a host-side exchange with the peers): this is the
        // call that may block when a peer is gone, and it touches no handle's stream
        const ncclResult_t r = R->AllGather(buf, buf + 1, 1, ncclInt32, c->comm, c->stream);
        if (r != ncclSuccess) rc = rmav_fail(RMAV_ERR_HIP, "ncclAllGather failed: %s", R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
    }
    if (rc == RMAV_OK && hipEventRecord(ev, c->stream) != hipSuccess) rc = rmav_fail(RMAV_ERR_HIP, "hipEventRecord failed");
    if (rc == RMAV_OK) {
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) break;
            (void)hipGetLastError();
            if (q != hipErrorNotReady) { rc = rmav_fail(RMAV_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(q)); break; }
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (timeout_s >= 0 && (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) {
                rc = rmav_fail(RMAV_ERR_TIMEOUT, "the warm-up collective did not complete within %.3f s", timeout_s);
                break;
            }
            timespec nap = {0, 50000};
            nanosleep(&nap, nullptr);
        }
    }
    // (on a time-out the collective may still be in flight: the buffer and the event are left to the process, not freed under it)
    if (rc != RMAV_ERR_TIMEOUT) {
        if (ev) (void)hipEventDestroy(ev);
        (void)hipFree(buf);
    }
    return rc;
}

int rmav_pack_stats(rmav_handle h, int64_t cmax, int32_t *send_out) {
    CHECK_HANDLE(h);
    if (!(h->flags & RMAV_F_TRACK_EPISODES))
        return rmav_fail(RMAV_ERR_INVALID, "handle was created without RMAV_F_TRACK_EPISODES");
    if (!send_out || cmax < h->n) return rmav_fail(RMAV_ERR_INVALID, "send_out is NULL or cmax < num_envs");
    hipLaunchKernelGGL(k_pack_stats, dim3((unsigned)((cmax + 255) / 256)), dim3(256), 0, h->stream,
                       (const float *)h->last_ret, (const EnvRec *)h->rec, h->n, cmax, send_out);
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

namespace {
// shard of rank c->rank out of n_total, checked against the handle
int check_shard(rmav_handle h, rmav_comm c, int64_t n_total, int64_t *cmax_out) {
    if (!c || c->magic != kCommMagic) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_comm");
    if (!(h->flags & RMAV_F_TRACK_EPISODES))
        return rmav_fail(RMAV_ERR_INVALID, "handle was created without RMAV_F_TRACK_EPISODES");
    if (c->device != h->device) return rmav_fail(RMAV_ERR_INVALID, "communicator and handle live on different devices");
    const int64_t W = c->world, base = n_total / W, rem = n_total % W;
    if (n_total <= 0 || base == 0) return rmav_fail(RMAV_ERR_INVALID, "n_total must be >= the number of ranks");
    const int64_t count = base + (c->rank < rem ? 1 : 0), start = c->rank * base + (c->rank < rem ? c->rank : rem);
    if (h->n != count || (int64_t)h->env_base != start)
        return rmav_fail(RMAV_ERR_INVALID, "rank %d of %d must own envs [%lld, %lld) of %lld; the handle owns [%llu, %llu)", c->rank,
                    c->world, (long long)start, (long long)(start + count), (long long)n_total,
                    (unsigned long long)h->env_base, (unsigned long long)(h->env_base + (uint64_t)h->n));
    *cmax_out = base + (rem ? 1 : 0);
    return RMAV_OK;
}
}  // namespace

namespace {
// Common front of _arm and _post: shard check, buffers, and the buffer pair of the next post (host-side back pressure).
int exchange_slot(rmav_handle h, rmav_comm c, int64_t n_total, int64_t *cmax_out, int *slot_out) {
    int64_t cmax = 0;
    if (int rc = check_shard(h, c, n_total, &cmax)) return rc;
    RcclApi *R = rccl();
    if (!R) return rmav_fail(RMAV_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
    if (cmax > c->cmax) {   // (re)allocate the buffer pairs
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        for (int k = 0; k < c->depth; ++k) {
            if (c->send[k]) (void)hipFree(c->send[k]);
            if (c->recv[k]) (void)hipFree(c->recv[k]);
            c->send[k] = c->recv[k] = nullptr;
            c->used[k] = false;
            if (hipMalloc((void **)&c->send[k], (size_t)(2 * cmax) * sizeof(int32_t)) != hipSuccess ||
                hipMalloc((void **)&c->recv[k], (size_t)(2 * cmax) * sizeof(int32_t) * (size_t)c->world) != hipSuccess) {
                (void)hipGetLastError();
                c->cmax = 0;
                return rmav_fail(RMAV_ERR_ALLOC, "device allocation for the exchange buffers failed");
            }
            // an armed launch writes only this rank's envs: the padding up to cmax stays zero from here on
            HIP_TRY(hipMemsetAsync(c->send[k], 0, (size_t)(2 * cmax) * sizeof(int32_t), c->stream));
        }
        if (c->arrive) (void)hipFree(c->arrive);
        c->arrive = nullptr;
        const size_t words = (size_t)((cmax + 31) / 32);
        if (hipMalloc((void **)&c->arrive, words * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            c->cmax = 0;
            return rmav_fail(RMAV_ERR_ALLOC, "device allocation for the exchange buffers failed");
        }
        HIP_TRY(hipMemsetAsync(c->arrive, 0, words * sizeof(uint32_t), c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->cmax = cmax;
    }
    const int k = c->posts % c->depth;
    // The gather that last used this buffer pair (`depth` posts ago) must have finished before the pack overwrites its
    // send half.  In a GPU-bound loop the host runs far ahead of the device, so with two pairs that gather has usually not
    // even started when the host gets here, and a device-side wait (hipStreamWaitEvent = a barrier packet in the COMPUTE
    // stream) was inserted in front of nearly every pack: +8 us per rollout, measured.  So the HOST waits instead - back
    // pressure that bounds its lead to `depth` rollouts (>= 0.5 ms of queued GPU work at depth 8) and puts nothing into
    // the compute stream.
    if (c->used[k] && hipEventQuery(c->done[k]) != hipSuccess) {
        (void)hipGetLastError();
        HIP_TRY(hipEventSynchronize(c->done[k]));
    }
    *cmax_out = cmax;
    *slot_out = k;
    return RMAV_OK;
}
}  // namespace

int rmav_allgather_stats_arm(rmav_handle h, rmav_comm c, int64_t n_total) {
    CHECK_HANDLE(h);
    if (h->xchg.armed) return rmav_fail(RMAV_ERR_INVALID, "an exchange is already armed on this handle: post it first");
    if (c && c->magic == kCommMagic && c->armed_by && c->armed_by != h)
        return rmav_fail(RMAV_ERR_INVALID, "this communicator is armed by another handle: post that exchange first");
    int64_t cmax = 0;
    int k = 0;
    if (int rc = exchange_slot(h, c, n_total, &cmax, &k)) return rc;
    h->xchg.armed = true;
    h->xchg.fired = false;
    h->xchg.stale = false;
    h->xchg.comm = c;
    c->armed_by = h;
    h->xchg.slot = k;
    h->xchg.cmax = cmax;
    h->xchg.seq = (uint32_t)(c->posts + 1);
    h->xchg.expected = 0;
    return RMAV_OK;
}

int rmav_allgather_stats_post(rmav_handle h, rmav_comm c, int64_t n_total) {
    CHECK_HANDLE(h);
    int64_t cmax = 0;
    int k = 0;
    RcclApi *R = rccl();
    const bool armed = h->xchg.armed;
    if (armed && h->xchg.comm != c) return rmav_fail(RMAV_ERR_INVALID, "the handle's armed exchange belongs to another communicator");
    if (armed) {   // allocated and back-pressured when it was armed
        if (!R) return rmav_fail(RMAV_ERR_NO_DEVICE, "librccl.so.1 could not be loaded");
        if (int rc = check_shard(h, c, n_total, &cmax)) return rc;
        if (cmax != h->xchg.cmax) return rmav_fail(RMAV_ERR_INVALID, "n_total differs from the armed exchange's");
        cmax = h->xchg.cmax;
        k = h->xchg.slot;
        h->xchg.armed = false;
        h->xchg.comm = nullptr;
        c->armed_by = nullptr;
    } else if (int rc = exchange_slot(h, c, n_total, &cmax, &k)) {
        return rc;
    }
    // (the gather that last used this buffer pair has finished - exchange_slot waited for it - so its time-out word is history)
    c->timeout_seq[k] = 0;
    c->slot_seq[k] = (uint32_t)(c->posts + 1);
    c->armed_slot[k] = armed && h->xchg.fired && !h->xchg.stale;
    if (c->armed_slot[k]) {
        // the rollout launch itself wrote the snapshot and its wavefronts' arrival words: nothing enters the compute stream.
        // The wait is bounded (k_wait_arrivals: 2 s from the moment the armed launch begins): past that the waiter poisons this
        // rank's payload, notes the post number in the pair's time-out word and lets the gather go ahead - the peers get their
        // collective either way, and only THIS post reports RMAV_ERR_TIMEOUT.
        hipLaunchKernelGGL(k_wait_arrivals, dim3(1), dim3(256), 0, c->stream, (const uint32_t *)c->arrive, h->xchg.expected,
                           h->xchg.seq, h->xchg.no_start ? (const uint32_t *)nullptr : (const uint32_t *)c->started, kArrivalWaitTicks,
                           kArrivalTotalTicks, c->timeout_seq_dev + k, c->send[k], cmax);
        HIP_TRY(hipGetLastError());
    } else {
        hipLaunchKernelGGL(k_pack_stats, dim3((unsigned)((cmax + 255) / 256)), dim3(256), 0, h->stream,
                           (const float *)h->last_ret, (const EnvRec *)h->rec, h->n, cmax, c->send[k]);
        HIP_TRY(hipGetLastError());
        if (c->flag) {
            const uint32_t seq = (uint32_t)(c->posts + 1);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, h->stream, c->flag, seq);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamWaitValue32(c->stream, c->flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        } else {
            HIP_TRY(hipEventRecord(c->ready[k], h->stream));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ready[k], 0));
        }
    }
    RCCL_TRY(R->AllGather(c->send[k], c->recv[k], (size_t)(2 * cmax), ncclInt32, c->comm, c->stream));
    HIP_TRY(hipEventRecord(c->done[k], c->stream));
    c->used[k] = true;
    c->posts += 1;
    return RMAV_OK;
}

int rmav_allgather_stats_result(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out) {
    CHECK_HANDLE(h);
    int64_t cmax = 0;
    if (int rc = check_shard(h, c, n_total, &cmax)) return rc;
    if (!returns_out || !lengths_out) return rmav_fail(RMAV_ERR_INVALID, "returns_out / lengths_out are required (device pointers)");
    if (c->posts == 0 || cmax != c->cmax) return rmav_fail(RMAV_ERR_INVALID, "no exchange of this size has been posted");
    const int k = (c->posts - 1) % c->depth;
    // (known only if the waiter has already run; rmav_allgather_stats_wait knows for certain.  Either way the payload of a
    // timed-out post is poisoned - return NaN, length -1 for this rank's envs - on every rank.)
    if (c->armed_slot[k] && c->timeout_seq[k] == c->slot_seq[k])
        return rmav_fail(RMAV_ERR_TIMEOUT, "the armed rollout launch of this exchange did not complete within 2 s of starting");
    HIP_TRY(hipStreamWaitEvent(h->stream, c->done[k], 0));
    hipLaunchKernelGGL(k_unpack_stats, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, h->stream,
                       (const int32_t *)c->recv[k], n_total, (int32_t)c->world, cmax, returns_out, lengths_out);
    HIP_TRY(hipGetLastError());
    return RMAV_OK;
}

int rmav_allgather_stats_wait(rmav_comm c, double timeout_s) {
    if (!c || c->magic != kCommMagic) return rmav_fail(RMAV_ERR_INVALID, "invalid rmav_comm");
    if (c->posts == 0) return RMAV_OK;
    DeviceGuard guard(c->device);
    const int k = (c->posts - 1) % c->depth;
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        const hipError_t e = hipEventQuery(c->done[k]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) return rmav_fail(RMAV_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (timeout_s >= 0 && (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s)
            return rmav_fail(RMAV_ERR_TIMEOUT, "the posted exchange did not complete within %.3f s", timeout_s);
        timespec nap = {0, 50000};
        nanosleep(&nap, nullptr);
    }
    if (c->armed_slot[k] && c->timeout_seq[k] == c->slot_seq[k])
        return rmav_fail(RMAV_ERR_TIMEOUT, "the armed rollout launch of this exchange did not complete within 2 s of starting");
    return RMAV_OK;
}

int rmav_allgather_stats(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out) {
    if (!returns_out || !lengths_out) return rmav_fail(RMAV_ERR_INVALID, "returns_out / lengths_out are required (device pointers)");
    if (int rc = rmav_allgather_stats_post(h, c, n_total)) return rc;
    return rmav_allgather_stats_result(h, c, n_total, returns_out, lengths_out);
}

}  // extern "C"
