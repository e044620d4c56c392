// rmav_kernels.hpp - HIP kernels of the batched quadrotor path (gfx950 / CDNA4, wave64).
//
// Data layout in HBM (per handle, N envs):
//   state      f32 [nS][N]   struct-of-arrays, updated in place; lane i of a wavefront owns env i,
//                            so every load/store of a component is one fully coalesced 256-byte
//                            wave transaction
//   rec        EnvRec [N]    ONE 16-byte record per env of what a lane needs only when its episode ENDS: steps_beyond_done
//                            (-1 = None), resets drawn so far (the RNG counter), the episode clock at the running episode's
//                            start, the last finished episode's length - one sector in, one out per termination (round 6;
//                            four separate arrays before: three scattered sectors in, four out)
//   ep_ret, last_ret   f32 [N]  running return (in / out every step when tracking), last finished episode's return
//   totals     {u64,f64,u64} [ceil(N/64)]  per-wavefront partial sums of finished episodes
//   env_time   f64 [N]       RMAV_REINMAV only: each env's clock
//   pe[3]      f32 [N]       optional per-env mass / load mass / tether length (domain randomisation)
// Caller buffers: actions [T][nA][N] | [T][N][nA], obs [T][nS][N] | [T][N][nS], rew f32 [T][N],
// done u8 [T][N].  INVARIANT: every kernel stores exactly 0 or 1 into `done` (all done stores below are `cond ? 1 : 0`):
// the Python layer views the bytes as bool without a conversion pass (vec_env.py).
//
// One kernel template covers step (n_steps = 1) and the fused rollout (n_steps = T, state held in
// registers between steps, so per-step HBM traffic shrinks to actions-in + trajectory-out).  The action
// source is a template parameter: caller buffer, counter RNG, the reference's geometric controller, or a
// Gaussian MLP policy evaluated in-kernel (fp32 VALU / bf16 MFMA).
// Per-step physics constants arrive as kernel arguments (scalar registers via s_load), not LDS: they are
// wave-uniform, ~200 bytes, and an LDS copy would cost a barrier per launch for nothing.  LDS is used where
// the constants are too big for SGPRs (the policy weights, 30-42 KB, staged once per launch), for the hand-over
// tiles between the integrator and the memory wavefront of small-batch fused rollouts, and for transposing
// batch-major obs tiles of big launches.
#pragma once

#include "rmav_math.hpp"
#include "rmav_policy.hpp"
#include "rmav_policy_mfma.hpp"
#include "rmav_policy_mfma32.hpp"

namespace rmav {

enum : int { ACT_BUFFER = 0, ACT_RANDOM = 1, ACT_CONTROLLER = 2, ACT_POLICY = 3, ACT_POLICY_BF16 = 4,
              // internal: ACT_RANDOM / ACT_CONTROLLER with a second, "memory" wavefront per 64 envs (see k_rollout)
              ACT_RANDOM_SPLIT = 5, ACT_CONTROLLER_SPLIT = 6,
              // internal: ACT_BUFFER, and the launch ends by evaluating control() on the state it leaves behind
              // (rmav_step_control: the gym-shaped single env gets step()'s outputs and the NEXT control() in one launch)
              ACT_BUFFER_CTRL = 7,
              // the fp32 policy on the fp32-input matrix instructions (rmav_policy_mfma32.hpp)
              ACT_POLICY_F32M = 8,
              // internal: ACT_BUFFER with the memory wavefront prefetching the caller's actions into the hand-over tile
              ACT_BUFFER_SPLIT = 9,
              // the f16 matrix-core actor (RMAV_POLICY_F16_MFMA): only as an (actor, critic) wavefront pair, rmav_policy_pair.hpp
              ACT_POLICY_F16 = 10,
              // ... and the shared-trunk form (one net with a mean head and a value head: RMAV_POLICY_F16_SHARED)
              ACT_POLICY_F16_SHARED = 11 };
constexpr bool is_mfma_policy(int mode) { return mode == ACT_POLICY_BF16 || mode == ACT_POLICY_F32M; }
constexpr bool is_policy(int mode) { return mode == ACT_POLICY || is_mfma_policy(mode); }
constexpr bool is_split(int mode) { return mode == ACT_RANDOM_SPLIT || mode == ACT_CONTROLLER_SPLIT || mode == ACT_BUFFER_SPLIT; }
// split mode whose memory wavefront hands actions TO the integrator (it draws them).  The caller's actions (ACT_BUFFER_SPLIT) were
// fetched by the memory wavefront as well until round 6: on gfx950 loads and stores share ONE counter (vmcnt) and return out of order
// with respect to each other, so a wavefront that both prefetches and stores can only wait for a load with vmcnt(0) - i.e. for every
// outstanding trajectory store (profiles/r06/kernel_sweep.md: 59 % of the wavefronts' cycles in waits).  Now the INTEGRATOR fetches
// them - it has no store in its step loop any more: its rare termination stores moved to the memory wavefront - in bursts of
// buf_prefetch<NA>() steps through a private LDS ring (split_self_fetch).
constexpr bool split_feeds_actions(int mode) { return mode == ACT_RANDOM_SPLIT; }
constexpr bool split_self_fetch(int mode) { return mode == ACT_BUFFER_SPLIT; }
constexpr bool is_buffer(int mode) { return mode == ACT_BUFFER || mode == ACT_BUFFER_CTRL; }
// kernels whose first workgroup publishes "begun" for an armed statistics exchange (see the top of k_rollout)
constexpr bool publishes_start(int mode) { return mode != ACT_CONTROLLER_SPLIT; }
// Split modes: env-steps per hand-over, and the LDS words of the double-buffered tiles
// (actions: helper -> integrator, only when the helper draws them; obs + reward + done [+ actions]: integrator -> helper)
// 1: the tiles of 8 pairs take 70 KB (quad3d) instead of 136 KB of the CU's 160 KB, so a communication kernel's
// workgroups (the overlapped statistics exchange) can share the CU with a rollout workgroup - with 2, a rollout next to a
// 60-80 us co-resident kernel measured +22..29 us at 131 072 envs, with 1 +10..11 us - and alone it is as fast or faster
// (profiles/r02/handover_chunk.md).
#ifndef RMAV_SPLIT_CHUNK
#define RMAV_SPLIT_CHUNK 1
#endif
constexpr int kSplitChunk = RMAV_SPLIT_CHUNK;
// Split modes: (integrator, memory wavefront) pairs per workgroup - a launch parameter (blockDim.x / 128, 1 .. 8).
// The pairs of one workgroup own ADJACENT 64-env groups and run on one CU, so its memory wavefronts store adjacent
// 256-byte segments of every trajectory column (the store-only microbenchmark tools/micro/store_patterns.hip absorbs
// 1 KiB-per-CU segments 6 % faster than 256-byte ones at 65 536 envs on cold buffers, and is at its fastest with
// 128-256 large workgroups).  Measured on cold trajectory buffers, 64-step launches (profiles/r02/split_group_ab.md,
// wg_scan.md): 1 -> 4 pairs at 65 536 envs: quadrotor3d 46.2 -> 45.5 us, slung load 74.3 -> 73.4, 2-D 45.5 -> 43.6,
// 2-D slung load 67.8 -> 65.1; 4 -> 8 pairs at 131 072 envs: quadrotor3d 102 -> 90 us (the one-wavefront kernel: 101),
// but 45 -> 64 us at 65 536 envs, where 8 pairs leave half of the CUs empty and put two integrators on every SIMD.
// The sweet spot is ONE workgroup per CU: the host launches ceil(N / 16 384) pairs per workgroup (256 workgroups).
constexpr int kSplitGroupMax = 8;   // 1024 threads
// ... which caps the kernel at 128 VGPRs.  Round 2's controller-driven slung-load integrators (fp64 controller on top of
// the fp64 step) needed 145-147 and stayed at 4 pairs; with the spare reset state in LDS (SplitTile::SPARE), dirty flags
// instead of copies of the loaded counters and the 8-coefficient atan2 of the 2-D controller they take 109 / 124, so
// every kind runs up to 8 pairs (tests/test_resource_usage.py pins the budgets).
template <int K, bool DRAWS> constexpr int split_group_cap() { return kSplitGroupMax; }
#ifndef RMAV_KBLOCK
#define RMAV_KBLOCK 256
#endif
constexpr int kBlock = RMAV_KBLOCK;  // upper bound (launch bounds); the launch may use 64/128/256 (.. RMAV_KBLOCK)
template <int K, int MODE> constexpr int rollout_threads_max() {   // launch bounds of k_rollout<K, MODE, *>
    return is_split(MODE) ? 128 * split_group_cap<K, split_feeds_actions(MODE)>() : kBlock;
}
template <int NS, int NA, bool DRAWS = true> struct SplitTile {
    static constexpr int CH = kSplitChunk;   // env-steps per hand-over
    static constexpr int A_HALF = DRAWS ? CH * NA * 64 : 0, A_WORDS = 2 * A_HALF;
    // one env-step of outputs: obs (feature-major [c][lane], or env-major [lane][c] with an odd row stride when the
    // trajectory is batch-major - both conflict-free to write), then reward[64], done[64]
    static constexpr int OBS_STRIDE = NS | 1, REW = OBS_STRIDE * 64, DONE = REW + 64;
    static constexpr int ACT = DONE + 64;   // actions [c][lane], only when the integrator computes them
    static constexpr int O_ROW = ACT + (DRAWS ? 0 : NA * 64), O_HALF = CH * O_ROW, O_WORDS = 2 * O_HALF;
    static constexpr int WORDS = A_WORDS + O_WORDS;
    // + (3-D slung load only, see spare_in_lds) the integrator's spare reset state [c][lane], drawn once per
    // launch and consumed by the first termination of a lane, laid out after the tiles of all pairs
    static constexpr int SPARE = NS * 64;
};
// Where the integrator of a two-wavefront kernel keeps its spare reset state.  In registers it is a predicated copy when a
// lane terminates; in LDS it frees NS = 16 registers of the step loop - what the controller-driven 3-D slung-load integrator
// (fp64 controller on top of the fp64 step: 140 VGPRs with it, 124 without) needs to fit the 128-register budget of 8 pairs per
// workgroup - but puts an LDS round trip into the reset path, which nearly every step of a 4-pair workgroup takes (one
// barrier for all pairs; 97 % of its steps have a terminating lane somewhere): 2-D kinds at 65 536 envs +5 %.  So only there.
// (The random- and caller-action 3-D slung-load integrators sit at 127-131 registers with it: LDS as well; measured equal.)
template <int K, int MODE> constexpr bool spare_in_lds() { return K == QUAD3D_SL && is_split(MODE); }
#ifndef RMAV_BUF_PREFETCH
#define RMAV_BUF_PREFETCH 4   // env-steps per burst of the integrator's action fetch (ACT_BUFFER_SPLIT); a power of two
#endif
// ... and twice that for the 2-action kinds, whose steps are half as long
template <int NA> constexpr int buf_prefetch() { return NA <= 2 ? 2 * RMAV_BUF_PREFETCH : RMAV_BUF_PREFETCH; }
// LDS words of the integrator's private action ring [buf_prefetch][NA][lane] (ACT_BUFFER_SPLIT), after the tiles and spares of all pairs
template <int NA> constexpr int buf_ring_words() { return buf_prefetch<NA>() * NA * 64; }
template <int K, int MODE> constexpr int split_words_per_pair() {
    using T = SplitTile<Dims<K>::NS, Dims<K>::NA, split_feeds_actions(MODE)>;
    return T::WORDS + (spare_in_lds<K, MODE>() ? T::SPARE : 0) + (split_self_fetch(MODE) ? buf_ring_words<Dims<K>::NA>() : 0);
}
// F_LEAN (set by the host for the two-wavefront kernels): feature-major trajectories whose every array spans < 4 GiB, so the
// memory wavefront can address them with ONE descriptor per array and a 32-bit scalar step offset (see the lean drain below)
enum : uint32_t { F_AUTO_RESET = 1u, F_TRACK = 2u, F_AOS = 4u, F_LEAN = 8u };


// What a lane needs only when its env's episode ends, as ONE 16-byte record per env (b128 / b96 accesses):
//   sbd        steps_beyond_done, -1 = None (quadrotor3d.py:68,112-122)
//   reset_cnt  resets drawn so far = the Philox counter of the next fresh state
//   ep_start   value of the handle's episode clock when the running episode began: length = clock - ep_start (ep_clock0 below)
//   last_len   length of the most recently finished episode (0: none yet); its return is last_ret[i]
// Round 6 (profiles/r06/step_shape_*.md: a kernel of k_step's memory shape, 1.3 % of the lanes ending an episode): as four
// arrays a termination costs three scattered 32-byte sectors in and four partial sectors out (+ the return); as one record
// one sector in, one 16-byte store out - 1 048 576 envs 24.0 -> 21.6 us, 262 144 envs 5.8 (eager dword loads) -> 5.4.
struct alignas(16) EnvRec {
    int32_t sbd;
    uint32_t reset_cnt;
    uint32_t ep_start;
    int32_t last_len;
};
static_assert(sizeof(EnvRec) == 16, "one b128 access per record");

struct Totals {
    unsigned long long episodes;
    double return_sum;
    unsigned long long length_sum;
};

struct RolloutArgs {
    float *state;
    int64_t n;
    int64_t pitch;          // elements between the feature columns of the trajectory arrays (act_in / act_out / obs_out [T][dim][pitch],
                            // rew_out / done_out [T][pitch]): N, or rmav_rollout_pitched's; batch-major arrays always N
    const float *act_in;
    float *act_out;
    float *obs_out;
    float *rew_out;
    uint8_t *done_out;
    EnvRec *rec;            // per-env termination record {sbd, reset_cnt, ep_start, last_len} (see EnvRec)
    float *ep_ret;
    float *last_ret;
    Totals *totals;
    uint64_t seed;
    uint64_t env_base;
    uint64_t t0;
    int32_t n_steps;
    uint32_t flags;
    float act_lo, act_hi;
    // ACT_POLICY only
    const float *pe[3];     // optional per-env constants (mass, load mass, tether length), nullptr = shared
    double *env_time;       // REINMAV only: the env's own clock t [N] (fp64: it decides 50 vs 51 sub-steps)
    const float *policy_w;  // packed weights (rmav_policy.hpp layout), device memory
    float *logp_out;        // [n_steps][N]
    float *val_out;         // [n_steps + 1][N]
    float *ctrl_out;        // ACT_BUFFER_CTRL only: control() of the state after the last step, [nA][N] | [N][nA]
    // this launch covers envs [slice_first, slice_first + slice_count) of the handle's N (slice_count = 0: all of them);
    // trajectory pitches stay what they are.  slice_first is a multiple of 64.
    uint32_t slice_first, slice_count;
    // statistics exchange armed for this launch (rmav_allgather_stats_arm), nullptr otherwise: every wavefront snapshots
    // its envs' last-episode statistics into the exchange's send buffer [2][xcmax] and then publishes xseq in its word of
    // xarrive - the communicator's stream polls those words (k_wait_arrivals), so the compute stream carries no pack or
    // signal kernel at all
    int32_t *xsend;
    int64_t xcmax;
    uint32_t *xarrive;
    uint32_t xseq;
    // ... and the launch's first workgroup publishes xseq here as soon as it runs: the bound of the communicator stream's wait
    // counts from THIS moment, not from when the waiter started (an armed launch may sit behind seconds of queued work)
    uint32_t *xstarted;
    // k_step of a batch that fits ONE wavefront, outputs in the handle's pinned host block: the wavefront ends by publishing
    // done_seq in this pinned word (system-scope release), and the host spins on it instead of going through
    // hipStreamSynchronize (the kernel's end-of-kernel release + the completion signal + the runtime's wait: ~4 us of a
    // ~15 us host round trip).  nullptr otherwise.
    uint32_t *done_flag;
    uint32_t done_seq;
};

// Episode lengths are not stored: every env keeps the value the episode clock had when its running episode began (ep_start).
// The clock is the low 32 bits of the handle's step counter t (rmav_seed / rmav_set_step_count, which move t, shift ep_start
// by the same amount) - a value every stepping kernel has in scalar registers anyway, so the scheme costs no register.  A fused kernel turns that into a
// running length once (clock0 - ep_start), counts in registers as before, and writes clock0 + n_steps - length back; the
// single-step kernel touches ep_start only in lanes whose episode ends - no per-step read-modify-write of a length array
// (8 B of 24 B of bookkeeping traffic per env-step; round 5).  32-bit wrap-around is harmless: only differences are used.
__device__ __forceinline__ uint32_t ep_clock0(const RolloutArgs &a) { return (uint32_t)a.t0; }

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// SoA accesses go through buffer instructions: a 128-bit resource descriptor (base address) and the
// per-component offset c*N*4 live in scalar registers (SALU arithmetic, issued beside the VALU stream),
// and the lane supplies only its 32-bit byte offset:   buffer_load_dword v, v_off, s[rsrc], s_coff offen.
// With plain pointers hipcc rebuilt a 64-bit per-lane address with VALU ops for every one of the
// 14 loads / 16-31 stores of a step (35 v_lshl_add_u64 per step, 88 -> 59 VGPRs for quad3d after the
// change).  Limits: lane offset 4*N and column offset 4*nS*N must fit 32 bits, hence N <= 2^25 envs
// per handle (rmav_create enforces it).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    // raw buffer (stride 0), bounds check disabled (num_records = 2^32-1; lanes are predicated by
    // li < N), word 3 = 0x00020000: the gfx9/CDNA raw-dword format
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
// same, with the hardware range check armed: stores at byte offsets >= `bytes` are dropped (no per-lane predicate)
__device__ __forceinline__ rsrc_t make_rsrc_bounded(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// read-once data (a caller's action buffer): non-temporal
__device__ __forceinline__ float buf_ld_nt(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 2));
}
__device__ __forceinline__ int32_t buf_ld_i32(rsrc_t r, uint32_t voff, uint32_t soff) {
    return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st_i32(rsrc_t r, uint32_t voff, uint32_t soff, int32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, r, voff, soff, 0);
}
// EnvRec accesses: the lane's record sits at byte offset 16 * env (N <= 2^25: fits 32 bits).  b96 = {sbd, reset_cnt, ep_start},
// b64 = {sbd, reset_cnt}; last_len (offset 12) is written on its own by the kernels that keep the rest in registers.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t rec_ld4(rsrc_t r, uint32_t env) { return __builtin_amdgcn_raw_buffer_load_b128(r, env * 16u, 0, 0); }
__device__ __forceinline__ u32x3_t rec_ld3(rsrc_t r, uint32_t env) { return __builtin_amdgcn_raw_buffer_load_b96(r, env * 16u, 0, 0); }
__device__ __forceinline__ u32x2_t rec_ld2(rsrc_t r, uint32_t env) { return __builtin_amdgcn_raw_buffer_load_b64(r, env * 16u, 0, 0); }
__device__ __forceinline__ void rec_st4(rsrc_t r, uint32_t env, u32x4_t v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, env * 16u, 0, 0); }
__device__ __forceinline__ void rec_st3(rsrc_t r, uint32_t env, u32x3_t v) { __builtin_amdgcn_raw_buffer_store_b96(v, r, env * 16u, 0, 0); }
__device__ __forceinline__ void rec_st2(rsrc_t r, uint32_t env, u32x2_t v) { __builtin_amdgcn_raw_buffer_store_b64(v, r, env * 16u, 0, 0); }
// one field (word w of the record) on its own: for kernels at their register limit, where a 2- / 3-register tuple would spill
__device__ __forceinline__ uint32_t rec_ld_word(rsrc_t r, uint32_t env, int w) { return __builtin_amdgcn_raw_buffer_load_b32(r, env * 16u, 4 * w, 0); }
__device__ __forceinline__ void rec_st_word(rsrc_t r, uint32_t env, int w, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, env * 16u, 4 * w, 0); }
__device__ __forceinline__ void rec_st_last_len(rsrc_t r, uint32_t env, int32_t v) { __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, r, env * 16u, 12, 0); }

// Cache policy of the SoA trajectory stores - a template parameter because the policy bits are instruction
// immediates (a wave-uniform runtime switch around three copies of the stores measured 9 % SLOWER than no
// choice at all: code size and SGPR pressure in the step loop).  Measured, 64-step quadrotor3d rollouts:
//   ST_WRITE_THROUGH (sc0 sc1): the line leaves the L2 at once.  Best when one launch's trajectory is small
//       (65 536 envs: 65.3 -> 62.5 us): nothing is left dirty for the end-of-kernel L2 write-back.
//   ST_STREAM (nt): best when one launch writes more than the 256 MB Infinity Cache absorbs
//       (262 144 envs: 232 -> 206 us, 1 M envs: 892 -> 772 us).
//   ST_DEFAULT in between and for single-step launches.
//   ST_AOS_LDS: big launches that want the obs trajectory as [T][N][nS] (gym / torch batch-major tensors).  A
//       lane storing its own nS floats touches every 128-byte line of the wavefront's 64 x nS x 4 byte span
//       once per component (40 / 64 byte lane stride: 1 M envs 21 % (quadrotor3d) / 32 % (slung load) slower
//       than SoA).  Here each full wavefront transposes its obs tile through LDS (row stride nS | 1 words:
//       conflict-free writes) and emits it as nS fully coalesced 256-byte `nt` stores.
enum : int { ST_DEFAULT = 0, ST_WRITE_THROUGH = 1, ST_STREAM = 2, ST_AOS_LDS = 3 };
template <int ST> struct StoreAux {
    static constexpr int value = (ST == ST_WRITE_THROUGH) ? 17 : ((ST == ST_STREAM || ST == ST_AOS_LDS) ? 2 : 0);
};
// LDS words per wavefront of the ST_AOS_LDS obs tile
template <int NS> struct AosTile { static constexpr int STRIDE = NS | 1, WORDS = 64 * STRIDE; };

template <int AUX>
__device__ __forceinline__ void buf_st_aux(rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, AUX);
}

// Wide drain of the hand-over tiles, built and measured, NOT the default: the memory wavefront
// reads 4 consecutive envs of a column with one ds_read_b128 and stores them with one buffer_store_dwordx4 - lane
// (cg = lane >> 4, eq = lane & 15) handles envs 4 eq .. 4 eq + 3 of column 4 j + cg - so a 64-env step of quadrotor3d
// leaves in 10 memory instructions instead of 28.  Measured (profiles/r02/wide_drain_ab.md, cold buffers, three
// interleaved repetitions): random-action rollouts 0-8 % slower, controller-driven ones 0-4 % faster; and the memory
// system absorbs 16-byte and 4-byte stores at the same rate (tools/micro/store_patterns.hip).  -DRMAV_WIDE_DRAIN=1
// builds it; both drains pass the parity suite bit for bit.

#ifndef RMAV_WIDE_DRAIN
#define RMAV_WIDE_DRAIN 0
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int AUX>
__device__ __forceinline__ void buf_st4_aux(rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {
    u32x4_t d;
    d[0] = __builtin_bit_cast(uint32_t, v.x);
    d[1] = __builtin_bit_cast(uint32_t, v.y);
    d[2] = __builtin_bit_cast(uint32_t, v.z);
    d[3] = __builtin_bit_cast(uint32_t, v.w);
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, AUX);
}
// C feature-major columns of a tile (column c at words [64 c, 64 c + 64)) -> C SoA columns of `r` (pitch `col` bytes),
// full 64-env wavefront starting at byte offset `wave_off` of each column
template <int AUX, int C>
__device__ __forceinline__ void wide_cols(const float *tile, rsrc_t r, uint32_t wave_off, uint32_t col, uint32_t lane) {
    const uint32_t cg = lane >> 4, eq = lane & 15u;
#pragma unroll
    for (int j = 0; 4 * j < C; ++j) {
        if (4 * j + 3 < C || 4 * j + (int)cg < C) {
            const float4 v = *reinterpret_cast<const float4 *>(tile + (4 * j + cg) * 64u + 4u * eq);
            buf_st4_aux<AUX>(r, wave_off + 16u * eq + cg * col, (uint32_t)(4 * j) * col, v);
        }
    }
}

// FIXED: the launch's options are the usual ones - feature-major trajectories, episode statistics tracked, auto-reset - and known
// at compile time.  The step loop tests each of them with a scalar compare + branch per env-step otherwise (loop-invariant, but the
// compiler does not unswitch a loop of this size), and where a lone wavefront's issue rate IS the step time (one pair per SIMD:
// ~5 cycles per instruction of any kind, profiles/r04/issue_rate.md) those ~10 instructions are 4-9 % of the launch
// (65 536 envs: quadrotor3d 41.8 -> 40.2 us, quadrotor2d-slungload 42.3 -> 38.9, controller-driven quadrotor3d 52.4 -> 47.5).
// Instantiated for the two-wavefront kernels with their default store policy; every other combination takes the runtime flags.
template <int K, int MODE, int ST = ST_DEFAULT, bool FIXED = false>
__global__ __launch_bounds__((rollout_threads_max<K, MODE>())) void k_rollout(const RolloutArgs a, const typename Env<K>::P p_shared,
                                                    const ParamsT<double> pc_shared) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    constexpr int AUX = StoreAux<ST>::value;
    // ACT_RANDOM_SPLIT: 128-thread workgroups, both wavefronts address the same 64 envs
    constexpr bool SPLIT = is_split(MODE), DRAWS = split_feeds_actions(MODE);   // DRAWS: the hand-over has an action tile
    [[maybe_unused]] constexpr int CH = SplitTile<NS, NA, DRAWS>::CH;   // env-steps per hand-over (split modes)
    // SPLIT: G pairs per workgroup; threads [0, 64 G) are the integrators, [64 G, 128 G) their memory wavefronts
    const uint32_t split_g = SPLIT ? (blockDim.x >> 7) : 1u;
    // (Alternating the two roles between the halves by workgroup index was measured in round 4 - no difference with one pair per
    // workgroup, slower with several, profiles/r04/two_d_kinds.md: which wavefronts share a SIMD is not what bounds them.)
    const bool upper_half = SPLIT && (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= split_g;
    const bool split_helper = upper_half;
    const uint32_t split_local = threadIdx.x - (upper_half ? 64u * split_g : 0u);
    const uint32_t gi = a.slice_first + (SPLIT ? blockIdx.x * (64u * split_g) + split_local : blockIdx.x * blockDim.x + threadIdx.x);
    const uint32_t slice_end = a.slice_count ? a.slice_first + a.slice_count : (uint32_t)a.n;
    // SPLIT: this pair's hand-over tiles
    [[maybe_unused]] float *lds_p = lds_w;
    if constexpr (SPLIT)
        lds_p = lds_w + (uint32_t)__builtin_amdgcn_readfirstlane(split_local >> 6) * SplitTile<NS, NA, DRAWS>::WORDS;
    const int64_t n = a.n;
    // ACT_POLICY_F32M: 32 envs per wavefront, env = column n of the wavefront's tile, simulated by both half-waves
    constexpr bool HALF = (MODE == ACT_POLICY_F32M);
    // The MFMA actor needs all 64 lanes of a wavefront to take part (lane l and lane l ^ 32 exchange state),
    // so in that mode lanes past the end of the batch become clones of env N-1: they compute and store
    // exactly what that env's lane does; only the episode totals must not count them.
    const uint32_t ge = HALF ? ((gi >> 6) << 5) + (gi & 31u) : gi;                      // env this lane works on
    const bool valid = ge < slice_end;
    const uint32_t li = ((is_mfma_policy(MODE) || SPLIT) && !valid) ? slice_end - 1u : ge;   // local env index
    const uint32_t col = (uint32_t)n * 4u;                      // bytes between components of an SoA block
    // trajectory arrays: [T][dim][pitch] feature-major (rmav_rollout_pitched; pitch = N otherwise, and always for batch-major)
    const int64_t tn = a.pitch;
    const uint32_t tcol = (uint32_t)tn * 4u;
    const uint32_t off = li * 4u;                               // this lane's byte offset inside a column
    const bool aos = !FIXED && (a.flags & F_AOS) != 0;
    const bool track = FIXED || (a.flags & F_TRACK) != 0;
    const bool auto_reset = FIXED || (a.flags & F_AUTO_RESET) != 0;

    // (the host rejects n_steps <= 0; the two-wavefront barrier protocol below needs at least one step.  Only there:
    // the same guard in front of the one-wavefront variants made hipcc restructure their step loop and cost them
    // 33-38 VGPRs, i.e. two to three wavefronts per SIMD of occupancy)
    if constexpr (is_split(MODE)) {
        if (a.n_steps <= 0) return;
    }

    // armed statistics exchange: "this launch has begun" - the communicator stream's 2 s bound counts from here.  Not in the
    // controller-driven two-wavefront kernels: their integrators (fp64 controller on the fp64 step, 124 - 128 registers) spill to
    // scratch with one more live value up here, and the same store inside their memory wavefront makes hipcc wrap that wavefront's
    // buffer accesses in waterfall loops (tests/test_resource_usage.py catches both); the host knows (kPublishesStart) and bounds
    // those launches by the waiter's overall limit only.
    if constexpr (publishes_start(MODE)) {
        if (a.xsend && blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(a.xstarted, a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    unsigned int fin_n = 0, fin_len = 0;
    float fin_ret = 0.0f;

    // ACT_RANDOM_SPLIT.  At C2 (65 536 envs = one wavefront per SIMD) the single-wavefront kernel is bound twice
    // over: ONE wavefront issues its ~340 instructions per env-step (VALU + SALU + branches, one stream) at ~5
    // cycles each - 0.72 us per env-step with every output switched off, against 0.34 us per wavefront-step when
    // 16 wavefronts share a SIMD - and then stalls on its own burst of 16 stores, while a store-only kernel
    // with the same pattern drains the trajectory at 7 TB/s (tools/micro/write_ceiling.hip: 36 us per 64 steps).
    // So each 64 envs get a second, "memory" wavefront:
    //   helper  (wave 1): draws the actions (Philox - a third of the instruction stream, independent of the
    //                     state) CH env-steps ahead into an LDS tile, and issues EVERY trajectory
    //                     store: actions directly, obs / reward / done from a second LDS tile the integrator
    //                     fills.  It is the only wavefront that ever waits on the memory pipeline.
    //   integrator (wave 0): state in registers, reads actions from LDS, writes its outputs to LDS.
    // Both tiles are double-buffered; one s_barrier per env-step swaps the halves of both.  The memory wavefront runs TWO
    // steps ahead with the actions, so that the integrator can fetch A(k+1) from LDS while it integrates step k (the LDS
    // round trip behind the barrier used to sit on its critical path: ~150 of ~1100 cycles per step):
    //   helper:     fill A(0), A(1) | B0 |               B0x | fill A(2)                | B1 | fill A(3), drain O(0) | B2 | ... | B(nc) | drain O(nc-1)
    //   integrator:                   B0 | read A(0)   | B0x | read A(1), A(0) -> O(0)  | B1 | read A(2), A(1) -> O(1) | B2 | ... | B(nc)
    // (B0x keeps fill A(2) - same half as A(0) - behind the integrator's first read.)
    // Same Philox counters, same arithmetic: same bits as ACT_RANDOM.  Lanes past the end of the batch are clones
    // of env N-1 (as in the MFMA mode) so that every lane of both wavefronts reaches every barrier.
    // ACT_CONTROLLER_SPLIT is the same arrangement without the draws: the integrator evaluates the controller and
    // hands the action over with its other outputs; the helper only drains.
    if constexpr (SPLIT) {
        using ST_ = SplitTile<NS, NA, DRAWS>;
        // The integrator's dependent chain is the critical path: issue priority 1 for it where that measured faster (same box,
        // two repetitions: quadrotor2d at 65 536 envs 36.9 -> 35.4 us, quadrotor3d at 131 072 envs 92.0 -> 90.5; NOT quadrotor3d
        // with one pair per SIMD: 44.4 -> 44.8, so the 3-D kinds get it only from 6 pairs per workgroup up; priority for the MEMORY
        // wavefront instead: 43.2 -> 45.0).
        if (!split_helper && (K == QUAD2D || K == QUAD2D_SL || split_g >= 6u)) __builtin_amdgcn_s_setprio(1);
        if (split_helper) {
            const uint64_t env_id = a.env_base + (uint64_t)li;
            const uint32_t lane = threadIdx.x & 63u;
            const int32_t T = a.n_steps;
            const int32_t nc = (T + CH - 1) / CH;
            // batch-major obs: output dword 64 q + lane of this wavefront is component e % NS of its env e / NS
            const uint32_t wave_first = __builtin_amdgcn_readfirstlane(gi - lane);
            // (the end of the slice again, as a 64-bit value of its own: derived from the kernel-scope `slice_end` - 32-bit, or
            // widened - hipcc stopped treating the trajectory descriptors below as wave-uniform and wrapped every store of
            // this wavefront in a waterfall loop: 243 v_readfirstlane in the quadrotor3d kernel instead of 3.
            // tests/test_resource_usage.py counts them.)
            const int64_t end64 = a.slice_count ? (int64_t)a.slice_first + (int64_t)a.slice_count : n;
            const uint32_t n_here = (uint64_t)wave_first + 64u <= (uint64_t)end64 ? 64u
                                    : ((uint64_t)wave_first < (uint64_t)end64 ? (uint32_t)(end64 - wave_first) : 0u);
            const uint32_t aos_bytes = n_here * (uint32_t)(NS * 4);   // clones past the end of the batch store nothing
            // wave-uniform: the ragged last wavefront drains dword-wise, and so does a batch whose column pitch or done
            // pointer would misalign the 16-byte / packed-byte stores
            [[maybe_unused]] const bool wide = !aos && n_here == 64u && (tn & 3) == 0 &&
                                               (reinterpret_cast<uintptr_t>(a.done_out) & 3u) == 0;
            uint32_t aos_rd[NS];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const uint32_t e = 64u * q + lane;
                aos_rd[q] = (e / NS) * ST_::OBS_STRIDE + (e % NS);
            }
            // ACT_BUFFER_SPLIT: this wavefront also writes the last-episode statistics when an env's episode ends (the integrator's
            // only stores inside its step loop - it must not have any, see split_self_fetch): the same running sums from the same
            // rewards in the same order, so the same bits
            [[maybe_unused]] float er_m = 0.0f;
            [[maybe_unused]] int32_t el_m = 0;
            if constexpr (split_self_fetch(MODE)) {
                if (track) {
                    er_m = buf_ld(make_rsrc(a.ep_ret), off, 0);
                    el_m = (int32_t)(ep_clock0(a) - rec_ld_word(make_rsrc(a.rec), li, 2));
                    // wait for the two loads HERE, before the first store is in flight: left to the first use inside the drain loop the
                    // compiler's s_waitcnt vmcnt(0) would sit in the loop and wait for every outstanding trajectory store, every step
                    asm volatile("" : "+v"(er_m), "+v"(el_m));
                }
            }
            static_assert(!(RMAV_WIDE_DRAIN && split_self_fetch(MODE)), "the wide drain does not carry the episode statistics");
            auto episode_end = [&](float rw, float dn) {
                if constexpr (split_self_fetch(MODE)) {
                    if (track) {
                        er_m += rw;
                        el_m += 1;
                        if (dn != 0.0f) {
                            buf_st(make_rsrc(a.last_ret), off, 0, er_m);
                            rec_st_last_len(make_rsrc(a.rec), li, el_m);
                            er_m = 0.0f;
                            el_m = 0;
                        }
                    }
                }
            };
            auto fill = [&](int32_t c) {   // actions of chunk c: draw, hand over, write the action trajectory
                float *buf = lds_p + (c & 1) * ST_::A_HALF + lane;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int32_t k = c * CH + j;
                    if (k < T) {
                        float act[NA];
                        random_action<K>(a.seed, env_id, a.t0 + (uint64_t)k, a.act_lo, a.act_hi, act);
#pragma unroll
                        for (int q = 0; q < NA; ++q) buf[(j * NA + q) * 64] = act[q];
                        if (a.act_out) {
                            float *dst_step = a.act_out + (int64_t)k * NA * tn;
                            if (aos) {
                                float *dst = dst_step + (int64_t)li * NA;
#pragma unroll
                                for (int q = 0; q < NA; ++q) dst[q] = act[q];
                            } else if (RMAV_WIDE_DRAIN && wide) {
                                // the tile the integrator will read is also the transposition buffer (LDS executes one
                                // wavefront's accesses in order)
                                wide_cols<AUX, NA>(buf - lane + j * NA * 64, make_rsrc(dst_step), wave_first * 4u, tcol, lane);
                            } else {
                                const rsrc_t ra = make_rsrc(dst_step);
#pragma unroll
                                for (int q = 0; q < NA; ++q) buf_st_aux<AUX>(ra, off, (uint32_t)q * tcol, act[q]);
                            }
                        }
                    }
                }
            };
            auto drain = [&](int32_t c) {  // obs / reward / done of chunk c: LDS -> trajectory
                const float *buf = lds_p + ST_::A_WORDS + (c & 1) * ST_::O_HALF + lane;
                if (RMAV_WIDE_DRAIN && wide) {
                    const float *tile = buf - lane;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int32_t k = c * CH + j;
                        if (k < T) {
                            const float *row = tile + j * ST_::O_ROW;
                            if constexpr (!DRAWS) {
                                if (a.act_out)
                                    wide_cols<AUX, NA>(row + ST_::ACT, make_rsrc(a.act_out + (int64_t)k * NA * tn), wave_first * 4u, tcol, lane);
                            }
                            if (a.obs_out) wide_cols<AUX, NS>(row, make_rsrc(a.obs_out + (int64_t)k * NS * tn), wave_first * 4u, tcol, lane);
                        }
                    }
                    // reward and done of the chunk's CH steps in one instruction each: lanes [16 j, 16 j + 16) take step j
                    const uint32_t sj = lane >> 4, eq = lane & 15u;
                    const int32_t k0 = c * CH;
                    if (sj < (uint32_t)CH && k0 + (int32_t)sj < T) {
                        const float *row = tile + sj * ST_::O_ROW;
                        if (a.rew_out) {
                            const float4 v = *reinterpret_cast<const float4 *>(row + ST_::REW + 4u * eq);
                            buf_st4_aux<AUX>(make_rsrc(a.rew_out + (int64_t)k0 * tn), (wave_first + 4u * eq) * 4u + sj * tcol, 0u, v);
                        }
                        if (a.done_out) {
                            const float4 d = *reinterpret_cast<const float4 *>(row + ST_::DONE + 4u * eq);
                            const uint32_t bytes = (d.x != 0.0f ? 1u : 0u) | (d.y != 0.0f ? 0x100u : 0u) | (d.z != 0.0f ? 0x10000u : 0u) |
                                                   (d.w != 0.0f ? 0x1000000u : 0u);
                            __builtin_amdgcn_raw_buffer_store_b32(bytes, make_rsrc(a.done_out + (int64_t)k0 * tn),
                                                                  wave_first + 4u * eq + sj * (uint32_t)tn, 0u, 0);
                        }
                    }
                    return;
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int32_t k = c * CH + j;
                    if (k < T) {
                        const float *row = buf + j * ST_::O_ROW;
                        if constexpr (!DRAWS) {
                            if (a.act_out) {
                                float *dst_step = a.act_out + (int64_t)k * NA * tn;
                                float av[NA];
#pragma unroll
                                for (int q = 0; q < NA; ++q) av[q] = row[ST_::ACT + q * 64];
                                if (aos) {
                                    float *dst = dst_step + (int64_t)li * NA;
#pragma unroll
                                    for (int q = 0; q < NA; ++q) dst[q] = av[q];
                                } else {
                                    const rsrc_t ra = make_rsrc(dst_step);
#pragma unroll
                                    for (int q = 0; q < NA; ++q) buf_st_aux<AUX>(ra, off, (uint32_t)q * tcol, av[q]);
                                }
                            }
                        }
                        if (a.obs_out) {
                            float *dst_step = a.obs_out + (int64_t)k * NS * tn;
                            float o[NS];
                            if (aos) {
                                // batch-major: the integrator wrote [env][c]; read it back in output order, so the
                                // wavefront's 64 x NS floats leave as NS contiguous 256-byte stores.  The buffer
                                // descriptor covers exactly this wavefront's valid rows: clones are range-checked away.
                                const float *tile = row - lane;
                                const rsrc_t ro = make_rsrc_bounded(dst_step + (int64_t)wave_first * NS, aos_bytes);
#pragma unroll
                                for (int q = 0; q < NS; ++q) o[q] = tile[aos_rd[q]];
#pragma unroll
                                // the per-q term rides in the instruction's immediate offset (256 q <= 3840 < 4096), which the
                                // range check covers; an SGPR soffset is NOT range-checked and would let the clone rows
                                // of a ragged last wavefront land past this wavefront's valid rows
                                for (int q = 0; q < NS; ++q) buf_st_aux<AUX>(ro, lane * 4u + 256u * q, 0u, o[q]);
                            } else {
                                const rsrc_t ro = make_rsrc(dst_step);
#pragma unroll
                                for (int q = 0; q < NS; ++q) o[q] = row[q * 64];
#pragma unroll
                                for (int q = 0; q < NS; ++q) buf_st_aux<AUX>(ro, off, (uint32_t)q * tcol, o[q]);
                            }
                        }
                        if (a.rew_out) buf_st_aux<AUX>(make_rsrc(a.rew_out + (int64_t)k * tn), off, 0, row[ST_::REW]);
                        if (a.done_out)
                            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(row[ST_::DONE] != 0.0f ? 1 : 0),
                                                                 make_rsrc(a.done_out + (int64_t)k * tn), li, 0, 0);
                        episode_end(row[ST_::REW], row[ST_::DONE]);
                    }
                }
            };
            // Lean addressing (the common case: feature-major trajectories below 4 GiB per array).  The generic drain above
            // rebuilds a descriptor per array and step from an advancing 64-bit pointer and tests every optional output with a
            // scalar branch: ~45 scalar / branch instructions per env-step on a wavefront whose ~7 cycles per issued instruction
            // ARE the step time (SQ counters, profiles/r02/sq_counters.md: 76 SALU per 64 envs and step for both wavefronts).
            // Here each array has ONE descriptor for the whole launch - a missing output gets num_records = 0, so the hardware
            // range check drops its stores and no branch is needed - the component offsets q * 4N sit in vector registers
            // (computed once), and a step advances one scalar offset per array.
            if ((a.flags & F_LEAN) != 0) {
                static_assert(CH == 1, "one env-step per hand-over");
                constexpr int NQ = NS > NA ? NS : NA;
                uint32_t voff[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q) voff[q] = off + (uint32_t)q * tcol;
                const rsrc_t rA = a.act_out ? make_rsrc(a.act_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rO = a.obs_out ? make_rsrc(a.obs_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rR = a.rew_out ? make_rsrc(a.rew_out) : make_rsrc_bounded(a.state, 0u);
                const rsrc_t rD = a.done_out ? make_rsrc(a.done_out) : make_rsrc_bounded(a.state, 0u);
                const uint32_t sA = (uint32_t)NA * tcol, sO = (uint32_t)NS * tcol, sR = tcol, sD = (uint32_t)tn;
                auto drain_l = [&](int32_t k) {   // obs / reward / done (and the controller's action) of step k: LDS -> trajectory
                    const float *row = lds_p + ST_::A_WORDS + (k & 1) * ST_::O_HALF + lane;
                    float o[NS];
#pragma unroll
                    for (int q = 0; q < NS; ++q) o[q] = row[q * 64];
                    const float rw = row[ST_::REW], dn = row[ST_::DONE];
                    if constexpr (!DRAWS) {
                        float av[NA];
#pragma unroll
                        for (int q = 0; q < NA; ++q) av[q] = row[ST_::ACT + q * 64];
#pragma unroll
                        for (int q = 0; q < NA; ++q) buf_st_aux<AUX>(rA, voff[q], (uint32_t)k * sA, av[q]);
                    }
#pragma unroll
                    for (int q = 0; q < NS; ++q) buf_st_aux<AUX>(rO, voff[q], (uint32_t)k * sO, o[q]);
                    buf_st_aux<AUX>(rR, off, (uint32_t)k * sR, rw);
                    __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(dn != 0.0f ? 1 : 0), rD, li, (uint32_t)k * sD, 0);
                    episode_end(rw, dn);
                };
                {
                    [[maybe_unused]] uint32_t blk[4] = {0u, 0u, 0u, 0u};   // 2-action kinds: the Philox block of the current pair of steps
                    [[maybe_unused]] bool blk_valid = false;
                    auto fill_l = [&](int32_t k) {   // actions of step k: draw, hand over, write the action trajectory
                        float *buf = lds_p + (k & 1) * ST_::A_HALF + lane;
                        float act[NA];
                        const uint64_t t = a.t0 + (uint64_t)k;
                        if constexpr (action_pairs<K>()) {
                            if ((t & 1u) == 0 || !blk_valid) {   // wave-uniform: one draw serves steps 2 j and 2 j + 1
                                random_block<K>(a.seed, env_id, t, blk);
                                blk_valid = true;
                            }
                            action_from_block<K>(blk, t, a.act_lo, a.act_hi, act);
                        } else {
                            random_action<K>(a.seed, env_id, t, a.act_lo, a.act_hi, act);
                        }
#pragma unroll
                        for (int q = 0; q < NA; ++q) buf[q * 64] = act[q];
#pragma unroll
                        for (int q = 0; q < NA; ++q) buf_st_aux<AUX>(rA, voff[q], (uint32_t)k * sA, act[q]);
                    };
                    if constexpr (DRAWS) {
                        fill_l(0);
                        if (nc >= 2) fill_l(1);
                    }
                    __syncthreads();                                   // B0
                    if constexpr (DRAWS) __syncthreads();              // B0x
                    if (nc >= 2) {
                        if constexpr (DRAWS) {
                            if (nc >= 3) fill_l(2);
                        }
                        __syncthreads();                               // B1
                    }
                    int32_t c = 2;
                    if constexpr (DRAWS && action_pairs<K>()) {
                        // One Philox block holds the actions of steps 2 j and 2 j + 1.  Written as `draw when t is even` inside
                        // fill_l the compiler hoists the draw out of the branch and every step pays its 20 quarter-rate multiplies
                        // (a third of the vector-pipe time of a 2-D pair): here the parity is in the structure of the loop instead.
                        auto fill_half = [&](int32_t k, const uint32_t (&b)[4], uint32_t odd) {
                            float *buf = lds_p + (k & 1) * ST_::A_HALF + lane;
                            float act[NA];
                            action_from_block<K>(b, (uint64_t)odd, a.act_lo, a.act_hi, act);
#pragma unroll
                            for (int q = 0; q < NA; ++q) buf[q * 64] = act[q];
#pragma unroll
                            for (int q = 0; q < NA; ++q) buf_st_aux<AUX>(rA, voff[q], (uint32_t)k * sA, act[q]);
                        };
                        if (c + 1 < nc && ((a.t0 + (uint64_t)(c + 1)) & 1u) != 0) {   // reach an even step
                            fill_l(c + 1);
                            drain_l(c - 2);
                            __syncthreads();                           // Bc
                            ++c;
                        }
                        for (; c + 2 < nc; c += 2) {
                            uint32_t b2[4];
                            random_block<K>(a.seed, env_id, a.t0 + (uint64_t)(c + 1), b2);
                            fill_half(c + 1, b2, 0u);
                            drain_l(c - 2);
                            __syncthreads();                           // Bc
                            fill_half(c + 2, b2, 1u);
                            drain_l(c - 1);
                            __syncthreads();                           // B(c + 1)
                        }
                        blk_valid = false;
                    }
                    for (; c + 1 < nc; ++c) {                          // steady state: no guards, no optional-output branches
                        if constexpr (DRAWS) fill_l(c + 1);
                        drain_l(c - 2);
                        __syncthreads();                               // Bc
                    }
                    if (nc >= 3) {                                     // c = nc - 1: nothing left to draw
                        drain_l(nc - 3);
                        __syncthreads();                               // B(nc - 1)
                    }
                }
                if (nc >= 2) drain_l(nc - 2);
                __syncthreads();                                   // B(nc)
                drain_l(nc - 1);
                return;
            }
            {
                if constexpr (DRAWS) {
                    fill(0);
                    if (nc >= 2) fill(1);
                }
                __syncthreads();                                   // B0
                if constexpr (DRAWS) __syncthreads();              // B0x
                for (int32_t c = 1; c < nc; ++c) {
                    if constexpr (DRAWS) {
                        if (c + 1 < nc) fill(c + 1);
                    }
                    if (c >= 2) drain(c - 2);
                    __syncthreads();                               // Bc
                }
            }
            if (nc >= 2) drain(nc - 2);
            __syncthreads();                                   // B(nc)
            drain(nc - 1);
            return;
        }
    }

    // ACT_POLICY: stage the policy weights into LDS once per launch (every thread of the block helps)
    if constexpr (is_policy(MODE)) {
        constexpr int NW4 = (MODE == ACT_POLICY ? PolicyLayout<NS>::TOTAL : MODE == ACT_POLICY_F32M ? Mfma32Layout::TOTAL : MfmaLayout::TOTAL) / 4;
        const float4 *src = reinterpret_cast<const float4 *>(a.policy_w);
        float4 *dst = reinterpret_cast<float4 *>(lds_w);
        for (int q = threadIdx.x; q < NW4; q += blockDim.x) dst[q] = src[q];
        __syncthreads();
        if constexpr (MODE == ACT_POLICY_BF16) {
            scale_biases_for_tanh();
            __syncthreads();
        }
    }

    if (li < slice_end) {
        const rsrc_t r_state = make_rsrc(a.state);
        float s[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) s[c] = buf_ld(r_state, off, (uint32_t)c * col);
        float er = 0.0f;
        int32_t el = 0;
        // steps_beyond_done and the reset counter ride in registers for the whole launch: loading them
        // on demand (only lanes that terminate need them) would put one or two dependent HBM round
        // trips into every step of every wavefront that has a finishing lane (~57 % of them at the
        // 1.3 %/step termination rate of random actions).  One access of the env's record (EnvRec) brings both, and the
        // episode's start when tracking.
        const rsrc_t r_rec = make_rsrc(a.rec);
        int32_t sb;
        uint32_t rc;
        constexpr bool REC_WORDS = is_policy(MODE);   // the one-wavefront actors sit at their register limit: word accesses, no tuples
        if (track) er = buf_ld(make_rsrc(a.ep_ret), off, 0);
        if constexpr (REC_WORDS) {
            sb = (int32_t)rec_ld_word(r_rec, li, 0);
            rc = rec_ld_word(r_rec, li, 1);
            if (track) el = (int32_t)(ep_clock0(a) - rec_ld_word(r_rec, li, 2));
        } else if (track) {
            const u32x3_t q = rec_ld3(r_rec, li);
            sb = (int32_t)q.x;
            rc = q.y;
            el = (int32_t)(ep_clock0(a) - q.z);
        } else {
            const u32x2_t q = rec_ld2(r_rec, li);
            sb = (int32_t)q.x;
            rc = q.y;
        }
        // (written back unconditionally at the end of the launch: neither copies of the loaded values - two registers of the
        // step loop - nor per-lane dirty masks - four scalar instructions per step - for 8 B per env and LAUNCH)
        const uint64_t env_id = a.env_base + (uint64_t)li;
        // per-env (domain-randomised) constants override the shared kernel arguments for this lane
        typename Env<K>::P pl = p_shared;
        ParamsT<double> pcl = pc_shared;
        if constexpr (K != REINMAV) {
            if (a.pe[0] || a.pe[1] || a.pe[2]) {
                const double m = a.pe[0] ? (double)a.pe[0][li] : (double)pc_shared.mass;
                const double ml = a.pe[1] ? (double)a.pe[1][li] : (double)pc_shared.load_mass;
                const double L = a.pe[2] ? (double)a.pe[2][li] : (double)pc_shared.L;
                override_params(pl, m, ml, L);
                override_params(pcl, m, ml, L);
            }
        }
        const typename Env<K>::P &p = pl;
        const ParamsT<double> &pc = pcl;
        double tenv = 0.0;
        if constexpr (K == REINMAV) tenv = a.env_time[li];

        // Spare reset state.  PMC counters (profiles/r01) show the fused kernel is instruction-issue bound
        // at C2: one wavefront per SIMD, ~4 cycles per instruction, SQ_ACTIVE_INST_ANY = 75 % of
        // SQ_WAVE_CYCLES.  Only ~1.3 % of the lanes terminate per step, but 57 % of the wavefronts
        // contain one, and each of those then executes the three Philox calls of reset_state() for the
        // whole wavefront: ~128 of the ~305 VALU instructions of an average step.  The state an env
        // will be reset to depends only on (seed, env id, reset counter), so for multi-step launches it
        // is drawn ONCE up front (all lanes busy, amortised over the launch) and the in-loop reset
        // becomes a predicated register copy.  A second termination of the same env inside one launch
        // falls back to drawing on demand.  Same counters, same bits either way.
        // (spare_in_lds<K, MODE>: one kernel keeps it in LDS instead, so that it does not occupy NS registers in the step loop)
        constexpr bool SPARE_LDS = spare_in_lds<K, MODE>();
        [[maybe_unused]] float spare[SPARE_LDS ? 1 : NS];
        [[maybe_unused]] float *lds_spare = nullptr;
        if constexpr (SPARE_LDS)
            lds_spare = lds_w + split_g * SplitTile<NS, NA, DRAWS>::WORDS +
                        (uint32_t)__builtin_amdgcn_readfirstlane(split_local >> 6) * SplitTile<NS, NA, DRAWS>::SPARE + (threadIdx.x & 63u);
        bool have_spare = false;
        if (K != REINMAV && auto_reset && a.n_steps >= 8) {   // ReinmavEnv.reset() is a no-op (reinmav_env.py:348-351)
            if constexpr (SPARE_LDS) {
                float sp[NS];
                reset_state<K>(a.seed, env_id, rc, sp);
#pragma unroll
                for (int c = 0; c < NS; ++c) lds_spare[c * 64] = sp[c];   // read back by this lane only: no barrier needed
            } else {
                reset_state<K>(a.seed, env_id, rc, spare);
            }
            have_spare = true;
        }

        // ST_AOS_LDS: this wavefront's transposition tile and, per output dword j of the lane, where in the
        // tile the element lives (output element e = 64 j + lane is component e % NS of the wave's env e / NS)
        [[maybe_unused]] float *tile = nullptr;
        [[maybe_unused]] uint32_t tile_rd[NS];
        [[maybe_unused]] bool full_wave = false;
        [[maybe_unused]] uint32_t wave_obs_base = 0;
        if constexpr (ST == ST_AOS_LDS) {
            const uint32_t lane = threadIdx.x & 63u;
            // readfirstlane: tell the compiler these are wave-uniform (SGPR offsets, no waterfall loops)
            const uint32_t wave_first = __builtin_amdgcn_readfirstlane(gi - lane);
            tile = lds_w + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * AosTile<NS>::WORDS;
            full_wave = wave_first + 64u <= slice_end;
            wave_obs_base = wave_first * (uint32_t)(NS * 4);
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const uint32_t e = 64u * j + lane;
                tile_rd[j] = (e / NS) * AosTile<NS>::STRIDE + (e % NS);
            }
        }

        // uniform cursors into the time-major trajectory buffers, advanced once per step
        const float *act_in = a.act_in;
        float *act_out = (!is_buffer(MODE) && !SPLIT) ? a.act_out : nullptr;   // SPLIT: the memory wavefront writes them
        float *obs_out = a.obs_out;
        float *rew_out = a.rew_out;
        uint8_t *done_out = a.done_out;

        float *logp_out = a.logp_out;
        float *val_out = a.val_out;
        float pol_std[4] = {0.f, 0.f, 0.f, 0.f};
        float pol_logp0 = 0.0f;   // - sum(logstd) - NA/2 * ln(2 pi)
        if constexpr (is_policy(MODE)) {
            float sl = 0.0f;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                const float ls = lds_w[(MODE == ACT_POLICY ? PolicyLayout<NS>::LOGSTD : MODE == ACT_POLICY_F32M ? Mfma32Layout::LOGSTD : MfmaLayout::LOGSTD) + c];
                pol_std[c] = expf(ls);
                sl += ls;
            }
            pol_logp0 = -sl - 0.5f * (float)NA * 1.8378770664093453f;
        }

        // ACT_BUFFER: actions are prefetched one step ahead
        float act_pre[NA];
        auto load_actions = [&](const float *src_step, float (&dst)[NA]) {
            if (aos) {
                const float *src = src_step + (int64_t)li * NA;
#pragma unroll
                for (int c = 0; c < NA; ++c) dst[c] = src[c];
            } else {
                const rsrc_t r = make_rsrc(src_step);
#pragma unroll
                for (int c = 0; c < NA; ++c) dst[c] = buf_ld(r, off, (uint32_t)c * tcol);
            }
        };
        if constexpr (is_buffer(MODE)) load_actions(act_in, act_pre);
        // ACT_BUFFER_SPLIT: the caller's actions in bursts of D steps.  ring_fetch(k0) issues the loads of steps k0 .. k0 + D - 1 into
        // registers; D steps later ring_put() parks them in this lane's column of a private LDS ring (read back by this lane only: no
        // barrier) and the next burst is issued - so every load has D steps (~2 - 5 us) to arrive, and the only vmcnt wait of the
        // loop sees loads alone (this wavefront stores nothing inside the loop).
        [[maybe_unused]] float ring_pre[split_self_fetch(MODE) ? buf_prefetch<NA>() : 1][NA];
        [[maybe_unused]] float *lds_ring = nullptr;
        [[maybe_unused]] auto ring_fetch = [&](int32_t k0) {
            if constexpr (split_self_fetch(MODE)) {
#pragma unroll
                for (int d = 0; d < buf_prefetch<NA>(); ++d)
                    if (k0 + d < a.n_steps) load_actions(a.act_in + (int64_t)(k0 + d) * NA * tn, ring_pre[d]);
            }
        };
        [[maybe_unused]] auto ring_put = [&]() {
            if constexpr (split_self_fetch(MODE)) {
#pragma unroll
                for (int d = 0; d < buf_prefetch<NA>(); ++d)
#pragma unroll
                    for (int c = 0; c < NA; ++c) lds_ring[(d * NA + c) * 64] = ring_pre[d][c];
            }
        };
        if constexpr (split_self_fetch(MODE)) {
            using STB = SplitTile<NS, NA, DRAWS>;
            lds_ring = lds_w + split_g * (STB::WORDS + (spare_in_lds<K, MODE>() ? STB::SPARE : 0)) +
                       (uint32_t)__builtin_amdgcn_readfirstlane(split_local >> 6) * buf_ring_words<NA>() + (threadIdx.x & 63u);
            ring_fetch(0);
            ring_put();
            ring_fetch(buf_prefetch<NA>());
        }
        // two-wavefront modes whose memory wavefront supplies the actions: A(k + 1) is fetched from the hand-over tile while
        // step k is integrated (see the protocol above)
        [[maybe_unused]] float act_nx[NA];
        if constexpr (split_feeds_actions(MODE)) {
            static_assert(CH == 1, "one env-step per hand-over");
            __syncthreads();                                       // B0: A(0) and A(1) are in the tile
            const float *buf = lds_p + (threadIdx.x & 63u);
#pragma unroll
            for (int c = 0; c < NA; ++c) act_nx[c] = buf[c * 64];
            __syncthreads();                                       // B0x (waits for the read above: lgkmcnt(0) precedes s_barrier)
        }

        for (int32_t k = 0; k < a.n_steps; ++k) {
            float act[NA];
            if constexpr (is_mfma_policy(MODE)) {
                float x[16], mean[4], val0, z[4];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = (c < NS) ? s[c] : 0.0f;
                if constexpr (MODE == ACT_POLICY_BF16) policy_forward_mfma(x, mean, val0);
                else policy_forward_mfma32<NS>(x, mean, val0);
                gaussian4(a.seed, env_id, a.t0 + (uint64_t)k, z);
                float q = 0.0f;
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    act[c] = rfma(pol_std[c], z[c], mean[c]);
                    q = rfma(z[c], z[c], q);
                }
                buf_st(make_rsrc(logp_out), off, 0, rfma(-0.5f, q, pol_logp0));
                buf_st(make_rsrc(val_out), off, 0, val0);
                logp_out += n;
                val_out += n;
            } else if constexpr (MODE == ACT_POLICY) {
                using PL = PolicyLayout<NS>;
                XVec<PL::NSP> x;
#pragma unroll
                for (int c = 0; c < PL::NSP; ++c) x.v[c] = (c < NS) ? s[c] : 0.0f;
                const float4 m4 = mlp_forward<NS>(x, 0u);
                const float4 v4 = mlp_forward<NS>(x, (uint32_t)PL::NET);
                const float mean[4] = {m4.x, m4.y, m4.z, m4.w};
                const float val[1] = {v4.x};
                float z[4];
                gaussian4(a.seed, env_id, a.t0 + (uint64_t)k, z);
                float q = 0.0f;
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    act[c] = rfma(pol_std[c], z[c], mean[c]);
                    q = rfma(z[c], z[c], q);
                }
                buf_st(make_rsrc(logp_out), off, 0, rfma(-0.5f, q, pol_logp0));
                buf_st(make_rsrc(val_out), off, 0, val[0]);
                logp_out += n;
                val_out += n;
            } else if constexpr (is_buffer(MODE)) {
                // the action of step k was fetched while step k-1 was integrated (see the prefetch below): with one
                // wavefront per SIMD an action load issued at the top of its own step exposes a full memory
                // round trip per step (1.40 -> 1.26 us per env-step batch at 65 536 envs)
#pragma unroll
                for (int c = 0; c < NA; ++c) act[c] = act_pre[c];
                act_in += (int64_t)NA * tn;
                if (k + 1 < a.n_steps) load_actions(act_in, act_pre);
            } else if constexpr (MODE == ACT_RANDOM) {
                random_action<K>(a.seed, env_id, a.t0 + (uint64_t)k, a.act_lo, a.act_hi, act);
            } else if constexpr (split_feeds_actions(MODE)) {
#pragma unroll
                for (int c = 0; c < NA; ++c) act[c] = act_nx[c];
                // A(k + 1): written before B(k), its half is not rewritten before B(k + 1).  (Past the last step this reads a
                // stale half and the values are never used.)
                const float *buf = lds_p + ((k + 1) & 1) * SplitTile<NS, NA, true>::A_HALF + (threadIdx.x & 63u);
#pragma unroll
                for (int c = 0; c < NA; ++c) act_nx[c] = buf[c * 64];
            } else if constexpr (MODE == ACT_CONTROLLER_SPLIT) {
                if ((k % CH) == 0) __syncthreads();   // B(k / chunk): the output tile swaps halves
                env_control<K>(s, pc, act);
            } else if constexpr (split_self_fetch(MODE)) {
                if ((k % CH) == 0) __syncthreads();   // B(k / chunk), as the controller-driven integrator
                constexpr int D = buf_prefetch<NA>();
                static_assert((D & (D - 1)) == 0, "burst length: a power of two");
                const int32_t slot = k & (D - 1);
                if (slot == 0 && k > 0) {   // wave-uniform.  Burst boundary: the loads issued D steps ago go into the ring, the next D are issued
                    ring_put();
                    ring_fetch(k + D);
                }
                // (read at the top of its own step: fetching it one step ahead, as the random-action integrator does with its tile,
                // measured the same and costs the slung-load integrators the four registers they do not have)
                const float *rb = lds_ring + slot * (NA * 64);
#pragma unroll
                for (int c = 0; c < NA; ++c) act[c] = rb[c * 64];
            } else if constexpr (K == REINMAV) {
#pragma unroll
                for (int c = 0; c < NA; ++c) act[c] = 0.0f;   // the built-in controller runs inside every sub-step
            } else {
                env_control<K>(s, pc, act);
            }

            float dist = 0.0f;
            bool done;
            float r;
            if constexpr (K == REINMAV) {
                float fm0[4];
                Env<K>::step(s, act, MODE == ACT_CONTROLLER, tenv, p, fm0);
                if (MODE == ACT_CONTROLLER) {
#pragma unroll
                    for (int c = 0; c < NA; ++c) act[c] = fm0[c];   // reported action = command of sub-step 0
                }
                done = true;   // reinmav_env.py:110
                r = 90.0f;     // reinmav_env.py:111-116: 100 - 10, every step
            } else {
                Env<K>::step(s, act, p, dist, done);
                // reward / steps_beyond_done machine  (quadrotor3d.py:112-122 and siblings)
                r = -dist;
                if (done) {
                    r = (sb < 0) ? 1.0f : 0.0f;
                    sb = (sb < 0) ? 0 : sb + 1;
                }
            }
            if (act_out) {
                if (aos) {   // 8 / 16 bytes per lane, contiguous across the wavefront: already coalesced
                    float *dst = act_out + (int64_t)li * NA;
#pragma unroll
                    for (int c = 0; c < NA; ++c) {
                        if constexpr (ST == ST_AOS_LDS) __builtin_nontemporal_store(act[c], dst + c);
                        else dst[c] = act[c];
                    }
                } else {
                    const rsrc_t ra = make_rsrc(act_out);
#pragma unroll
                    for (int c = 0; c < NA; ++c) buf_st_aux<AUX>(ra, off, (uint32_t)c * tcol, act[c]);
                }
                act_out += (int64_t)NA * tn;
            }

            // End of an episode (statistics) and auto-reset.  A lane that terminates again in the same launch has no spare reset
            // state left: draw one first, skipped with ONE wave-uniform branch; what remains on the common path is a single predicated
            // copy (the nested form - copy the spare OR draw - cost a dozen exec-mask instructions per step).  The draw costs the
            // wavefront its ~130 instructions whatever the number of lanes in it, so EVERY lane without a spare takes one then (its
            // reset counter already names its next episode): one draw per wavefront serves all the lanes that have used theirs up,
            // instead of one draw per second termination (the 2-D kinds under random actions: an on-demand draw in most steps).
            //
            // Controller-driven and caller-action rollouts (SKIP_QUIET): all of it sits behind ONE more wave-uniform branch, and a
            // step in which no lane of the wavefront finishes (every step of a hovering rollout) skips the ~25 predicated
            // instructions of the episode hand-off and the reset copy: 65 536 envs, controller-driven, same box, quadrotor3d
            // 47.4 -> 45.8 us, quadrotor2d 34.8 -> 31.2, quadrotor3d-slungload 71.5 -> 66.3.  Under random actions most wavefronts
            // have a finishing lane in most steps and the extra branch costs 1-3 %, so there the block stays predicated.  (Two
            // copies of the same statements rather than shared lambdas: the register allocation of the 1024-thread kernels
            // is tight enough to spill with the latter - tests/test_resource_usage.py.)
            constexpr bool SKIP_QUIET = MODE == ACT_CONTROLLER || MODE == ACT_CONTROLLER_SPLIT || is_buffer(MODE) || MODE == ACT_BUFFER_SPLIT;
            if constexpr (SKIP_QUIET) {
                if (track) {
                    er += r;
                    el += 1;
                }
                if (__ballot(done) != 0) {
                    if (track && done) {
                        if constexpr (!split_self_fetch(MODE)) {   // (ACT_BUFFER_SPLIT: the memory wavefront writes them)
                            buf_st(make_rsrc(a.last_ret), off, 0, er);
                            rec_st_last_len(make_rsrc(a.rec), li, el);
                        }
                        if (valid && !(HALF && (threadIdx.x & 32u))) {   // (the second copy of an env does not count)
                            fin_n += 1;
                            fin_len += (unsigned int)el;
                            fin_ret += er;
                        }
                        er = 0.0f;
                        el = 0;
                    }
                    if (K != REINMAV && auto_reset) {
                        const bool rst = done;
                        if (__ballot(rst && !have_spare) != 0) {
                            if (!have_spare) {
                                float sp[NS];
                                reset_state<K>(a.seed, env_id, rc, sp);
#pragma unroll
                                for (int c = 0; c < NS; ++c) {
                                    if constexpr (SPARE_LDS) lds_spare[c * 64] = sp[c];
                                    else spare[c] = sp[c];
                                }
                                have_spare = true;
                            }
                        }
                        if (rst) {
#pragma unroll
                            for (int c = 0; c < NS; ++c) {
                                if constexpr (SPARE_LDS) s[c] = lds_spare[c * 64];
                                else s[c] = spare[c];
                            }
                            have_spare = false;
                            rc += 1;
                        }
                    }
                }
            } else {
                if (track) {
                    er += r;
                    el += 1;
                    if (done) {
                        buf_st(make_rsrc(a.last_ret), off, 0, er);
                        rec_st_last_len(make_rsrc(a.rec), li, el);
                        if (valid && !(HALF && (threadIdx.x & 32u))) {   // (the second copy of an env does not count)
                            fin_n += 1;
                            fin_len += (unsigned int)el;
                            fin_ret += er;
                        }
                        er = 0.0f;
                        el = 0;
                    }
                }
                if (K != REINMAV && auto_reset) {
                    const bool rst = done;
                    if (__ballot(rst && !have_spare) != 0) {
                        if (!have_spare) {
                            float sp[NS];
                            reset_state<K>(a.seed, env_id, rc, sp);
#pragma unroll
                            for (int c = 0; c < NS; ++c) {
                                if constexpr (SPARE_LDS) lds_spare[c * 64] = sp[c];
                                else spare[c] = sp[c];
                            }
                            have_spare = true;
                        }
                    }
                    if (rst) {
#pragma unroll
                        for (int c = 0; c < NS; ++c) {
                            if constexpr (SPARE_LDS) s[c] = lds_spare[c * 64];
                            else s[c] = spare[c];
                        }
                        have_spare = false;
                        rc += 1;
                    }
                }
            }
            if constexpr (SPLIT) {
                // hand obs / reward / done (and the controller's action) to the memory wavefront; it drains this
                // half two barriers later
                using ST_ = SplitTile<NS, NA, DRAWS>;
                float *row = lds_p + ST_::A_WORDS + ((k / CH) & 1) * ST_::O_HALF + (k % CH) * ST_::O_ROW +
                             (threadIdx.x & 63u);
                {   // (handed over also when no obs trajectory was asked for: a uniform branch here costs every step)
                    if (aos) {   // env-major for the batch-major drain
                        float *mine = row + (threadIdx.x & 63u) * (ST_::OBS_STRIDE - 1);
#pragma unroll
                        for (int c = 0; c < NS; ++c) mine[c] = s[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < NS; ++c) row[c * 64] = s[c];
                    }
                }
                row[ST_::REW] = r;
                row[ST_::DONE] = done ? 1.0f : 0.0f;
                if constexpr (!DRAWS) {
#pragma unroll
                    for (int c = 0; c < NA; ++c) row[ST_::ACT + c * 64] = act[c];
                }
                if constexpr (split_feeds_actions(MODE)) __syncthreads();   // B(k + 1): O(k) handed over, A(k + 2) may be written
            } else if (obs_out) {
                if (ST == ST_AOS_LDS && full_wave) {
                    // all 64 lanes are here (full_wave is wave-uniform); LDS executes one wavefront's
                    // instructions in order, the fences only pin the compiler's ordering
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    float *row = tile + (threadIdx.x & 63u) * AosTile<NS>::STRIDE;
#pragma unroll
                    for (int c = 0; c < NS; ++c) row[c] = s[c];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const rsrc_t ro = make_rsrc(obs_out);
                    const uint32_t voff = (threadIdx.x & 63u) * 4u;
#pragma unroll
                    for (int j = 0; j < NS; ++j) buf_st_aux<AUX>(ro, voff, wave_obs_base + 256u * j, tile[tile_rd[j]]);
                } else if (aos) {
                    float *dst = obs_out + (int64_t)li * NS;
#pragma unroll
                    for (int c = 0; c < NS; ++c) dst[c] = s[c];
                } else {
                    const rsrc_t ro = make_rsrc(obs_out);
#pragma unroll
                    for (int c = 0; c < NS; ++c) buf_st_aux<AUX>(ro, off, (uint32_t)c * tcol, s[c]);
                }
                obs_out += (int64_t)NS * tn;
            }
            if (!SPLIT && rew_out) {
                buf_st_aux<AUX>(make_rsrc(rew_out), off, 0, r);
                rew_out += tn;
            }
            if (!SPLIT && done_out) {
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(done ? 1 : 0), make_rsrc(done_out), li, 0, 0);
                done_out += tn;
            }
        }
        if constexpr (MODE == ACT_CONTROLLER_SPLIT || split_self_fetch(MODE)) __syncthreads();   // B(nc): the last step's outputs are in LDS

        if constexpr (is_mfma_policy(MODE)) {   // bootstrap value of the state the rollout ends in
            float x[16], mean[4], val0;
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = (c < NS) ? s[c] : 0.0f;
            if constexpr (MODE == ACT_POLICY_BF16) policy_forward_mfma(x, mean, val0);
            else policy_forward_mfma32<NS>(x, mean, val0);
            buf_st(make_rsrc(val_out), off, 0, val0);
        }
        if constexpr (MODE == ACT_POLICY) {   // bootstrap value of the state the rollout ends in
            using PL = PolicyLayout<NS>;
            XVec<PL::NSP> x;
#pragma unroll
            for (int c = 0; c < PL::NSP; ++c) x.v[c] = (c < NS) ? s[c] : 0.0f;
            const float4 v4 = mlp_forward<NS>(x, (uint32_t)PL::NET);
            buf_st(make_rsrc(val_out), off, 0, v4.x);
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) buf_st(r_state, off, (uint32_t)c * col, s[c]);
        if constexpr (MODE == ACT_BUFFER_CTRL && K != REINMAV) {   // control() of the state this launch leaves behind
            float a2[NA];
            env_control<K>(s, pc, a2);
            if (aos) {
                float *dst = a.ctrl_out + (int64_t)li * NA;
#pragma unroll
                for (int c = 0; c < NA; ++c) dst[c] = a2[c];
            } else {
                const rsrc_t rc2 = make_rsrc(a.ctrl_out);
#pragma unroll
                for (int c = 0; c < NA; ++c) buf_st(rc2, off, (uint32_t)c * col, a2[c]);
            }
        }
        if constexpr (K == REINMAV) a.env_time[li] = tenv;
        if (track) buf_st(make_rsrc(a.ep_ret), off, 0, er);
        if constexpr (is_policy(MODE)) {
            rec_st_word(make_rsrc(a.rec), li, 0, (uint32_t)sb);
            rec_st_word(make_rsrc(a.rec), li, 1, rc);
            if (track) rec_st_word(make_rsrc(a.rec), li, 2, ep_clock0(a) + (uint32_t)a.n_steps - (uint32_t)el);
        } else if (track) {
            rec_st3(make_rsrc(a.rec), li, u32x3_t{(uint32_t)sb, rc, ep_clock0(a) + (uint32_t)a.n_steps - (uint32_t)el});
        } else {
            rec_st2(make_rsrc(a.rec), li, u32x2_t{(uint32_t)sb, rc});
        }
    }

    if (track) {
        // Episode totals: each wavefront owns one slot of a [ceil(N/64)] partials array, so the adds never
        // contend (same-address device atomics cost ~12 ns each: ~600 finishing waves per step made the
        // single-step kernel 20 us slower than its memory time).  The adds are result-less atomics:
        // fire-and-forget at the L2, no load -> add -> store round trip at the tail of the kernel.
        // rmav_episode_totals sums the slots.
        Totals *slot = a.totals + (gi >> 6);
        if (a.n_steps == 1) {
            // single-step launches are latency-bound (~4.5 us): three 6-deep shuffle reductions at the tail of
            // the kernel cost ~0.4 us, while on average fewer than one lane per wavefront finishes an episode
            // - let those lanes add to the wave's slot themselves.
            if (fin_n != 0) {
                atomicAdd(&slot->episodes, (unsigned long long)fin_n);
                atomicAdd(&slot->length_sum, (unsigned long long)fin_len);
                atomicAdd(&slot->return_sum, (double)fin_ret);
            }
        } else if (__ballot(fin_n != 0) != 0) {
            // (the matrix-core actors keep ds_bpermute out of their kernels: rmav_policy_mfma.hpp, xor32)
            const unsigned int wn = is_mfma_policy(MODE) ? wave_sum_x(fin_n) : wave_sum(fin_n);
            const unsigned int wl = is_mfma_policy(MODE) ? wave_sum_x(fin_len) : wave_sum(fin_len);
            const float wr = is_mfma_policy(MODE) ? wave_sum_x(fin_ret) : wave_sum(fin_ret);
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&slot->episodes, (unsigned long long)wn);
                atomicAdd(&slot->length_sum, (unsigned long long)wl);
                atomicAdd(&slot->return_sum, (double)wr);
            }
        }
    }

    if (a.xsend) {   // wave-uniform.  Snapshot for the statistics exchange, then this wavefront's arrival word.
        // The loads see this lane's own stores of the step loop (same address, same lane: program order); the stores are
        // agent-scope (written through to the level the other XCDs' wavefronts and the next kernel read from), and
        // vmcnt(0) = they have been acknowledged there before the arrival word goes out.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (valid && !(HALF && (threadIdx.x & 32u))) {
            const float lr = a.last_ret[li];
            const int32_t ll = a.rec[li].last_len;
            __hip_atomic_store(a.xsend + li, __builtin_bit_cast(int32_t, lr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.xsend + a.xcmax + li, ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // one word per wavefront that owns envs (its lane 0 owns the first of them): 64 envs each, 32 in the fp32-MFMA mode
        if ((threadIdx.x & 63u) == 0 && valid)
            __hip_atomic_store(a.xarrive + (HALF ? (ge >> 5) : (ge >> 6)), a.xseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// reset_state<K> for the lanes of a wavefront that need it, computed cooperatively (all 64 lanes must be active).
// Philox4x32-10 is ~20 quarter-rate 32x32->64 multiplies per call and reset_state needs ceil(nS / 4) calls (the counter
// word c3 selects the block of four components); executed by a whole wavefront for the one or two lanes whose env has
// just terminated, that is ~1 us of a ~4.5 us single-step launch at one wavefront per SIMD.  Here every group of four
// lanes serves ONE terminating lane instead: lane 4g + b draws block b for the g-th terminating lane (env id and reset
// counter of that lane, so the same counters and the same bits), and the terminating lane picks its components up
// with ds_bpermute.  One Philox call per wavefront covers up to 16 terminations (more: the loop goes round again).
template <int K>
__device__ __forceinline__ void reset_state_wave(uint64_t seed, uint64_t env_id_lane0, uint32_t rc, bool need,
                                                 float (&s)[Dims<K>::NS]) {
    constexpr int NS = Dims<K>::NS;
    static_assert(NS <= 16, "four lanes x four components per terminating lane");
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t m = __ballot(need);                      // wave-uniform
    while (m) {
        // source lane of this lane's group: the (lane >> 2)-th set bit of m, if there is one
        uint64_t t = m;
        for (uint32_t q = lane >> 2; q != 0 && t != 0; --q) t &= t - 1;
        const uint32_t src = t ? (uint32_t)__builtin_ctzll(t) : 0u;
        const uint32_t rc_src = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)rc);
        const uint64_t env = env_id_lane0 + src;
        uint32_t r[4];
        philox4x32_10((uint32_t)env, (uint32_t)(env >> 32), rc_src, (1u << 24) | (lane & 3u), (uint32_t)seed,
                      (uint32_t)(seed >> 32), r);
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) u[i] = rfma(1.0f / 8388608.0f, (float)(r[i] >> 8), -1.0f);
        // a needing lane with k set bits of m below it (k < 16) takes component c from lane 4k + c / 4, register c % 4
        const uint32_t k = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        const bool mine = need && ((m >> lane) & 1ull) && k < 16u;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            const float v = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)((4u * (k & 15u) + (uint32_t)(c >> 2)) << 2),
                                                                                  __builtin_bit_cast(int, u[c & 3])));
            if (mine) s[c] = v;
        }
        for (int q = 0; q < 16 && m; ++q) m &= m - 1;   // the 16 lowest set bits are served
    }
}

// One env-step per launch (rmav_step, rmav_step_control, the fused = 0 loop of rmav_rollout with caller actions).
// At BASELINE's 65 536 envs such a launch is latency-bound (a few us against 0.8 us of HBM time), so this is
// k_rollout<K, ACT_BUFFER> at n_steps = 1 re-cut for latency, same arithmetic and same bits:
//   * a small straight-line body (no step loop, action prefetch, spare reset state, store-policy or tile variants):
//     less code to fetch into an instruction cache that is cold at every kernel boundary;
//   * outputs that do not depend on the reset state (reward, done, episode accumulators, last-episode stores,
//     steps_beyond_done) are issued BEFORE the Philox draws of reset_state(), whose ~130 instructions then run
//     under the stores' flight time; only the state / obs stores wait for them;
//   * no atomics: the wavefront's slot of the episode totals is read with scalar loads at the top of the kernel
//     (in flight beside the state loads) and rewritten by one lane with plain stores; the per-wave sums are
//     gathered with v_readlane over the set bits of the `done` ballot (usually one bit) instead of shuffle
//     reductions.  (k_rollout adds to the same slots with atomics; launches are stream-ordered, so they mix.)
//
// LAZY (the default from 786 432 envs; rmav_set_tuning(RMAV_TUNE_STEP_LAZY, 0 | 1) overrides): steps_beyond_done and the reset counter are
// needed only by lanes whose env terminates (~1.3 % per step), so they are loaded under that predicate AFTER `done` is known -
// 8 B per env-step less HBM-side traffic (the launch fetches 1.28 x its algorithmic bytes otherwise, of which these two
// arrays are 0.08), but a dependent memory round trip on the critical path of every wavefront that has a finishing lane
// (57 % of them), in a kernel whose whole duration is ~1.5 round trips above the launch floor.  profiles/r03/step_lazy_ab.md.
// What the first loads of k_step need, as LEADING scalar kernel arguments (16 dwords): with -mllvm -amdgpu-kernarg-preload-count=16
// (Makefile) the command processor puts them into scalar registers when it launches a wavefront, so the state / action loads issue
// without first waiting for an s_load of the argument block - one scalar-cache round trip off the critical path of a launch that
// lasts ~1.5 memory round trips (round 6; tools/micro/launch_floor.hip with and without the flag: 2.90 -> 2.72 us for a 14-load /
// 11-store kernel at 65 536 threads).  Aggregates (RolloutArgs, the constants) cannot be preloaded and follow as before; the fields
// duplicated here are read from these arguments only.
struct StepHot {   // (documentation of the argument order; passed as separate scalars)
    float *state; int64_t n; const float *act_in; int64_t pitch; uint32_t block, flags; float *ep_ret; EnvRec *rec;
};
template <int K, bool CTRL, bool LAZY, int ST>
__device__ __forceinline__ void step_body(const RolloutArgs &a, const uint32_t block_pl, const typename Env<K>::P &p_shared, const ParamsT<double> &pc_shared) {
    static_assert(K != REINMAV, "ReinmavEnv steps go through k_rollout");
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    constexpr int AUX = StoreAux<ST>::value;   // cache policy of the per-env stores (RMAV_TUNE_STEP_STORE; measured in profiles/r04/step_store_policy.md)
    const uint32_t gi = blockIdx.x * block_pl + threadIdx.x;
    const int64_t n = a.n;
    const bool valid = gi < (uint64_t)n;
    // lanes past the end of the batch are clones of env N-1 that store nothing: all 64 lanes of every wavefront
    // stay active, which the cooperative reset needs
    const uint32_t li = valid ? gi : (uint32_t)n - 1u;
    const uint32_t col = (uint32_t)n * 4u, off = li * 4u;
    const uint32_t tcol = (uint32_t)a.pitch * 4u;   // trajectory columns (actions in, obs out): rmav_rollout_pitched
    const bool aos = (a.flags & F_AOS) != 0;
    const bool track = (a.flags & F_TRACK) != 0;
    const bool auto_reset = (a.flags & F_AUTO_RESET) != 0;

    // This wavefront's slot of the episode totals is read at the top of the kernel and rewritten by one lane at the end (no atomics).
    // Through the scalar cache the read shares lgkmcnt with the late kernel-argument loads, so the `s_waitcnt lgkmcnt(0)` in front of
    // the bookkeeping loads also waits for that memory round trip; as a VECTOR load issued behind all other loads it does not.
    // Same-box A/B (profiles/r05/step_totals_ab.md): the vector form is 2.5 - 6 % faster for the 3-D kinds up to 262 144 envs (65 536:
    // 4.32 - 4.47 -> 4.18 - 4.24 us) and 1 - 3.5 % slower for the 2-D kinds and the lazy kernel of the big batches - so it is chosen by kind.
    constexpr bool kVectorTotals = !LAZY && (K == QUAD3D || K == QUAD3D_SL);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(gi >> 6);
    Totals tot = {0ull, 0.0, 0ull};
    if constexpr (!kVectorTotals) {
        if (track) tot = a.totals[wave];
    }

    const rsrc_t r_state = make_rsrc(a.state);
    float s[NS], act[NA];
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] = buf_ld(r_state, off, (uint32_t)c * col);
    if (aos) {
        // batch-major actions [N][nA] (what a policy network / the VecEnv hands over): ONE 8- / 16-byte load per lane when the
        // buffer is aligned for it (torch tensors are), instead of nA dword loads that each touch all of the wavefront's lines
        const float *src = a.act_in + (int64_t)li * NA;
        if ((reinterpret_cast<uintptr_t>(a.act_in) & (NA * 4 - 1)) == 0) {   // wave-uniform (NA is 2 or 4)
            if constexpr (NA == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src);
                act[0] = v.x; act[1] = v.y; act[2] = v.z; act[3] = v.w;
            } else {
                const float2 v = *reinterpret_cast<const float2 *>(src);
                act[0] = v.x; act[1] = v.y;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NA; ++c) act[c] = src[c];
        }
    } else {
        const rsrc_t r = make_rsrc(a.act_in);
#pragma unroll
        for (int c = 0; c < NA; ++c) act[c] = LAZY ? buf_ld_nt(r, off, (uint32_t)c * tcol) : buf_ld(r, off, (uint32_t)c * tcol);   // big batches: read-once, non-temporal (1 048 576 envs -2 %)
    }
    float er = 0.0f;
    if (track) er = buf_ld(make_rsrc(a.ep_ret), off, 0);
    // needed only by lanes whose env terminates in this step: steps_beyond_done (the terminal reward), the reset counter
    // (the Philox counter of the fresh state) and the episode's start (its length)
    // - ONE 16-byte record per env (EnvRec): one b128 load, and one b128 store when the episode ends
    int32_t sb = -1;
    uint32_t rc = 0, es = 0;
    int32_t ll = 0;   // the record's last_len: rewritten with the record (kept when the handle does not track episodes)
    if constexpr (!LAZY) {
        const u32x4_t q = rec_ld4(make_rsrc(a.rec), li);
        sb = (int32_t)q.x;
        rc = q.y;
        es = q.z;
        ll = (int32_t)q.w;
    }
    if constexpr (kVectorTotals) if (track) {
        const Totals *tp = a.totals + (gi >> 6);
        asm volatile("" : "+v"(tp));          // keep it a vector address (vmcnt, in order), not an s_load
        tot = *tp;
    }
    typename Env<K>::P pl = p_shared;
    ParamsT<double> pcl = pc_shared;
    if (a.pe[0] || a.pe[1] || a.pe[2]) {
        const double m = a.pe[0] ? (double)a.pe[0][li] : (double)pc_shared.mass;
        const double ml = a.pe[1] ? (double)a.pe[1][li] : (double)pc_shared.load_mass;
        const double L = a.pe[2] ? (double)a.pe[2][li] : (double)pc_shared.L;
        override_params(pl, m, ml, L);
        override_params(pcl, m, ml, L);
    }

    float dist = 0.0f;
    bool done;
    Env<K>::step(s, act, pl, dist, done);
    // Where and how the step's outputs are stored (same values on both paths): reward, done, running return
    auto store_scalars = [&](float r_, bool done_, float er_) {
        if (a.rew_out) buf_st_aux<AUX>(make_rsrc(a.rew_out), off, 0, r_);
        if (a.done_out) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(done_ ? 1 : 0), make_rsrc(a.done_out), li, 0, AUX);
        if (track) buf_st_aux<AUX>(make_rsrc(a.ep_ret), off, 0, er_);
    };
    // ... the state in place, and the obs copy when the caller wants one
    // (Batch-major obs through an LDS transpose - as the fused kernels do for big launches - was built and measured here in
    // round 3: QuadrotorVecEnv.step at 65 536 envs 4.91-5.02 us with it, 4.99-5.01 without; not kept.)
    auto store_state = [&]() {
#pragma unroll
        for (int c = 0; c < NS; ++c) buf_st_aux<AUX>(r_state, off, (uint32_t)c * col, s[c]);
        if (a.obs_out) {
            if (aos) {
                float *dst = a.obs_out + (int64_t)li * NS;
#pragma unroll
                for (int c = 0; c < NS; ++c) dst[c] = s[c];
            } else {
                const rsrc_t ro = make_rsrc(a.obs_out);
#pragma unroll
                for (int c = 0; c < NS; ++c) buf_st_aux<AUX>(ro, off, (uint32_t)c * tcol, s[c]);
            }
        }
    };
    // (Round 6, measured and NOT kept - profiles/r06/step_early_stores_ab.md: storing the outputs of the lanes whose env goes on
    // right here, under the execution mask, ahead of the finishing lanes' record round trip and reset draw, and the finishing
    // lanes' afterwards.  Every store of the 57 % of wavefronts with a finishing lane then leaves as two partial-line pieces:
    // 1 048 576 envs 23.0 -> 25.9 us, 4 194 304 envs 85.1 -> 99.9.)
    if constexpr (LAZY) {
        if (done) {   // only the finishing lanes fetch: ONE 32-byte sector each
            const u32x4_t q = rec_ld4(make_rsrc(a.rec), li);
            sb = (int32_t)q.x;
            rc = q.y;
            es = q.z;
            ll = (int32_t)q.w;
        }
    }
    // reward / steps_beyond_done machine  (quadrotor3d.py:112-122 and siblings)
    float r = -dist;
    if (done) {
        r = (sb < 0) ? 1.0f : 0.0f;
        sb = (sb < 0) ? 0 : sb + 1;
    }
    bool fin = false;
    float fin_ret = 0.0f;
    int32_t fin_len = 0;
    // everything that does not need the reset state goes out first
    if (valid) {
        if (track) {
            er += r;
            if (done) {
                const uint32_t clk = ep_clock0(a) + 1u;       // the episode clock after this step
                const int32_t el = (int32_t)(clk - es);       // steps of the episode that ends here
                buf_st(make_rsrc(a.last_ret), off, 0, er);
                ll = el;
                es = clk;                                     // the next episode starts now
                fin = true;
                fin_ret = er;
                fin_len = el;
                er = 0.0f;
            }
        }
        store_scalars(r, done, er);
        if (done)   // the whole record in one 16-byte store
            rec_st4(make_rsrc(a.rec), li, u32x4_t{(uint32_t)sb, auto_reset ? rc + 1u : rc, es, (uint32_t)ll});
    }
    if (auto_reset)   // wave-uniform; every lane takes part
        reset_state_wave<K>(a.seed, a.env_base + (uint64_t)(gi - (threadIdx.x & 63u)), rc, done && valid, s);
    if (valid) store_state();
    if constexpr (CTRL) {   // control() of the state this launch leaves behind
        float a2[NA];
        env_control<K>(s, pcl, a2);
        if (valid) {
            if (aos) {
                float *dst = a.ctrl_out + (int64_t)li * NA;
#pragma unroll
                for (int c = 0; c < NA; ++c) dst[c] = a2[c];
            } else {
                const rsrc_t rc2 = make_rsrc(a.ctrl_out);
#pragma unroll
                for (int c = 0; c < NA; ++c) buf_st(rc2, off, (uint32_t)c * col, a2[c]);
            }
        }
    }
    if (track) {
        uint64_t m = __ballot(fin);
        if (m != 0) {   // wave-uniform
            uint32_t cnt = 0, len = 0;
            float ret = 0.0f;
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                cnt += 1u;
                len += (uint32_t)__builtin_amdgcn_readlane(fin_len, l);
                ret += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin_ret), l));
            }
            if ((threadIdx.x & 63u) == 0u) {
                Totals *slot = a.totals + wave;
                slot->episodes = tot.episodes + cnt;
                slot->return_sum = tot.return_sum + (double)ret;
                slot->length_sum = tot.length_sum + len;
            }
        }
    }
    if (a.done_flag) {   // wave-uniform; the launch is a single wavefront (the host guarantees it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every lane's output stores have been acknowledged
        if (gi == 0) __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The two entry points of step_body.  k_step: the eager variant (below 786 432 envs, where a launch is latency-bound) with the
// preloaded leading arguments (StepHot).  k_step_big: the lazy variant of the big batches WITHOUT them - there the command processor's
// extra work per wavefront launch (16 384 wavefronts at 1 048 576 envs) costs more than the saved scalar load: same-box A/B
// 21.7 -> 22.6 us with preloading, against 4.25 -> 3.93 us at 65 536 envs and 6.32 -> 5.95 at 262 144 (profiles/r06/step_preload_ab.md).
template <int K, bool CTRL, bool LAZY = false, int ST = ST_DEFAULT>
__global__ __launch_bounds__(kBlock) void k_step(float *state_pl, int64_t n_pl, const float *act_pl, int64_t pitch_pl, uint32_t block_pl, uint32_t flags_pl,
                                                 float *ep_ret_pl, EnvRec *rec_pl, const RolloutArgs a_in, const typename Env<K>::P p_shared,
                                                 const ParamsT<double> pc_shared) {
    // the preloaded copies replace the struct's fields (same values: launch_step_k fills both); every other field is still
    // loaded from the argument block when it is first used
    RolloutArgs a = a_in;
    a.state = state_pl;
    a.n = n_pl;
    a.act_in = act_pl;
    a.pitch = pitch_pl;
    a.flags = flags_pl;
    a.ep_ret = ep_ret_pl;
    a.rec = rec_pl;
    step_body<K, CTRL, LAZY, ST>(a, block_pl, p_shared, pc_shared);
}
template <int K, bool LAZY, int ST>
__global__ __launch_bounds__(kBlock) void k_step_big(const RolloutArgs a, const typename Env<K>::P p_shared, const ParamsT<double> pc_shared) {
    step_body<K, false, LAZY, ST>(a, blockDim.x, p_shared, pc_shared);
}

// reset() of every env
template <int K>
__global__ __launch_bounds__(kBlock) void k_reset(float *state, int64_t n, EnvRec *rec,
                                                  float *ep_ret, uint32_t ep_clock, float *obs_out,
                                                  uint64_t seed, uint64_t env_base, uint32_t flags) {
    constexpr int NS = Dims<K>::NS;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s[NS];
    EnvRec q = rec[i];
    const uint32_t rc = q.reset_cnt;
    if constexpr (K == REINMAV) {   // ReinmavEnv.reset() returns the current state unchanged (reinmav_env.py:348-351)
#pragma unroll
        for (int c = 0; c < NS; ++c) s[c] = state[(int64_t)c * n + i];
    } else {
        reset_state<K>(seed, env_base + (uint64_t)i, rc, s);
#pragma unroll
        for (int c = 0; c < NS; ++c) state[(int64_t)c * n + i] = s[c];
    }
    q.reset_cnt = rc + 1;
    if (flags & F_TRACK) {
        ep_ret[i] = 0.0f;
        q.ep_start = ep_clock;   // running length 0
    }
    rec[i] = q;                  // (steps_beyond_done is NOT cleared: quadrotor3d.py:182-185 does not)
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
                                                    uint32_t flags, const ParamsT<double> pc_shared,
                                                    const float *pe_mass, const float *pe_lmass, const float *pe_L) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ParamsT<double> pc = pc_shared;
    if (pe_mass || pe_lmass || pe_L)
        override_params(pc, pe_mass ? (double)pe_mass[i] : pc_shared.mass, pe_lmass ? (double)pe_lmass[i] : pc_shared.load_mass,
                        pe_L ? (double)pe_L[i] : pc_shared.L);
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

// ReinmavEnv: the built-in controller's command (F, Mx, My, Mz) at the env's current (state, t)
[[maybe_unused]] static __global__ __launch_bounds__(kBlock) void k_control_reinmav(const float *state, const double *env_time, int64_t n,
                                                            float *act_out, uint32_t flags, const ReinmavP p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s[13], fm[4];
#pragma unroll
    for (int c = 0; c < 13; ++c) s[c] = state[(int64_t)c * n + i];
    double R[3][3];
    {
        const double q[4] = {s[6], s[7], s[8], s[9]};
        reinmav_quat2mat(q, R);
    }
    reinmav_controller(p, s, R, env_time[i], fm);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (flags & F_AOS) act_out[i * 4 + c] = (float)fm[c];
        else act_out[(int64_t)c * n + i] = (float)fm[c];
    }
}

// [dim][n] <-> [n][dim]
[[maybe_unused]] static __global__ __launch_bounds__(kBlock) void k_soa_to_aos(const float *src, float *dst, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < dim; ++c) dst[i * dim + c] = src[(int64_t)c * n + i];
}
[[maybe_unused]] static __global__ __launch_bounds__(kBlock) void k_aos_to_soa(const float *src, float *dst, int64_t n, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int c = 0; c < dim; ++c) dst[(int64_t)c * n + i] = src[i * dim + c];
}

}  // namespace rmav
