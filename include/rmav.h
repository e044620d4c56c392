/* rmav.h - C ABI of the MI355X-native batched quadrotor dynamics path (librmav.so).
 *
 * This is the drop-in boundary for reinmav-gym's native environments.  The reference has no FFI
 * (it is pure Python); what a binding replaces is the body of its five native gym.Env classes in
 * gym_reinmav/envs/native/ of the reference repository:
 *
 *   reference interface (file:line)                               entry point here
 *   ---------------------------------------------------------------------------------------------
 *   Quadrotor3D.__init__            quadrotor3d.py:44-74          rmav_create(RMAV_QUAD3D, ...)
 *   Quadrotor3DSlungload.__init__   quadrotor3d_slungload.py:44-80  rmav_create(RMAV_QUAD3D_SL, ...)
 *   Quadrotor2D.__init__            quadrotor2d.py:43-67          rmav_create(RMAV_QUAD2D, ...)
 *   Quadrotor2DSlungload.__init__   quadrotor2d_slungload.py:43-73  rmav_create(RMAV_QUAD2D_SL, ...)
 *   ReinmavEnv.__init__ / .step()   reinmav_env.py:53-84, 99-126  rmav_create(RMAV_REINMAV, ...),
 *                                                                 rmav_rollout(RMAV_ACT_CONTROLLER), rmav_get_time
 *   (hard-coded physics literals in each __init__)                rmav_default_params / rmav_params
 *   .seed(seed)                     quadrotor3d.py:77-79          rmav_seed
 *   .reset()                        quadrotor3d.py:182-185        rmav_reset
 *   .step(action)                   quadrotor3d.py:81-124         rmav_step        (batch of N envs)
 *                                   quadrotor3d_slungload.py:87-167
 *                                   quadrotor2d.py:74-113
 *                                   quadrotor2d_slungload.py:79-154
 *   .control()                      quadrotor3d.py:126-180        rmav_control
 *                                   quadrotor2d.py:115-138
 *   one iteration "action = env.control(); env.step(action)" of the reference's smoke tests
 *                                   test/test_quadrotor2d.py:17-18  rmav_control_step (one launch), or rmav_step_control
 *                                   test/test_quadrotor3d.py:16-17  (step + the NEXT control() in one launch)
 *   .state / .steps_beyond_done     quadrotor3d.py:104,68         rmav_get_state / rmav_set_state,
 *                                                                 rmav_get_sbd / rmav_set_sbd
 *   the test loop "control -> step -> reset on done"              rmav_rollout(RMAV_ACT_CONTROLLER)
 *                                   test/test_quadrotor3d.py:16-22
 *   baselines VecEnv rollouts driven by gym_reinmav/run.py:89,190-211
 *                                                                 rmav_rollout(RMAV_ACT_BUFFER|RANDOM), rmav_rollout_chunked
 *   baselines ppo2 Runner.run(): model.step(obs) + env.step(a)    rmav_rollout_policy (+ rmav_pack_policy)
 *   baselines Monitor episode statistics (info['episode'])        rmav_episode_totals / _buffers
 *   baselines ppo2 Runner.run() advantage pass (GAE lambda)       rmav_gae, rmav_normalize
 *   MPI rank probe / data-parallel workers gym_reinmav/run.py:18-21,177-182
 *                                                                 rmav_comm_* + rmav_allgather_stats (RCCL over xGMI)
 *
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns an rmav_status (0 = ok, < 0 = error) unless noted;
 *     rmav_last_error() returns a thread-local message for the last failure; nothing throws or
 *     aborts across this boundary.
 *   - a handle owns the device-resident env state (fp32, struct-of-arrays) of N independent envs on
 *     one GPU and one HIP stream; it is NOT thread-safe; distinct handles are independent.
 *   - `mem` says where every caller-supplied pointer of that call lives: RMAV_HOST (the library
 *     stages - calls that move <= 256 KiB go zero-copy through a pinned, device-mapped block owned by the
 *     handle: one launch + one synchronise (single-step calls of <= 64 envs wait on a pinned completion word the kernel
 *     writes instead), which is what the gym-shaped single env uses; bulk calls go
 *     through device scratch - and synchronises before returning) or RMAV_DEVICE (HIP device
 *     pointers, e.g. torch tensor data_ptr(); work is enqueued on the handle's stream and the call
 *     returns without synchronising).
 *   - `layout`: RMAV_SOA = [dim][N] (component-major, the native device layout, coalesced) or
 *     RMAV_AOS = [N][dim] (what gym / a policy network hands over).  Trajectory buffers of
 *     rmav_rollout are time-major: [T][dim][N] (SOA) or [T][N][dim] (AOS).
 *   - there is no CPU implementation behind this ABI: rmav_create fails with RMAV_ERR_NO_DEVICE
 *     when no GPU is visible.
 *
 * RNG streams (Philox4x32-10, key = (seed_lo, seed_hi), counter = (env_lo, env_hi, c2, c3), where
 * env is the GLOBAL env id = env_id_base + local index, so results do not depend on how envs are
 * sharded over GPUs):
 *   reset : c2 = index of this env's reset (0 for the first),  c3 = (1<<24) | j ; block j supplies
 *           state components 4j..4j+3 as  2*u - 1,  u = (x>>8) * 2^-24          (U[-1,1), as
 *           quadrotor3d.py:184 draws every state component)
 *   action: block index b = the handle's global step counter t (4-action kinds) or t >> 1 (2-action kinds, which use
 *           draws 2 (t & 1) and 2 (t & 1) + 1 of the block, i.e. one Philox call per two steps);
 *           c2 = low 32 bits of b, c3 = (2<<24) | (bits 32..47 of b) << 8 ; component i = fma(act_hi-act_lo, u_i, act_lo)
 *   policy noise (rmav_rollout_policy): as "action" with tag 3; (r0,r1) and (r2,r3) -> Box-Muller:
 *           u1 = ((r>>8)+1) * 2^-24, u2 = (r'>>8) * 2^-24, z = sqrt(-2 ln u1) * (cos, sin)(2 pi u2)
 */
#ifndef RMAV_H
#define RMAV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMAV_VERSION 101 /* 0.1.1: rmav_params.g_vec */

typedef struct rmav_env_s *rmav_handle;

enum rmav_kind {
    RMAV_QUAD2D = 0,
    RMAV_QUAD2D_SL = 1,
    RMAV_QUAD3D = 2,
    RMAV_QUAD3D_SL = 3,
    /* ReinmavEnv (reinmav_env.py:51-352, id 'reinmav-v0'): 13-state rigid body [x y z dx dy dz qw qx qy qz p q r]
     * with thrust + body torques -> motor mixing/clamp -> linear and angular acceleration, 50-or-51 Euler
     * sub-steps of 1/5000 s per step, reward 90 and done = 1 on every step, reset() a no-op.  The reference's
     * step() takes no action: that behaviour is RMAV_ACT_CONTROLLER (built-in PD controller on a min-jerk
     * trajectory, evaluated every sub-step).  With RMAV_ACT_BUFFER / RANDOM / POLICY the 4 action components
     * are (F, Mx, My, Mz) held over the step (an extension).  Each env carries its own clock (rmav_get_time). */
    RMAV_REINMAV = 4
};

enum rmav_status {
    RMAV_OK = 0,
    RMAV_ERR_INVALID = -1,   /* bad argument */
    RMAV_ERR_NO_DEVICE = -2, /* no usable GPU */
    RMAV_ERR_HIP = -3,       /* a HIP runtime call failed (message has the HIP error string) */
    RMAV_ERR_ALLOC = -4,     /* device / host allocation failed */
    RMAV_ERR_TIMEOUT = -5    /* a bounded wait expired (rmav_allgather_stats_wait, an armed exchange whose launch never completed) */
};

enum rmav_mem { RMAV_HOST = 0, RMAV_DEVICE = 1 };
enum rmav_layout { RMAV_SOA = 0, RMAV_AOS = 1 };
enum rmav_action_mode {
    RMAV_ACT_BUFFER = 0,    /* actions read from a caller buffer */
    RMAV_ACT_RANDOM = 1,    /* uniform in [act_lo, act_hi) from the counter RNG, generated in-kernel */
    RMAV_ACT_CONTROLLER = 2, /* the reference's geometric controller, evaluated in-kernel */
    RMAV_ACT_POLICY = 3,     /* Gaussian MLP policy evaluated in-kernel, fp32 (rmav_rollout_policy only) */
    RMAV_ACT_POLICY_BF16 = 4 /* the same policy on the matrix cores: bf16 operands, fp32 accumulate */
};
enum rmav_integrator { RMAV_INT_EULER = 0, RMAV_INT_RK4 = 1 };
enum rmav_policy_precision {
    RMAV_POLICY_FP32 = 0,       /* fp32 FMAs on the vector ALU */
    RMAV_POLICY_BF16_MFMA = 1,  /* bf16 operands, fp32 accumulate on the matrix cores */
    RMAV_POLICY_FP32_MFMA = 2,  /* fp32 operands and accumulate on the fp32-input matrix instructions: same precision
                                   class as RMAV_POLICY_FP32 (only the summation order differs), ~2x its speed */
    RMAV_POLICY_F16_MFMA = 3,   /* f16 operands (11-bit mantissa), fp32 accumulate, tanh folded into the next layer's weights
                                   (csrc/rmav_policy_pair.hpp): ~8x closer to the fp32 policy than bf16 and faster.
                                   Weight buffer: rmav_pack_policy_f16 */
    RMAV_POLICY_F16_SHARED = 4  /* a DIFFERENT architecture, same arithmetic as RMAV_POLICY_F16_MFMA: ONE 2x64 tanh trunk with a mean head
                                   and a scalar value head on its latent - baselines' value_network = 'shared', what ppo2 builds for an env
                                   type without a defaults entry (the native envs of gym_reinmav: env_type 'native'); the other precisions
                                   evaluate a policy net and a separate value net (value_network = 'copy', baselines' MuJoCo default).
                                   Weight buffer: rmav_policy_weight_count_shared() floats = ONE net of the bf16 fragment layout with
                                   output rows 0..3 = the mean head, row 4 = the value head, then logstd [4]; built by rmav_pack_policy_f16 */
};

/* rmav_create flags */
#define RMAV_F_AUTO_RESET 1u     /* VecEnv semantics: a done env is reset inside step; the returned
                                    obs is the post-reset obs (baselines DummyVecEnv.step_wait) */
#define RMAV_F_TRACK_EPISODES 2u /* keep per-env episode return/length (baselines Monitor) */

/* Physics and controller constants; defaults are the literals in each reference __init__
 * (rmav_default_params).  Doubles, so that e.g. dt is the same 0.01 the reference uses. */
typedef struct rmav_params {
    double mass;          /* quadrotor3d.py:45 */
    double load_mass;     /* quadrotor3d_slungload.py:46 */
    double dt;            /* quadrotor3d.py:46 */
    double g;             /* RMAV_REINMAV: the scalar self.gravity (reinmav_env.py:58).  Quadrotor kinds: not read (see g_vec);
                             rmav_default_params fills 9.8 = |g_vec| for information */
    double tether_length; /* quadrotor3d_slungload.py:58 / quadrotor2d_slungload.py:53 */
    double pos_limit;     /* episode ends when |pos| > pos_limit (which body: see DESIGN.md) */
    double vel_limit;     /* ... or |vel| > vel_limit */
    double thrust_scale;  /* quadrotor2d.py:75 (10 for quad2d, 1 otherwise) */
    int32_t clamp_thrust; /* quadrotor2d.py:76-77 (1 for quad2d) */
    int32_t integrator;   /* RMAV_REINMAV only: RMAV_INT_EULER (0, the reference: reinmav_env.py:90-98) or
                             RMAV_INT_RK4 (1: classical Runge-Kutta over the same sub-step grid, command held over
                             each sub-step; an option the reference does not have) */
    double ref_pos[3];    /* controller set-point  quadrotor3d.py:51 */
    double ref_vel[3];    /* quadrotor3d.py:52 */
    double kp, kv, tau;   /* controller gains  quadrotor3d.py:143-145, quadrotor2d.py:116-118 */
    double act_lo, act_hi; /* action Box bounds (quadrotor3d.py:70 etc.); used by RMAV_ACT_RANDOM only */
    double g_vec[3];      /* quadrotor kinds: the gravity VECTOR self.g, added component-wise by step() and subtracted by the 3-D
                             control() - quadrotor3d.py:47,96-99,162: (0, 0, -9.8); 2-D kinds use [0..1] - quadrotor2d.py:46,88:
                             (0, -9.8) - and their control() keeps the reference's literal (0, 9.8) (quadrotor2d.py:130) */
} rmav_params;

typedef struct rmav_ep_totals {
    uint64_t episodes;   /* finished episodes since creation / last clear */
    double return_sum;   /* sum of their returns */
    uint64_t length_sum; /* sum of their lengths */
} rmav_ep_totals;

/* ---- library-level ------------------------------------------------------------------------- */
int rmav_version(void);              /* returns RMAV_VERSION */
const char *rmav_last_error(void);   /* thread-local, never NULL */
int rmav_device_count(void);         /* number of visible GPUs, 0 if none (never negative) */
int rmav_state_dim(int kind);        /* 5, 9, 10, 16, 13; -1 for a bad kind */
int rmav_action_dim(int kind);       /* 2, 2, 4, 4, 4 */
int rmav_algorithmic_bytes(int kind); /* bytes per env-step of SURVEY.md 8(d): 53, 85, 101, 149, 125 */
/* reading_2d selects how the unparsable quadrotor2d.py:95-98 is read: 'B' -> |p|>3 or |v|>2
 * (default when 0 is passed), 'A' -> |p|>3 or |v|>10.  Ignored for other kinds. */
int rmav_default_params(int kind, int reading_2d, rmav_params *out);

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Creates n_envs envs of `kind` on GPU `device`; env i has global id env_id_base + i.  All envs are
 * reset once (reset index 0), like the reference constructors (quadrotor3d.py:73-74).
 * params may be NULL (defaults).  hip_stream may be NULL (the handle creates its own non-blocking
 * stream) or a hipStream_t the caller owns (e.g. torch.cuda.current_stream().cuda_stream; pass
 * hipStreamLegacy == (void*)1 to name the legacy default stream, whose handle is 0). */
int rmav_create(rmav_handle *out, int kind, int64_t n_envs, int device, uint64_t seed,
                uint64_t env_id_base, uint32_t flags, const rmav_params *params, void *hip_stream);
int rmav_destroy(rmav_handle h);
/* Re-keys the RNG and rewinds the reset / step counters (gym Env.seed).  Does not touch state. */
int rmav_seed(rmav_handle h, uint64_t seed);
int rmav_get_params(rmav_handle h, rmav_params *out);
int rmav_set_params(rmav_handle h, const rmav_params *in);
/* Rebind the handle to another HIP stream the caller owns (NULL: back to a stream owned by the
 * handle).  Pending work on the old stream is waited for first.  Lets a caller capture rmav_step /
 * rmav_rollout launches (mem = RMAV_DEVICE: no allocation, no synchronisation) into a hipGraph on its
 * capture stream. */
int rmav_set_stream(rmav_handle h, void *hip_stream);
/* Per-env (domain-randomised) physical constants.  The reference hard-codes one value per class
 * (quadrotor3d_slungload.py:45-59); RL users randomise them per env.  values: N floats (one per env)
 * or NULL to go back to the shared value of rmav_params.  Quadrotor kinds only. */
enum rmav_env_param { RMAV_PARAM_MASS = 0, RMAV_PARAM_LOAD_MASS = 1, RMAV_PARAM_TETHER_LENGTH = 2 };
int rmav_set_env_param(rmav_handle h, int which, const float *values, int mem);
/* Explicit, per-handle overrides of the launch heuristics (DESIGN.md section 4 states the automatic rules and the
 * measurements behind them).  -1 = automatic (the default for every key).  Results never depend on these: every
 * variant produces the same bits (tests/test_gpu_parity.py::test_rollout_bits_do_not_depend_on_kernel_variant). */
enum rmav_tuning_key {
    RMAV_TUNE_SPLIT = 0,           /* fused rollouts: 0 = one wavefront per 64 envs, 1 = integrator + memory wavefront pairs */
    RMAV_TUNE_SLICE = 1,           /* batches beyond the two-wavefront capacity: 0 = one launch, 1 = one launch per slice */
    RMAV_TUNE_STORE_POLICY = 2,    /* trajectory stores: 0 write-back, 1 write-through, 2 non-temporal, 3 LDS-transposed AoS */
    RMAV_TUNE_SPLIT_GROUP = 3,     /* (integrator, memory wavefront) pairs per workgroup, 1 .. 8 */
    RMAV_TUNE_BLOCK = 4,           /* workgroup size of the one-wavefront kernels: 64 | 128 | 256 */
    RMAV_TUNE_STEP_KERNEL = 5,     /* 0: single-step calls use the rollout kernel at n_steps = 1 instead of k_step */
    RMAV_TUNE_SPLIT_MIN_STEPS = 6, /* shortest fused launch that may use the two-wavefront kernel (default 2) */
    RMAV_TUNE_LEAN = 7,            /* 0: the two-wavefront kernel's memory wavefront uses the generic (pointer-advancing) drain */
    RMAV_TUNE_STEP_LAZY = 8,       /* 1: k_step loads steps_beyond_done / reset counters only in lanes whose env terminates */
    RMAV_TUNE_SLICE_ENVS = 9,      /* E >= 64: fused rollouts as two-wavefront launches over slices of at most E envs */
    RMAV_TUNE_HOST_FLAG = 10,      /* 0: host-pointer single-wavefront steps wait with hipStreamSynchronize instead of the pinned completion word */
    RMAV_TUNE_POLICY_PAIR = 11,    /* RMAV_POLICY_BF16_MFMA: 0 = one wavefront per 64 envs (round 3's kernel) instead of the (actor, critic) pair */
    RMAV_TUNE_PAIR_GROUP = 12,     /* (actor, critic) wavefront pairs per workgroup of the matrix-core actors, 1 .. 4 */
    RMAV_TUNE_STEP_STORE = 13,     /* cache policy of k_step's per-env stores: 0 write-back (default), 1 write-through, 2 non-temporal */
    RMAV_TUNE_ROLE_SWAP = 14,      /* two-wavefront kernels: 1 + s = alternate which half of a workgroup integrates by bit s of the workgroup index */
    RMAV_TUNE_FIXED_FLAGS = 15,    /* two-wavefront kernels: 0 = never take the variant with the usual launch options compiled in (bits are the same) */
    RMAV_TUNE_COUNT = 16
};
int rmav_set_tuning(rmav_handle h, int key, int value);
int rmav_get_tuning(rmav_handle h, int key, int *value_out);
int64_t rmav_num_envs(rmav_handle h); /* < 0 on a bad handle */
int rmav_sync(rmav_handle h);         /* waits for everything enqueued on the handle's stream */

/* ---- the hot path -------------------------------------------------------------------------- */
/* Reset every env (fresh U[-1,1) state); steps_beyond_done is NOT cleared (the reference's reset
 * does not).  obs_out (nS*N floats) may be NULL. */
int rmav_reset(rmav_handle h, float *obs_out, int mem, int layout);

/* One step of all N envs.  actions: nA*N floats.  Outputs (each may be NULL): obs_out nS*N floats,
 * rew_out N floats, done_out N bytes (0/1). */
int rmav_step(rmav_handle h, const float *actions, float *obs_out, float *rew_out,
              uint8_t *done_out, int mem, int layout);

/* Geometric controller: actions_out (nA*N floats) <- control(state). */
int rmav_control(rmav_handle h, float *actions_out, int mem, int layout);

/* One iteration of the reference's test loop (test/test_quadrotor3d.py:16-17: action = env.control();
 * env.step(action)) in ONE launch: actions_out (nullable) receives the controller's action, the other outputs
 * are those of rmav_step.  Same results as rmav_control followed by rmav_step. */
int rmav_control_step(rmav_handle h, float *actions_out, float *obs_out, float *rew_out, uint8_t *done_out,
                      int mem, int layout);
/* rmav_step, and the launch ends by evaluating control() on the state it leaves behind: next_actions_out
 * (nA*N floats, required) = what rmav_control would return if called next.  Lets a gym-shaped caller that
 * alternates control() / step(action) pay one launch + one synchronise per iteration (the class caches the
 * action).  Quadrotor kinds only. */
int rmav_step_control(rmav_handle h, const float *actions, float *obs_out, float *rew_out, uint8_t *done_out,
                      float *next_actions_out, int mem, int layout);

/* n_steps steps of all N envs.  fused != 0: one kernel launch with the state held in registers
 * across the steps; fused == 0: n_steps launches of the single-step kernel (same results).
 * actions_in: [n_steps][nA][N] for RMAV_ACT_BUFFER, otherwise ignored (may be NULL).
 * Optional trajectory outputs: actions_out [n_steps][nA][N], obs_out [n_steps][nS][N] (obs after
 * each step, post auto-reset), rew_out [n_steps][N], done_out [n_steps][N]. */
int rmav_rollout(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in,
                 float *actions_out, float *obs_out, float *rew_out, uint8_t *done_out, int mem,
                 int layout, int fused);

/* rmav_rollout with a column pitch: device pointers, feature-major arrays whose feature columns are `pitch` elements apart -
 * actions_in / actions_out [n_steps][nA][pitch], obs_out [n_steps][nS][pitch], rew_out / done_out [n_steps][pitch]; env i is
 * element i of every column, elements [N, pitch) are never written.  Same results as rmav_rollout.  Why: with the plain layout
 * a batch size that is not a multiple of 16 starts every column off a 64-byte line, every wavefront's 256-byte store ends in
 * partial lines, and the library has to fall back to write-back stores (65 599 envs: 67.7 us per 64-step launch against 48.8
 * for 65 600 on one box; 1 048 575: 1 564 against 730).  rmav_trajectory_pitch() = N rounded up to a multiple of 64 (so that
 * the byte-wide done rows start on a line as well) takes the batch size out of it; any pitch >= N is accepted. */
int64_t rmav_trajectory_pitch(rmav_handle h);
int rmav_rollout_pitched(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t pitch, int fused);

/* rmav_rollout (fused) with CHUNK-MAJOR trajectory arrays: the env range is cut into chunks of chunk_envs (a multiple of 64; the last
 * one may be shorter) and chunk c's trajectory is a dense array of its own,
 *   actions_in / actions_out [n_chunks][n_steps][nA][chunk_envs], obs_out [n_chunks][n_steps][nS][chunk_envs], rew_out / done_out
 *   [n_chunks][n_steps][chunk_envs]    (env i = chunk i / chunk_envs, column i % chunk_envs; device pointers; same values as rmav_rollout).
 * Why: the fused rollout of quadrotor3d is bound by its trajectory stores, and a launch over 65 536 envs writing ONE dense region is what
 * the store stream of this GPU likes best; slicing a big batch into such launches does not help as long as every launch writes a strided
 * half / quarter of arrays laid out for the whole batch (a 65 536-env launch into arrays of pitch 131 072: 53 us instead of 40 - 42), with
 * chunk-major arrays it does.  Measured (profiles/r05/chunk_probe.md, one box, 64-step launches): quadrotor3d random actions 131 072 envs
 * 91.2 -> 87.1 us (0.72 -> 0.75 of the 8 TB/s roofline), 262 144 envs 190.2 -> 175.2 (0.69 -> 0.75), 1 048 576 unchanged; the slung-load
 * kinds, controller-driven rollouts and the 2-D kinds are 3 - 10 % SLOWER chunked (their launches are not store-bound), so
 * rmav_chunk_envs() recommends a chunk only for quadrotor3d beyond 65 536 envs and returns N rounded up to a multiple of 64 (one chunk = the
 * plain layout, with that column pitch when N % 64 != 0) otherwise.
 * A learner that flattens (step, env) samples anyway - PPO2 does - consumes the chunks as they are.
 * n_steps >= 2; chunk_envs a multiple of 64, at most the two-wavefront kernel's capacity (131 072; 65 536 for controller-driven slung-load). */
int64_t rmav_chunk_envs(rmav_handle h);
int rmav_rollout_chunked(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t chunk_envs);

/* PPO2-style rollout with the policy inside the kernel (the caller loop of gym_reinmav/run.py:63-68:
 * baselines ppo2 Runner = model.step(obs) -> env.step(actions), network='mlp').  Policy: two 64-unit tanh
 * layers -> Gaussian mean (state-independent log-std), plus a value net of the same shape.  All pointers
 * are DEVICE pointers, layout is SoA, nothing synchronises (capturable in a hipGraph).
 * weights: rmav_policy_weight_count(kind) floats, 16-byte aligned, layout (H = 64, NSP = nS rounded up
 *   to a multiple of 4), policy net then value net, each:
 *     W1 [H][NSP] (row = hidden unit, zero padded) | b1 [H] | W2T [H][H] (W2T[i][j] = W2[j][i]) | b2 [H] |
 *     W3T [H][4] (W3T[j][k] = W3[k][j], zero padded to 4 outputs) | b3 [4]
 *   then logstd [4] (zero padded).
 * Per step t: a = mean(obs_t) + exp(logstd) * z_t with z_t standard normal from the counter RNG
 * (stream tag 3, Box-Muller; see csrc/rmav_policy.hpp), logp_out[t] = log N(a; mean, std),
 * value_out[t] = V(obs_t); value_out[n_steps] = V(obs after the last step) for bootstrapping.
 * actions_out [n_steps][nA][N], obs_out [n_steps][nS][N], rew_out / done_out [n_steps][N] may be NULL.
 * precision = RMAV_POLICY_BF16_MFMA evaluates the same two nets with v_mfma_f32_32x32x16_bf16 (bf16
 * weights and activations, fp32 accumulation; means / values within ~1e-2 of the fp32 policy).  Its weight
 * buffer is rmav_policy_weight_count_bf16() floats of pre-arranged MFMA fragments: per net
 *   A1 [2][64 lanes][8 bf16] | A2 [2][4][64][8] | A3 [4][64][8] | b1 [64] | b2 [64] | b3 [32] (fp32)
 * then logstd [4]; fragment (.., lane = (m = lane & 31, h = lane >> 5), j) holds
 *   layer 1: W1p[32 Mt + m][8 h + j]              (W1 zero-padded to 16 inputs)
 *   layer 2: W2 [32 Mt + m][rowmap(s, h, j)]
 *   layer 3: W3p[m][rowmap(s, h, j)]               (W3 zero-padded to 32 outputs)
 *   rowmap(s, h, j) = 32 (s >> 1) + (r & 3) + 8 (r >> 2) + 4 h,  r = 8 (s & 1) + j
 * (csrc/rmav_policy_mfma.hpp explains why; gym_reinmav_amd.ppo.pack_policy_weights_bf16 builds it). */
int64_t rmav_policy_weight_count(int kind);
int64_t rmav_policy_weight_count_bf16(void);
/* RMAV_POLICY_FP32_MFMA: rmav_policy_weight_count_f32_mfma() floats of pre-arranged A operands of
 * v_mfma_f32_32x32x2_f32, per net (policy, then value):
 *   A1 [2 T][2 sq][64 lanes][4]          lane (m, h), entry j: W1p[32 T + m][2 (4 sq + j) + h]   (W1 zero-padded to 16 inputs)
 *   A2 [2 To][2 Tin][4 rq][64 lanes][4]  lane (m, h), entry j: W2[32 To + m][32 Tin + row(4 rq + j, h)]
 *   W3 [2 h][4 outputs][32]              entry 16 Tin + r:      W3p[o][32 Tin + row(r, h)]        (W3 zero-padded to 4 outputs)
 *   b1 [64] | b2 [64] | b3 [4]
 * then logstd [4];  row(r, h) = (r & 3) + 8 (r >> 2) + 4 h  (csrc/rmav_policy_mfma32.hpp explains why;
 * gym_reinmav_amd.ppo.pack_policy_weights_f32_mfma builds it). */
int64_t rmav_policy_weight_count_f32_mfma(void);
int64_t rmav_policy_weight_count_shared(void);   /* RMAV_POLICY_F16_SHARED */
/* Builds such a weight buffer on the device in ONE launch on the handle's stream: with `flat` = the concatenation of the
 * n_params (<= 16) parameter tensors `params[k]` (DEVICE pointers in a HOST array; sizes[k] elements each) followed by zeros,
 * weights_out[i] = flat[idx_lo[i]] when idx_hi[i] < 0, else the two bf16 roundings of flat[idx_lo[i]] (low half) and
 * flat[idx_hi[i]] (high half) in one 32-bit word.  idx_lo / idx_hi: int32 [n_out] on the DEVICE - the fixed permutation of a
 * layout above (gym_reinmav_amd.ppo._PolicyPacker builds them once).  Replaces the chain of small tensor operations a
 * learner would otherwise run before every rollout (baselines: model.step reads the live variables; here the actor's copy
 * is re-derived from the learner's parameters). */
int rmav_pack_policy(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                     const int32_t *idx_hi, int64_t n_out, float *weights_out);
/* RMAV_POLICY_F16_MFMA: the bf16 layout above with f16 pairs in the fragment words (same idx_lo / idx_hi maps, n_out =
 * rmav_policy_weight_count_bf16()), and the fragments of layers 2 and 3 pre-multiplied (in fp32, before the one rounding to
 * f16) by -2 k and -2, k = 2 log2(e): the kernel hands r = 1 / (1 + e^(2z)) = (1 - tanh z) / 2 to the next layer instead of
 * tanh z and derives the matching biases b' = b + rowsum(W) from these rounded weights when it stages them
 * (gym_reinmav_amd.ppo.pack_policy_weights_f16 is the torch form of the same buffer). */
int rmav_pack_policy_f16(rmav_handle h, int n_params, const float *const *params, const int64_t *sizes, const int32_t *idx_lo,
                         const int32_t *idx_hi, int64_t n_out, float *weights_out);
int rmav_rollout_policy(rmav_handle h, int32_t n_steps, const float *weights, float *actions_out,
                        float *obs_out, float *rew_out, uint8_t *done_out, float *logp_out,
                        float *value_out, int precision);

/* ---- learner-side passes over a trajectory (DEVICE pointers, enqueued on the handle's stream) -------- */
/* Generalised advantage estimation, the backward pass of baselines ppo2 Runner.run():
 *   delta_t = reward_scale * r_t + gamma V_{t+1} (1 - done_t) - V_t,  A_t = delta_t + gamma lam (1 - done_t) A_{t+1}
 * rew [n_steps][N], done u8 [n_steps][N] (1 = the episode ended with step t), values [n_steps + 1][N]
 * (values[n_steps] = bootstrap value; exactly what rmav_rollout_policy writes); adv_out, ret_out [n_steps][N]
 * (ret = A + V).  sums_out (nullable): 2 doubles on the device <- (sum A, sum A^2) over all n_steps*N samples,
 * for the advantage normalisation (all-reduce them across ranks first when data parallel).  fp32 FMAs;
 * agrees with a float64 per-env recursion to ~1e-6 relative. */
int rmav_gae(rmav_handle h, int32_t n_steps, const float *rew, const uint8_t *done, const float *values,
             float gamma, float lam, float reward_scale, float *adv_out, float *ret_out, double *sums_out);
/* x[i] <- (x[i] - mean) * rstd for i < count (x 16-byte aligned): advantage normalisation in place. */
int rmav_normalize(rmav_handle h, float *x, int64_t count, float mean, float rstd);

/* ---- multi-GPU: the path's one collective ------------------------------------------ */
/* Envs shard over ranks by contiguous ranges of GLOBAL env id: rank r of W owns base + (r < rem) envs
 * starting at r*base + min(r, rem), base = n_total / W, rem = n_total % W (create each rank's handle with
 * env_id_base = that start).  The data path needs no communication; the only exchange is the all-gather of
 * per-env episode statistics once per rollout.  The communicator wraps an RCCL communicator (librccl.so.1 is
 * loaded on first use; one process per GPU).  Rank 0 calls rmav_comm_unique_id and hands the 128 bytes to the
 * other ranks out of band (file, MPI, a torch store ...); every rank then calls rmav_comm_create. */
typedef struct rmav_comm_s *rmav_comm;
#define RMAV_COMM_ID_BYTES 128
/* Optional, before any other rmav_comm_* call of the process: resolve the five collective entry points (ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclGetErrorString) from THIS shared object instead of librccl.so.1 - a
 * site's own RCCL build, or the test suite's stand-in that lets two rank processes share ONE GPU (tests/stub_rccl). */
int rmav_comm_use_library(const char *path);
int rmav_comm_unique_id(void *id_out /* RMAV_COMM_ID_BYTES bytes, host */);
int rmav_comm_create(rmav_comm *out, const void *id, int rank, int world, int device);
int rmav_comm_destroy(rmav_comm c);
/* What the communicator is: rank / world as passed to rmav_comm_create, and what the collective library itself reports for its
 * communicator (ncclCommUserRank / ncclCommCount; -1 when the library does not export them).  Any pointer may be NULL.  Lets a
 * launcher assert that RCCL really connected `world` ranks (the role of the MPI rank probe of gym_reinmav/run.py:18-21,177-182). */
int rmav_comm_info(rmav_comm c, int *rank_out, int *world_out, int *lib_rank_out, int *lib_world_out);
/* One tiny all-gather on the communicator's own stream, awaited on the HOST for at most timeout_s seconds (< 0: no limit):
 * RMAV_OK, or RMAV_ERR_TIMEOUT.  RCCL connects its transports inside the FIRST collective's enqueue - a host-side exchange
 * with the peers that blocks when one of them is gone - so a caller that wants a bounded set-up runs rmav_comm_create +
 * rmav_comm_warmup on a helper thread and joins it with a deadline (gym_reinmav_amd.distributed.NativeStatsExchange does);
 * no handle and no handle's stream is involved. */
int rmav_comm_warmup(rmav_comm c, double timeout_s);
/* returns_out f32 [n_total], lengths_out i32 [n_total] (DEVICE pointers) <- return / length of every env's most
 * recently finished episode, in global env order, on every rank.  Enqueued on the handle's stream (pack ->
 * ncclAllGather over xGMI -> unpack); does not synchronise.  Needs RMAV_F_TRACK_EPISODES. */
int rmav_allgather_stats(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out);
/* The same exchange in two halves, so that it overlaps the next rollout: _post packs this rank's payload on the
 * handle's stream (a stream-ordered snapshot) and runs the ncclAllGather on the communicator's OWN (high-priority)
 * stream behind a signal word; up to eight exchanges may be in flight, and a ninth post blocks the HOST until the
 * oldest one has finished (back pressure - nothing is ever inserted into the handle's stream).  _result makes the
 * handle's stream wait for the most recently posted gather and unpacks it.  rmav_allgather_stats = _post followed
 * by _result.  ~15 us of host time per post (two small launches, one event record, one RCCL enqueue). */
int rmav_allgather_stats_post(rmav_handle h, rmav_comm c, int64_t n_total);
/* Optional, BEFORE the rollout whose statistics the next _post will exchange: the next fused rmav_rollout /
 * rmav_rollout_policy launch of `h` then writes the snapshot itself (every wavefront stores its envs' statistics into
 * the exchange's send buffer and publishes an arrival word; the communicator's stream polls those), so that _post puts
 * NOTHING into the handle's stream - no pack kernel, no signal kernel (~8 us per post at 131 072 envs).  Same snapshot,
 * same result.  Only a call that is ONE fused launch over all of the handle's envs takes the snapshot; if none happens
 * between _arm and _post (single-step launches incl. rmav_rollout(fused = 0), a sliced launch), or if another stepping
 * launch follows the one that took it, _post packs as usual.  The communicator stream's wait for the armed launch is
 * bounded: 2 s counted from the moment that launch BEGINS on the device (it may sit behind any amount of queued work first;
 * controller-driven rollouts of <= 131 072 envs publish no start word and are bounded by the waiter's overall 10 min only).
 * Past that the waiter poisons this rank's payload - return NaN, length -1 for each of its envs, on every rank - and the
 * collective is issued all the same, so no peer hangs; rmav_allgather_stats_wait (and _result, once the waiter has run)
 * return RMAV_ERR_TIMEOUT for THAT post only: later posts on the communicator are unaffected.  One armed exchange per handle
 * and per communicator at a time; _post with the same communicator consumes it; destroying the communicator or the handle
 * disarms the other. */
int rmav_allgather_stats_arm(rmav_handle h, rmav_comm c, int64_t n_total);
int rmav_allgather_stats_result(rmav_handle h, rmav_comm c, int64_t n_total, float *returns_out, int32_t *lengths_out);
/* HOST-side bounded wait for the most recently posted exchange (polls its completion event; touches no stream):
 * RMAV_OK once the gather has finished, RMAV_ERR_TIMEOUT after timeout_s seconds (< 0: no limit).  Lets a caller probe a
 * freshly created communicator - post, wait with a deadline, only then _result - without ever parking a handle's stream
 * behind a collective that may never complete. */
int rmav_allgather_stats_wait(rmav_comm c, double timeout_s);
/* The send side of that exchange alone, for callers that own the collective (torch.distributed over RCCL):
 * send_out i32 [2][cmax] (DEVICE) <- bit patterns of the per-env last returns, then the last lengths, zero padded
 * from num_envs to cmax (the largest shard).  A stream-ordered snapshot in one small launch, so the next
 * rollout may overwrite the per-env arrays while the collective is still in flight. */
int rmav_pack_stats(rmav_handle h, int64_t cmax, int32_t *send_out);

/* ---- state access (also the env checkpoint) ------------------------------------------------ */
int rmav_get_state(rmav_handle h, float *out, int mem, int layout);      /* nS*N floats */
int rmav_set_state(rmav_handle h, const float *in, int mem, int layout);
int rmav_get_sbd(rmav_handle h, int32_t *out, int mem); /* steps_beyond_done per env, -1 = None */
int rmav_set_sbd(rmav_handle h, const int32_t *in, int mem);
int rmav_get_reset_counts(rmav_handle h, uint32_t *out, int mem); /* resets drawn so far per env */
int rmav_set_reset_counts(rmav_handle h, const uint32_t *in, int mem);
int rmav_get_time(rmav_handle h, double *out, int mem);  /* RMAV_REINMAV: per-env clock t (reinmav_env.py:72) */
int rmav_set_time(rmav_handle h, const double *in, int mem);
int rmav_get_step_count(rmav_handle h, uint64_t *out); /* global step counter t */
int rmav_set_step_count(rmav_handle h, uint64_t t);

/* ---- episode statistics (needs RMAV_F_TRACK_EPISODES) -------------------------------------- */
int rmav_episode_totals(rmav_handle h, rmav_ep_totals *out, int clear); /* synchronises */
/* Per-env return / length of the most recently finished episode (0 / 0 if none yet) and of the
 * running episode.  Any pointer may be NULL.  This is the payload of the per-rollout all-gather. */
int rmav_episode_buffers(rmav_handle h, float *last_return, int32_t *last_length,
                         float *cur_return, int32_t *cur_length, int mem);

#ifdef __cplusplus
}
#endif
#endif /* RMAV_H */
