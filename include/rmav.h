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
    RMAV_ACT_POLICY = 3,     /* Gaussian MLP policy evaluated in-kernel (rmav_rollout_policy only: rmav_ppo.h) */
    RMAV_ACT_POLICY_BF16 = 4 /* the same policy on the matrix cores */
};
enum rmav_integrator { RMAV_INT_EULER = 0, RMAV_INT_RK4 = 1 };

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
/* Explicit, per-handle overrides of the launch rules (DESIGN.md section 4 states the automatic rules and what was measured).
 * -1 = automatic, the default of every key.  Results never depend on them: every variant writes the same bits
 * (tests/test_gpu_parity.py::test_kernel_selection_variants_give_the_same_bits, ::test_single_step_variants_give_the_same_bits). */
enum rmav_tuning_key {
    RMAV_TUNE_SPLIT = 0,        /* fused rollouts: 0 = one wavefront per 64 envs, 1 = (integrator, memory) wavefront pairs */
    RMAV_TUNE_SLICE = 1,        /* batches beyond the pair kernel's capacity: 0 = one launch, 1 = one launch per balanced slice */
    RMAV_TUNE_STORE_POLICY = 2, /* trajectory stores: 0 write-back, 1 write-through, 2 non-temporal, 3 LDS-transposed batch-major obs */
    RMAV_TUNE_SPLIT_GROUP = 3,  /* (integrator, memory) pairs per workgroup, 1 .. 8 */
    RMAV_TUNE_BLOCK = 4,        /* workgroup size of the one-wavefront kernels: 64 | 128 | 256 */
    RMAV_TUNE_STEP_LAZY = 5,    /* rmav_step: 1 = the env's termination record is loaded only in lanes whose episode ends */
    RMAV_TUNE_STEP_STORE = 6,   /* rmav_step's per-env stores: 0 write-back, 1 write-through, 2 non-temporal */
    RMAV_TUNE_POLICY_PAIR = 7,  /* RMAV_POLICY_BF16_MFMA: 0 = one wavefront per 64 envs instead of the (actor, critic) pair */
    RMAV_TUNE_PAIR_GROUP = 8,   /* (actor, critic) pairs per workgroup of the matrix-core actors, 1 .. 4 */
    RMAV_TUNE_COUNT = 9
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
 * element i of every column, elements [N, pitch) are never written.  Same results as rmav_rollout.  Keeps the fast store path
 * for batch sizes that are not a multiple of 16.  rmav_trajectory_pitch() = N rounded up to a multiple
 * of 64; any pitch >= N is accepted. */
int64_t rmav_trajectory_pitch(rmav_handle h);
int rmav_rollout_pitched(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t pitch, int fused);

/* rmav_rollout (fused) with CHUNK-MAJOR trajectory arrays: the env range is cut into chunks of chunk_envs (a multiple of 64; the
 * last one may be shorter) and chunk c's trajectory is a dense array of its own,
 *   actions_in / actions_out [n_chunks][n_steps][nA][chunk_envs], obs_out [n_chunks][n_steps][nS][chunk_envs],
 *   rew_out / done_out [n_chunks][n_steps][chunk_envs]   (env i = chunk i / chunk_envs, column i % chunk_envs; device pointers).
 * Same values as rmav_rollout; one launch per chunk, each writing one dense region - the layout this GPU stores fastest for
 * big quadrotor3d batches (131 072 envs: 0.65 -> 0.78 of the HBM roofline).  rmav_chunk_envs() is the
 * recommended chunk: 65 536 for quadrotor3d beyond 65 536 envs, otherwise N rounded up to a multiple of 64 (one chunk = the
 * plain layout, with that column pitch when N % 64 != 0).  A learner that flattens (step, env) samples - PPO2 does - consumes
 * the chunks as they are.  chunk_envs < N needs n_steps >= 2 and chunk_envs <= the pair kernel's capacity (131 072; 65 536
 * for controller-driven slung-load); RMAV_ACT_BUFFER does not echo the caller's actions (actions_out must be NULL). */
int64_t rmav_chunk_envs(rmav_handle h);
int rmav_rollout_chunked(rmav_handle h, int32_t n_steps, int action_mode, const float *actions_in, float *actions_out,
                         float *obs_out, float *rew_out, uint8_t *done_out, int64_t chunk_envs);

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

/* The two extensions of the path, each in its own header (included here, so that `#include "rmav.h"` declares everything):
 *   rmav_ppo.h   SURVEY 8(f1): the PPO2 rollout loop with the policy inside the kernel, GAE, advantage normalisation
 *   rmav_comm.h  SURVEY 8(e): the path's one collective - the all-gather of per-env episode statistics over RCCL / xGMI */
#include "rmav_ppo.h"
#include "rmav_comm.h"

#endif /* RMAV_H */
