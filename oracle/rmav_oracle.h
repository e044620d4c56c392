/* rmav_oracle.h - CPU restatement (fp64, plain C) of reinmav-gym's native quadrotor hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (reinmav-gym_amd/) may include,
 * link or call this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time the reference's algorithm without the reference's Python.
 *
 * Every function cites the reference lines it follows (paths relative to the reference repo root,
 * gym_reinmav/envs/native/).  Arithmetic is fp64 like the reference (NumPy float64) and follows
 * the reference's order of operations.
 *
 * Pinning status: PINNED against the reference's own step()/control() executed in the authoring
 * container (oracle/ref_harness.py loads the four reference files by path; tests/golden/ holds
 * the resulting vectors, tests/golden/make_golden.py regenerates them).  The reference ships no
 * golden vectors or asserting tests of its own (the reference test scripts only print a wall-clock time).  The one
 * third-party piece on the path, pyquaternion (requirements.txt:1 "pyquaternion>0.9", un-pinned,
 * absent from the reference tree and from this image) is restated from its published algorithm
 * in oracle/ref_harness.py and cross-checked against scipy.spatial.transform.Rotation; that
 * dependency is the only part of the pin that is not the reference's own code.
 */
#ifndef RMAV_ORACLE_H
#define RMAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORACLE_QUAD2D = 0, ORACLE_QUAD2D_SL = 1, ORACLE_QUAD3D = 2, ORACLE_QUAD3D_SL = 3 };

/* Physics / controller constants.  Defaults are the literals of each reference __init__. */
typedef struct oracle_params {
    double mass;          /* quadrotor3d.py:45 */
    double load_mass;     /* quadrotor3d_slungload.py:46 */
    double dt;            /* quadrotor3d.py:46 */
    double g;             /* |g_vec| = 9.8, informational (mirrors rmav_params.g); the dynamics read g_vec */
    double tether_length; /* quadrotor3d_slungload.py:58 (1.5), quadrotor2d_slungload.py:53 (0.5) */
    double pos_limit;     /* |pos| > pos_limit terminates */
    double vel_limit;     /* |vel| > vel_limit terminates */
    double thrust_scale;  /* quadrotor2d.py:75 multiplies the thrust command by 10 */
    int clamp_thrust;     /* quadrotor2d.py:76-77 clamps the scaled thrust at >= 0 */
    double ref_pos[3];    /* controller set-point */
    double ref_vel[3];
    double kp, kv, tau;   /* controller gains */
    double g_vec[3];      /* self.g: (0,0,-9.8) quadrotor3d.py:47, quadrotor3d_slungload.py:48; 2-D kinds [0..1] = (0,-9.8)
                             quadrotor2d.py:46, quadrotor2d_slungload.py:47.  Added as a vector like the reference does */
} oracle_params;

int oracle_state_dim(int kind);  /* 5, 9, 10, 16 */
int oracle_action_dim(int kind); /* 2, 2, 4, 4 */
/* reading_2d: 'B' (default; |p|>3 or |v|>2) or 'A' (|p|>3 or |v|>10); ignored for other kinds. */
int oracle_default_params(int kind, int reading_2d, oracle_params *p);

/* One env, one step.  s/s_out: nS doubles in the reference's state order; a: nA doubles.
 * sbd: in/out steps_beyond_done, -1 encodes Python None.  Returns 0, or -1 for a bad kind. */
int oracle_step(int kind, const oracle_params *p, const double *s, const double *a, double *s_out,
                double *reward, int *done, int *sbd);

/* Same with the tether branch of the slung-load kinds forced: force_taut = 1 (taut), 0 (slack),
 * -1 (decide by |tether| >= L like the reference).  The kinematic projection leaves |tether| = L up
 * to rounding, so in closed loop the next step's branch is decided by the last bit of a norm (and,
 * in the reference, by whether NumPy's BLAS dot uses FMA); tests accept either branch there. */
int oracle_step_branch(int kind, const oracle_params *p, const double *s, const double *a,
                       double *s_out, double *reward, int *done, int *sbd, int force_taut);

/* Geometric controller of the reference's test loop: state -> action. */
int oracle_control(int kind, const oracle_params *p, const double *s, double *a_out);

/* ---- counter-based RNG shared (by specification) with the HIP path ------------------------- */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* u32 -> float in [0,1): (x >> 8) * 2^-24 */
float oracle_u01(uint32_t x);
/* State drawn at the episode-th reset of global env env_id: every component U[-1,1)
 * (quadrotor3d.py:182-185 draws uniform(-1,1) for all nS components). */
void oracle_reset_state(int kind, uint64_t seed, uint64_t env_id, uint32_t episode, float *s_out);
/* Random action of global env env_id at global step t: component i = fmaf(hi-lo, u_i, lo). */
void oracle_random_action(int kind, uint64_t seed, uint64_t env_id, uint64_t t, float lo, float hi,
                          float *a_out);

/* ---- batched drivers (AoS, used for parity of the batched path and as the CPU baseline) ----- */
/* n envs, one step each; s is updated in place ([n][nS]); a is [n][nA]; sbd is [n].
 * If round_f32 != 0 the inputs are taken as fp32-representable and outputs are rounded to fp32
 * (storage model of the HIP path); arithmetic is fp64 either way. */
int oracle_batch_step(int kind, const oracle_params *p, int64_t n, double *s, const double *a,
                      double *reward, uint8_t *done, int32_t *sbd, int round_f32);

/* CPU-baseline workload: n envs x steps, random actions from the counter RNG in [lo,hi), auto
 * reset on done (fresh U[-1,1) state), state stored as fp32 between steps.  Returns the number of
 * env-steps executed; *ret_sum accumulates rewards (so the work cannot be optimised away). */
int64_t oracle_rollout_random(int kind, const oracle_params *p, int64_t n, int64_t steps,
                              uint64_t seed, uint64_t env_id_base, float lo, float hi, float *state,
                              int32_t *sbd, uint32_t *episode, uint64_t t0, double *ret_sum,
                              int64_t *n_done);

/* ---- ReinmavEnv (reinmav_env.py): 13-state rigid body, built-in PD controller + min-jerk trajectory ---- */
typedef struct oracle_reinmav_params {
    double arm_length, mass, gravity, min_force, max_force; /* reinmav_env.py:55-59 */
    double inertia[3][3], inv_inertia[3][3];                /* :60-63 */
    double dt, ds, t_max;                                   /* :73, :91, :129 */
    double kp[3], kd[3], kp_rot[3], kd_rot[3];              /* :312-315 */
} oracle_reinmav_params;
void oracle_reinmav_default_params(oracle_reinmav_params *p);
/* state order (reinmav_env.py:79): x y z dx dy dz qw qx qy qz p q r */
/* trj_gen + stateToQd + controller (reinmav_env.py:128-136, 292-337): fm = (F, Mx, My, Mz) */
void oracle_reinmav_controller(const oracle_reinmav_params *p, const double s[13], double t, double fm[4]);
/* quad_eq_of_motion2 (reinmav_env.py:203-264): motor mixing + clamp, rigid-body derivative */
void oracle_reinmav_derivative(const oracle_reinmav_params *p, const double s[13], const double fm[4],
                               double sdot[13]);
/* step() (reinmav_env.py:99-126) = myODE (:90-98): Euler sub-steps of ds over np.arange(t, t+dt, ds)
 * (50 or 51 of them, decided by fp64 rounding of t), then t += dt.  action == NULL: the built-in
 * controller is evaluated at every sub-step like the reference; otherwise (F, Mx, My, Mz) is held over the
 * step (an extension: the reference has no action input).  reward is the reference's constant 90.0 and
 * done is always 1.  Returns the number of sub-steps taken. */
int oracle_reinmav_step(const oracle_reinmav_params *p, double s[13], double *t, const double *action,
                        double *reward, int *done);
/* Same with classical RK4 sub-steps (command held over each sub-step) - NOT in the reference; it checks the
 * product's optional RMAV_INT_RK4 integrator. */
int oracle_reinmav_step_rk4(const oracle_reinmav_params *p, double s[13], double *t, const double *action,
                            double *reward, int *done);

#ifdef __cplusplus
}
#endif
#endif
