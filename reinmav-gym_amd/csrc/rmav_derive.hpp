// rmav_derive.hpp - host-side derivation of the per-kind kernel constants from rmav_params.
// All derived values are computed in fp64 and rounded once to the arithmetic type R.
#pragma once

#include <cstring>

#include "../../include/rmav.h"
#include "rmav_math.hpp"

namespace rmav {

// two_d: the 2-D kinds, whose controller adds the literal (0, 9.8) (quadrotor2d.py:130) whatever self.g is
template <typename R> inline ParamsT<R> derive(const rmav_params &q, bool two_d) {
    ParamsT<R> p;
    memset(&p, 0, sizeof(p));
    p.inv_mass = (R)(1.0 / q.mass);
    p.mass = (R)q.mass;
    p.load_mass = (R)q.load_mass;
    p.inv_mtot = (R)(1.0 / (q.mass + q.load_mass));
    p.dt = (R)q.dt;
    p.half_dt2 = (R)(0.5 * q.dt * q.dt);
    for (int i = 0; i < 3; ++i) {
        p.gv[i] = (R)q.g_vec[i];
        p.ff[i] = two_d ? (R)(i == 1 ? 9.8 : 0.0) : (R)(-q.g_vec[i]);
    }
    p.L = (R)q.tether_length;
    p.mL = (R)(q.mass * q.tether_length);
    p.pos_limit = (R)q.pos_limit;
    p.vel_limit = (R)q.vel_limit;
    p.thrust_scale = (R)q.thrust_scale;
    p.kp = (R)q.kp;
    p.kv = (R)q.kv;
    p.two_over_tau = (R)(2.0 / q.tau);
    p.neg_inv_tau = (R)(-1.0 / q.tau);
    for (int i = 0; i < 3; ++i) {
        p.ref_pos[i] = (R)q.ref_pos[i];
        p.ref_vel[i] = (R)q.ref_vel[i];
    }
    p.clamp_thrust = q.clamp_thrust;
    return p;
}

// ReinmavEnv constants: mass / gravity / dt come from rmav_params, the rest are the reference's literals
// (reinmav_env.py:55-63, :91, :129, :312-315).
inline ReinmavP derive_reinmav(const rmav_params &q) {
    ReinmavP p;
    memset(&p, 0, sizeof(p));
    p.arm_length = 0.0860;
    p.mass = q.mass;
    p.gravity = q.g;
    p.min_force4 = 0.0 / 4.0;
    p.max_force4 = 3.5316 / 4.0;
    const double I[3][3] = {{0.00025, 0, 2.55e-06}, {0, 0.000232, 0}, {2.55e-06, 0, 0.0003738}};
    memcpy(p.inertia, I, sizeof(I));
    const double c00 = I[1][1] * I[2][2] - I[1][2] * I[2][1], c01 = I[1][2] * I[2][0] - I[1][0] * I[2][2],
                 c02 = I[1][0] * I[2][1] - I[1][1] * I[2][0];
    const double det = I[0][0] * c00 + I[0][1] * c01 + I[0][2] * c02;
    p.inv_inertia[0][0] = c00 / det;
    p.inv_inertia[0][1] = (I[0][2] * I[2][1] - I[0][1] * I[2][2]) / det;
    p.inv_inertia[0][2] = (I[0][1] * I[1][2] - I[0][2] * I[1][1]) / det;
    p.inv_inertia[1][0] = c01 / det;
    p.inv_inertia[1][1] = (I[0][0] * I[2][2] - I[0][2] * I[2][0]) / det;
    p.inv_inertia[1][2] = (I[0][2] * I[1][0] - I[0][0] * I[1][2]) / det;
    p.inv_inertia[2][0] = c02 / det;
    p.inv_inertia[2][1] = (I[0][1] * I[2][0] - I[0][0] * I[2][1]) / det;
    p.inv_inertia[2][2] = (I[0][0] * I[1][1] - I[0][1] * I[1][0]) / det;
    p.dt = q.dt;
    p.ds = 1.0 / 5000;
    p.t_max = 4.0;
    const double kp[3] = {10, 10, 35}, kd[3] = {5, 5, 22}, kpr[3] = {100, 100, 100}, kdr[3] = {.1, .1, .1};
    memcpy(p.kp, kp, sizeof(kp));
    memcpy(p.kd, kd, sizeof(kd));
    memcpy(p.kp_rot, kpr, sizeof(kpr));
    memcpy(p.kd_rot, kdr, sizeof(kdr));
    p.rk4 = (q.integrator == RMAV_INT_RK4) ? 1 : 0;
    return p;
}

template <int K> inline typename Env<K>::P derive_env(const rmav_params &q) {
    if constexpr (K == REINMAV) return derive_reinmav(q);
    else return derive<typename Env<K>::R>(q, K == QUAD2D || K == QUAD2D_SL);
}

}  // namespace rmav
