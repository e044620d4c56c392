// rmav_derive.hpp - host-side derivation of the per-kind kernel constants from rmav_params.
// All derived values are computed in fp64 and rounded once to the arithmetic type R.
#pragma once

#include <cstring>

#include "../../include/rmav.h"
#include "rmav_math.hpp"

namespace rmav {

template <typename R> inline ParamsT<R> derive(const rmav_params &q) {
    ParamsT<R> p;
    memset(&p, 0, sizeof(p));
    p.inv_mass = (R)(1.0 / q.mass);
    p.mass = (R)q.mass;
    p.load_mass = (R)q.load_mass;
    p.inv_mtot = (R)(1.0 / (q.mass + q.load_mass));
    p.dt = (R)q.dt;
    p.half_dt2 = (R)(0.5 * q.dt * q.dt);
    p.g = (R)q.g;
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

}  // namespace rmav
