// rmav_math.hpp - per-env (per-lane) arithmetic of the batched quadrotor path (five env kinds), gfx950.
//
// One environment lives in the registers of one wavefront lane; everything here is lane-local
// register math (the 3x3 / quaternion products are ~100-250 flops, far below the HBM time of the
// 53-149 bytes an env-step moves).  Functions are __host__ __device__ so that the test-only
// helper tests/hostmath.hip can run the SAME source on the host cores of a GPU-less box; the
// shipped library only ever launches them inside kernels.
//
// Arithmetic types (R):
//   quad2d, quad3d        : R = float.  Worst error vs the fp64 reference ~1e-7 * max(1,|y|).
//   quad2d_sl, quad3d_sl  : R = double on fp32 storage.  The tether projection
//                           v_l - ((v_l - v).e) e  cancels (quadrotor3d_slungload.py:128,
//                           quadrotor2d_slungload.py:115) and fp32 misses the 1e-6 bar there;
//                           CDNA4 runs fp64 FMA at half the fp32 rate, which this HBM-bound path
//                           does not notice.  sin/cos stay fp32 (|d| <= 6e-8 on a unit vector).
//   controllers           : R = double (gains of 10..50 amplify fp32 rounding past 1e-6).
//   reinmav               : R = double incl. asin/atan2/sincos (attitude loop gain kp_rot/I ~ 4e5 1/s^2).
// All multiply-adds are written as explicit fma() and the library is built with
// -ffp-contract=off, so the single-step kernel, the fused rollout kernel and the host test build
// produce the same bits for the same inputs (up to libm's sinf/cosf/atan2).
//
// Reference citations are relative to gym_reinmav/envs/native/ of ethz-asl/reinmav-gym.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RMAV_HD __host__ __device__ __forceinline__

namespace rmav {

enum : int { QUAD2D = 0, QUAD2D_SL = 1, QUAD3D = 2, QUAD3D_SL = 3, REINMAV = 4 };

template <int K> struct Dims;
template <> struct Dims<QUAD2D>    { static constexpr int NS = 5,  NA = 2; using real = float;  };
template <> struct Dims<QUAD2D_SL> { static constexpr int NS = 9,  NA = 2; using real = double; };
template <> struct Dims<QUAD3D>    { static constexpr int NS = 10, NA = 4; using real = float;  };
template <> struct Dims<QUAD3D_SL> { static constexpr int NS = 16, NA = 4; using real = double; };
template <> struct Dims<REINMAV>   { static constexpr int NS = 13, NA = 4; using real = double; };

// ---- scalar helpers -----------------------------------------------------------------------------
RMAV_HD float  rfma(float a, float b, float c)    { return __builtin_fmaf(a, b, c); }
RMAV_HD double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// Square roots and reciprocal square roots.  The host test build uses the correctly rounded libm forms.
// fp32 (quad2d / quad3d) on the device: the hardware's 1-ulp v_sqrt_f32 / v_rsq_f32.  hipcc's IEEE-exact sqrtf and 1/x are 12-14
// dependent instructions each (scale, v_sqrt/v_rcp, two fma refinements, fix-ups), the fp32 step has three of
// them on its critical path, and at one wavefront per SIMD that chain - not the issue rate - sets the step
// time.  1 ulp = 6e-8 relative, an order of magnitude inside the 1e-6 parity bar (measured worst error of the
// whole step vs the fp64 oracle stays < 3e-7).  The host test build keeps the correctly rounded versions, so host
// and device may differ in the last bit of a norm; the parity tests compare both with the oracle, not with
// each other.  The hardware instructions flush denormal inputs, hence the FLT_MIN guard.
RMAV_HD float  root(float x)  {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}
RMAV_HD float inv_sqrt(float x) {   // 1/sqrt(x)
#if defined(__HIP_DEVICE_COMPILE__)
    // branch-free: a denormal argument is scaled into the normal range first (2^64, exact) and the result scaled back (2^32)
    // - two selects instead of a divergent branch around the IEEE divide / square root sequence, which sat in the step loop of
    // every 3-D kind (quat_normalise) as ~10 scalar / branch instructions per env-step
    const bool tiny = x < 1.17549435e-38f;
    const float r = __builtin_amdgcn_rsqf(tiny ? x * 18446744073709551616.0f : x);
    return tiny ? r * 4294967296.0f : r;
#else
    return 1.0f / __builtin_sqrtf(x);
#endif
}
// fp64 on the device: v_rsq_f64 is a ~single-precision estimate; hipcc's correctly rounded sqrt / divide
// wrap it in 15-25 more fp64 instructions (half rate).  One Newton step takes the estimate to <= 1e-12
// relative, and every fp64 result of these kinds is rounded to fp32 storage (6e-8) at the end of the step, so
// the extra digits of the exact sequences are never seen.  Branch-free: for 0, inf, NaN and negative x the
// raw estimate already is what 1/sqrt(x) returns (inf, 0, NaN, NaN) and is passed through.
RMAV_HD double inv_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * y), 0.5 * y, 0.5);
    const double r = __builtin_fma(y, e, y);
    return (x > 0.0 && x < __builtin_inf()) ? r : y;
#else
    return 1.0 / __builtin_sqrt(x);
#endif
}
RMAV_HD double root(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = inv_sqrt(x), g = x * y;
    const double r = __builtin_fma(__builtin_fma(-g, g, x), 0.5 * y, g);   // one correction: g + (x - g^2) * y/2
    return (x > 0.0 && x < __builtin_inf()) ? r : (x == 0.0 ? x : (x > 0.0 ? x : y));   // 0 -> 0, inf -> inf, else NaN
#else
    return __builtin_sqrt(x);
#endif
}
RMAV_HD float  rabs(float x)  { return __builtin_fabsf(x); }
RMAV_HD double rabs(double x) { return __builtin_fabs(x); }

// Constants of one env kind in the arithmetic type R, derived once on the host in fp64.
template <typename R> struct ParamsT {
    R inv_mass;      // 1/mass
    R mass;
    R load_mass;
    R inv_mtot;      // 1/(mass+load_mass)
    R dt;
    R half_dt2;      // 0.5*dt*dt
    R gv[3];         // gravity VECTOR (quadrotor3d.py:47 self.g = (0, 0, -9.8); 2-D kinds: (0, -9.8), quadrotor2d.py:46), added component-wise
    R ff[3];         // controller feed-forward: 3-D kinds -self.g (quadrotor3d.py:162), 2-D kinds the LITERAL (0, 9.8) of quadrotor2d.py:130
    R L;             // tether length
    R mL;            // mass * tether length
    R pos_limit, vel_limit;
    R thrust_scale;
    R kp, kv;
    R two_over_tau;  // 3-D controller  quadrotor3d.py:173
    R neg_inv_tau;   // 2-D controller  quadrotor2d.py:133
    R ref_pos[3], ref_vel[3];
    int32_t clamp_thrust;
    int32_t _pad;
};

// Re-derive the mass / tether dependent constants for one lane (per-env domain randomisation); same
// formulas as the host-side derive() in rmav_derive.hpp, evaluated in fp64 and rounded once to R.
template <typename R> RMAV_HD void override_params(ParamsT<R> &p, double mass, double load_mass, double L) {
    p.inv_mass = (R)(1.0 / mass);
    p.mass = (R)mass;
    p.load_mass = (R)load_mass;
    p.inv_mtot = (R)(1.0 / (mass + load_mass));
    p.L = (R)L;
    p.mL = (R)(mass * L);
}

// ---- Philox4x32-10 (counter RNG; stream layout documented in include/rmav.h) ----------------------
// a ^ b ^ c: one v_bitop3_b32 (truth table 0x96) on gfx950 - hipcc leaves the two xors of a Philox round unfused, and the
// random-action draw is the largest single block of the rollout's instruction stream (64 -> 44 vector instructions per call)
RMAV_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}
RMAV_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                           uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

RMAV_HD float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// reset(): every state component ~ U[-1,1)   (quadrotor3d.py:182-185 and the three siblings)
template <int K>
RMAV_HD void reset_state(uint64_t seed, uint64_t env_id, uint32_t reset_idx,
                         float (&s)[Dims<K>::NS]) {
    constexpr int NS = Dims<K>::NS;
#pragma unroll
    for (int j = 0; j * 4 < NS; ++j) {
        uint32_t r[4];
        philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), reset_idx, (1u << 24) | (uint32_t)j,
                      (uint32_t)seed, (uint32_t)(seed >> 32), r);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (j * 4 + i < NS) s[j * 4 + i] = rfma(1.0f / 8388608.0f, (float)(r[i] >> 8), -1.0f);  // = fma(2, u01(r), -1), exact in fp32
    }
}

// Random actions (RMAV_ACT_RANDOM).  One Philox4x32-10 call yields four draws: the 4-action kinds use block index t, the
// 2-action kinds (quadrotor2d, quadrotor2d-slungload) use block index t >> 1 and take the half (t & 1) of it - the draw is
// the largest single block of a random-action rollout's instruction stream (20 quarter-rate 32x32->64 multiplies per call),
// and half of it was thrown away for the 2-D kinds.
template <int K> constexpr bool action_pairs() { return Dims<K>::NA <= 2; }
template <int K>
RMAV_HD void random_block(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t (&r)[4]) {
    const uint64_t b = action_pairs<K>() ? (t >> 1) : t;
    philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)b,
                  (2u << 24) | ((uint32_t)((b >> 32) & 0xFFFFu) << 8), (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
}
template <int K>
RMAV_HD void action_from_block(const uint32_t (&r)[4], uint64_t t, float lo, float hi, float (&a)[Dims<K>::NA]) {
    // = fma(hi - lo, u01(r), lo) bit for bit: u01 is k * 2^-24 with k < 2^24, both scalings by 2^-24 are exact, and the
    // fma rounds the same real number once - written this way it is one multiply per draw less
    const float scale = (hi - lo) * (1.0f / 16777216.0f);
    // the half of the block as a bit-select on the values ((t & 1) ? r[2 + i] : r[i] is turned into an indexed load from a
    // copy of r[] in scratch memory by hipcc)
    const uint32_t m = action_pairs<K>() ? (uint32_t)0 - (uint32_t)(t & 1u) : 0u;
#pragma unroll
    for (int i = 0; i < Dims<K>::NA; ++i) {
        const uint32_t x = action_pairs<K>() ? ((r[i & 1] & ~m) | (r[2 + (i & 1)] & m)) : r[i];
        a[i] = rfma(scale, (float)(x >> 8), lo);
    }
}
template <int K>
RMAV_HD void random_action(uint64_t seed, uint64_t env_id, uint64_t t, float lo, float hi,
                           float (&a)[Dims<K>::NA]) {
    uint32_t r[4];
    random_block<K>(seed, env_id, t, r);
    action_from_block<K>(r, t, lo, hi, a);
}

// ---- quaternion pieces (pyquaternion semantics the reference relies on) ------------------------
// Quaternion._normalise(): q/|q| unless |1-|q|^2| < 1e-14 or |q| is 0 (or NaN).
template <typename R> RMAV_HD void quat_normalise(const R (&q)[4], R (&o)[4]) {
    const R n2 = rfma(q[0], q[0], rfma(q[1], q[1], rfma(q[2], q[2], q[3] * q[3])));
    // n2 == 0 (or NaN): left alone, like _normalise.  Written as a select: no divergent branch in the step loop
    const R s = (!(rabs(R(1) - n2) < R(1e-14)) && n2 > R(0)) ? inv_sqrt(n2) : R(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = q[i] * s;
}

// rotation_matrix . e3 for a normalised (w,x,y,z): the body z axis in world frame.
template <typename R> RMAV_HD void quat_body_z(const R (&q)[4], R (&b)[3]) {
    const R w = q[0], x = q[1], y = q[2], z = q[3];
    const R t0 = rfma(x, z, w * y);
    const R t1 = rfma(y, z, -(w * x));
    b[0] = t0 + t0;
    b[1] = t1 + t1;
    b[2] = rfma(w, w, rfma(z, z, -rfma(x, x, y * y)));
}

// q' = q_raw + dt * 0.5 * qn (x) (0, w)     quadrotor3d.py:101-102  (normalised qn, raw q)
template <typename R>
RMAV_HD void quat_integrate(const R (&q_raw)[4], const R (&qn)[4], const R (&w)[3], R dt, R (&o)[4]) {
    const R hw = R(0.5) * qn[0], hx = R(0.5) * qn[1], hy = R(0.5) * qn[2], hz = R(0.5) * qn[3];
    const R e0 = -rfma(hx, w[0], rfma(hy, w[1], hz * w[2]));
    const R e1 = rfma(hw, w[0], rfma(hy, w[2], -(hz * w[1])));
    const R e2 = rfma(hz, w[0], rfma(hw, w[1], -(hx * w[2])));
    const R e3 = rfma(hx, w[1], rfma(hw, w[2], -(hy * w[0])));
    o[0] = rfma(e0, dt, q_raw[0]);
    o[1] = rfma(e1, dt, q_raw[1]);
    o[2] = rfma(e2, dt, q_raw[2]);
    o[3] = rfma(e3, dt, q_raw[3]);
}

// ---- the four step() bodies ----------------------------------------------------------------------
// Each updates the fp32 state in place and returns the distance the reward uses and `done`
// (reward / steps_beyond_done bookkeeping is identical for all kinds and lives in the kernel).

template <int K> struct Env;

// Quadrotor3D.step  quadrotor3d.py:81-124
template <> struct Env<QUAD3D> {
    using R = float;
    using P = ParamsT<R>;
    static RMAV_HD void step(float (&s)[10], const float (&a)[4], const ParamsT<R> &p, float &dist,
                             bool &done) {
        const R q[4] = {s[3], s[4], s[5], s[6]};
        const R w[3] = {a[1], a[2], a[3]};
        R qn[4], b[3], qo[4];
        quat_normalise(q, qn);                       // :96 rotation_matrix normalises
        quat_body_z(qn, b);
        const R k = a[0] * p.inv_mass;               // :96 thrust/mass
        const R acc[3] = {rfma(k, b[0], p.gv[0]), rfma(k, b[1], p.gv[1]), rfma(k, b[2], p.gv[2])};   // :96 ... + self.g
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const R v = s[7 + i];
            s[i] = rfma(acc[i], p.half_dt2, rfma(v, p.dt, s[i]));   // :98 (old velocity)
            s[7 + i] = rfma(acc[i], p.dt, v);                        // :99
        }
        quat_integrate(q, qn, w, p.dt, qo);          // :101-102
#pragma unroll
        for (int i = 0; i < 4; ++i) s[3 + i] = qo[i];
        const R np = root(rfma(s[0], s[0], rfma(s[1], s[1], s[2] * s[2])));
        const R nv = root(rfma(s[7], s[7], rfma(s[8], s[8], s[9] * s[9])));
        done = (np > p.pos_limit) || (nv > p.vel_limit);   // :106-110 (the "< -thr" clauses are dead)
        dist = np;                                   // :113 reward = -|pos|
    }
};

// Quadrotor3DSlungload.step  quadrotor3d_slungload.py:87-167
template <> struct Env<QUAD3D_SL> {
    using R = double;
    using P = ParamsT<R>;
    static RMAV_HD void step(float (&s)[16], const float (&a)[4], const ParamsT<R> &p, float &dist,
                             bool &done) {
        R pos[3] = {s[0], s[1], s[2]};
        R vel[3] = {s[7], s[8], s[9]};
        R lp[3] = {s[10], s[11], s[12]};
        R lv[3] = {s[13], s[14], s[15]};
        const R thrust = a[0];
        // The attitude (normalise, body z axis, quaternion integration) has no cancellation in it and runs in fp32 exactly as
        // in Quadrotor3D - fp64 vector FMAs are half rate here; only the translation / tether arithmetic below needs fp64.
        // Worst obs error vs the fp64 oracle 8.8e-8 (all-fp64: 6.0e-8; bar 1e-6), two-wavefront rollout at 131 072 envs -6.6 %.
        float qnf[4], bf[3], qof[4];
        const float qf[4] = {s[3], s[4], s[5], s[6]};
        const float wf[3] = {a[1], a[2], a[3]};
        quat_normalise(qf, qnf);
        quat_body_z(qnf, bf);
        quat_integrate(qf, qnf, wf, (float)p.dt, qof);                        // :122-123 / :144-145
        const R b[3] = {bf[0], bf[1], bf[2]};
        const R tv[3] = {lp[0] - pos[0], lp[1] - pos[1], lp[2] - pos[2]};     // :101
        const R dd = rfma(tv[0], tv[0], rfma(tv[1], tv[1], tv[2] * tv[2]));
        const bool taut = dd >= p.L * p.L;                                    // :104  |tv| >= L, without the root
        const R k = thrust * p.inv_mass;
        R acc[3] = {rfma(k, b[0], p.gv[0]), rfma(k, b[1], p.gv[1]), rfma(k, b[2], p.gv[2])};   // :118 / :140
        R la[3] = {p.gv[0], p.gv[1], p.gv[2]};                                // :134 slack: a_l = g
        if (taut) {
            const R inv_d = inv_sqrt(dd);
            const R u[3] = {tv[0] * inv_d, tv[1] * inv_d, tv[2] * inv_d};     // :102
            const R c = p.mL * rfma(lv[0], lv[0], rfma(lv[1], lv[1], lv[2] * lv[2]));
            // :110 inner(u, thrust_vec - c) with the scalar c broadcast over the vector
            const R sc = rfma(u[0], rfma(thrust, b[0], -c),
                              rfma(u[1], rfma(thrust, b[1], -c), u[2] * rfma(thrust, b[2], -c)));
            const R f = sc * p.inv_mtot;                                      // :111
            la[0] = rfma(f, u[0], p.gv[0]);
            la[1] = rfma(f, u[1], p.gv[1]);
            la[2] = rfma(f, u[2], p.gv[2]);
            // :115 T = m_l * |a_l - g| * u ;  a_l - g = f*u with |u| = 1, so the norm is |f| (to ~1e-15; no root needed)
            const R tn = p.load_mass * rabs(f);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[i] = rfma(tn * u[i], p.inv_mass, acc[i]);   // :118 + T/mass
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lp[i] = rfma(la[i], p.half_dt2, rfma(lv[i], p.dt, lp[i]));        // :112 / :136
            lv[i] = rfma(la[i], p.dt, lv[i]);                                 // :113 / :137
            pos[i] = rfma(acc[i], p.half_dt2, rfma(vel[i], p.dt, pos[i]));    // :119 / :141
            vel[i] = rfma(acc[i], p.dt, vel[i]);                              // :120 / :142
        }
        if (taut) {                                                           // :126-128 projection
            const R e[3] = {lp[0] - pos[0], lp[1] - pos[1], lp[2] - pos[2]};
            const R inv_n = inv_sqrt(rfma(e[0], e[0], rfma(e[1], e[1], e[2] * e[2])));
            const R dir[3] = {e[0] * inv_n, e[1] * inv_n, e[2] * inv_n};
            const R pr = rfma(lv[0] - vel[0], dir[0],
                              rfma(lv[1] - vel[1], dir[1], (lv[2] - vel[2]) * dir[2]));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                lp[i] = rfma(dir[i], p.L, pos[i]);
                lv[i] = rfma(-pr, dir[i], lv[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            s[i] = (float)pos[i];
            s[7 + i] = (float)vel[i];
            s[10 + i] = (float)lp[i];
            s[13 + i] = (float)lv[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s[3 + i] = qof[i];
        // the two norms of the termination test and the reward: fp32, from the stored state, as in Quadrotor3D
        const float nlp = root(rfma(s[10], s[10], rfma(s[11], s[11], s[12] * s[12])));
        const float nv = root(rfma(s[7], s[7], rfma(s[8], s[8], s[9] * s[9])));
        done = (nlp > (float)p.pos_limit) || (nv > (float)p.vel_limit);   // :149-153 load position, quad velocity
        dist = nlp;                                                        // :156 reward = -|load_pos|
    }
};

// sin and cos of the 2-D kinds' pitch angle (quadrotor2d.py:88, quadrotor2d_slungload.py:93: cos / sin of theta + pi/2).  libm's
// sincosf is ~45 vector instructions; this one is ~25 (quadrotor2d 64-step launches at 65 536 envs 35.7 -> 35.0 us: the integrator wavefront's
// instruction count is what its step time is made of at one pair per SIMD - a dependent chain issues as fast as an independent
// one, profiles/r04/issue_rate.md - but the Philox draws and the hand-over are most of it): k = rint(x 2/pi), r = x - k pi/2 in two fma steps (pi/2 = HI + MID, products
// exact inside the fma), 3-coefficient minimax polynomials in r^2 on [-pi/4, pi/4] (truncation 8e-9 / 6e-10), quadrant by bit
// operations.  Absolute error <= 1.2e-7 for |x| < 32 768 (tests/test_hostmath.py vs libm's fp64); beyond that, and for inf / NaN,
// libm (the angle is never wrapped, quirk Q9, but an env is long terminated before it gets there).
RMAV_HD void fast_sincosf(float x, float &sn, float &cs) {
    if (!(rabs(x) < 32768.0f)) {   // rare (and inf / NaN): the exact-reduction path of libm
        sincosf(x, &sn, &cs);
        return;
    }
    const float kf = __builtin_rintf(x * 0.63661977236758134f);
    float r = rfma(kf, -1.5707963705062866f, x);
    r = rfma(kf, 4.371138828673793e-08f, r);
    const int32_t q = (int32_t)kf;
    const float z = r * r;
    const float ps = rfma(z, rfma(z, -1.9587950374e-04f, 8.3327488974e-03f), -1.6666664183e-01f);
    const float pc = rfma(z, rfma(z, 2.4547991416e-05f, -1.3888303656e-03f), 4.1666664183e-02f);
    const float s0 = rfma(r * z, ps, r);
    const float c0 = rfma(z, rfma(z, pc, -0.5f), 1.0f);
    // quadrant q & 3: (sin, cos) = (s, c), (c, -s), (-s, -c), (-c, s)
    const bool swap = (q & 1) != 0;
    const uint32_t sb = __builtin_bit_cast(uint32_t, swap ? c0 : s0) ^ (((uint32_t)q & 2u) << 30);
    const uint32_t cb = __builtin_bit_cast(uint32_t, swap ? s0 : c0) ^ ((((uint32_t)q + 1u) & 2u) << 30);
    sn = __builtin_bit_cast(float, sb);
    cs = __builtin_bit_cast(float, cb);
}

// Quadrotor2D.step  quadrotor2d.py:74-113
template <> struct Env<QUAD2D> {
    using R = float;
    using P = ParamsT<R>;
    static RMAV_HD void step(float (&s)[5], const float (&a)[2], const ParamsT<R> &p, float &dist,
                             bool &done) {
        R thrust = p.thrust_scale * a[0];                         // :75
        if (p.clamp_thrust && thrust < R(0)) thrust = R(0);       // :76-77
        float sn, cs;
        fast_sincosf(s[2], sn, cs);
        // (cos(th+pi/2), sin(th+pi/2)) = (-sin th, cos th)      :88
        const R k = thrust * p.inv_mass;
        const R acc[2] = {rfma(k, -sn, p.gv[0]), rfma(k, cs, p.gv[1])};   // :88 ... + self.g
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const R v = s[3 + i];
            s[i] = rfma(acc[i], p.half_dt2, rfma(v, p.dt, s[i]));  // :89 (old velocity)
            s[3 + i] = rfma(acc[i], p.dt, v);                      // :90
        }
        s[2] = rfma(a[1], p.dt, s[2]);                            // :91 (no wrap)
        const R np = root(rfma(s[0], s[0], s[1] * s[1]));
        const R nv = root(rfma(s[3], s[3], s[4] * s[4]));
        done = (np > p.pos_limit) || (nv > p.vel_limit);          // :95-98 under the chosen reading
        dist = np;                                                // :102
    }
};

// Quadrotor2DSlungload.step  quadrotor2d_slungload.py:79-154  (velocity-first updates)
template <> struct Env<QUAD2D_SL> {
    using R = double;
    using P = ParamsT<R>;
    static RMAV_HD void step(float (&s)[9], const float (&a)[2], const ParamsT<R> &p, float &dist,
                             bool &done) {
        R pos[2] = {s[0], s[1]};
        R vel[2] = {s[3], s[4]};
        R lp[2] = {s[5], s[6]};
        R lv[2] = {s[7], s[8]};
        R thrust = p.thrust_scale * (R)a[0];                      // :80 (scale 1, no clamp by default)
        if (p.clamp_thrust && thrust < R(0)) thrust = R(0);
        float sn, cs;
        sincosf(s[2], &sn, &cs);   // (fast_sincosf costs the controller-driven two-wavefront integrator of this kind a spill: resource test)
        const R dir[2] = {-(R)sn, (R)cs};
        const R tv[2] = {lp[0] - pos[0], lp[1] - pos[1]};         // :92
        const R dd = rfma(tv[0], tv[0], tv[1] * tv[1]);
        const bool taut = dd >= p.L * p.L;                        // :95  |tv| >= L, without the root
        const R k = thrust * p.inv_mass;
        R acc[2] = {rfma(k, dir[0], p.gv[0]), rfma(k, dir[1], p.gv[1])};   // :107 / :128
        R la[2] = {p.gv[0], p.gv[1]};                             // :123
        if (taut) {
            const R inv_d = inv_sqrt(dd);
            const R u[2] = {tv[0] * inv_d, tv[1] * inv_d};        // :93
            const R c = p.mL * rfma(lv[0], lv[0], lv[1] * lv[1]);
            const R sc = rfma(u[0], rfma(thrust, dir[0], -c), u[1] * rfma(thrust, dir[1], -c));  // :97
            const R f = sc * p.inv_mtot;                          // :98
            la[0] = rfma(f, u[0], p.gv[0]);
            la[1] = rfma(f, u[1], p.gv[1]);
            const R tn = p.load_mass * rabs(f);                   // :102 |a_l - g| = |f u| = |f|
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = rfma(tn * u[i], p.inv_mass, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            lv[i] = rfma(la[i], p.dt, lv[i]);                                  // :99 / :124
            lp[i] = rfma(la[i], p.half_dt2, rfma(lv[i], p.dt, lp[i]));         // :100 / :125 (new v)
            vel[i] = rfma(acc[i], p.dt, vel[i]);                               // :108 / :129
            pos[i] = rfma(acc[i], p.half_dt2, rfma(vel[i], p.dt, pos[i]));     // :109 / :130 (new v)
        }
        const float th = rfma(a[1], (float)p.dt, s[2]);                        // :110 / :131
        if (taut) {                                                            // :113-115
            const R e[2] = {lp[0] - pos[0], lp[1] - pos[1]};
            const R inv_n = inv_sqrt(rfma(e[0], e[0], e[1] * e[1]));
            const R dr[2] = {e[0] * inv_n, e[1] * inv_n};
            const R pr = rfma(lv[0] - vel[0], dr[0], (lv[1] - vel[1]) * dr[1]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                lp[i] = rfma(dr[i], p.L, pos[i]);
                lv[i] = rfma(-pr, dr[i], lv[i]);
            }
        }
        s[0] = (float)pos[0]; s[1] = (float)pos[1]; s[2] = th;
        s[3] = (float)vel[0]; s[4] = (float)vel[1];
        s[5] = (float)lp[0];  s[6] = (float)lp[1];
        s[7] = (float)lv[0];  s[8] = (float)lv[1];
        const float nlp = root(rfma(s[5], s[5], s[6] * s[6]));    // fp32 norms of the stored state, as in Quadrotor2D
        const float nlv = root(rfma(s[7], s[7], s[8] * s[8]));
        done = (nlp > (float)p.pos_limit) || (nlv > (float)p.vel_limit);   // :136-140 load pos, load vel
        dist = root(rfma(s[0], s[0], s[1] * s[1]));               // :143 reward = -|quad pos|
    }
};

// ---- geometric controllers (always fp64) -----------------------------------------------------------

// Quadrotor3D.control  quadrotor3d.py:126-180  (= quadrotor3d_slungload.py:169-226)
template <int NS>
RMAV_HD void control_3d(const float (&s)[NS], const ParamsT<double> &p, float (&a)[4]) {
    using R = double;
    R ad[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)                                   // :155-162
        ad[i] = rfma(p.kp, (R)s[i] - p.ref_pos[i], p.kv * ((R)s[7 + i] - p.ref_vel[i])) + p.ff[i];   // :162 ... - self.g
    // acc2quat :127-141 ; yc = (0,1,0)
    const R inv_n = inv_sqrt(rfma(ad[0], ad[0], rfma(ad[1], ad[1], ad[2] * ad[2])));
    R zb[3] = {ad[0] * inv_n, ad[1] * inv_n, ad[2] * inv_n};
    R xb[3] = {zb[2], R(0), -zb[0]};                              // cross(yc, zb)
    const R inv_x = inv_sqrt(rfma(xb[0], xb[0], xb[2] * xb[2]));
    xb[0] *= inv_x;
    xb[2] *= inv_x;
    const R yb[3] = {rfma(zb[1], xb[2], -(zb[2] * xb[1])), rfma(zb[2], xb[0], -(zb[0] * xb[2])),
                     rfma(zb[0], xb[1], -(zb[1] * xb[0]))};       // cross(zb, xb)
    const R inv_z = inv_sqrt(rfma(zb[0], zb[0], rfma(zb[1], zb[1], zb[2] * zb[2])));
#pragma unroll
    for (int i = 0; i < 3; ++i) zb[i] *= inv_z;
    // Quaternion(matrix=[xb yb zb]) : trace method on m = R^T  (m[i][j] = R[j][i])
    // m00 = xb0, m01 = xb1, m02 = xb2 ; m10 = yb0, ... ; m20 = zb0, ...
    const R m00 = xb[0], m01 = xb[1], m02 = xb[2];
    const R m10 = yb[0], m11 = yb[1], m12 = yb[2];
    const R m20 = zb[0], m21 = zb[1], m22 = zb[2];
    R t, qd[4];
    if (m22 < R(0)) {
        if (m00 > m11) {
            t = R(1) + m00 - m11 - m22;
            qd[0] = m12 - m21; qd[1] = t; qd[2] = m01 + m10; qd[3] = m20 + m02;
        } else {
            t = R(1) - m00 + m11 - m22;
            qd[0] = m20 - m02; qd[1] = m01 + m10; qd[2] = t; qd[3] = m12 + m21;
        }
    } else {
        if (m00 < -m11) {
            t = R(1) - m00 - m11 + m22;
            qd[0] = m01 - m10; qd[1] = m20 + m02; qd[2] = m12 + m21; qd[3] = t;
        } else {
            t = R(1) + m00 + m11 + m22;
            qd[0] = t; qd[1] = m12 - m21; qd[2] = m20 - m02; qd[3] = m01 - m10;
        }
    }
    const R kq = R(0.5) * inv_sqrt(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) qd[i] *= kq;
    // error_att = conj(q_raw) (x) q_des   :169   (the stored quaternion is NOT normalised here)
    const R cw = s[3], cx = -(R)s[4], cy = -(R)s[5], cz = -(R)s[6];
    const R qe0 = rfma(cw, qd[0], -rfma(cx, qd[1], rfma(cy, qd[2], cz * qd[3])));
    const R qe1 = rfma(cx, qd[0], rfma(cw, qd[1], rfma(cy, qd[3], -(cz * qd[2]))));
    const R qe2 = rfma(cy, qd[0], rfma(cz, qd[1], rfma(cw, qd[2], -(cx * qd[3]))));
    const R qe3 = rfma(cz, qd[0], rfma(cx, qd[2], rfma(cw, qd[3], -(cy * qd[1]))));
    // np.sign: +1 / -1 / 0, NaN propagates
    const R sg = (qe0 > R(0)) ? R(1) : ((qe0 < R(0)) ? R(-1) : qe0);
    const R kw = p.two_over_tau * sg;                             // :173
    const R q[4] = {s[3], s[4], s[5], s[6]};
    R qn[4], b[3];
    quat_normalise(q, qn);
    quat_body_z(qn, b);
    a[0] = (float)rfma(ad[0], b[0], rfma(ad[1], b[1], ad[2] * b[2]));   // :176
    a[1] = (float)(kw * qe1);
    a[2] = (float)(kw * qe2);
    a[3] = (float)(kw * qe3);
}

// atan2 for the 2-D controller, absolute error <= 3e-13 (checked against libm over 4e5 random arguments and the axes;
// the result is multiplied by 1 / tau = 10 and rounded to fp32, so 1e-11 would do).  libm's correctly rounded fp64 atan2 is
// ~70 fp64 instructions - one IEEE division, a degree-19 polynomial whose 19 coefficients hipcc parks in 38 vector registers
// - and was half of the controller-driven 2-D step.  Here: ONE reciprocal (v_rcp_f64 + two Newton steps) of a reduced
// argument |u| <= tan(pi / 8) - atan(t) = pi / 4 + atan((t - 1) / (t + 1)) folded into the quotient's numerator and
// denominator - and a degree-7 polynomial in u^2 (Chebyshev-node fit of atan(sqrt z) / sqrt z on [0, tan^2(pi / 8)]).
RMAV_HD double fast_atan2(double y, double x) {
    const double ax = rabs(x), ay = rabs(y);
    const double mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    const bool big = mn > 0.41421356237309503 * mx;
    const double num = big ? mn - mx : mn;
    double den = big ? mn + mx : mx;
    den = den > 0.0 ? den : 1.0;                           // atan2(0, 0) = 0
#if defined(__HIP_DEVICE_COMPILE__)
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-den, rc, 1.0), rc, rc);
    const double u = num * rc;
#else
    const double u = num / den;
#endif
    const double z = u * u;
    double p = -0.03765508211963291;
    p = rfma(p, z, 0.0697419671381827);
    p = rfma(p, z, -0.08992552776397131);
    p = rfma(p, z, 0.11103456904403586);
    p = rfma(p, z, -0.14285386554069893);
    p = rfma(p, z, 0.19999993053581264);
    p = rfma(p, z, -0.33333333276922894);
    p = rfma(p, z, 0.9999999999992449);
    double r = rfma(u, p, big ? 0.7853981633974483 : 0.0);
    r = ay > ax ? 1.5707963267948966 - r : r;
    r = x < 0.0 ? 3.141592653589793 - r : r;
    return y < 0.0 ? -r : r;
}

// Quadrotor2D.control  quadrotor2d.py:115-138  (= quadrotor2d_slungload.py:156-183)
template <int NS>
RMAV_HD void control_2d(const float (&s)[NS], const ParamsT<double> &p, float (&a)[2]) {
    using R = double;
    // :130 "+ np.array([0.0, 9.8])": a literal in the reference, NOT self.g - ff carries it
    const R ax = rfma(p.kp, (R)s[0] - p.ref_pos[0], p.kv * ((R)s[3] - p.ref_vel[0])) + p.ff[0];
    const R ay = rfma(p.kp, (R)s[1] - p.ref_pos[1], p.kv * ((R)s[4] - p.ref_vel[1])) + p.ff[1];
    const R th_d = fast_atan2(ay, ax) - R(1.5707963267948966);    // :131
    a[1] = (float)(p.neg_inv_tau * ((R)s[2] - th_d));             // :132-133
    a[0] = (float)(p.mass * root(rfma(ax, ax, ay * ay)));   // :134
}

// ================================================================================================
// ReinmavEnv  (reinmav_env.py): 13-state rigid body  [x y z dx dy dz qw qx qy qz p q r]  (:79),
// thrust + body torques -> motor mixing with clamp -> linear / angular acceleration (:203-264), a
// built-in PD position/attitude controller tracking a min-jerk trajectory (:128-136, :306-337), 50-or-51
// explicit-Euler sub-steps of 1/5000 s per step (:90-98).  The reference's step() takes no action.
// fp64 arithmetic on fp32 state: the attitude loop has gain kp_rot/I ~ 4e5 1/s^2, so fp32 rounding of the
// Euler angles would show up as 1e-4 rad/s in the body rates.
// ================================================================================================
struct ReinmavP {
    double arm_length, mass, gravity, min_force4, max_force4;  // force limits per rotor (:59, :212)
    double inertia[3][3], inv_inertia[3][3];
    double dt, ds, t_max;
    double kp[3], kd[3], kp_rot[3], kd_rot[3];
    int32_t rk4;   // 0: explicit Euler sub-steps (the reference), 1: RK4 sub-steps
    int32_t _pad;
};

RMAV_HD void reinmav_quat2mat(const double (&q)[4], double (&m)[3][3]) {   // quat2mat :267-290
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double Nq = rfma(w, w, rfma(x, x, rfma(y, y, z * z)));
    if (!(Nq > 2.220446049250313e-16)) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m[i][j] = (i == j) ? 1.0 : 0.0;
        return;
    }
    const double sc = 2.0 / Nq;
    const double X = x * sc, Y = y * sc, Z = z * sc;
    const double wX = w * X, wY = w * Y, wZ = w * Z;
    const double xX = x * X, xY = x * Y, xZ = x * Z;
    const double yY = y * Y, yZ = y * Z, zZ = z * Z;
    m[0][0] = 1.0 - (yY + zZ); m[0][1] = xY - wZ;         m[0][2] = xZ + wY;
    m[1][0] = xY + wZ;         m[1][1] = 1.0 - (xX + zZ); m[1][2] = yZ - wX;
    m[2][0] = xZ - wY;         m[2][1] = yZ + wX;         m[2][2] = 1.0 - (xX + yY);
}

// trj_gen (:128-136) + stateToQd / RotToRPY (:292-304, :341-346) + controller (:306-337) -> (F, Mx, My, Mz)
// R = quat2mat(attitude) is passed in: the controller and the dynamics of one sub-step use the same matrix.
RMAV_HD void reinmav_controller(const ReinmavP &p, const double (&s)[13], const double (&R)[3][3], double t,
                                double (&fm)[4]) {
    // RotToRPY :341-346.  phi = asin(R12) lies in [-pi/2, pi/2], so cos(phi) = sqrt(1 - R12^2) >= 0, and dividing
    // both atan2 arguments by that positive number (as the reference does) does not change the angle: the two
    // divisions and the cos() are dropped (differences at the 1e-16 level, far below the 1e-6 parity bar; at
    // cos(phi) == 0 the reference divides by zero and returns NaN/pi/2 - a gimbal-lock state it never reaches).
    const double phi = asin(R[1][2]);
    const double psi = atan2(-R[1][0], R[1][1]);
    const double theta = atan2(-R[0][2], R[2][2]);
    const double tm = p.t_max;
    double u = t < tm ? t : tm;
    u = (u > 0.0 ? u : 0.0) / tm;
    const double u2 = u * u, u3 = u2 * u, u4 = u2 * u2, u5 = u4 * u;
    const double pos = 10.0 * u3 - 15.0 * u4 + 6.0 * u5;
    const double vel = (30 / tm) * u2 - (60 / tm) * u3 + (30 / tm) * u4;
    const double acc = (60 / (tm * tm)) * u - (180 / (tm * tm)) * u2 + (120 / (tm * tm)) * u3;
    double ddr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ddr[i] = acc + p.kd[i] * (vel - s[3 + i]) + p.kp[i] * (pos - s[i]);
    const double psi_des = pos, dpsi_des = vel;
    double sp, cp;
    sincos(psi_des, &sp, &cp);
    const double phi_des = 1 / p.gravity * (ddr[0] * sp - ddr[1] * cp);
    const double theta_des = 1 / p.gravity * (ddr[0] * cp + ddr[1] * sp);
    fm[0] = p.mass * (p.gravity + ddr[2]);
    fm[1] = p.kp_rot[0] * (phi_des - phi) - p.kd_rot[0] * s[10];
    fm[2] = p.kp_rot[1] * (theta_des - theta) - p.kd_rot[1] * s[11];
    fm[3] = p.kp_rot[2] * (psi_des - psi) + p.kd_rot[2] * (dpsi_des - s[12]);
}

// quad_eq_of_motion2 (:203-264): sdot = f(s, F, M)
RMAV_HD void reinmav_derivative(const ReinmavP &p, const double (&s)[13], const double (&bRw)[3][3], const double (&fm)[4],
                                double (&sd)[13]) {
    const double L = p.arm_length, k = 0.5 / L;
    double T[4] = {0.25 * fm[0] - k * fm[2], 0.25 * fm[0] + k * fm[1], 0.25 * fm[0] + k * fm[2], 0.25 * fm[0] - k * fm[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // np.maximum(np.minimum(T, max/4), min/4)
        T[i] = T[i] < p.max_force4 ? T[i] : p.max_force4;
        T[i] = T[i] > p.min_force4 ? T[i] : p.min_force4;
    }
    const double force = T[0] + T[1] + T[2] + T[3];
    const double mom[3] = {L * T[1] - L * T[3], L * T[2] - L * T[0], fm[3]};
    const double q[4] = {s[6], s[7], s[8], s[9]};
    const double im = 1.0 / p.mass;
    const double acc[3] = {im * (bRw[2][0] * force), im * (bRw[2][1] * force), im * (bRw[2][2] * force - p.mass * p.gravity)};
    const double pw = s[10], qw = s[11], rw = s[12];
    const double qerr = 2.0 * (1.0 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));   // K_quat = 2 (:242)
    const double qd[4] = {-0.5 * (-pw * q[1] - qw * q[2] - rw * q[3]) + qerr * q[0],
                          -0.5 * (pw * q[0] - rw * q[2] + qw * q[3]) + qerr * q[1],
                          -0.5 * (qw * q[0] + rw * q[1] - pw * q[3]) + qerr * q[2],
                          -0.5 * (rw * q[0] - qw * q[1] + pw * q[2]) + qerr * q[3]};
    const double w[3] = {pw, qw, rw};
    double Iw[3], rhs[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Iw[i] = p.inertia[i][0] * w[0] + p.inertia[i][1] * w[1] + p.inertia[i][2] * w[2];
    rhs[0] = mom[0] - (w[1] * Iw[2] - w[2] * Iw[1]);
    rhs[1] = mom[1] - (w[2] * Iw[0] - w[0] * Iw[2]);
    rhs[2] = mom[2] - (w[0] * Iw[1] - w[1] * Iw[0]);
    sd[0] = s[3]; sd[1] = s[4]; sd[2] = s[5];
#pragma unroll
    for (int i = 0; i < 3; ++i) sd[3 + i] = acc[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) sd[6 + i] = qd[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        sd[10 + i] = p.inv_inertia[i][0] * rhs[0] + p.inv_inertia[i][1] * rhs[1] + p.inv_inertia[i][2] * rhs[2];
}

// one sub-step of length ds with the command held:  explicit Euler (the reference, :98) or classical RK4
RMAV_HD void reinmav_substep(const ReinmavP &p, double (&s)[13], const double (&R)[3][3], const double (&fm)[4]) {
    const double ds = p.ds;
    double k1[13];
    reinmav_derivative(p, s, R, fm, k1);
    if (!p.rk4) {
#pragma unroll
        for (int i = 0; i < 13; ++i) s[i] = rfma(ds, k1[i], s[i]);
        return;
    }
    double k2[13], k3[13], k4[13], y[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) y[i] = rfma(0.5 * ds, k1[i], s[i]);
    double Ry[3][3];
    {
        const double qy[4] = {y[6], y[7], y[8], y[9]};
        reinmav_quat2mat(qy, Ry);
    }
    reinmav_derivative(p, y, Ry, fm, k2);
#pragma unroll
    for (int i = 0; i < 13; ++i) y[i] = rfma(0.5 * ds, k2[i], s[i]);
    {
        const double qy[4] = {y[6], y[7], y[8], y[9]};
        reinmav_quat2mat(qy, Ry);
    }
    reinmav_derivative(p, y, Ry, fm, k3);
#pragma unroll
    for (int i = 0; i < 13; ++i) y[i] = rfma(ds, k3[i], s[i]);
    {
        const double qy[4] = {y[6], y[7], y[8], y[9]};
        reinmav_quat2mat(qy, Ry);
    }
    reinmav_derivative(p, y, Ry, fm, k4);
#pragma unroll
    for (int i = 0; i < 13; ++i) s[i] = rfma(ds / 6.0, (k1[i] + 2.0 * k2[i]) + (2.0 * k3[i] + k4[i]), s[i]);
}

template <> struct Env<REINMAV> {
    using R = double;
    using P = ReinmavP;
    // step() :99-126.  use_controller: evaluate the built-in controller at every sub-step (the reference);
    // otherwise hold the caller's (F, Mx, My, Mz) over the step.  fm0 returns the command of sub-step 0.
    static RMAV_HD void step(float (&sf)[13], const float (&a)[4], bool use_controller, double &t, const ReinmavP &p,
                             float (&fm0)[4]) {
        double s[13];
#pragma unroll
        for (int i = 0; i < 13; ++i) s[i] = sf[i];
        // np.arange(t, t+dt, ds): ceil((stop-start)/step) values  start + i*delta, delta = (start+step)-start
        const double start = t, stop = t + p.dt;
        int n = (int)ceil((stop - start) / p.ds);
        const double delta = (start + p.ds) - start;
        double fm[4] = {a[0], a[1], a[2], a[3]};
        for (int i = 0; i < n; ++i) {
            double R[3][3];
            {
                const double q[4] = {s[6], s[7], s[8], s[9]};
                reinmav_quat2mat(q, R);
            }
            if (use_controller) {
                const double ti = (i == 0) ? start : ((i == 1) ? start + p.ds : start + i * delta);
                reinmav_controller(p, s, R, ti, fm);
            }
            if (i == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) fm0[c] = (float)fm[c];
            }
            reinmav_substep(p, s, R, fm);
        }
        t = t + p.dt;   // :119
#pragma unroll
        for (int i = 0; i < 13; ++i) sf[i] = (float)s[i];
    }
};

template <int K>
RMAV_HD void env_control(const float (&s)[Dims<K>::NS], const ParamsT<double> &p,
                         float (&a)[Dims<K>::NA]) {
    if constexpr (K == QUAD3D || K == QUAD3D_SL) control_3d<Dims<K>::NS>(s, p, a);
    else if constexpr (K == QUAD2D || K == QUAD2D_SL) control_2d<Dims<K>::NS>(s, p, a);
}

}  // namespace rmav
