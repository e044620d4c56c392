/* rmav_oracle.c - fp64 CPU restatement of reinmav-gym's native quadrotor step()/control()/reset().
 *
 * TEST INFRASTRUCTURE ONLY - see rmav_oracle.h.  Reference line citations are relative to
 * gym_reinmav/envs/native/ of the reference repository.
 */
#include "rmav_oracle.h"

#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const int k_state_dim[4] = {5, 9, 10, 16};
static const int k_action_dim[4] = {2, 2, 4, 4};

int oracle_state_dim(int kind) { return (kind < 0 || kind > 3) ? -1 : k_state_dim[kind]; }
int oracle_action_dim(int kind) { return (kind < 0 || kind > 3) ? -1 : k_action_dim[kind]; }

int oracle_default_params(int kind, int reading_2d, oracle_params *p) {
    if (kind < 0 || kind > 3 || !p) return -1;
    memset(p, 0, sizeof(*p));
    p->mass = 1.0;      /* quadrotor3d.py:45, quadrotor2d.py:44 */
    p->load_mass = 0.1; /* quadrotor3d_slungload.py:46, quadrotor2d_slungload.py:45 */
    p->dt = 0.01;       /* quadrotor3d.py:46 */
    p->g = 9.8;
    p->g_vec[kind <= ORACLE_QUAD2D_SL ? 1 : 2] = -9.8; /* quadrotor3d.py:47 g=(0,0,-9.8); quadrotor2d.py:46 g=(0,-9.8) */
    p->thrust_scale = 1.0;
    p->clamp_thrust = 0;
    p->kp = -5.0; /* quadrotor3d.py:143, quadrotor2d.py:116 */
    p->kv = -4.0; /* quadrotor3d.py:144, quadrotor2d.py:117 */
    switch (kind) {
    case ORACLE_QUAD2D:
        /* quadrotor2d.py:95-98 (does not parse as shipped; see ref_harness.py).
         * reading B: |p|>3.0 or |v|>10.0 or |v|>vel_threshold(2.0)  ==  |p|>3 or |v|>2
         * reading A: |p|>3.0 or |v|>10.0 */
        p->pos_limit = 3.0;
        p->vel_limit = (reading_2d == 'A') ? 10.0 : 2.0;
        p->thrust_scale = 10.0; /* quadrotor2d.py:75 */
        p->clamp_thrust = 1;    /* quadrotor2d.py:76-77 */
        p->tau = 0.1;           /* quadrotor2d.py:118 */
        break;
    case ORACLE_QUAD2D_SL:
        p->tether_length = 0.5; /* quadrotor2d_slungload.py:53 */
        p->pos_limit = 2.0;     /* quadrotor2d_slungload.py:56 */
        p->vel_limit = 10.0;    /* quadrotor2d_slungload.py:57 */
        p->tau = 0.1;           /* quadrotor2d_slungload.py:159 */
        break;
    case ORACLE_QUAD3D:
        p->pos_limit = 3.0;  /* quadrotor3d.py:55 */
        p->vel_limit = 10.0; /* quadrotor3d.py:56 */
        p->ref_pos[2] = 2.0; /* quadrotor3d.py:51 */
        p->tau = 0.3;        /* quadrotor3d.py:145 */
        break;
    case ORACLE_QUAD3D_SL:
        p->tether_length = 1.5; /* quadrotor3d_slungload.py:58 */
        p->pos_limit = 3.0;     /* quadrotor3d_slungload.py:55 */
        p->vel_limit = 10.0;    /* quadrotor3d_slungload.py:56 */
        p->ref_pos[2] = 1.0;    /* quadrotor3d_slungload.py:52 */
        p->tau = 0.3;           /* quadrotor3d_slungload.py:187 */
        break;
    }
    return 0;
}

/* ---- pyquaternion semantics used by the 3-D files (published 0.9.x algorithm) ---------------- */

/* Quaternion._normalise(): q <- q/|q| unless |1-|q|^2| < 1e-14 or |q| == 0. */
static void quat_normalise(const double q[4], double out[4]) {
    double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    if (!(fabs(1.0 - n2) < 1e-14)) {
        double n = sqrt(n2);
        if (n > 0) {
            for (int i = 0; i < 4; ++i) out[i] = q[i] / n;
            return;
        }
    }
    for (int i = 0; i < 4; ++i) out[i] = q[i];
}

/* Quaternion.rotation_matrix.dot([0,0,1]) for an already-normalised q: third column of
 * (Q . Qbar^T)[1:,1:], summed in the k=0..3 order np.dot uses. */
static void quat_body_z(const double q[4], double b3[3]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    b3[0] = x * z + w * y + (-z) * (-x) + y * w;
    b3[1] = y * z + z * y + w * (-x) + (-x) * w;
    b3[2] = z * z + (-y) * y + x * (-x) + w * w;
}

/* Quaternion.__mul__: _q_matrix(a) . b  (Hamilton product). */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
    double w = a[0], x = a[1], y = a[2], z = a[3];
    o[0] = w * b[0] + (-x) * b[1] + (-y) * b[2] + (-z) * b[3];
    o[1] = x * b[0] + w * b[1] + (-z) * b[2] + y * b[3];
    o[2] = y * b[0] + z * b[1] + w * b[2] + (-x) * b[3];
    o[3] = z * b[0] + (-y) * b[1] + x * b[2] + w * b[3];
}

/* Quaternion.derivative(rate) = 0.5 * self * Quaternion(vector=rate). */
static void quat_derivative(const double qn[4], const double rate[3], double o[4]) {
    double half[4] = {0.5, 0.0, 0.0, 0.0}, h[4], r[4] = {0.0, rate[0], rate[1], rate[2]};
    quat_mul(half, qn, h);
    quat_mul(h, r, o);
}

/* Quaternion(matrix=R): validation (rtol 1e-5, atol 1e-8) then the 4-branch trace method on R^T.
 * Returns 0 on success, -1 where pyquaternion raises ValueError (output filled with NaN). */
static int quat_from_matrix(const double R[3][3], double q[4]) {
    /* np.allclose(R.R^T, I) and np.isclose(det, 1) */
    int ok = 1;
    for (int i = 0; i < 3 && ok; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = R[i][0] * R[j][0] + R[i][1] * R[j][1] + R[i][2] * R[j][2];
            double e = (i == j) ? 1.0 : 0.0;
            if (!(fabs(v - e) <= 1e-8 + 1e-5 * fabs(e))) ok = 0;
        }
    double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) -
                 R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                 R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (!(fabs(det - 1.0) <= 1e-8 + 1e-5)) ok = 0;
    if (!ok) {
        q[0] = q[1] = q[2] = q[3] = NAN;
        return -1;
    }
    double m[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = R[j][i];
    double t;
    if (m[2][2] < 0) {
        if (m[0][0] > m[1][1]) {
            t = 1 + m[0][0] - m[1][1] - m[2][2];
            q[0] = m[1][2] - m[2][1]; q[1] = t; q[2] = m[0][1] + m[1][0]; q[3] = m[2][0] + m[0][2];
        } else {
            t = 1 - m[0][0] + m[1][1] - m[2][2];
            q[0] = m[2][0] - m[0][2]; q[1] = m[0][1] + m[1][0]; q[2] = t; q[3] = m[1][2] + m[2][1];
        }
    } else {
        if (m[0][0] < -m[1][1]) {
            t = 1 - m[0][0] - m[1][1] + m[2][2];
            q[0] = m[0][1] - m[1][0]; q[1] = m[2][0] + m[0][2]; q[2] = m[1][2] + m[2][1]; q[3] = t;
        } else {
            t = 1 + m[0][0] + m[1][1] + m[2][2];
            q[0] = t; q[1] = m[1][2] - m[2][1]; q[2] = m[2][0] - m[0][2]; q[3] = m[0][1] - m[1][0];
        }
    }
    double k = 0.5 / sqrt(t);
    for (int i = 0; i < 4; ++i) q[i] *= k;
    return 0;
}

static double norm2(const double *v, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}
static double inner(const double *a, const double *b, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* Reward / steps_beyond_done state machine, identical in all four files
 * (quadrotor3d.py:112-122, quadrotor3d_slungload.py:155-165, quadrotor2d.py:101-111,
 * quadrotor2d_slungload.py:142-152).  reset() never clears steps_beyond_done, so the terminal
 * reward 1.0 is paid once per env object lifetime. */
static double reward_machine(int done, double dist, int *sbd) {
    if (!done) return -dist;
    if (*sbd < 0) {
        *sbd = 0;
        return 1.0;
    }
    *sbd += 1;
    return 0.0;
}

/* ---- Quadrotor3D.step  quadrotor3d.py:81-124 --------------------------------------------------- */
static void step_quad3d(const oracle_params *p, const double *s, const double *a, double *o,
                        double *reward, int *done, int *sbd) {
    const double dt = p->dt;
    double thrust = a[0];                       /* :82 */
    const double *w = a + 1;                    /* :83 */
    double pos[3] = {s[0], s[1], s[2]};         /* :89 */
    double att[4] = {s[3], s[4], s[5], s[6]};   /* :90 */
    double vel[3] = {s[7], s[8], s[9]};         /* :91 */
    const double *g = p->g_vec;                /* self.g */
    double qn[4], b3[3], acc[3], qd[4];
    quat_normalise(att, qn);                    /* :96 rotation_matrix normalises the Quaternion */
    quat_body_z(qn, b3);
    for (int i = 0; i < 3; ++i) acc[i] = thrust / p->mass * b3[i] + g[i];                 /* :96 */
    for (int i = 0; i < 3; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;   /* :98 */
    for (int i = 0; i < 3; ++i) vel[i] = vel[i] + acc[i] * dt;                            /* :99 */
    quat_derivative(qn, w, qd);                 /* :101 (normalised q) */
    for (int i = 0; i < 4; ++i) att[i] = att[i] + qd[i] * dt; /* :102 (raw att) */
    o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
    o[3] = att[0]; o[4] = att[1]; o[5] = att[2]; o[6] = att[3];
    o[7] = vel[0]; o[8] = vel[1]; o[9] = vel[2];
    double np_ = norm2(pos, 3), nv = norm2(vel, 3);
    *done = (np_ < -p->pos_limit) || (np_ > p->pos_limit) || (nv < -p->vel_limit) ||
            (nv > p->vel_limit);                /* :106-110 */
    *reward = reward_machine(*done, np_, sbd);  /* :112-122 */
}

/* ---- Quadrotor3DSlungload.step  quadrotor3d_slungload.py:87-167 ------------------------------- */
static void step_quad3d_sl(const oracle_params *p, const double *s, const double *a, double *o,
                           double *reward, int *done, int *sbd, int force_taut) {
    const double dt = p->dt, L = p->tether_length;
    double thrust = a[0];
    const double *w = a + 1;
    double pos[3] = {s[0], s[1], s[2]};
    double att[4] = {s[3], s[4], s[5], s[6]};
    double vel[3] = {s[7], s[8], s[9]};
    double lp[3] = {s[10], s[11], s[12]};
    double lv[3] = {s[13], s[14], s[15]};
    const double *g = p->g_vec;                /* self.g */
    double tv[3], u[3], qn[4], b3[3], acc[3], qd[4], la[3];
    for (int i = 0; i < 3; ++i) tv[i] = lp[i] - pos[i];          /* :101 */
    double d = norm2(tv, 3);
    for (int i = 0; i < 3; ++i) u[i] = tv[i] / d;                /* :102 */
    quat_normalise(att, qn);
    quat_body_z(qn, b3);
    if (force_taut < 0 ? (d >= L) : force_taut) {                /* :104 taut */
        double thr_vec[3], tmp[3], T[3], ldir[3], dlp[3], dv[3];
        for (int i = 0; i < 3; ++i) thr_vec[i] = thrust * b3[i]; /* :109 */
        double c = p->mass * L * inner(lv, lv, 3);
        for (int i = 0; i < 3; ++i) tmp[i] = thr_vec[i] - c;     /* :110 vector - scalar */
        double sc = inner(u, tmp, 3);
        for (int i = 0; i < 3; ++i) la[i] = sc * u[i];
        for (int i = 0; i < 3; ++i) la[i] = (1 / (p->mass + p->load_mass)) * la[i] + g[i]; /* :111 */
        for (int i = 0; i < 3; ++i) lp[i] = lp[i] + lv[i] * dt + 0.5 * la[i] * dt * dt;    /* :112 */
        for (int i = 0; i < 3; ++i) lv[i] = lv[i] + la[i] * dt;                            /* :113 */
        for (int i = 0; i < 3; ++i) tmp[i] = -g[i] + la[i];
        double tn = norm2(tmp, 3);
        for (int i = 0; i < 3; ++i) T[i] = p->load_mass * tn * u[i];                       /* :115 */
        for (int i = 0; i < 3; ++i) acc[i] = thrust / p->mass * b3[i] + g[i] + T[i] / p->mass; /* :118 */
        for (int i = 0; i < 3; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;
        for (int i = 0; i < 3; ++i) vel[i] = vel[i] + acc[i] * dt;
        quat_derivative(qn, w, qd);
        for (int i = 0; i < 4; ++i) att[i] = att[i] + qd[i] * dt;                          /* :123 */
        for (int i = 0; i < 3; ++i) dlp[i] = lp[i] - pos[i];
        double dn = norm2(dlp, 3);
        for (int i = 0; i < 3; ++i) ldir[i] = dlp[i] / dn;                                 /* :126 */
        for (int i = 0; i < 3; ++i) lp[i] = pos[i] + ldir[i] * L;                          /* :127 */
        for (int i = 0; i < 3; ++i) dv[i] = lv[i] - vel[i];
        double pr = inner(dv, ldir, 3);
        for (int i = 0; i < 3; ++i) lv[i] = lv[i] - pr * ldir[i];                          /* :128 */
    } else {                                                     /* :131 slack */
        for (int i = 0; i < 3; ++i) la[i] = g[i];
        for (int i = 0; i < 3; ++i) lp[i] = lp[i] + lv[i] * dt + 0.5 * la[i] * dt * dt;    /* :136 */
        for (int i = 0; i < 3; ++i) lv[i] = lv[i] + la[i] * dt;
        for (int i = 0; i < 3; ++i) acc[i] = thrust / p->mass * b3[i] + g[i];              /* :140 */
        for (int i = 0; i < 3; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;
        for (int i = 0; i < 3; ++i) vel[i] = vel[i] + acc[i] * dt;
        quat_derivative(qn, w, qd);
        for (int i = 0; i < 4; ++i) att[i] = att[i] + qd[i] * dt;
    }
    o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
    o[3] = att[0]; o[4] = att[1]; o[5] = att[2]; o[6] = att[3];
    o[7] = vel[0]; o[8] = vel[1]; o[9] = vel[2];
    o[10] = lp[0]; o[11] = lp[1]; o[12] = lp[2];
    o[13] = lv[0]; o[14] = lv[1]; o[15] = lv[2];
    double nlp = norm2(lp, 3), nv = norm2(vel, 3);
    *done = (nlp < -p->pos_limit) || (nlp > p->pos_limit) || (nv < -p->vel_limit) ||
            (nv > p->vel_limit);                 /* :149-153 load position, quad velocity */
    *reward = reward_machine(*done, nlp, sbd);   /* :155-165 reward -|load_pos| */
}

/* ---- Quadrotor2D.step  quadrotor2d.py:74-113 ---------------------------------------------------- */
static void step_quad2d(const oracle_params *p, const double *s, const double *a, double *o,
                        double *reward, int *done, int *sbd) {
    const double dt = p->dt;
    double thrust = p->thrust_scale * a[0];               /* :75 */
    if (p->clamp_thrust && thrust < 0.0) thrust = 0.0;    /* :76-77 */
    double w = a[1];                                      /* :78 */
    double pos[2] = {s[0], s[1]}, att = s[2], vel[2] = {s[3], s[4]};
    const double *g = p->g_vec;                /* self.g (2 components) */
    double dir[2] = {cos(att + M_PI / 2), sin(att + M_PI / 2)};
    double acc[2];
    for (int i = 0; i < 2; ++i) acc[i] = thrust / p->mass * dir[i] + g[i];               /* :88 */
    for (int i = 0; i < 2; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;  /* :89 */
    for (int i = 0; i < 2; ++i) vel[i] = vel[i] + acc[i] * dt;                           /* :90 */
    att = att + w * dt;                                                                  /* :91 */
    o[0] = pos[0]; o[1] = pos[1]; o[2] = att; o[3] = vel[0]; o[4] = vel[1];
    double np_ = norm2(pos, 2), nv = norm2(vel, 2);
    *done = (np_ > p->pos_limit) || (nv > p->vel_limit);  /* :95-98 under the chosen reading */
    *reward = reward_machine(*done, np_, sbd);            /* :101-111 */
}

/* ---- Quadrotor2DSlungload.step  quadrotor2d_slungload.py:79-154 (velocity-first updates) ------- */
static void step_quad2d_sl(const oracle_params *p, const double *s, const double *a, double *o,
                           double *reward, int *done, int *sbd, int force_taut) {
    const double dt = p->dt, L = p->tether_length;
    double thrust = p->thrust_scale * a[0];               /* :80 (scale 1: no x10, no clamp) */
    if (p->clamp_thrust && thrust < 0.0) thrust = 0.0;
    double w = a[1];
    double pos[2] = {s[0], s[1]}, att = s[2], vel[2] = {s[3], s[4]};
    double lp[2] = {s[5], s[6]}, lv[2] = {s[7], s[8]};
    const double *g = p->g_vec;                /* self.g (2 components) */
    double tv[2], u[2], la[2], acc[2];
    for (int i = 0; i < 2; ++i) tv[i] = lp[i] - pos[i];   /* :92 */
    double d = norm2(tv, 2);
    for (int i = 0; i < 2; ++i) u[i] = tv[i] / d;         /* :93 */
    double dir[2] = {cos(att + M_PI / 2), sin(att + M_PI / 2)};
    if (force_taut < 0 ? (d >= L) : force_taut) {         /* :95 taut */
        double thr_vec[2], tmp[2], T[2], ldir[2], dlp[2], dv[2];
        for (int i = 0; i < 2; ++i) thr_vec[i] = thrust * dir[i];                          /* :96 */
        double c = p->mass * L * inner(lv, lv, 2);
        for (int i = 0; i < 2; ++i) tmp[i] = thr_vec[i] - c;                               /* :97 */
        double sc = inner(u, tmp, 2);
        for (int i = 0; i < 2; ++i) la[i] = sc * u[i];
        for (int i = 0; i < 2; ++i) la[i] = (1 / (p->mass + p->load_mass)) * la[i] + g[i]; /* :98 */
        for (int i = 0; i < 2; ++i) lv[i] = lv[i] + la[i] * dt;                            /* :99 */
        for (int i = 0; i < 2; ++i) lp[i] = lp[i] + lv[i] * dt + 0.5 * la[i] * dt * dt;    /* :100 */
        for (int i = 0; i < 2; ++i) tmp[i] = -g[i] + la[i];
        double tn = norm2(tmp, 2);
        for (int i = 0; i < 2; ++i) T[i] = p->load_mass * tn * u[i];                       /* :102 */
        for (int i = 0; i < 2; ++i) acc[i] = thrust / p->mass * dir[i] + g[i] + T[i] / p->mass; /* :107 */
        for (int i = 0; i < 2; ++i) vel[i] = vel[i] + acc[i] * dt;                         /* :108 */
        for (int i = 0; i < 2; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt; /* :109 */
        att = att + w * dt;
        for (int i = 0; i < 2; ++i) dlp[i] = lp[i] - pos[i];
        double dn = norm2(dlp, 2);
        for (int i = 0; i < 2; ++i) ldir[i] = dlp[i] / dn;                                 /* :113 */
        for (int i = 0; i < 2; ++i) lp[i] = pos[i] + ldir[i] * L;                          /* :114 */
        for (int i = 0; i < 2; ++i) dv[i] = lv[i] - vel[i];
        double pr = inner(dv, ldir, 2);
        for (int i = 0; i < 2; ++i) lv[i] = lv[i] - pr * ldir[i];                          /* :115 */
    } else {                                              /* :118 slack */
        for (int i = 0; i < 2; ++i) la[i] = g[i];
        for (int i = 0; i < 2; ++i) lv[i] = lv[i] + la[i] * dt;                            /* :124 */
        for (int i = 0; i < 2; ++i) lp[i] = lp[i] + lv[i] * dt + 0.5 * la[i] * dt * dt;    /* :125 */
        for (int i = 0; i < 2; ++i) acc[i] = thrust / p->mass * dir[i] + g[i];             /* :128 */
        for (int i = 0; i < 2; ++i) vel[i] = vel[i] + acc[i] * dt;
        for (int i = 0; i < 2; ++i) pos[i] = pos[i] + vel[i] * dt + 0.5 * acc[i] * dt * dt;
        att = att + w * dt;
    }
    o[0] = pos[0]; o[1] = pos[1]; o[2] = att; o[3] = vel[0]; o[4] = vel[1];
    o[5] = lp[0]; o[6] = lp[1]; o[7] = lv[0]; o[8] = lv[1];
    double nlp = norm2(lp, 2), nlv = norm2(lv, 2);
    *done = (nlp < -p->pos_limit) || (nlp > p->pos_limit) || (nlv < -p->vel_limit) ||
            (nlv > p->vel_limit);                      /* :136-140 load position, load velocity */
    *reward = reward_machine(*done, norm2(pos, 2), sbd); /* :142-152 reward -|quad pos| */
}

int oracle_step_branch(int kind, const oracle_params *p, const double *s, const double *a,
                       double *s_out, double *reward, int *done, int *sbd, int force_taut) {
    switch (kind) {
    case ORACLE_QUAD2D: step_quad2d(p, s, a, s_out, reward, done, sbd); return 0;
    case ORACLE_QUAD2D_SL: step_quad2d_sl(p, s, a, s_out, reward, done, sbd, force_taut); return 0;
    case ORACLE_QUAD3D: step_quad3d(p, s, a, s_out, reward, done, sbd); return 0;
    case ORACLE_QUAD3D_SL: step_quad3d_sl(p, s, a, s_out, reward, done, sbd, force_taut); return 0;
    }
    return -1;
}

int oracle_step(int kind, const oracle_params *p, const double *s, const double *a, double *s_out,
                double *reward, int *done, int *sbd) {
    return oracle_step_branch(kind, p, s, a, s_out, reward, done, sbd, -1);
}

/* ---- Quadrotor3D.control  quadrotor3d.py:126-180 (= quadrotor3d_slungload.py:169-226) ---------- */
static int control_3d(const oracle_params *p, const double *s, double *a_out) {
    double pos[3] = {s[0], s[1], s[2]};
    double att[4] = {s[3], s[4], s[5], s[6]};
    double vel[3] = {s[7], s[8], s[9]};
    const double *g = p->g_vec;                /* self.g */
    double ad[3];
    for (int i = 0; i < 3; ++i) {
        double ep = pos[i] - p->ref_pos[i], ev = vel[i] - p->ref_vel[i]; /* :155-156 */
        double fb = p->kp * ep + p->kv * ev;                             /* :160 */
        ad[i] = 0.0 + fb - g[i];                                         /* :162 */
    }
    /* acc2quat :127-141 */
    double n = norm2(ad, 3), zb[3], xb[3], yb[3];
    for (int i = 0; i < 3; ++i) zb[i] = ad[i] / n;
    /* np.cross((0,1,0), zb) */
    xb[0] = 1.0 * zb[2] - 0.0 * zb[1];
    xb[1] = 0.0 * zb[0] - 0.0 * zb[2];
    xb[2] = 0.0 * zb[1] - 1.0 * zb[0];
    double nx = norm2(xb, 3);
    for (int i = 0; i < 3; ++i) xb[i] = xb[i] / nx;
    yb[0] = zb[1] * xb[2] - zb[2] * xb[1];
    yb[1] = zb[2] * xb[0] - zb[0] * xb[2];
    yb[2] = zb[0] * xb[1] - zb[1] * xb[0];
    double nz = norm2(zb, 3);
    for (int i = 0; i < 3; ++i) zb[i] = zb[i] / nz;
    double R[3][3] = {{xb[0], yb[0], zb[0]}, {xb[1], yb[1], zb[1]}, {xb[2], yb[2], zb[2]}};
    double qdes[4];
    int rc = quat_from_matrix(R, qdes); /* :139 */
    double conj[4] = {att[0], -att[1], -att[2], -att[3]}, qe[4];
    quat_mul(conj, qdes, qe);           /* :169 conj(raw q) * q_des */
    double sg = (qe[0] > 0) ? 1.0 : ((qe[0] < 0) ? -1.0 : (qe[0] == 0 ? 0.0 : NAN));
    double k = (2 / p->tau) * sg;       /* :173 */
    double qn[4], b3[3];
    quat_normalise(att, qn);
    quat_body_z(qn, b3);
    a_out[0] = ad[0] * b3[0] + ad[1] * b3[1] + ad[2] * b3[2]; /* :176 */
    a_out[1] = k * qe[1];
    a_out[2] = k * qe[2];
    a_out[3] = k * qe[3];
    return rc;
}

/* ---- Quadrotor2D.control  quadrotor2d.py:115-138 (= quadrotor2d_slungload.py:156-183) ---------- */
static int control_2d(const oracle_params *p, const double *s, double *a_out) {
    double ad[2];
    const double g9[2] = {0.0, 9.8};              /* :130 a literal np.array([0.0, 9.8]), not self.g */
    for (int i = 0; i < 2; ++i) {
        double ep = s[i] - p->ref_pos[i], ev = s[3 + i] - p->ref_vel[i]; /* :127-128 */
        ad[i] = p->kp * ep + p->kv * ev + g9[i];                         /* :130 */
    }
    double th_d = atan2(ad[1], ad[0]) - M_PI / 2; /* :131 */
    double e = s[2] - th_d;                       /* :132 */
    a_out[1] = (-1 / p->tau) * e;                 /* :133 */
    a_out[0] = p->mass * norm2(ad, 2);            /* :134 */
    return 0;
}

int oracle_control(int kind, const oracle_params *p, const double *s, double *a_out) {
    switch (kind) {
    case ORACLE_QUAD2D:
    case ORACLE_QUAD2D_SL: return control_2d(p, s, a_out);
    case ORACLE_QUAD3D:
    case ORACLE_QUAD3D_SL: return control_3d(p, s, a_out);
    }
    return -1;
}

/* ---- Philox4x32-10 (Salmon et al., SC'11; Random123 constants) -------------------------------- */
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float oracle_u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

/* Counter layout (specification shared with the HIP path, see include/rmav.h "RNG streams"):
 *   key = (seed_lo, seed_hi); ctr = (env_lo, env_hi, c2, (tag << 24) | (hi16 << 8) | block)
 *   reset : tag 1, c2 = episode index, hi16 = 0, block j yields components 4j..4j+3
 *   action: tag 2, block index b = t for the 4-action kinds, b = t >> 1 for the 2-action kinds (which take draws 2 (t & 1),
 *           2 (t & 1) + 1 of the block); c2 = low 32 bits of b, hi16 = bits 32..47 of b */
void oracle_reset_state(int kind, uint64_t seed, uint64_t env_id, uint32_t episode, float *s_out) {
    int nS = k_state_dim[kind];
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int j = 0; j * 4 < nS; ++j) {
        uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), episode,
                           (1u << 24) | (uint32_t)j};
        uint32_t r[4];
        oracle_philox4x32_10(ctr, key, r);
        for (int i = 0; i < 4 && j * 4 + i < nS; ++i)
            s_out[j * 4 + i] = 2.0f * oracle_u01(r[i]) - 1.0f; /* exact in fp32 */
    }
}

void oracle_random_action(int kind, uint64_t seed, uint64_t env_id, uint64_t t, float lo, float hi,
                          float *a_out) {
    int nA = k_action_dim[kind];
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int pairs = nA <= 2;
    const uint64_t b = pairs ? (t >> 1) : t;
    uint32_t ctr[4] = {(uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)b,
                       (2u << 24) | ((uint32_t)((b >> 32) & 0xFFFFu) << 8)};
    uint32_t r[4];
    oracle_philox4x32_10(ctr, key, r);
    const int first = pairs ? 2 * (int)(t & 1u) : 0;
    for (int i = 0; i < nA; ++i) a_out[i] = fmaf(hi - lo, oracle_u01(r[first + i]), lo);
}

/* ---- batched drivers ---------------------------------------------------------------------------- */
int oracle_batch_step(int kind, const oracle_params *p, int64_t n, double *s, const double *a,
                      double *reward, uint8_t *done, int32_t *sbd, int round_f32) {
    if (kind < 0 || kind > 3) return -1;
    int nS = k_state_dim[kind], nA = k_action_dim[kind];
    for (int64_t e = 0; e < n; ++e) {
        double o[16], r;
        int d, sb = sbd[e];
        oracle_step(kind, p, s + e * nS, a + e * nA, o, &r, &d, &sb);
        for (int i = 0; i < nS; ++i) s[e * nS + i] = round_f32 ? (double)(float)o[i] : o[i];
        reward[e] = round_f32 ? (double)(float)r : r;
        done[e] = (uint8_t)d;
        sbd[e] = sb;
    }
    return 0;
}

int64_t oracle_rollout_random(int kind, const oracle_params *p, int64_t n, int64_t steps,
                              uint64_t seed, uint64_t env_id_base, float lo, float hi, float *state,
                              int32_t *sbd, uint32_t *episode, uint64_t t0, double *ret_sum,
                              int64_t *n_done) {
    if (kind < 0 || kind > 3) return -1;
    int nS = k_state_dim[kind], nA = k_action_dim[kind];
    double acc = 0.0;
    int64_t nd = 0;
    /* envs are independent, so the env loop is the outer (and, with OMP_NUM_THREADS > 1, the parallel)
     * one: every env runs its `steps` steps; per-env results do not depend on the thread count */
#pragma omp parallel for reduction(+ : acc, nd) schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        for (int64_t k = 0; k < steps; ++k) {
            float af[4];
            double s[16], a[4], o[16], r;
            int d, sb = sbd[e];
            oracle_random_action(kind, seed, env_id_base + (uint64_t)e, t0 + (uint64_t)k, lo, hi, af);
            for (int i = 0; i < nA; ++i) a[i] = af[i];
            for (int i = 0; i < nS; ++i) s[i] = state[e * nS + i];
            oracle_step(kind, p, s, a, o, &r, &d, &sb);
            sbd[e] = sb;
            acc += r;
            if (d) {
                ++nd;
                oracle_reset_state(kind, seed, env_id_base + (uint64_t)e, episode[e], state + e * nS);
                episode[e] += 1;
            } else {
                for (int i = 0; i < nS; ++i) state[e * nS + i] = (float)o[i];
            }
        }
    }
    if (ret_sum) *ret_sum += acc;
    if (n_done) *n_done += nd;
    return n * steps;
}

/* ================================================================================================
 * ReinmavEnv  (reinmav_env.py)
 * ================================================================================================ */
void oracle_reinmav_default_params(oracle_reinmav_params *p) {
    memset(p, 0, sizeof(*p));
    p->arm_length = 0.0860; /* :55 */
    p->mass = 0.1800;       /* :56 */
    p->gravity = 9.8100;    /* :57 */
    p->min_force = 0.0;     /* :58 */
    p->max_force = 3.5316;  /* :59 */
    const double I[3][3] = {{0.00025, 0, 2.55e-06}, {0, 0.000232, 0}, {2.55e-06, 0, 0.0003738}}; /* :60-62 */
    memcpy(p->inertia, I, sizeof(I));
    /* Inertia.getI() (:63): 3x3 inverse by cofactors */
    double c00 = I[1][1] * I[2][2] - I[1][2] * I[2][1], c01 = I[1][2] * I[2][0] - I[1][0] * I[2][2],
           c02 = I[1][0] * I[2][1] - I[1][1] * I[2][0];
    double det = I[0][0] * c00 + I[0][1] * c01 + I[0][2] * c02;
    p->inv_inertia[0][0] = c00 / det;
    p->inv_inertia[0][1] = (I[0][2] * I[2][1] - I[0][1] * I[2][2]) / det;
    p->inv_inertia[0][2] = (I[0][1] * I[1][2] - I[0][2] * I[1][1]) / det;
    p->inv_inertia[1][0] = c01 / det;
    p->inv_inertia[1][1] = (I[0][0] * I[2][2] - I[0][2] * I[2][0]) / det;
    p->inv_inertia[1][2] = (I[0][2] * I[1][0] - I[0][0] * I[1][2]) / det;
    p->inv_inertia[2][0] = c02 / det;
    p->inv_inertia[2][1] = (I[0][1] * I[2][0] - I[0][0] * I[2][1]) / det;
    p->inv_inertia[2][2] = (I[0][0] * I[1][1] - I[0][1] * I[1][0]) / det;
    p->dt = 1.0 / 100;  /* :73 */
    p->ds = 1.0 / 5000; /* :91 */
    p->t_max = 4.0;     /* :129 */
    const double kp[3] = {10, 10, 35}, kd[3] = {5, 5, 22}, kpr[3] = {100, 100, 100}, kdr[3] = {.1, .1, .1};
    memcpy(p->kp, kp, sizeof(kp));
    memcpy(p->kd, kd, sizeof(kd));
    memcpy(p->kp_rot, kpr, sizeof(kpr));
    memcpy(p->kd_rot, kdr, sizeof(kdr));
}

/* quat2mat (:267-290) */
static void reinmav_quat2mat(const double q[4], double m[3][3]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double Nq = w * w + x * x + y * y + z * z;
    if (!(Nq > 2.220446049250313e-16)) { /* np.where(Nq > eps, mat, eye) */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m[i][j] = (i == j);
        return;
    }
    double s = 2.0 / Nq;
    double X = x * s, Y = y * s, Z = z * s;
    double wX = w * X, wY = w * Y, wZ = w * Z;
    double xX = x * X, xY = x * Y, xZ = x * Z;
    double yY = y * Y, yZ = y * Z, zZ = z * Z;
    m[0][0] = 1.0 - (yY + zZ); m[0][1] = xY - wZ;         m[0][2] = xZ + wY;
    m[1][0] = xY + wZ;         m[1][1] = 1.0 - (xX + zZ); m[1][2] = yZ - wX;
    m[2][0] = xZ - wY;         m[2][1] = yZ + wX;         m[2][2] = 1.0 - (xX + yY);
}

/* trj_gen (:128-136) -> [pos x3, vel x3, acc x3, pos, vel] */
static void reinmav_trj(const oracle_reinmav_params *p, double t, double d[11]) {
    double tm = p->t_max;
    t = fmax(0.0, fmin(t, tm));
    t = t / tm;
    double pos = 10.0 * pow(t, 3) - 15.0 * pow(t, 4) + 6.0 * pow(t, 5);
    double vel = (30 / tm) * pow(t, 2) - (60 / tm) * pow(t, 3) + (30 / tm) * pow(t, 4);
    double acc = (60 / (tm * tm)) * t - (180 / (tm * tm)) * pow(t, 2) + (120 / (tm * tm)) * pow(t, 3);
    d[0] = d[1] = d[2] = pos;
    d[3] = d[4] = d[5] = vel;
    d[6] = d[7] = d[8] = acc;
    d[9] = pos;
    d[10] = vel;
}

void oracle_reinmav_controller(const oracle_reinmav_params *p, const double s[13], double t, double fm[4]) {
    /* stateToQd (:292-304) + RotToRPY (:341-346) */
    double R[3][3];
    const double q[4] = {s[6], s[7], s[8], s[9]};
    reinmav_quat2mat(q, R);
    double phi = asin(R[1][2]);
    double psi = atan2(-R[1][0] / cos(phi), R[1][1] / cos(phi));
    double theta = atan2(-R[0][2] / cos(phi), R[2][2] / cos(phi));
    double d[11];
    reinmav_trj(p, t, d);
    /* controller (:306-337) */
    double ddr[3];
    for (int i = 0; i < 3; ++i) {
        double ep = d[i] - s[i], ev = d[3 + i] - s[3 + i];
        ddr[i] = d[6 + i] + p->kd[i] * ev + p->kp[i] * ep;
    }
    double psi_des = d[9], dpsi_des = d[10];
    double u1 = p->mass * (p->gravity + ddr[2]);
    double phi_des = 1 / p->gravity * (ddr[0] * sin(psi_des) - ddr[1] * cos(psi_des));
    double theta_des = 1 / p->gravity * (ddr[0] * cos(psi_des) + ddr[1] * sin(psi_des));
    fm[0] = u1;
    fm[1] = p->kp_rot[0] * (phi_des - phi) - p->kd_rot[0] * s[10];
    fm[2] = p->kp_rot[1] * (theta_des - theta) - p->kd_rot[1] * s[11];
    fm[3] = p->kp_rot[2] * (psi_des - psi) + p->kd_rot[2] * (dpsi_des - s[12]);
}

void oracle_reinmav_derivative(const oracle_reinmav_params *p, const double s[13], const double fm[4],
                               double sdot[13]) {
    const double L = p->arm_length;
    /* motor mixing (:206-216) */
    const double A[4][3] = {{0.25, 0, -0.5 / L}, {0.25, 0.5 / L, 0.}, {0.25, 0, 0.5 / L}, {0.25, -0.5 / L, 0}};
    double T[4];
    for (int i = 0; i < 4; ++i) {
        double v = A[i][0] * fm[0] + A[i][1] * fm[1] + A[i][2] * fm[2];
        T[i] = fmax(fmin(v, p->max_force / 4.0), p->min_force / 4.0);
    }
    double force = 1.0 * T[0] + 1.0 * T[1] + 1.0 * T[2] + 1.0 * T[3];
    double mom[3] = {0.0 * T[0] + L * T[1] + 0.0 * T[2] + (-L) * T[3], (-L) * T[0] + 0.0 * T[1] + L * T[2] + 0. * T[3],
                     fm[3]};
    /* :233-241 */
    const double q[4] = {s[6], s[7], s[8], s[9]};
    double bRw[3][3];
    reinmav_quat2mat(q, bRw);
    /* accel = 1/m * (wRb . (0,0,F) - (0,0,m g)),  wRb = bRw^T  (:240) */
    double acc[3];
    for (int i = 0; i < 3; ++i) {
        double v = bRw[0][i] * 0.0 + bRw[1][i] * 0.0 + bRw[2][i] * force;
        acc[i] = 1.0 / p->mass * (v - (i == 2 ? p->mass * p->gravity : 0.0));
    }
    /* qdot (:243-246) */
    double pw = s[10], qw = s[11], rw = s[12];
    double quaterror = 1 - (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double Om[4][4] = {{0, -pw, -qw, -rw}, {pw, 0, -rw, qw}, {qw, rw, 0, -pw}, {rw, -qw, pw, 0}};
    double qdot[4];
    for (int i = 0; i < 4; ++i) {
        double v = 0;
        for (int j = 0; j < 4; ++j) v += (-1.0 / 2 * Om[i][j]) * q[j];
        qdot[i] = v + 2.0 * quaterror * q[i];
    }
    /* pqrdot = invI (M - w x (I w))  (:248-250) */
    double Iw[3], w[3] = {pw, qw, rw}, tmp[3], rhs[3];
    for (int i = 0; i < 3; ++i) Iw[i] = p->inertia[i][0] * w[0] + p->inertia[i][1] * w[1] + p->inertia[i][2] * w[2];
    tmp[0] = w[1] * Iw[2] - w[2] * Iw[1];
    tmp[1] = w[2] * Iw[0] - w[0] * Iw[2];
    tmp[2] = w[0] * Iw[1] - w[1] * Iw[0];
    for (int i = 0; i < 3; ++i) rhs[i] = mom[i] - tmp[i];
    sdot[0] = s[3]; sdot[1] = s[4]; sdot[2] = s[5];
    sdot[3] = acc[0]; sdot[4] = acc[1]; sdot[5] = acc[2];
    for (int i = 0; i < 4; ++i) sdot[6 + i] = qdot[i];
    for (int i = 0; i < 3; ++i)
        sdot[10 + i] = p->inv_inertia[i][0] * rhs[0] + p->inv_inertia[i][1] * rhs[1] + p->inv_inertia[i][2] * rhs[2];
}

int oracle_reinmav_step(const oracle_reinmav_params *p, double s[13], double *t, const double *action,
                        double *reward, int *done) {
    /* np.arange(t, t+dt, ds): length ceil((stop-start)/step); values start + i*delta with
     * delta = (start+step) - start  (NumPy's fill for doubles) */
    const double start = *t, stop = *t + p->dt;
    int n = (int)ceil((stop - start) / p->ds);
    if (n < 0) n = 0;
    const double delta = (start + p->ds) - start;
    for (int i = 0; i < n; ++i) {
        double ti = (i == 0) ? start : ((i == 1) ? start + p->ds : start + i * delta);
        double fm[4], sdot[13];
        if (action) memcpy(fm, action, sizeof(fm));
        else oracle_reinmav_controller(p, s, ti, fm);
        oracle_reinmav_derivative(p, s, fm, sdot);
        for (int k = 0; k < 13; ++k) s[k] = s[k] + p->ds * sdot[k]; /* :98 */
    }
    *reward = 100.0 - 10.0; /* :111-116 */
    *done = 1;              /* :110 */
    *t = *t + p->dt;        /* :119 */
    return n;
}

int oracle_reinmav_step_rk4(const oracle_reinmav_params *p, double s[13], double *t, const double *action,
                            double *reward, int *done) {
    const double start = *t, stop = *t + p->dt, ds = p->ds;
    int n = (int)ceil((stop - start) / ds);
    if (n < 0) n = 0;
    const double delta = (start + ds) - start;
    for (int i = 0; i < n; ++i) {
        double ti = (i == 0) ? start : ((i == 1) ? start + ds : start + i * delta);
        double fm[4], k1[13], k2[13], k3[13], k4[13], y[13];
        if (action) memcpy(fm, action, sizeof(fm));
        else oracle_reinmav_controller(p, s, ti, fm);
        oracle_reinmav_derivative(p, s, fm, k1);
        for (int k = 0; k < 13; ++k) y[k] = s[k] + 0.5 * ds * k1[k];
        oracle_reinmav_derivative(p, y, fm, k2);
        for (int k = 0; k < 13; ++k) y[k] = s[k] + 0.5 * ds * k2[k];
        oracle_reinmav_derivative(p, y, fm, k3);
        for (int k = 0; k < 13; ++k) y[k] = s[k] + ds * k3[k];
        oracle_reinmav_derivative(p, y, fm, k4);
        for (int k = 0; k < 13; ++k) s[k] = s[k] + ds / 6.0 * (k1[k] + 2.0 * k2[k] + 2.0 * k3[k] + k4[k]);
    }
    *reward = 100.0 - 10.0;
    *done = 1;
    *t = *t + p->dt;
    return n;
}
