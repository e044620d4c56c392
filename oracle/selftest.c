/* selftest.c - sanitizer run of the oracle (ASan + UBSan): known-answer vectors and a short batched
 * workload.  Built and run by tests/test_oracle_sanitizers.py; TEST INFRASTRUCTURE ONLY. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rmav_oracle.h"

static int fails = 0;
#define CHECK(c)                                                            \
    do {                                                                    \
        if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } \
    } while (0)

int main(void) {
    /* Philox4x32-10 known answers (Random123) */
    uint32_t c0[4] = {0, 0, 0, 0}, k0[2] = {0, 0}, o[4];
    oracle_philox4x32_10(c0, k0, o);
    CHECK(o[0] == 0x6627e8d5u && o[1] == 0xe169c58du && o[2] == 0xbc57ac4cu && o[3] == 0x9b00dbd8u);
    /* SURVEY 8a known-answer vector for Quadrotor3D.step */
    oracle_params p;
    CHECK(oracle_default_params(ORACLE_QUAD3D, 'B', &p) == 0);
    double s[10] = {.1, -.2, .3, .9, .1, -.2, .3, .5, -.4, .2}, a[4] = {9, .1, -.2, .3}, out[16], r;
    int d, sbd = -1;
    CHECK(oracle_step(ORACLE_QUAD3D, &p, s, a, out, &r, &d, &sbd) == 0);
    CHECK(fabs(out[0] - 0.10485789473684212) < 1e-15 && fabs(r + 0.3792366205113136) < 1e-15 && d == 0 && sbd == -1);
    CHECK(oracle_step(9, &p, s, a, out, &r, &d, &sbd) == -1);
    /* every kind: batched random workload with auto-reset (exercises reset / action streams, all branches) */
    for (int kind = 0; kind < 4; ++kind) {
        const int n = 257, nS = oracle_state_dim(kind);
        float *st = (float *)malloc(sizeof(float) * n * nS);
        int32_t *sb = (int32_t *)malloc(sizeof(int32_t) * n);
        uint32_t *ep = (uint32_t *)malloc(sizeof(uint32_t) * n);
        for (int e = 0; e < n; ++e) {
            oracle_reset_state(kind, 7, (uint64_t)e, 0, st + e * nS);
            sb[e] = -1;
            ep[e] = 1;
        }
        oracle_default_params(kind, 'B', &p);
        double ret = 0;
        int64_t nd = 0;
        int64_t k = oracle_rollout_random(kind, &p, n, 300, 7, 0, -10.f, 10.f, st, sb, ep, 0, &ret, &nd);
        CHECK(k == (int64_t)n * 300 && nd > 0 && isfinite(ret));
        double sd[16], ad[4] = {1, 2, 3, 4}, ctrl[4];
        for (int i = 0; i < nS; ++i) sd[i] = st[i];
        CHECK(oracle_control(kind, &p, sd, ctrl) == 0);
        free(st); free(sb); free(ep);
        (void)ad;
    }
    /* ReinmavEnv: 20 steps from the initial state */
    oracle_reinmav_params rp;
    oracle_reinmav_default_params(&rp);
    double rs[13] = {0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0}, t = 0, rew;
    for (int i = 0; i < 20; ++i) {
        int n = oracle_reinmav_step(&rp, rs, &t, NULL, &rew, &d);
        CHECK((n == 50 || n == 51) && rew == 90.0 && d == 1);
    }
    CHECK(fabs(t - 0.2) < 1e-12 && isfinite(rs[0]));
    printf(fails ? "selftest: %d failure(s)\n" : "selftest ok\n", fails);
    return fails ? 1 : 0;
}
