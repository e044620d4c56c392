// hostmath.hip - TEST-ONLY helper: runs the per-lane arithmetic of the HIP kernels
// (reinmav-gym_amd/csrc/rmav_math.hpp, the very same __host__ __device__ source) on the host CPU.
//
// Why: the authoring container has no GPU.  This lets `pytest -m "not gpu"` measure the rounding
// error of the kernels' fp32 / mixed arithmetic against the fp64 oracle before any GPU run.
// It is built into tests/hostmath/_build/ and is NOT part of librmav.so; the product has no CPU
// path (rmav_create fails without a GPU).
#include "../../reinmav-gym_amd/csrc/rmav_derive.hpp"
#include "../../reinmav-gym_amd/csrc/rmav_math.hpp"

using namespace rmav;

namespace {
template <int K>
void step_k(const rmav_params &q, float *s, const float *a, float *dist, int *done) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    using R = typename Env<K>::R;
    const ParamsT<R> p = derive<R>(q, K == QUAD2D || K == QUAD2D_SL);
    float ss[NS], aa[NA];
    for (int i = 0; i < NS; ++i) ss[i] = s[i];
    for (int i = 0; i < NA; ++i) aa[i] = a[i];
    bool d;
    Env<K>::step(ss, aa, p, *dist, d);
    *done = d ? 1 : 0;
    for (int i = 0; i < NS; ++i) s[i] = ss[i];
}
template <int K> void control_k(const rmav_params &q, const float *s, float *a) {
    constexpr int NS = Dims<K>::NS, NA = Dims<K>::NA;
    const ParamsT<double> p = derive<double>(q, K == QUAD2D || K == QUAD2D_SL);
    float ss[NS], aa[NA];
    for (int i = 0; i < NS; ++i) ss[i] = s[i];
    env_control<K>(ss, p, aa);
    for (int i = 0; i < NA; ++i) a[i] = aa[i];
}
template <int K> void reset_k(uint64_t seed, uint64_t env, uint32_t idx, float *s) {
    float ss[Dims<K>::NS];
    reset_state<K>(seed, env, idx, ss);
    for (int i = 0; i < Dims<K>::NS; ++i) s[i] = ss[i];
}
template <int K> void action_k(uint64_t seed, uint64_t env, uint64_t t, float lo, float hi, float *a) {
    float aa[Dims<K>::NA];
    random_action<K>(seed, env, t, lo, hi, aa);
    for (int i = 0; i < Dims<K>::NA; ++i) a[i] = aa[i];
}
}  // namespace

extern "C" {
// n envs, AoS: s [n][nS] in/out, a [n][nA]
int hm_step(int kind, const rmav_params *q, int64_t n, float *s, const float *a, float *dist, int *done) {
    static const int nS[4] = {5, 9, 10, 16}, nA[4] = {2, 2, 4, 4};
    for (int64_t e = 0; e < n; ++e) {
        float *se = s + e * nS[kind];
        const float *ae = a + e * nA[kind];
        switch (kind) {
        case QUAD2D: step_k<QUAD2D>(*q, se, ae, dist + e, done + e); break;
        case QUAD2D_SL: step_k<QUAD2D_SL>(*q, se, ae, dist + e, done + e); break;
        case QUAD3D: step_k<QUAD3D>(*q, se, ae, dist + e, done + e); break;
        case QUAD3D_SL: step_k<QUAD3D_SL>(*q, se, ae, dist + e, done + e); break;
        default: return -1;
        }
    }
    return 0;
}
int hm_control(int kind, const rmav_params *q, int64_t n, const float *s, float *a) {
    static const int nS[4] = {5, 9, 10, 16}, nA[4] = {2, 2, 4, 4};
    for (int64_t e = 0; e < n; ++e) {
        const float *se = s + e * nS[kind];
        float *ae = a + e * nA[kind];
        switch (kind) {
        case QUAD2D: control_k<QUAD2D>(*q, se, ae); break;
        case QUAD2D_SL: control_k<QUAD2D_SL>(*q, se, ae); break;
        case QUAD3D: control_k<QUAD3D>(*q, se, ae); break;
        case QUAD3D_SL: control_k<QUAD3D_SL>(*q, se, ae); break;
        default: return -1;
        }
    }
    return 0;
}
// ReinmavEnv: n envs, s [n][13] in/out, t [n] in/out, a [n][4] or NULL (-> built-in controller), fm0 [n][4] out
int hm_reinmav_step(const rmav_params *q, int64_t n, float *s, double *t, const float *a, float *fm0) {
    const ReinmavP p = derive_reinmav(*q);
    for (int64_t e = 0; e < n; ++e) {
        float ss[13], aa[4] = {0, 0, 0, 0}, f0[4];
        for (int i = 0; i < 13; ++i) ss[i] = s[e * 13 + i];
        if (a) for (int i = 0; i < 4; ++i) aa[i] = a[e * 4 + i];
        Env<REINMAV>::step(ss, aa, a == nullptr, t[e], p, f0);
        for (int i = 0; i < 13; ++i) s[e * 13 + i] = ss[i];
        for (int i = 0; i < 4; ++i) fm0[e * 4 + i] = f0[i];
    }
    return 0;
}
int hm_reset_state(int kind, uint64_t seed, uint64_t env, uint32_t idx, float *s) {
    switch (kind) {
    case QUAD2D: reset_k<QUAD2D>(seed, env, idx, s); break;
    case QUAD2D_SL: reset_k<QUAD2D_SL>(seed, env, idx, s); break;
    case QUAD3D: reset_k<QUAD3D>(seed, env, idx, s); break;
    case QUAD3D_SL: reset_k<QUAD3D_SL>(seed, env, idx, s); break;
    default: return -1;
    }
    return 0;
}
void hm_fast_sincosf(int64_t n, const float *x, float *sn, float *cs) {   // the 2-D kinds' sin / cos
    for (int64_t i = 0; i < n; ++i) fast_sincosf(x[i], sn[i], cs[i]);
}
double hm_fast_atan2(double y, double x) { return fast_atan2(y, x); }   // the 2-D controller's atan2 (host form: true division)
int hm_random_action(int kind, uint64_t seed, uint64_t env, uint64_t t, float lo, float hi, float *a) {
    switch (kind) {
    case QUAD2D: action_k<QUAD2D>(seed, env, t, lo, hi, a); break;
    case QUAD2D_SL: action_k<QUAD2D_SL>(seed, env, t, lo, hi, a); break;
    case QUAD3D: action_k<QUAD3D>(seed, env, t, lo, hi, a); break;
    case QUAD3D_SL: action_k<QUAD3D_SL>(seed, env, t, lo, hi, a); break;
    default: return -1;
    }
    return 0;
}
}
