"""The kernels' per-lane arithmetic (csrc/rmav_math.hpp, compiled for the host by the test-only helper
tests/hostmath) against the fp64 oracle and the reference's golden vectors: the fp32 / mixed-precision
design meets the 1e-6 * max(1,|y|) bar before any GPU run."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from util import CTRL_TOL, KINDS, NA, NS, TOL, near_threshold, random_cases, scaled_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def hm(built):
    return C.CDLL(os.path.join(ROOT, "tests", "hostmath", "_build", "libhostmath.so"))


def _step(hm, kind, s, a, reading=None):
    from gym_reinmav_amd import _abi as A

    k = A.KIND_BY_NAME[kind]
    p = A.default_params(k, reading)
    s2 = np.ascontiguousarray(s, dtype=np.float32).copy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    n = len(s2)
    dist = np.zeros(n, np.float32)
    done = np.zeros(n, np.int32)
    assert hm.hm_step(k, C.byref(p), C.c_int64(n), s2.ctypes.data_as(FP), a.ctypes.data_as(FP),
                      dist.ctypes.data_as(FP), done.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return s2, dist, done.astype(bool)


@pytest.mark.parametrize("kind", KINDS)
def test_step_arithmetic_vs_golden(hm, kind, golden):
    g = golden[kind]
    s2, dist, done = _step(hm, kind, g["step_s"], g["step_a"])
    assert scaled_err(s2, g["step_s2"]).max() <= TOL
    ok = near_threshold(kind, g["step_s2"])
    assert np.array_equal(done | ok, g["step_d"] | ok)
    alive = ~g["step_d"] & ~done
    assert scaled_err(-dist[alive], g["step_r"][alive]).max() <= TOL


@pytest.mark.parametrize("kind", KINDS)
def test_step_arithmetic_vs_oracle_random(hm, kind):
    s, a = random_cases(kind, 20000, seed=21)
    s2, dist, done = _step(hm, kind, s, a)
    o2, r, d, _ = O.batch_step(kind, s.astype(np.float64), a.astype(np.float64))
    assert scaled_err(s2, o2).max() <= TOL
    ok = near_threshold(kind, o2)
    assert np.array_equal(done | ok, d | ok)
    alive = ~d & ~done
    assert scaled_err(-dist[alive], r[alive]).max() <= TOL


@pytest.mark.parametrize("kind", KINDS)
def test_control_arithmetic(hm, kind, golden):
    from gym_reinmav_amd import _abi as A

    k = A.KIND_BY_NAME[kind]
    p = A.default_params(k)
    g = golden[kind]
    cs = g["ctrl_s"].astype(np.float32)
    ca = np.zeros((len(cs), NA[kind]), np.float32)
    assert hm.hm_control(k, C.byref(p), C.c_int64(len(cs)), cs.ctypes.data_as(FP), ca.ctypes.data_as(FP)) == 0
    assert scaled_err(ca, g["ctrl_a"]).max() <= CTRL_TOL


@pytest.mark.parametrize("kind", KINDS)
def test_rng_streams_bit_exact(hm, kind):
    from gym_reinmav_amd import _abi as A

    k = A.KIND_BY_NAME[kind]
    for env in (0, 1, 65535, 2**40 + 3):
        for idx in (0, 1, 77):
            s = np.zeros(NS[kind], np.float32)
            hm.hm_reset_state(k, C.c_uint64(9), C.c_uint64(env), C.c_uint32(idx), s.ctypes.data_as(FP))
            assert np.array_equal(s, O.reset_state(kind, 9, env, idx))
        for t in (0, 4, 5, 2**33 + 1, 2**33 + 2):
            a = np.zeros(NA[kind], np.float32)
            hm.hm_random_action(k, C.c_uint64(9), C.c_uint64(env), C.c_uint64(t), C.c_float(-10), C.c_float(10),
                                a.ctypes.data_as(FP))
            assert np.array_equal(a, O.random_action(kind, 9, env, t, -10.0, 10.0))


def test_fast_sincosf_vs_libm(hm):
    """The 2-D kinds' sin / cos (two-fma reduction + 3-coefficient polynomials, csrc/rmav_math.hpp): absolute error <= 1.5e-7
    against fp64 libm of the SAME fp32 argument over [-32768, 32768], dense near 0, at the multiples of pi/4 and their fp32
    neighbours; beyond the range and for inf / NaN it must be libm itself.  (The step multiplies it by thrust / mass <= 100
    and dt = 0.01: 1.5e-7 on the velocity, inside the 1e-6 bar with the rest of the step's rounding.)"""
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-4, 4, 400000), rng.uniform(-100, 100, 200000), rng.uniform(-32768, 32768, 400000),
                        rng.uniform(-1e-3, 1e-3, 10000), np.arange(-2000, 2001) * (np.pi / 4)]).astype(np.float32)
    x = np.concatenate([x, np.nextafter(x[-4001:], np.float32(np.inf)), np.nextafter(x[-4001:], np.float32(-np.inf)),
                        np.array([0.0, -0.0, 32767.998, -32767.998], np.float32)])
    sn, cs = np.empty_like(x), np.empty_like(x)
    hm.hm_fast_sincosf(C.c_int64(len(x)), x.ctypes.data_as(FP), sn.ctypes.data_as(FP), cs.ctypes.data_as(FP))
    xd = x.astype(np.float64)
    es, ec = np.abs(sn - np.sin(xd)).max(), np.abs(cs - np.cos(xd)).max()
    assert es <= 1.5e-7 and ec <= 1.5e-7, (es, ec)
    assert np.abs(sn * sn + cs * cs - 1.0).max() <= 4e-7
    big = np.array([32768.0, -1e6, 3.4e38, np.inf, -np.inf, np.nan], np.float32)
    sb, cb = np.empty_like(big), np.empty_like(big)
    hm.hm_fast_sincosf(C.c_int64(len(big)), big.ctypes.data_as(FP), sb.ctypes.data_as(FP), cb.ctypes.data_as(FP))
    with np.errstate(invalid="ignore"):
        ref_s, ref_c = np.sin(big.astype(np.float64)), np.cos(big.astype(np.float64))
    assert np.array_equal(np.isnan(sb), np.isnan(ref_s)) and np.array_equal(np.isnan(cb), np.isnan(ref_c))
    fin = np.isfinite(big)
    assert np.abs(sb[fin] - ref_s[fin]).max() <= 1e-6 and np.abs(cb[fin] - ref_c[fin]).max() <= 1e-6


def test_fast_atan2_vs_libm(hm):
    """The 2-D controller's atan2 (one reciprocal + an 8-coefficient polynomial, csrc/rmav_math.hpp): absolute error
    <= 1e-12 against libm over the plane, the axes, the octant boundaries and tiny / huge magnitudes (1e-11 would do:
    the result is multiplied by 1 / tau = 10 and rounded to fp32)."""
    import math

    hm.hm_fast_atan2.restype = C.c_double
    hm.hm_fast_atan2.argtypes = [C.c_double, C.c_double]
    rng = np.random.RandomState(0)
    pts = [(y, x) for y, x in rng.uniform(-60, 60, (200000, 2))]
    pts += [(y * 10.0 ** e, x * 10.0 ** e) for (y, x), e in zip(rng.uniform(-1, 1, (2000, 2)), rng.randint(-12, 12, 2000))]
    t8 = math.tan(math.pi / 8)
    for s1 in (-1.0, 1.0):
        for s2 in (-1.0, 1.0):
            pts += [(0.0, s2), (s1, 0.0), (s1, s2), (s1 * t8, s2), (s1, s2 * t8), (s1 * (t8 + 1e-15), s2), (s1 * 1e-300, s2),
                    (s1 * 3.0, s2 * 3.0000000001)]
    pts.append((0.0, 0.0))
    worst = max(abs(hm.hm_fast_atan2(y, x) - math.atan2(y, x)) for y, x in pts)
    assert worst <= 1e-12, worst
