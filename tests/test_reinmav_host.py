"""ReinmavEnv (reinmav_env.py): oracle vs the reference's recorded run, and the kernel arithmetic on the host."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
from util import TOL, scaled_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reinmav.npz")))


def test_oracle_reproduces_reference_run(gold, built):
    """The reference's own 400-step run (test/test_reinmav.py): every step teacher-forced from the recorded
    state and clock, incl. the 50-vs-51 sub-step quirk of np.arange(t, t+dt, 1/5000)."""
    ns = set()
    for s, t, s2, t2 in zip(gold["run_s"], gold["run_t"], gold["run_s2"], gold["run_t2"]):
        o, tn, r, d, n = O.reinmav_step(s, t)
        assert np.abs(o - s2).max() < 1e-11 and tn == t2 and r == 90.0 and d is True
        ns.add(n)
    assert ns == {50, 51}
    # free-running from the initial state the oracle stays on the reference trajectory (stable closed loop)
    s, t = O.REINMAV_INIT_STATE.copy(), 0.0
    for k in range(400):
        s, t, _, _, _ = O.reinmav_step(s, t)
    assert np.abs(s - gold["run_s2"][-1]).max() < 1e-9
    assert np.abs(s[:3] - 1.0).max() < 2e-3   # the min-jerk trajectory ends at (1,1,1), yaw 1 rad


def test_oracle_perturbed_steps_and_controller(gold, built):
    for s, t, s2, t2, fm in zip(gold["step_s"], gold["step_t"], gold["step_s2"], gold["step_t2"], gold["ctrl_fm"]):
        o, tn, _, _, _ = O.reinmav_step(s, t)
        assert scaled_err(o, s2).max() < 1e-10 and tn == t2
        assert scaled_err(O.reinmav_controller(s, t), fm).max() < 1e-12


def test_oracle_vs_live_reference(built):
    import ref_harness as rh

    if not rh.available():
        pytest.skip("reference tree not mounted (GPU box)")
    env = rh.RefReinmav()
    rng = np.random.RandomState(3)
    for i in range(40):
        s = O.REINMAV_INIT_STATE + rng.normal(scale=0.05, size=13)
        t = float(rng.uniform(0, 5))
        env.set(s, t)
        s2, r, d, t2 = env.step()
        o, tn, _, _, _ = O.reinmav_step(s, t)
        assert scaled_err(o, s2).max() < 1e-10 and tn == t2


def test_kernel_arithmetic_on_host(gold, built):
    from gym_reinmav_amd import _abi as A

    hm = C.CDLL(os.path.join(ROOT, "tests", "hostmath", "_build", "libhostmath.so"))
    p = A.default_params(A.REINMAV)
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    for key_s, key_t in (("run_s", "run_t"), ("step_s", "step_t")):
        s = gold[key_s].astype(np.float32)
        t = gold[key_t].copy()
        exp, texp, _ = O.reinmav_batch_step(s.astype(np.float64), t)
        fm0 = np.zeros((len(s), 4), np.float32)
        out = s.copy()
        assert hm.hm_reinmav_step(C.byref(p), C.c_int64(len(s)), out.ctypes.data_as(fp), t.ctypes.data_as(dp), None,
                                  fm0.ctypes.data_as(fp)) == 0
        assert scaled_err(out, exp).max() <= TOL and np.array_equal(t, texp)
        ctrl = np.array([O.reinmav_controller(si, ti) for si, ti in zip(s.astype(np.float64), gold[key_t])])
        assert scaled_err(fm0, ctrl).max() <= TOL
    # external action held over the step
    rng = np.random.RandomState(1)
    a = rng.uniform(0, 3, (len(s), 4)).astype(np.float32)
    a[:, 1:] *= 0.01
    t = gold["step_t"].copy()
    exp, texp, _ = O.reinmav_batch_step(s.astype(np.float64), t, actions=a.astype(np.float64))
    out = s.copy()
    hm.hm_reinmav_step(C.byref(p), C.c_int64(len(s)), out.ctypes.data_as(fp), t.ctypes.data_as(dp), a.ctypes.data_as(fp),
                       fm0.ctypes.data_as(fp))
    assert scaled_err(out, exp).max() <= TOL and np.array_equal(fm0, a)


def test_rk4_kernel_arithmetic_on_host(gold, built):
    from gym_reinmav_amd import _abi as A

    hm = C.CDLL(os.path.join(ROOT, "tests", "hostmath", "_build", "libhostmath.so"))
    p = A.default_params(A.REINMAV)
    p.integrator = A.INT_RK4
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    s = gold["step_s"].astype(np.float32)
    t = gold["step_t"].copy()
    exp, texp, _ = O.reinmav_batch_step(s.astype(np.float64), t, rk4=True)
    out, fm0 = s.copy(), np.zeros((len(s), 4), np.float32)
    assert hm.hm_reinmav_step(C.byref(p), C.c_int64(len(s)), out.ctypes.data_as(fp), t.ctypes.data_as(dp), None,
                              fm0.ctypes.data_as(fp)) == 0
    assert scaled_err(out, exp).max() <= TOL and np.array_equal(t, texp)
