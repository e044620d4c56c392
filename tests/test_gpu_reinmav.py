"""ReinmavEnv on the GPU: parity with the oracle / the reference's recorded run, through the C ABI."""
import os

import numpy as np
import pytest

import oracle as O
from util import TOL, scaled_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def G(built):
    import torch

    assert torch.cuda.is_available()
    import gym_reinmav_amd as g

    return g


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "reinmav.npz")))


def test_reference_run_teacher_forced_and_free_running(G, gold):
    n = len(gold["run_s"])
    env = G.BatchedQuadrotor("reinmav", n, auto_reset=False, track_episodes=False)
    assert np.array_equal(env.get_state(), np.tile(O.REINMAV_INIT_STATE.astype(np.float32), (n, 1)))
    assert np.array_equal(env.get_time(), np.zeros(n))
    # all 400 recorded (state, t) pairs of the reference run at once, one step each, built-in controller
    env.set_state(gold["run_s"].astype(np.float32))
    env.set_time(gold["run_t"])
    tr = env.rollout(1, mode="controller", layout="aos", want=("actions", "obs", "rew", "done"))
    exp, texp, ns = O.reinmav_batch_step(gold["run_s"].astype(np.float32).astype(np.float64), gold["run_t"])
    assert set(ns) == {50, 51}
    assert scaled_err(tr["obs"][0], exp).max() <= TOL
    # vs the reference's own fp64 outputs: the recorded inputs are rounded to fp32 on entry, and the attitude
    # loop (kp_rot / kd_rot = 1000 1/s) turns a 6e-8 rad input rounding into ~1e-5 rad/s within one step
    assert scaled_err(tr["obs"][0], gold["run_s2"]).max() <= 3e-5
    assert np.array_equal(env.get_time(), texp) and np.array_equal(texp, gold["run_t2"])
    assert (tr["rew"] == 90.0).all() and (tr["done"] == 1).all()
    ctrl = np.array([O.reinmav_controller(s, t) for s, t in zip(gold["run_s"].astype(np.float32).astype(np.float64), gold["run_t"])])
    assert scaled_err(tr["actions"][0], ctrl).max() <= TOL
    env.close()
    # one env, 400 steps free running in ONE fused launch: ends where the reference ends
    one = G.BatchedQuadrotor("reinmav", 1, auto_reset=True, track_episodes=True)
    tr = one.rollout(400, mode="controller", layout="aos", fused=True, want=("obs",))
    assert np.abs(tr["obs"][-1, 0] - gold["run_s2"][-1]).max() < 2e-4
    assert np.abs(tr["obs"][:, 0] - gold["run_s2"]).max() < 1e-3
    assert abs(one.get_time()[0] - gold["run_t2"][-1]) < 1e-12
    tot = one.episode_totals()
    assert tot["episodes"] == 400 and abs(tot["return_sum"] - 400 * 90.0) < 1e-6   # done = True every step
    one.close()


@pytest.mark.parametrize("mode", ["controller", "buffer"])
def test_perturbed_batch_vs_oracle(G, gold, mode):
    rng = np.random.RandomState(5)
    n = 3000 + 17
    k = rng.randint(0, 400, n)
    s = (gold["run_s"][k] + rng.normal(scale=0.02, size=(n, 13))).astype(np.float32)
    t = np.where(rng.uniform(size=n) < 0.5, gold["run_t"][k], rng.uniform(0, 4.5, n))
    a = None
    if mode == "buffer":
        a = rng.uniform(0, 3.0, (n, 4)).astype(np.float32)
        a[:, 1:] = rng.normal(scale=0.02, size=(n, 3)).astype(np.float32)
    env = G.BatchedQuadrotor("reinmav", n, auto_reset=False, track_episodes=False)
    env.set_state(s)
    env.set_time(t)
    if mode == "controller":
        obs = env.rollout(1, mode="controller", layout="aos", want=("obs",))["obs"][0]
    else:
        obs, rew, done = env.step(a)
        assert (rew == 90.0).all() and done.all()
    exp, texp, _ = O.reinmav_batch_step(s.astype(np.float64), t, actions=None if a is None else a.astype(np.float64))
    assert scaled_err(obs, exp).max() <= TOL
    assert np.array_equal(env.get_time(), texp)
    # fused 5-step launch == five single-step launches, bit for bit
    for fused in (True, False):
        env.set_state(s)
        env.set_time(t)
        r = env.rollout(5, mode=mode, actions=None if a is None else np.broadcast_to(a.T, (5, 4, n)).copy(), layout="soa",
                        fused=fused, want=("obs",))
        ref = r["obs"] if fused else ref
        assert np.array_equal(r["obs"], ref)
    env.close()


def test_gym_shaped_reinmav_env(G, gold):
    """test/test_reinmav.py:12-25: gym.make('reinmav-v0'); 400 x env.step()."""
    env = G.make("reinmav-v0")
    assert env.mass == 0.18 and env.dt == 0.01 and env.arm_length == 0.086
    assert np.array_equal(env.reset(), O.REINMAV_INIT_STATE)
    s, t = O.REINMAV_INIT_STATE.copy(), 0.0
    for k in range(60):
        obs, reward, done, info = env.step()
        s, t, r, d, _ = O.reinmav_step(s.astype(np.float32).astype(np.float64), t)
        assert reward == 90.0 and done is True and info == {} and obs.shape == (13,)
        assert scaled_err(obs, s).max() <= TOL and env.t == t
        s = obs
    fm = env.control()
    assert scaled_err(fm, O.reinmav_controller(env.state, env.t)).max() <= TOL
    obs, reward, done, _ = env.step([1.8, 0.0, 0.0, 0.0])     # extension: explicit (F, Mx, My, Mz)
    exp, _, _, _, _ = O.reinmav_step(s.astype(np.float32).astype(np.float64), t, action=[np.float32(1.8), 0, 0, 0])
    assert scaled_err(obs, exp).max() <= TOL
    env.state = gold["run_s"][100]
    env.t = gold["run_t"][100]
    obs, _, _, _ = env.step()
    assert scaled_err(obs, gold["run_s2"][100]).max() <= 3e-5   # input rounding, see above
    env.close()


def test_rk4_option_vs_oracle_and_vs_euler(G, gold):
    """RMAV_INT_RK4 (an option beyond the reference, which only has explicit Euler): matches the oracle's RK4,
    and really is a different integrator than the default."""
    A = G._abi
    n = len(gold["step_s"])
    p = A.default_params(A.REINMAV)
    p.integrator = A.INT_RK4
    s, t = gold["step_s"].astype(np.float32), gold["step_t"]
    env = G.BatchedQuadrotor("reinmav", n, auto_reset=False, track_episodes=False, params=p)
    env.set_state(s)
    env.set_time(t)
    obs = env.rollout(1, mode="controller", layout="aos", want=("obs",))["obs"][0]
    exp, texp, _ = O.reinmav_batch_step(s.astype(np.float64), t, rk4=True)
    assert scaled_err(obs, exp).max() <= TOL and np.array_equal(env.get_time(), texp)
    euler, _, _ = O.reinmav_batch_step(s.astype(np.float64), t)
    assert np.abs(exp - euler).max() > 1e-9
    # on the reference's own (smooth) trajectory the two integrators agree closely at ds = 1/5000 s
    k = 200
    e1, _, _, _, _ = O.reinmav_step(gold["run_s"][k], gold["run_t"][k])
    e4, _, _, _, _ = O.reinmav_step(gold["run_s"][k], gold["run_t"][k], rk4=True)
    assert np.abs(e1 - e4).max() < 1e-3
    bad = A.default_params(A.REINMAV)
    bad.integrator = 7
    with pytest.raises(G.RmavError):
        G.BatchedQuadrotor("reinmav", 4, params=bad)
    env.close()


def test_trj_gen_is_the_reference_min_jerk_profile(G):
    """ReinmavEnv.trj_gen (reinmav_env.py:128-136): the quintic in s = clip(t, 0, 4) / 4 and its two derivatives, the same profile on
    x, y, z and yaw - [x, y, z, vx, vy, vz, ax, ay, az, yaw, yaw rate].  Closed-form spot values + the clipping at both ends."""
    env = G.make("reinmav-v0")
    d = env.trj_gen(2.0)                       # s = 1/2: pos 1/2, vel 15/32, acc 0
    assert len(d) == 11 and d[0] == d[1] == d[2] == d[9] == 0.5
    assert abs(d[3] - 0.46875) < 1e-15 and d[3] == d[4] == d[5] == d[10] and abs(d[6]) < 1e-15 and d[6] == d[7] == d[8]
    assert env.trj_gen(-1.0) == [0.0] * 11 and env.trj_gen(0.0) == [0.0] * 11
    end = env.trj_gen(9.0)
    assert end[0] == 1.0 and abs(end[3]) < 1e-15 and abs(end[6]) < 1e-13 and end == env.trj_gen(4.0)
    s = 0.3 / 4.0
    assert abs(env.trj_gen(0.3)[0] - (10 * s ** 3 - 15 * s ** 4 + 6 * s ** 5)) < 1e-16
    assert abs(env.trj_gen(0.3)[6] - (60 / 16 * s - 180 / 16 * s ** 2 + 120 / 16 * s ** 3)) < 1e-15
    env.close()
