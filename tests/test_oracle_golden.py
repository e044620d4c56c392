"""Pins the CPU oracle (oracle/rmav_oracle.c) against the golden vectors produced by the reference's
own step()/control() (tests/golden/make_golden.py), against the survey's known-answer vectors, and -
when /root/reference is mounted - against the reference executed live."""
import numpy as np
import pytest

import oracle as O
from util import KINDS, NA, NS, scaled_err

FP64_TOL = 1e-12


@pytest.mark.parametrize("kind", KINDS)
def test_step_matches_reference_vectors(kind, golden, built):
    g = golden[kind]
    s2, r, d, sbd = O.batch_step(kind, g["step_s"], g["step_a"])
    assert scaled_err(s2, g["step_s2"]).max() < FP64_TOL
    assert scaled_err(r, g["step_r"]).max() < FP64_TOL
    assert np.array_equal(d, g["step_d"])
    assert np.array_equal(sbd, np.where(g["step_d"], 0, -1))
    assert 0.03 < d.mean() < 0.6  # both branches are exercised


def test_quad2d_reading_A(golden, built):
    g = golden["quad2d"]
    p = O.default_params("quad2d", "A")
    s2, r, d, _ = O.batch_step("quad2d", g["step_s"], g["step_a"], params=p)
    assert scaled_err(s2, g["step_s2"]).max() < FP64_TOL
    assert np.array_equal(d, g["step_d_A"])
    assert scaled_err(r, g["step_r_A"]).max() < FP64_TOL
    assert (g["step_d_A"] != g["step_d"]).any()  # the two readings really differ on this set


@pytest.mark.parametrize("kind", KINDS)
def test_control_matches_reference_vectors(kind, golden, built):
    g = golden[kind]
    a = O.batch_control(kind, g["ctrl_s"])
    assert scaled_err(a, g["ctrl_a"]).max() < FP64_TOL


@pytest.mark.parametrize("kind", KINDS)
def test_tilted_gravity_matches_reference_vectors(kind, built):
    """self.g is a public vector the reference reads on every call (quadrotor3d.py:47,96-99,162; quadrotor2d.py:46,88): the
    oracle's g_vec against the reference's own step() / control() run with a TILTED vector assigned to the env object
    (tests/golden/gravity.npz, make_golden.py::make_gravity).  The 2-D control() ignores it (its literal (0, 9.8),
    quadrotor2d.py:130) - the fixture holds what the reference did."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gravity.npz"))
    p = O.default_params(kind)
    gv = g[kind + "_g"]
    for i, v in enumerate(gv):
        p.g_vec[i] = float(v)
    s2, r, d, _ = O.batch_step(kind, g[kind + "_s"], g[kind + "_a"], params=p)
    assert scaled_err(s2, g[kind + "_s2"]).max() < FP64_TOL
    assert scaled_err(r, g[kind + "_r"]).max() < FP64_TOL and np.array_equal(d, g[kind + "_d"])
    assert scaled_err(O.batch_control(kind, g[kind + "_s"], params=p), g[kind + "_ctrl"]).max() < FP64_TOL
    # ... and the default vector gives something else for step() (and, for the 3-D kinds, for control())
    s2d, *_ = O.batch_step(kind, g[kind + "_s"], g[kind + "_a"])
    assert scaled_err(s2d, g[kind + "_s2"]).max() > 1e-5
    cd = scaled_err(O.batch_control(kind, g[kind + "_s"]), g[kind + "_ctrl"]).max()
    assert (cd > 1e-3) if kind.startswith("quad3d") else (cd < FP64_TOL)


@pytest.mark.parametrize("kind", KINDS)
def test_lifetime_terminal_reward_once(kind, golden, built):
    """Q1: reset() never clears steps_beyond_done, so the terminal reward is 1.0 once per env lifetime."""
    g = golden[kind]
    sbd = None
    for s, a, s2, r, d, sb in zip(g["life_s"], g["life_a"], g["life_s2"], g["life_r"], g["life_d"], g["life_sbd"]):
        o, rr, dd, sbd = O.step(kind, s, a, sbd)
        assert scaled_err(o, s2).max() < FP64_TOL and abs(rr - r) < FP64_TOL and dd == d
        assert (-1 if sbd is None else sbd) == sb
    assert list(g["life_r"][g["life_d"]]) == [1.0, 0.0, 0.0]


@pytest.mark.parametrize("kind", KINDS)
def test_closed_loop_trajectories(kind, golden, built):
    """The reference test loop (control -> step -> reset on done, 400 steps, 4 seeds), replayed step by
    step from the recorded pre-step states."""
    g = golden[kind]
    for sfx in ([""] + (["_A"] if kind == "quad2d" else [])):
        p = O.default_params(kind, "A" if sfx else "B")
        S, A_, S2, R, D = (g["traj_" + k + sfx] for k in ("s", "a", "s2", "r", "d"))
        for e in range(S.shape[0]):
            a = O.batch_control(kind, S[e], params=p)
            assert scaled_err(a, A_[e]).max() < 1e-11
            sbd = None
            s2 = np.empty_like(S2[e])
            flips = 0
            for k in range(S.shape[1]):
                sb_in = sbd
                s2[k], r, d, sbd = O.step(kind, S[e, k], A_[e, k], sb_in, params=p)
                if scaled_err(s2[k], S2[e, k]).max() > 1e-11 and abs(O.tether_slack(kind, S[e, k], p)) < 1e-12:
                    # |tether| == L to the last bit right after a projection step: the reference's
                    # branch hangs on the rounding of NumPy's BLAS dot; the other branch must match
                    taut = O.tether_slack(kind, S[e, k], p) >= 0
                    s2[k], r, d, sbd = O.step(kind, S[e, k], A_[e, k], sb_in, params=p, force_taut=int(not taut))
                    flips += 1
                assert abs(r - R[e, k]) < 1e-11 and d == D[e, k]
            assert scaled_err(s2, S2[e]).max() < 1e-11
            assert flips < 40
            # free-running (no teacher forcing) between resets the fp64 restatement also tracks
            if kind not in ("quad2d", "quad3d"):
                continue  # the tether branch is a coin flip on the last bit in closed loop (see above)
            k0 = 0
            s = S[e, 0].copy()
            for k in range(min(60, S.shape[1])):
                if D[e, k]:
                    break
                a1 = O.control(kind, s, params=p)
                s, _, _, _ = O.step(kind, s, a1, None, params=p)
                k0 = k
            assert scaled_err(s, S2[e, k0]).max() < 1e-8


def test_known_answer_vectors(built):
    """SURVEY.md section 8a KATs (fp64, one step, steps_beyond_done=None)."""
    s3 = [.1, -.2, .3, .9, .1, -.2, .3, .5, -.4, .2]
    o, r, d, _ = O.step("quad3d", s3, [9, .1, -.2, .3])
    exp = [0.10485789473684212, -0.2041421052631579, 0.3019126315789474, 0.8992818151535404, 0.10046169025843832,
           -0.20092338051687664, 0.30138507077531496, 0.47157894736842104, -0.428421052631579, 0.18252631578947368]
    assert np.abs(o - exp).max() < 1e-15 and abs(r - -0.3792366205113136) < 1e-15 and not d
    o, r, d, _ = O.step("quad3d_sl", s3 + [.9, .8, -.9, .2, .1, -.3], [9, .1, -.2, .3])
    assert np.abs(o[10:13] - [0.784131713484662, 0.6523776998250504, -0.725256641476481]).max() < 1e-14
    assert np.abs(o[13:16] - [-0.05769207648966829, -0.22454806037021033, -0.008756713518731418]).max() < 1e-14
    assert abs(r - -1.2515815607970424) < 1e-14
    o, r, d, _ = O.step("quad3d_sl", s3 + [.6, .5, -.9, .2, .1, -.3], [9, .1, -.2, .3])  # slack
    assert np.abs(o[:10] - exp).max() < 1e-15
    assert np.abs(o[10:] - [0.602, 0.501, -0.90349, 0.2, 0.1, -0.398]).max() < 1e-15
    o, r, d, _ = O.step("quad2d", [.1, -.2, .3, .5, -.4], [.9, .7])
    assert np.abs(o - [0.10486701590700241, -0.2040600985798935, 0.307, 0.47340318140047943,
                       -0.41201971597869547]).max() < 1e-15
    o, r, d, _ = O.step("quad2d_sl", [.1, -.2, .3, .5, -.4, .3, -.9, .2, .1], [9, .7])
    assert np.abs(o - [0.10463469123517606, -0.20429804803927143, 0.307, 0.4756460823450701, -0.4198698692847627,
                       0.24100888025218947, -0.6853407494390036, 0.3310463719527571, -0.46086346128702516]).max() < 1e-14
    assert abs(r - -0.22953455304711115) < 1e-15


def test_philox_known_answers(built):
    """Random123 kat_vectors for philox4x32-10."""
    assert list(O.philox([0] * 4, [0] * 2)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(O.philox([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("kind", KINDS)
def test_reset_distribution(kind, built):
    """reset draws every component from U[-1,1) (quadrotor3d.py:184) - check range and moments."""
    s = O.reset_states(kind, 3, np.arange(4096), 0)
    assert s.shape == (4096, NS[kind]) and s.dtype == np.float32
    assert s.min() >= -1.0 and s.max() < 1.0
    assert abs(s.mean()) < 0.02 and abs(s.var() - 1 / 3) < 0.02
    assert not np.array_equal(s, O.reset_states(kind, 3, np.arange(4096), 1))
    a = O.random_actions(kind, 3, np.arange(2048), 17, 0.0, 10.0)
    assert a.shape == (2048, NA[kind]) and a.min() >= 0.0 and a.max() < 10.0 and abs(a.mean() - 5) < 0.2


def test_quaternion_stand_in_against_independent_implementations():
    """SURVEY 8c's one soft link: pyquaternion (requirements.txt:1, call sites quadrotor3d.py:94,96,101-102,139,166-169,176) is
    absent from the reference tree and from this image, so the reference's 3-D rows run on ref_harness's restatement of its
    0.9.x algorithm.  Every member of row a10 is checked against an independent implementation (SciPy's Rotation for
    rotation_matrix / __mul__ / Quaternion(matrix=), a component-wise Hamilton product for derivative) - needs no reference tree."""
    import ref_harness as rh

    e_rot, e_mul, e_mat, e_der = rh.selfcheck_quaternion(1500)
    assert e_rot < 1e-14 and e_mul < 1e-14 and e_mat < 1e-14 and e_der < 1e-14, (e_rot, e_mul, e_mat, e_der)
    # members with a fixed answer: copies do not normalise, Quaternion(Quaternion) shares storage, rotation_matrix normalises
    # self.q IN PLACE (what makes quadrotor3d.py:101's derivative see the unit quaternion), conjugate does not
    q = rh.Quaternion([2.0, 0.0, 0.0, 0.0])
    assert list(q.elements) == [2.0, 0.0, 0.0, 0.0] and rh.Quaternion(q).q is q.q
    assert list(q.conjugate.elements) == [2.0, -0.0, -0.0, -0.0]
    q.rotation_matrix
    assert list(q.elements) == [1.0, 0.0, 0.0, 0.0]
    almost = rh.Quaternion([1.0 + 1e-15, 0.0, 0.0, 0.0])      # |1 - |q|^2| < 1e-14: left alone
    almost.rotation_matrix
    assert almost.elements[0] == 1.0 + 1e-15
    with pytest.raises(ValueError):
        rh.Quaternion(matrix=np.eye(3) * 1.01)                 # the orthogonality check of the real package


def test_quaternion_stand_in_equals_the_real_package_when_it_is_importable():
    """Self-enabling: with the real pyquaternion importable (not in this image) the harness runs the reference on it
    (ref_harness._install_stubs) and the stand-in must agree with it member by member."""
    import ref_harness as rh

    diffs = rh.compare_with_real(1000)
    if diffs is None:
        pytest.skip("pyquaternion is not installed here: the stand-in is checked against SciPy only")
    assert max(diffs.values()) < 1e-15, diffs


def test_oracle_vs_live_reference(built):
    """Authoring container only: the oracle against the reference files executed now."""
    import ref_harness as rh

    if not rh.available():
        pytest.skip("reference tree not mounted (GPU box)")
    assert max(rh.selfcheck_quaternion(500)) < 1e-14
    rng = np.random.RandomState(99)
    for kind in KINDS:
        env = rh.RefEnv(kind)
        for i in range(300):
            s = rng.uniform(-2, 2, NS[kind])
            a = rng.uniform(-12, 12, NA[kind])
            env.set_state(s, sbd=None)
            c_ref = env.control()
            o_ref, r_ref, d_ref = env.step(a)
            o, r, d, _ = O.step(kind, s, a)
            assert scaled_err(o, o_ref).max() < FP64_TOL and abs(r - r_ref) < FP64_TOL and d == d_ref
            assert scaled_err(O.control(kind, s), c_ref).max() < 1e-11


def test_numpy_per_env_restatement_matches_golden(golden, built):
    """The per-env NumPy stand-in (oracle/numpy_ref.py, used as the interpreter-bound CPU baseline) against the
    reference's golden vectors and the C oracle."""
    from numpy_ref import Quadrotor3DNumpy

    g = golden["quad3d"]
    env = Quadrotor3DNumpy()
    for i in range(0, len(g["step_s"]), 3):
        env.state, env.steps_beyond_done = g["step_s"][i].copy(), None
        o, r, d, _ = env.step(g["step_a"][i])
        assert scaled_err(o, g["step_s2"][i]).max() < FP64_TOL and abs(r - g["step_r"][i]) < FP64_TOL and d == g["step_d"][i]
    for i in range(0, len(g["ctrl_s"]), 3):
        env.state = g["ctrl_s"][i].copy()
        assert scaled_err(env.control(), g["ctrl_a"][i]).max() < 1e-11
        assert scaled_err(env.control(), O.control("quad3d", g["ctrl_s"][i])).max() < 1e-11
