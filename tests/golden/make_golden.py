#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference itself (authoring container only).

The reference (ethz-asl/reinmav-gym, mounted read-only at /root/reference) ships no golden vectors
and no asserting tests, so these fixtures are produced by *running the reference's own
step()/control() code* through ``oracle/ref_harness.py`` (which loads the four files of
``gym_reinmav/envs/native/`` by path under small stand-ins for gym / pyquaternion / legacy NumPy).
The fixtures hold data only (inputs and the reference's outputs, fp64); no reference text.

    python tests/golden/make_golden.py          # rewrites tests/golden/<kind>.npz

Contents per kind (all float64 unless noted):
  step_s [n,nS], step_a [n,nA]          inputs (fp32-representable values)
  step_s2 [n,nS], step_r [n], step_d [n] reference outputs with steps_beyond_done=None
  ctrl_s [m,nS], ctrl_a [m,nA]          control() inputs/outputs
  life_s, life_a, life_s2, life_r, life_d, life_sbd   3-episode lifetime run (terminal reward 1,0,0;
                                        sbd = steps_beyond_done after the step, -1 = None)
  traj_s, traj_a, traj_s2, traj_r, traj_d  [4 seeds, 400 steps, ...] closed loop control()->step()
                                        with reset-on-done exactly like test/test_quadrotor3d.py:16-22
  (quad2d only) *_A variants of step_d/step_r/traj_* under reading A of quadrotor2d.py:95-98
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ref_harness as rh  # noqa: E402

NS = {"quad2d": 5, "quad2d_sl": 9, "quad3d": 10, "quad3d_sl": 16}
NA = {"quad2d": 2, "quad2d_sl": 2, "quad3d": 4, "quad3d_sl": 4}
# action Box of each env (quadrotor3d.py:70, quadrotor3d_slungload.py:75, quadrotor2d.py:62,
# quadrotor2d_slungload.py:68); step() never clips, so half the cases go beyond it.
BOX = {"quad2d": (-10, 10), "quad2d_sl": (-10, 10), "quad3d": (0, 10), "quad3d_sl": (-10, 10)}
LIMITS = {"quad2d": (3.0, 2.0), "quad2d_sl": (2.0, 10.0), "quad3d": (3.0, 10.0), "quad3d_sl": (3.0, 10.0)}
TETHER = {"quad2d_sl": 0.5, "quad3d_sl": 1.5}


def f32r(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def _pos_slice(kind):  # body whose position terminates the episode
    return {"quad2d": slice(0, 2), "quad2d_sl": slice(5, 7), "quad3d": slice(0, 3), "quad3d_sl": slice(10, 13)}[kind]


def _vel_slice(kind):  # body whose velocity terminates the episode
    return {"quad2d": slice(3, 5), "quad2d_sl": slice(7, 9), "quad3d": slice(7, 10), "quad3d_sl": slice(7, 10)}[kind]


def step_inputs(kind, rng, n_random=1024):
    nS, nA = NS[kind], NA[kind]
    lo, hi = BOX[kind]
    S, A = [], []
    for i in range(n_random):
        sc = 1.0 if i % 2 == 0 else 3.0
        S.append(rng.uniform(-sc, sc, nS))
        if i % 4 < 2:
            A.append(rng.uniform(lo, hi, nA))
        else:
            A.append(rng.uniform(-20, 20, nA))
    # --- forced branches -----------------------------------------------------------------------
    plim, vlim = LIMITS[kind]
    ps, vs = _pos_slice(kind), _vel_slice(kind)
    dim = ps.stop - ps.start
    for j in range(96):
        s = rng.uniform(-0.5, 0.5, nS)
        a = rng.uniform(lo, hi, nA)
        u = rng.normal(size=dim)
        u /= np.linalg.norm(u)
        mode = j % 6
        if mode == 0:      # far outside in position
            s[ps] = u * (plim + 0.5)
        elif mode == 1:    # position just inside / outside the limit
            s[ps] = u * (plim + (1e-4 if j % 12 < 6 else -1e-4))
            s[vs] = 0.0
        elif mode == 2:    # far outside in velocity
            s[vs] = u * (vlim + 1.0)
        elif mode == 3:    # velocity near the limit
            s[vs] = u * (vlim + (2e-2 if j % 12 < 6 else -2e-2))
        elif mode == 4:    # comfortably alive
            s[:] = rng.uniform(-0.2, 0.2, nS)
        else:              # zero action
            a[:] = 0.0
        S.append(s)
        A.append(a)
    if kind in TETHER:     # taut / slack / near the switch
        L = TETHER[kind]
        qs = slice(0, dim)
        for j in range(96):
            s = rng.uniform(-0.6, 0.6, nS)
            a = rng.uniform(lo, hi, nA)
            u = rng.normal(size=dim)
            u /= np.linalg.norm(u)
            d = [L * 1.3, L * 0.7, L + 1e-3, L - 1e-3, L * 1.01, L * 2.0][j % 6]
            s[ps] = s[qs] + u * d
            S.append(s)
            A.append(a)
    if kind == "quad2d":   # thrust clamp (quadrotor2d.py:76-77): negative commands
        for j in range(16):
            s = rng.uniform(-0.5, 0.5, nS)
            a = rng.uniform(-10, 0, nA)
            S.append(s)
            A.append(a)
    if kind in ("quad3d", "quad3d_sl"):   # unit quaternions and near-unit (normalise threshold 1e-14)
        for j in range(32):
            s = rng.uniform(-0.5, 0.5, nS)
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            s[3:7] = q
            S.append(s)
            A.append(rng.uniform(lo, hi, nA))
    return f32r(np.array(S)), f32r(np.array(A))


def run_steps(kind, S, A, reading="B"):
    env = rh.RefEnv(kind, reading_2d=reading)
    S2, R, D = [], [], []
    for s, a in zip(S, A):
        env.set_state(s, sbd=None)
        o, r, d = env.step(a)
        S2.append(o)
        R.append(r)
        D.append(d)
    return np.array(S2), np.array(R), np.array(D, dtype=bool)


def run_control(kind, rng, m=512):
    nS = NS[kind]
    env = rh.RefEnv(kind)
    S = []
    for i in range(m):
        sc = 1.0 if i % 2 == 0 else 3.0
        s = rng.uniform(-sc, sc, nS)
        if kind.startswith("quad3d") and i % 4 == 3:  # unit attitude
            s[3:7] /= np.linalg.norm(s[3:7])
        S.append(s)
    S = f32r(np.array(S))
    Aout = []
    for s in S:
        env.set_state(s)
        Aout.append(env.control())
    return S, np.array(Aout)


def run_lifetime(kind, rng, episodes=3, reading="B"):
    """Random actions, reset on done, never touching steps_beyond_done (Q1: rewards 1.0, 0.0, 0.0)."""
    lo, hi = BOX[kind]
    env = rh.RefEnv(kind, reading_2d=reading, seed=123)
    env.set_state(f32r(env.reset()))
    S, A, S2, R, D, SBD = [], [], [], [], [], []
    ep = 0
    while ep < episodes and len(S) < 4000:
        s = env.get_state()
        a = f32r(rng.uniform(lo, hi, NA[kind]))
        o, r, d = env.step(a)
        S.append(s); A.append(a); S2.append(o); R.append(r); D.append(d)
        SBD.append(-1 if env.sbd is None else env.sbd)
        if d:
            ep += 1
            env.set_state(f32r(env.reset()))
        else:
            env.set_state(f32r(o))  # fp32 storage model keeps the run replayable by fp32 paths
    # one extra step after done without reset: "step after done" branch (sbd increments, reward 0)
    return (np.array(S), np.array(A), np.array(S2), np.array(R), np.array(D, dtype=bool),
            np.array(SBD, dtype=np.int32))


def run_traj(kind, seeds=(0, 1, 2, 3), steps=400, reading="B"):
    """The reference test loop (test/test_quadrotor3d.py:16-22) without render()."""
    TS, TA, TS2, TR, TD = [], [], [], [], []
    for seed in seeds:
        env = rh.RefEnv(kind, reading_2d=reading, seed=seed)
        env.reset()
        S, A, S2, R, D = [], [], [], [], []
        for _ in range(steps):
            s = env.get_state()
            a = env.control()
            o, r, d = env.step(a)
            S.append(s); A.append(a); S2.append(o); R.append(r); D.append(d)
            if d:
                env.reset()
        TS.append(S); TA.append(A); TS2.append(S2); TR.append(R); TD.append(D)
    return np.array(TS), np.array(TA), np.array(TS2), np.array(TR), np.array(TD, dtype=bool)


def make_reinmav():
    """reinmav_env.py: the reference's own 400-step run (test/test_reinmav.py:16-22) plus perturbed single steps.
    Keys: run_s [400,13], run_t [400] (state / clock before each step), run_s2 [400,13], run_t2 [400];
    step_s [m,13], step_t [m], step_s2, step_t2; ctrl_s, ctrl_t, ctrl_fm [m,4] (built-in controller output)."""
    env = rh.RefReinmav()
    S, T, S2, T2 = [], [], [], []
    for _ in range(400):
        S.append(np.array(env.env.state, dtype=np.float64).ravel())
        T.append(env.env.t)
        s2, r, d, t2 = env.step()
        assert r == 90.0 and d is True
        S2.append(s2)
        T2.append(t2)
    out = {"run_s": np.array(S), "run_t": np.array(T), "run_s2": np.array(S2), "run_t2": np.array(T2)}
    rng = np.random.RandomState(15)
    ps, pt = [], []
    for i in range(192):
        k = rng.randint(0, 400)
        s = out["run_s"][k] + rng.normal(scale=0.02 if i % 2 else 0.002, size=13)
        ps.append(f32r(s))
        pt.append(float(rng.uniform(0, 4.5)) if i % 3 else float(out["run_t"][k]))
    ps, pt = np.array(ps), np.array(pt)
    ps2, pt2, fm = [], [], []
    for s, t in zip(ps, pt):
        fm.append(env.force_moment(s, t))
        env.set(s, t)
        s2, _, _, t2 = env.step()
        ps2.append(s2)
        pt2.append(t2)
    out.update({"step_s": ps, "step_t": pt, "step_s2": np.array(ps2), "step_t2": np.array(pt2), "ctrl_fm": np.array(fm)})
    path = os.path.join(HERE, "reinmav.npz")
    np.savez_compressed(path, **out)
    print("reinmav", {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")


# self.g is a public attribute the reference reads on every call (quadrotor3d.py:47,96-99,162; quadrotor2d.py:46,88): the same
# step / control cases with a TILTED gravity vector assigned to the env object - what rmav_params.g_vec carries (round 6)
GRAVITY = {"quad2d": (1.25, -8.5), "quad2d_sl": (-0.75, -9.0), "quad3d": (1.25, -0.75, -8.5), "quad3d_sl": (-1.0, 0.5, -9.25)}


def make_gravity():
    out = {}
    for kind in rh.KINDS:
        rng = np.random.RandomState({"quad2d": 21, "quad2d_sl": 22, "quad3d": 23, "quad3d_sl": 24}[kind])
        nS, nA = NS[kind], NA[kind]
        lo, hi = BOX[kind]
        n = 384
        S = f32r(np.array([rng.uniform(-(1.0 if i % 2 == 0 else 2.5), (1.0 if i % 2 == 0 else 2.5), nS) for i in range(n)]))
        A = f32r(np.array([rng.uniform(lo, hi, nA) for _ in range(n)]))
        if kind in TETHER:   # keep clear of the tether switch: the branch there hangs on the last bit of a norm
            L = TETHER[kind]
            dim = 2 if kind.startswith("quad2d") else 3
            ls = slice(5, 7) if dim == 2 else slice(10, 13)
            d = np.linalg.norm(S[:, ls] - S[:, 0:dim], axis=1)
            keep = np.abs(d - L) > 1e-3
            S, A = S[keep], A[keep]
        env = rh.RefEnv(kind)
        env.env.g = np.array(GRAVITY[kind], dtype=np.float64)
        S2, R, D, C = [], [], [], []
        for s_, a_ in zip(S, A):
            env.set_state(s_, sbd=None)
            C.append(env.control())
            env.set_state(s_, sbd=None)
            o, r, d = env.step(a_)
            S2.append(o)
            R.append(r)
            D.append(d)
        out[kind + "_g"] = np.array(GRAVITY[kind], dtype=np.float64)
        out[kind + "_s"], out[kind + "_a"] = S, A
        out[kind + "_s2"], out[kind + "_r"], out[kind + "_d"] = np.array(S2), np.array(R), np.array(D, dtype=bool)
        out[kind + "_ctrl"] = np.array(C)
    path = os.path.join(HERE, "gravity.npz")
    np.savez_compressed(path, **out)
    print("gravity", {k: v.shape for k, v in out.items() if k.endswith("_s")}, os.path.getsize(path), "bytes")


def main():
    assert rh.available(), "the reference tree is required to regenerate golden vectors"
    make_reinmav()
    make_gravity()
    errs = rh.selfcheck_quaternion()
    assert max(errs) < 1e-14, errs
    for kind in rh.KINDS:
        rng = np.random.RandomState({"quad2d": 11, "quad2d_sl": 12, "quad3d": 13, "quad3d_sl": 14}[kind])
        out = {}
        S, A = step_inputs(kind, rng)
        out["step_s"], out["step_a"] = S, A
        out["step_s2"], out["step_r"], out["step_d"] = run_steps(kind, S, A)
        out["ctrl_s"], out["ctrl_a"] = run_control(kind, rng)
        (out["life_s"], out["life_a"], out["life_s2"], out["life_r"], out["life_d"],
         out["life_sbd"]) = run_lifetime(kind, rng)
        (out["traj_s"], out["traj_a"], out["traj_s2"], out["traj_r"], out["traj_d"]) = run_traj(kind)
        if kind == "quad2d":
            s2a, out["step_r_A"], out["step_d_A"] = run_steps(kind, S, A, reading="A")
            assert np.array_equal(s2a, out["step_s2"])
            (out["traj_s_A"], out["traj_a_A"], out["traj_s2_A"], out["traj_r_A"],
             out["traj_d_A"]) = run_traj(kind, reading="A")
        path = os.path.join(HERE, kind + ".npz")
        np.savez_compressed(path, **out)
        print(kind, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes",
              "done frac", out["step_d"].mean(), "life rewards at done", out["life_r"][out["life_d"]])


if __name__ == "__main__":
    main()
