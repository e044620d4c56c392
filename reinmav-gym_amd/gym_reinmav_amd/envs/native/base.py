"""Shared implementation of the four gym-shaped single-env classes (batch = 1 on the GPU).

The reference classes (gym_reinmav/envs/native/quadrotor{2d,3d}[_slungload].py) are old-gym
``gym.Env`` duck types: ``seed(seed=None) -> [seed]``, ``reset() -> obs``,
``step(action) -> (obs, reward, done, {})``, ``control() -> action``, ``render``, ``close`` and public
attributes ``state``, ``steps_beyond_done``, ``mass``, ``dt``, ``g``, ``ref_pos``, ``ref_vel``,
``pos_threshold``, ``vel_threshold``, ``action_space``, ``observation_space``.  The classes here
keep that surface; the arithmetic runs in the HIP kernels through ``librmav.so`` (one env, one lane).
Observations and rewards are returned as float64 / Python float like the reference, but their
values are the fp32 results of the device path.
"""
from __future__ import annotations

import numpy as np

from ... import _abi as A
from ...core import BatchedQuadrotor
from ...spaces import Box

try:  # pragma: no cover
    import gym as _gym  # type: ignore

    _EnvBase = _gym.Env
except Exception:  # pragma: no cover
    try:
        import gymnasium as _gym  # type: ignore

        _EnvBase = _gym.Env
    except Exception:
        _EnvBase = object


class NativeQuadrotorEnv(_EnvBase):
    metadata = {"render.modes": ["human"]}
    _kind = None          # set by subclasses
    _action_box = None    # (low, high, dtype)
    _reading_2d = None

    def __init__(self, device: int = 0, seed=None):
        kind = A.KIND_BY_NAME[self._kind]
        nS, nA = A.STATE_DIM[kind], A.ACTION_DIM[kind]
        lo, hi, dt = self._action_box
        self.action_space = Box(low=lo, high=hi, shape=(nA,), dtype=dt)
        self.observation_space = Box(low=-10.0, high=10.0, shape=(nS,), dtype=dt)
        self.viewer = None
        self._seed_value = self._fresh_seed() if seed is None else int(seed)
        # gym.Env semantics: no auto-reset, no Monitor; the constructor seeds then resets
        # (quadrotor3d.py:73-74), which rmav_create does as well.
        self._batch = BatchedQuadrotor(kind, 1, device=device, seed=self._seed_value, auto_reset=False,
                                       track_episodes=False, reading_2d=self._reading_2d)
        p = self._batch.params
        self.mass, self.dt = p.mass, p.dt
        self.g = np.array([0.0, -p.g]) if nS in (5, 9) else np.array([0.0, 0.0, -p.g])
        dim = 2 if nS in (5, 9) else 3
        self.ref_pos = np.array(list(p.ref_pos)[:dim])
        self.ref_vel = np.array(list(p.ref_vel)[:dim])
        if nS in (9, 16):
            self.load_mass, self.tether_length = p.load_mass, p.tether_length
        # Lean per-call path: preallocated host arrays and cached ctypes pointers, so a step() is one ABI call
        # (= one kernel launch + one stream synchronise through the handle's pinned block) plus a few small
        # NumPy conversions.  The launch also evaluates control() on the new state (rmav_step_control), so the
        # reference's test loop "action = env.control(); env.step(action)" costs ONE launch per iteration:
        # control() returns the cached action until something else changes the state.
        import ctypes as C

        self._a = np.zeros((1, nA), np.float32)
        self._o = np.zeros((1, nS), np.float32)
        self._r = np.zeros(1, np.float32)
        self._d = np.zeros(1, np.uint8)
        self._c = np.zeros((1, nA), np.float32)
        vp = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
        self._pa, self._po, self._pr, self._pd, self._pc = vp(self._a), vp(self._o), vp(self._r), vp(self._d), vp(self._c)
        self._step_control = self._batch._lib.rmav_step_control
        self._hnd = self._batch._h
        self._ctrl_valid = False

    # -- gym.Env ----------------------------------------------------------------------------------------
    @staticmethod
    def _fresh_seed() -> int:
        return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).astype(np.uint64) @ np.array(
            [1, 1 << 32], dtype=np.uint64))

    def seed(self, seed=None):
        self._seed_value = self._fresh_seed() if seed is None else int(seed)
        self._ctrl_valid = False
        self._batch.seed(self._seed_value)
        return [self._seed_value]

    def reset(self):
        self._ctrl_valid = False
        return self._batch.reset()[0].astype(np.float64)

    def step(self, action):
        self._a[0, :] = action          # casts float64 -> float32, raises on a wrong length like the reference's unpacking
        if self._hnd is None:
            raise A.RmavError(A.ERR_INVALID, "step() on a closed env")
        rc = self._step_control(self._hnd, self._pa, self._po, self._pr, self._pd, self._pc, A.HOST, A.AOS)
        if rc < 0:
            A.check(rc)
        self._ctrl_valid = True
        return self._o[0].astype(np.float64), float(self._r[0]), bool(self._d[0]), {}

    def control(self):
        if self._ctrl_valid:            # evaluated by the last step()'s launch on the state it left behind
            return self._c[0].astype(np.float64)
        return self._batch.control()[0].astype(np.float64)

    def render(self, mode="human", close=False):
        raise NotImplementedError("rendering (pyglet / vpython in the reference) is outside the GPU hot path")

    def close(self):
        self._hnd = None                # the cached raw handle must not outlive the library's
        self._ctrl_valid = False
        self._batch.close()

    # -- public attributes of the reference ---------------------------------------------------------------
    @property
    def state(self):
        return self._batch.get_state()[0].astype(np.float64)

    @state.setter
    def state(self, s):
        self._ctrl_valid = False
        self._batch.set_state(np.asarray(s, dtype=np.float32).reshape(1, -1))

    @property
    def steps_beyond_done(self):
        v = int(self._batch.get_sbd()[0])
        return None if v < 0 else v

    @steps_beyond_done.setter
    def steps_beyond_done(self, v):
        self._ctrl_valid = False
        self._batch.set_sbd(np.array([-1 if v is None else int(v)], dtype=np.int32))

    @property
    def pos_threshold(self):
        return self._batch.params.pos_limit

    @property
    def vel_threshold(self):
        return self._batch.params.vel_limit
