"""``reinmav-v0`` - drop-in for the reference's ``ReinmavEnv`` (reinmav_env.py:51-352).

A 13-state rigid body ``[x y z dx dy dz qw qx qy qz p q r]`` flown by a built-in PD controller along a
min-jerk trajectory; ``step()`` takes **no action** in the reference (``test/test_reinmav.py:16-22`` calls
``env.step()`` 400 times), returns reward 90.0 and ``done=True`` every step, and ``reset()`` returns the
current state unchanged.  All of that is kept.  The dynamics (50-or-51 Euler sub-steps of 1/5000 s, motor
mixing with clamp, quaternion-norm feedback) run in the HIP kernel through ``librmav.so`` in fp64 on fp32
state.  Extension: ``step(action)`` with ``action = (F, Mx, My, Mz)`` replaces the built-in controller
for that step.  Plotting (matplotlib/TkAgg in the reference) is out of scope.
"""
from __future__ import annotations

import numpy as np

from ... import _abi as A
from ...core import BatchedQuadrotor
from ...spaces import Box
from .base import _EnvBase


class ReinmavEnv(_EnvBase):
    metadata = {"render.modes": ["human"]}

    def __init__(self, device: int = 0):
        self.arm_length, self.mass, self.gravity = 0.0860, 0.1800, 9.8100      # reinmav_env.py:55-57
        self.min_force, self.max_force = 0.0, 3.5316                             # :58-59
        self.Inertia = np.array([[0.00025, 0, 2.55e-06], [0, 0.000232, 0], [2.55e-06, 0, 0.0003738]])
        self.invInertia = np.linalg.inv(self.Inertia)
        self.dt = 1 / 100                                                        # :73
        self.init_state = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]   # :79
        self.action_space = Box(low=-np.inf, high=np.inf, shape=(4,), dtype=np.float32)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(13,), dtype=np.float32)
        self._batch = BatchedQuadrotor(A.REINMAV, 1, device=device, auto_reset=False, track_episodes=False)

    def seed(self, seed=None):
        return [seed]   # the env has no randomness (reinmav_env.py:86-89 only creates an unused RandomState)

    def step(self, action=None):
        if action is None:   # the reference: built-in controller, evaluated at every Euler sub-step
            tr = self._batch.rollout(1, mode="controller", layout="aos", want=("obs", "rew", "done"))
            return tr["obs"][0, 0].astype(np.float64), float(tr["rew"][0, 0]), bool(tr["done"][0, 0]), {}
        a = np.asarray(action, dtype=np.float32).reshape(1, 4)
        obs, rew, done = self._batch.step(a)
        return obs[0].astype(np.float64), float(rew[0]), bool(done[0]), {}

    def control(self):
        """(F, Mx, My, Mz) the built-in controller commands at the current (state, t)  (reinmav_env.py:306-337)."""
        return self._batch.control()[0].astype(np.float64)

    def trj_gen(self, t):
        """The min-jerk reference the built-in controller tracks (reinmav_env.py:128-136), host side, for callers that plot or
        inspect it: a quintic in s = clip(t, 0, 4 s) / 4 s and its first two time derivatives, the same profile on x, y, z (and
        yaw) -> [x, y, z, vx, vy, vz, ax, ay, az, yaw, yaw rate].  The device path evaluates the same polynomial every sub-step
        (csrc/rmav_math.hpp, Env<REINMAV>)."""
        T = 4.0
        s = min(max(float(t), 0.0), T) / T
        pos = s ** 3 * (10.0 + s * (-15.0 + 6.0 * s))
        vel = s ** 2 * (30.0 + s * (-60.0 + 30.0 * s)) / T
        acc = s * (60.0 + s * (-180.0 + 120.0 * s)) / T ** 2
        return [pos] * 3 + [vel] * 3 + [acc] * 3 + [pos, vel]

    def reset(self):
        return self.state   # reinmav_env.py:348-351: returns the current state, changes nothing

    def render(self, mode="human", close=False):
        raise NotImplementedError("plotting / rendering is outside the GPU hot path")

    def close(self):
        self._batch.close()

    @property
    def state(self):
        return self._batch.get_state()[0].astype(np.float64)

    @state.setter
    def state(self, s):
        self._batch.set_state(np.asarray(s, dtype=np.float32).reshape(1, 13))

    @property
    def t(self):
        return float(self._batch.get_time()[0])

    @t.setter
    def t(self, v):
        self._batch.set_time(float(v))
