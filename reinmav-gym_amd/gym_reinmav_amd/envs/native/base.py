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
        self.observation_space = Box(low=-10.0, high=10.0, shape=(nS,), dtype=np.float64)   # quadrotor3d.py:71 dtype=np.float
        self.viewer = None
        self._seed_value = self._fresh_seed() if seed is None else int(seed)
        # gym.Env semantics: no auto-reset, no Monitor; the constructor seeds then resets
        # (quadrotor3d.py:73-74), which rmav_create does as well.
        self._batch = BatchedQuadrotor(kind, 1, device=device, seed=self._seed_value, auto_reset=False,
                                       track_episodes=False, reading_2d=self._reading_2d)
        self._dim = 2 if nS in (5, 9) else 3
        self._has_load = nS in (9, 16)
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

    # The reference reads these on EVERY call (quadrotor3d.py:86 `self.mass`, :96-102 `self.dt`, `self.g`, :148 `self.ref_pos`,
    # :162 `self.ref_vel`, :107-110 the thresholds; quadrotor3d_slungload.py:101-128 `self.load_mass`, `self.tether_length`), so
    # assigning one of them - or writing an element of the array-valued ones - changes the next step() / control().  Here
    # they are views of the handle's rmav_params: a getter reads it, a setter writes it back through rmav_set_params (the
    # derived constants are recomputed at the next launch) and drops the cached control() action.
    def _set_param(self, **kv):
        p = self._batch.params
        for k, v in kv.items():
            if k in ("ref_pos", "ref_vel", "g_vec"):
                v = np.asarray(v, dtype=np.float64).reshape(-1)
                if v.shape != (self._dim,):
                    raise ValueError(f"{k} must have {self._dim} components")
                arr = getattr(p, k)
                for i in range(self._dim):
                    arr[i] = float(v[i])
            else:
                setattr(p, k, float(v))
        self._batch.params = p          # rmav_set_params validates (raises RmavError on e.g. mass <= 0)
        self._ctrl_valid = False

    def _vec(self, read, setter):
        """A snapshot of a vector-valued constant whose ELEMENT writes go through: `a[key] = value` applies that one assignment to
        the handle's CURRENT value (not to the snapshot's other, possibly stale, elements) and refreshes the snapshot."""
        a = _WriteThrough(read())

        def on_write(key, value):
            cur = np.array(read(), dtype=np.float64)
            cur[key] = value
            setter(cur)
            np.ndarray.__setitem__(a, slice(None), read())
        a._on_write = on_write
        return a

    def _read_vec(self, name):
        return np.array(list(getattr(self._batch.params, name))[:self._dim], dtype=np.float64)

    def _read_g(self):
        return self._read_vec("g_vec")

    mass = property(lambda self: self._batch.params.mass, lambda self, v: self._set_param(mass=v))
    dt = property(lambda self: self._batch.params.dt, lambda self, v: self._set_param(dt=v))
    pos_threshold = property(lambda self: self._batch.params.pos_limit, lambda self, v: self._set_param(pos_limit=v))
    vel_threshold = property(lambda self: self._batch.params.vel_limit, lambda self, v: self._set_param(vel_limit=v))

    def _load_attr(name):  # noqa: N805 - class-body helper: slung-load classes only, AttributeError elsewhere like the reference
        def get(self):
            if not self._has_load:
                raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")
            return getattr(self._batch.params, name)

        def set_(self, v):
            if not self._has_load:      # the reference would just grow an unused attribute
                object.__setattr__(self, "_unused_" + name, v)
                return
            self._set_param(**{name: v})
        return property(get, set_)

    load_mass = _load_attr("load_mass")
    tether_length = _load_attr("tether_length")
    del _load_attr

    @property
    def g(self):
        """Gravity VECTOR like quadrotor3d.py:47 / quadrotor2d.py:46: any direction, read by every step() (and by the 3-D control())."""
        return self._vec(self._read_g, self._set_g)

    @g.setter
    def g(self, v):
        self._set_g(v)

    def _set_g(self, v):
        self._set_param(g_vec=v)

    @property
    def ref_pos(self):
        return self._vec(lambda: self._read_vec("ref_pos"), lambda v: self._set_param(ref_pos=v))

    @ref_pos.setter
    def ref_pos(self, v):
        self._set_param(ref_pos=v)

    @property
    def ref_vel(self):
        return self._vec(lambda: self._read_vec("ref_vel"), lambda v: self._set_param(ref_vel=v))

    @ref_vel.setter
    def ref_vel(self, v):
        self._set_param(ref_vel=v)


class _WriteThrough(np.ndarray):
    """The array a vector-valued attribute getter hands out: ``env.ref_pos[2] = 1.0`` must move the set-point as it does in the
    reference (whose ``self.ref_pos`` IS the array control() reads), so element writes are pushed back to the handle.
    Arithmetic on it yields plain ndarrays."""

    _on_write = None

    def __new__(cls, values):
        return np.asarray(values, dtype=np.float64).view(cls)

    def __array_finalize__(self, obj):
        self._on_write = None           # views / results of arithmetic do not write through

    def __setitem__(self, key, value):
        if self._on_write is not None:
            self._on_write(key, value)      # validates, writes the handle, refreshes this snapshot
        else:
            np.ndarray.__setitem__(self, key, value)

    def __array_wrap__(self, out, context=None, return_scalar=False):
        out = np.asarray(out)
        return out[()] if return_scalar else out
