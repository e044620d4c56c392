"""Container-only loader for the reference's four native env files.  TEST INFRASTRUCTURE.

This module exists so that the oracle (``oracle/rmav_oracle.c``) can be pinned against outputs of the
reference *itself*, executed in the authoring container.  It never copies reference text: each file
is read from ``/root/reference`` at run time and ``exec``'d in a fresh module namespace whose
imports resolve to the small stand-ins defined below.  On the GPU box ``/root/reference`` does not
exist and :func:`available` returns False; nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py``
imports this file.

Stand-ins (all our own code):

* ``gym``          - minimal duck type (``Env``, ``spaces.Box``, ``utils.seeding.np_random``,
                     ``logger.warn``).  The reference only uses these names
                     (``gym_reinmav/envs/native/quadrotor3d.py:34-40``).
* ``pyquaternion`` - the reference depends on ``pyquaternion>0.9`` (``requirements.txt:1``), which is
                     neither vendored under /root/reference nor installed here.  ``Quaternion`` below
                     restates the published pyquaternion 0.9.x algorithm for exactly the members the
                     hot path calls: ctor from a 4-sequence (copy, no
                     normalise), ctor from a Quaternion (shares ``q``), ``Quaternion(vector=)``,
                     ``Quaternion(matrix=)`` (orthogonality check + 4-branch trace method),
                     ``rotation_matrix`` (normalises ``self.q`` first when ``|1-|q|^2| >= 1e-14``),
                     ``derivative``, ``conjugate``, ``__mul__``/``__rmul__`` (Hamilton), ``elements``.
                     :func:`selfcheck_quaternion` cross-checks it against the independent
                     ``scipy.spatial.transform.Rotation`` (``rotation_matrix``, ``__mul__``, ``Quaternion(matrix=)``)
                     and a component-wise Hamilton product (``derivative``).  When the REAL package is importable
                     :func:`_install_stubs` uses it instead (``QUATERNION_SOURCE`` says which ran) and
                     :func:`compare_with_real` checks stand-in == real member by member.
* legacy NumPy     - the reference predates NumPy 1.24: it uses ``np.float``
                     (``quadrotor3d.py:70-71``) and builds ragged ``np.array((scalar, array([th]),...))``
                     (``quadrotor2d.py:113``).  ``_LegacyNumpy`` forwards everything to real NumPy and
                     restores those two behaviours.

``quadrotor2d.py`` does not parse as shipped (``quadrotor2d.py:95-98``: line 96 lacks the trailing
backslash).  Reading "B" appends that one character in memory (done = |p|>3 or |v|>10 or |v|>2,
i.e. |p|>3 or |v|>2); reading "A" drops the two dangling lines (done = |p|>3 or |v|>10).
"""
from __future__ import annotations

import math
import os
import sys
import types
import warnings

import numpy as np

REF_ROOT = os.environ.get("RMAV_REFERENCE_ROOT", "/root/reference")
NATIVE_DIR = os.path.join(REF_ROOT, "gym_reinmav", "envs", "native")

KINDS = ("quad2d", "quad2d_sl", "quad3d", "quad3d_sl")
_FILES = {
    "quad2d": ("quadrotor2d.py", "Quadrotor2D"),
    "quad2d_sl": ("quadrotor2d_slungload.py", "Quadrotor2DSlungload"),
    "quad3d": ("quadrotor3d.py", "Quadrotor3D"),
    "quad3d_sl": ("quadrotor3d_slungload.py", "Quadrotor3DSlungload"),
}


def available() -> bool:
    return os.path.isdir(NATIVE_DIR)


# ------------------------------------------------------------------------------------------------
# pyquaternion stand-in (published 0.9.x algorithm, members used by the hot path only)
# ------------------------------------------------------------------------------------------------
class Quaternion:
    def __init__(self, *args, **kwargs):
        if len(args) == 1 and not kwargs:
            a = args[0]
            if isinstance(a, Quaternion):
                self.q = a.q  # shared storage, as in pyquaternion
                return
            try:
                r = float(a)
                self.q = np.array([r, 0.0, 0.0, 0.0])
                return
            except TypeError:
                pass
            self.q = self._seq(a, 4)
        elif len(args) == 4:
            self.q = self._seq(args, 4)
        elif not args and "matrix" in kwargs:
            self.q = self._from_matrix(np.asarray(kwargs["matrix"], dtype=float))
        elif not args and "array" in kwargs:
            self.q = self._seq(kwargs["array"], 4)
        elif not args and ("vector" in kwargs or "scalar" in kwargs):
            s = float(kwargs.get("scalar", 0.0) or 0.0)
            v = kwargs.get("vector", None)
            v = np.zeros(3) if v is None else self._seq(v, 3)
            self.q = np.array([s, v[0], v[1], v[2]])
        elif not args and not kwargs:
            self.q = np.array([1.0, 0.0, 0.0, 0.0])
        else:
            raise TypeError("unsupported Quaternion constructor form in stand-in")

    @staticmethod
    def _seq(seq, n):
        if len(seq) != n:
            raise ValueError("Unexpected number of elements in sequence")
        return np.asarray([float(e) for e in seq])

    @staticmethod
    def _from_matrix(R, rtol=1e-05, atol=1e-08):
        if R.shape == (4, 4):
            R = R[:-1][:, :-1]
        if R.shape != (3, 3):
            raise ValueError("Invalid matrix shape")
        if not np.allclose(np.dot(R, R.conj().transpose()), np.eye(3), rtol=rtol, atol=atol):
            raise ValueError("Matrix must be orthogonal, i.e. its transpose should be its inverse")
        if not np.isclose(np.linalg.det(R), 1.0, rtol=rtol, atol=atol):
            raise ValueError("Matrix must be special orthogonal i.e. its determinant must be +1.0")
        m = R.conj().transpose()
        if m[2, 2] < 0:
            if m[0, 0] > m[1, 1]:
                t = 1 + m[0, 0] - m[1, 1] - m[2, 2]
                q = [m[1, 2] - m[2, 1], t, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]]
            else:
                t = 1 - m[0, 0] + m[1, 1] - m[2, 2]
                q = [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t, m[1, 2] + m[2, 1]]
        else:
            if m[0, 0] < -m[1, 1]:
                t = 1 - m[0, 0] - m[1, 1] + m[2, 2]
                q = [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t]
            else:
                t = 1 + m[0, 0] + m[1, 1] + m[2, 2]
                q = [t, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]]
        q = np.array(q).astype("float64")
        q *= 0.5 / math.sqrt(t)
        return q

    # -- algebra --------------------------------------------------------------------------------
    def _q_matrix(self):
        w, x, y, z = self.q
        return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])

    def _q_bar_matrix(self):
        w, x, y, z = self.q
        return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])

    def __mul__(self, other):
        if isinstance(other, Quaternion):
            return Quaternion(array=np.dot(self._q_matrix(), other.q))
        return self * Quaternion(other)

    def __rmul__(self, other):
        return Quaternion(other) * self

    def _sum_of_squares(self):
        return np.dot(self.q, self.q)

    @property
    def norm(self):
        return math.sqrt(self._sum_of_squares())

    def is_unit(self, tolerance=1e-14):
        return abs(1.0 - self._sum_of_squares()) < tolerance

    def _normalise(self):
        if not self.is_unit():
            n = self.norm
            if n > 0:
                self.q = self.q / n  # rebinds; a caller's array is untouched

    @property
    def rotation_matrix(self):
        self._normalise()
        product_matrix = np.dot(self._q_matrix(), self._q_bar_matrix().conj().transpose())
        return product_matrix[1:][:, 1:]

    def derivative(self, rate):
        rate = self._seq(rate, 3)
        return 0.5 * self * Quaternion(vector=rate)

    @property
    def conjugate(self):
        return Quaternion(scalar=self.q[0], vector=-self.q[1:4])

    @property
    def elements(self):
        return self.q

    @property
    def scalar(self):
        return self.q[0]

    @property
    def vector(self):
        return self.q[1:4]


def _hamilton(a, b):
    """Hamilton product written out component by component (independent of the matrix form the stand-in uses)."""
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw])


def selfcheck_quaternion(n: int = 2000, seed: int = 7, cls=None) -> tuple[float, float, float, float]:
    """Cross-check a Quaternion class (default: the stand-in) against independent implementations, over the members the hot
    path calls.

    Returns (max |R(q)e3 - scipy|, max |q1*q2 - scipy| up to sign, max |Quaternion(matrix=R) - scipy from_matrix| up to sign,
    max |q.derivative(w) - 0.5 q (x) (0, w)| against the component-wise Hamilton product)."""
    from scipy.spatial.transform import Rotation

    Q = cls or Quaternion
    rng = np.random.RandomState(seed)
    e_rot = e_mul = e_mat = e_der = 0.0
    for _ in range(n):
        q = rng.uniform(-1, 1, 4)
        p = rng.uniform(-1, 1, 4)
        b3 = Q(q).rotation_matrix.dot(np.array([0.0, 0.0, 1.0]))
        ref = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()[:, 2]
        e_rot = max(e_rot, float(np.abs(b3 - ref).max()))
        qu, pu = q / np.linalg.norm(q), p / np.linalg.norm(p)
        prod = (Q(qu) * Q(pu)).elements
        r = (Rotation.from_quat([qu[1], qu[2], qu[3], qu[0]]) * Rotation.from_quat([pu[1], pu[2], pu[3], pu[0]])).as_quat()
        r = np.array([r[3], r[0], r[1], r[2]])
        e_mul = max(e_mul, float(min(np.abs(prod - r).max(), np.abs(prod + r).max())))
        # Quaternion(matrix=) - the controller's acc2quat (quadrotor3d.py:139) - on a rotation matrix from an independent source
        R = Rotation.from_quat([pu[1], pu[2], pu[3], pu[0]]).as_matrix()
        qm = np.asarray(Q(matrix=R).elements, dtype=float)
        r = Rotation.from_matrix(R).as_quat()
        r = np.array([r[3], r[0], r[1], r[2]])
        e_mat = max(e_mat, float(min(np.abs(qm - r).max(), np.abs(qm + r).max())))
        # derivative (quadrotor3d.py:101) on the NON-unit q the env hands over after rotation_matrix normalised a copy
        w = rng.uniform(-10, 10, 3)
        d = np.asarray(Q(q).derivative(w).elements, dtype=float)
        e_der = max(e_der, float(np.abs(d - 0.5 * _hamilton(q, np.array([0.0, w[0], w[1], w[2]]))).max()))
    return e_rot, e_mul, e_mat, e_der


def real_pyquaternion():
    """The real ``pyquaternion`` package when it is importable (it is not in this image: requirements.txt:1 is un-vendored),
    else None.  Never returns the stand-in module that :func:`_install_stubs` plants in ``sys.modules``."""
    mod = sys.modules.get("pyquaternion")
    if mod is not None:
        return None if getattr(mod, "_rmav_stub", False) else mod
    try:
        import importlib

        mod = importlib.import_module("pyquaternion")
    except Exception:
        return None
    return mod


# which Quaternion the loaded reference classes run on: "stand-in" or "pyquaternion <version>" (set by _install_stubs)
QUATERNION_SOURCE = "stand-in"


def compare_with_real(n: int = 2000, seed: int = 11) -> dict | None:
    """Stand-in vs the REAL package on every member of row a10, when the package is importable (None otherwise):
    max absolute difference per member.  tests/test_oracle_golden.py asserts <= 1e-15-ish on each."""
    mod = real_pyquaternion()
    if mod is None:
        return None
    from scipy.spatial.transform import Rotation

    RQ = mod.Quaternion
    rng = np.random.RandomState(seed)
    out = {k: 0.0 for k in ("ctor", "rotation_matrix", "normalise_side_effect", "mul", "rmul", "derivative", "conjugate", "matrix")}

    def upd(k, a, b):
        out[k] = max(out[k], float(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)).max()))

    for i in range(n):
        q = rng.uniform(-1, 1, 4) * (1.0 if i % 2 else 3.0)
        p = rng.uniform(-1, 1, 4)
        w = rng.uniform(-10, 10, 3)
        a, b = Quaternion(q), RQ(q)
        upd("ctor", a.elements, b.elements)
        upd("rotation_matrix", a.rotation_matrix, b.rotation_matrix)
        upd("normalise_side_effect", a.elements, b.elements)          # rotation_matrix normalised self.q in place
        upd("mul", (Quaternion(q) * Quaternion(p)).elements, (RQ(q) * RQ(p)).elements)
        upd("rmul", (0.5 * Quaternion(q)).elements, (0.5 * RQ(q)).elements)
        upd("derivative", Quaternion(q).derivative(w).elements, RQ(q).derivative(w).elements)
        upd("conjugate", Quaternion(q).conjugate.elements, RQ(q).conjugate.elements)
        R = Rotation.from_quat(p / np.linalg.norm(p)).as_matrix()
        upd("matrix", Quaternion(matrix=R).elements, RQ(matrix=R).elements)
    return out


# ------------------------------------------------------------------------------------------------
# gym stand-in
# ------------------------------------------------------------------------------------------------
class _Box:
    def __init__(self, low=None, high=None, shape=None, dtype=None):
        self.low, self.high, self.shape, self.dtype = low, high, shape, dtype


class _Env:
    metadata = {}


WARNINGS: list[str] = []


def _install_stubs() -> None:
    if "gym" in sys.modules and getattr(sys.modules["gym"], "_rmav_stub", False):
        return
    gym = types.ModuleType("gym")
    gym._rmav_stub = True
    gym.Env = _Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = _Box
    error = types.ModuleType("gym.error")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    logger = types.ModuleType("gym.logger")
    logger.warn = lambda msg, *a: WARNINGS.append(str(msg))
    utils.seeding = seeding
    gym.spaces, gym.error, gym.utils, gym.logger = spaces, error, utils, logger
    # the REAL pyquaternion when the environment has it (RMAV_QUATERNION_STANDIN=1 forces the stand-in); the stand-in otherwise
    global QUATERNION_SOURCE
    pq = None if os.environ.get("RMAV_QUATERNION_STANDIN") == "1" else real_pyquaternion()
    if pq is not None:
        QUATERNION_SOURCE = "pyquaternion " + str(getattr(pq, "__version__", "?"))
    else:
        pq = types.ModuleType("pyquaternion")
        pq._rmav_stub = True
        pq.Quaternion = Quaternion
        QUATERNION_SOURCE = "stand-in"
    sys.modules.update(
        {
            "gym": gym,
            "gym.spaces": spaces,
            "gym.error": error,
            "gym.utils": utils,
            "gym.utils.seeding": seeding,
            "gym.logger": logger,
            "pyquaternion": pq,
        }
    )


class _LegacyNumpy(types.ModuleType):
    """Real NumPy plus ``np.float`` and the pre-1.24 handling of size-1 arrays inside ``np.array``."""

    def __init__(self):
        super().__init__("numpy_legacy_proxy")
        self.float = float

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(x, *a, **k):
        try:
            return np.array(x, *a, **k)
        except ValueError:
            return np.array([float(np.asarray(e).reshape(())) for e in x], *a, **k)


_CACHE: dict = {}


def load_class(kind: str, reading_2d: str = "B"):
    """Return the reference env class for ``kind`` (exec'd from /root/reference, never copied)."""
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    key = (kind, reading_2d if kind == "quad2d" else "-")
    if key in _CACHE:
        return _CACHE[key]
    _install_stubs()
    fname, cname = _FILES[kind]
    path = os.path.join(NATIVE_DIR, fname)
    with open(path, "r") as f:
        src = f.read()
    if kind == "quad2d":
        lines = src.split("\n")
        assert lines[95].rstrip().endswith("> 10.0"), "unexpected quadrotor2d.py layout"
        if reading_2d == "B":
            lines[95] = lines[95].rstrip() + " \\"
        elif reading_2d == "A":
            del lines[96:98]
        else:
            raise ValueError("reading_2d must be 'A' or 'B'")
        src = "\n".join(lines)
    mod = types.ModuleType("rmav_ref_" + kind + "_" + reading_2d)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        exec(compile(src, path, "exec"), mod.__dict__)
    finally:
        sys.dont_write_bytecode = old
    mod.np = _LegacyNumpy()
    mod.print = lambda *a, **k: None
    cls = getattr(mod, cname)
    _CACHE[key] = cls
    return cls


def load_reinmav_class():
    """The reference's fifth native env (reinmav_env.py): 13-state rigid body with a built-in PD controller
    and min-jerk trajectory; step() takes no action.  Needs a matplotlib stand-in (the file selects the
    TkAgg backend at import, reinmav_env.py:43-45)."""
    if not available():
        raise RuntimeError("reference tree not present (expected on the GPU box)")
    if "reinmav" in _CACHE:
        return _CACHE["reinmav"]
    _install_stubs()
    if "matplotlib" not in sys.modules or not getattr(sys.modules["matplotlib"], "_rmav_stub", False):
        mpl = types.ModuleType("matplotlib")
        mpl._rmav_stub = True
        mpl.use = lambda *a, **k: None
        plt = types.ModuleType("matplotlib.pyplot")
        mpl.pyplot = plt
        sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, plt
    path = os.path.join(NATIVE_DIR, "reinmav_env.py")
    with open(path, "r") as f:
        src = f.read()
    mod = types.ModuleType("rmav_ref_reinmav")
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        exec(compile(src, path, "exec"), mod.__dict__)
    finally:
        sys.dont_write_bytecode = old
    mod.print = lambda *a, **k: None
    _CACHE["reinmav"] = mod.ReinmavEnv
    return mod.ReinmavEnv


class RefReinmav:
    """Driver around one reference ReinmavEnv object (state: 13 floats + the env's own clock t)."""

    def __init__(self):
        warnings.simplefilter("ignore")
        self.env = load_reinmav_class()()

    def set(self, state, t):
        self.env.state = np.array(state, dtype=np.float64)
        self.env.t = float(t)

    def step(self):
        s, r, d, _ = self.env.step()
        return np.asarray(s, dtype=np.float64).ravel(), float(r), bool(d), float(self.env.t)

    def substep_derivative(self, state, t):
        return np.asarray(self.env.quad_eq_of_motion1(np.array(state, dtype=np.float64), float(t)), dtype=np.float64).ravel()

    def force_moment(self, state, t):
        qd = self.env.stateToQd(np.array(state, dtype=np.float64))
        F, M = self.env.controller(float(t), qd, self.env.trj_gen(float(t)))
        return np.array([float(np.asarray(F).ravel()[0])] + [float(m) for m in M])


def _flat(x) -> np.ndarray:
    return np.array([float(np.asarray(e).reshape(())) for e in x], dtype=np.float64)


class RefEnv:
    """Thin driver around one reference env object."""

    def __init__(self, kind: str, reading_2d: str = "B", seed: int | None = 0):
        warnings.simplefilter("ignore")
        self.kind = kind
        self.env = load_class(kind, reading_2d)()
        if seed is not None:
            self.env.seed(seed)

    # state is kept exactly as the reference keeps it (tuple / ndarray)
    def set_state(self, s, sbd="keep"):
        self.env.state = np.array(s, dtype=np.float64)
        if sbd != "keep":
            self.env.steps_beyond_done = sbd

    def get_state(self) -> np.ndarray:
        return _flat(self.env.state)

    def reset(self) -> np.ndarray:
        return _flat(self.env.reset())

    def step(self, a):
        obs, r, done, info = self.env.step(np.array(a, dtype=np.float64))
        return _flat(obs), float(np.asarray(r).reshape(())), bool(done)

    def control(self) -> np.ndarray:
        return _flat(self.env.control())

    @property
    def sbd(self):
        return self.env.steps_beyond_done
