"""ctypes front-end to the CPU oracle (``oracle/rmav_oracle.c``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package (``reinmav-gym_amd/``) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librmav_oracle.so")

KINDS = {"quad2d": 0, "quad2d_sl": 1, "quad3d": 2, "quad3d_sl": 3}
STATE_DIM = {"quad2d": 5, "quad2d_sl": 9, "quad3d": 10, "quad3d_sl": 16}
ACTION_DIM = {"quad2d": 2, "quad2d_sl": 2, "quad3d": 4, "quad3d_sl": 4}


class Params(C.Structure):
    _fields_ = [
        ("mass", C.c_double),
        ("load_mass", C.c_double),
        ("dt", C.c_double),
        ("g", C.c_double),
        ("tether_length", C.c_double),
        ("pos_limit", C.c_double),
        ("vel_limit", C.c_double),
        ("thrust_scale", C.c_double),
        ("clamp_thrust", C.c_int),
        ("ref_pos", C.c_double * 3),
        ("ref_vel", C.c_double * 3),
        ("kp", C.c_double),
        ("kv", C.c_double),
        ("tau", C.c_double),
        ("g_vec", C.c_double * 3),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rmav_oracle.c")
    hdr = os.path.join(_HERE, "rmav_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, hdr)
    )
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        L.oracle_default_params.argtypes = [C.c_int, C.c_int, C.POINTER(Params)]
        L.oracle_step.argtypes = [C.c_int, C.POINTER(Params), dp, dp, dp, dp, ip, ip]
        L.oracle_step_branch.argtypes = [C.c_int, C.POINTER(Params), dp, dp, dp, dp, ip, ip, C.c_int]
        L.oracle_control.argtypes = [C.c_int, C.POINTER(Params), dp, dp]
        L.oracle_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.oracle_philox4x32_10.restype = None
        L.oracle_u01.argtypes = [C.c_uint32]
        L.oracle_u01.restype = C.c_float
        L.oracle_reset_state.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, fp]
        L.oracle_reset_state.restype = None
        L.oracle_random_action.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float, C.c_float, fp]
        L.oracle_random_action.restype = None
        L.oracle_batch_step.argtypes = [C.c_int, C.POINTER(Params), C.c_int64, dp, dp, dp,
                                        C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_int]
        L.oracle_rollout_random.argtypes = [C.c_int, C.POINTER(Params), C.c_int64, C.c_int64, C.c_uint64,
                                            C.c_uint64, C.c_float, C.c_float, fp, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_uint32), C.c_uint64, dp, C.POINTER(C.c_int64)]
        L.oracle_rollout_random.restype = C.c_int64
        _lib = L
    return _lib


def default_params(kind: str, reading_2d: str = "B") -> Params:
    p = Params()
    rc = lib().oracle_default_params(KINDS[kind], ord(reading_2d), C.byref(p))
    assert rc == 0
    return p


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def step(kind: str, s, a, sbd: int | None = None, params: Params | None = None, force_taut: int = -1):
    """One env, one step.  Returns (s_next f64[nS], reward, done, sbd_next).
    force_taut: -1 = like the reference, 1 / 0 = force the taut / slack tether branch."""
    p = params or default_params(kind)
    s = np.ascontiguousarray(s, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert s.shape == (STATE_DIM[kind],) and a.shape == (ACTION_DIM[kind],)
    o = np.empty_like(s)
    r = C.c_double()
    d = C.c_int()
    sb = C.c_int(-1 if sbd is None else int(sbd))
    rc = lib().oracle_step_branch(KINDS[kind], C.byref(p), _dptr(s), _dptr(a), _dptr(o), C.byref(r), C.byref(d),
                                  C.byref(sb), int(force_taut))
    assert rc == 0
    return o, r.value, bool(d.value), (None if sb.value < 0 else sb.value)


TETHER = {"quad2d_sl": (slice(0, 2), slice(5, 7)), "quad3d_sl": (slice(0, 3), slice(10, 13))}


def tether_slack(kind: str, s, params: Params | None = None) -> np.ndarray:
    """|load_pos - pos| - L per env (0-d for one env); NaN for kinds without a tether."""
    s = np.asarray(s, dtype=np.float64)
    if kind not in TETHER:
        return np.full(s.shape[:-1], np.nan)
    p = params or default_params(kind)
    ps, ls = TETHER[kind]
    return np.linalg.norm(s[..., ls] - s[..., ps], axis=-1) - p.tether_length


def control(kind: str, s, params: Params | None = None) -> np.ndarray:
    p = params or default_params(kind)
    s = np.ascontiguousarray(s, dtype=np.float64)
    a = np.empty(ACTION_DIM[kind], dtype=np.float64)
    lib().oracle_control(KINDS[kind], C.byref(p), _dptr(s), _dptr(a))
    return a


def batch_step(kind: str, s, a, sbd=None, params: Params | None = None, round_f32: bool = False):
    """s [n,nS], a [n,nA] -> (s_next [n,nS], reward [n], done bool[n], sbd int32[n])."""
    p = params or default_params(kind)
    s = np.array(s, dtype=np.float64, order="C", copy=True)
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = s.shape[0]
    assert s.shape == (n, STATE_DIM[kind]) and a.shape == (n, ACTION_DIM[kind])
    sb = np.full(n, -1, dtype=np.int32) if sbd is None else np.array(sbd, dtype=np.int32, copy=True)
    r = np.empty(n, dtype=np.float64)
    d = np.empty(n, dtype=np.uint8)
    rc = lib().oracle_batch_step(KINDS[kind], C.byref(p), n, _dptr(s), _dptr(a), _dptr(r),
                                 d.ctypes.data_as(C.POINTER(C.c_uint8)), sb.ctypes.data_as(C.POINTER(C.c_int32)),
                                 int(round_f32))
    assert rc == 0
    return s, r, d.astype(bool), sb


def batch_control(kind: str, s, params: Params | None = None) -> np.ndarray:
    p = params or default_params(kind)
    s = np.ascontiguousarray(s, dtype=np.float64)
    out = np.empty((s.shape[0], ACTION_DIM[kind]), dtype=np.float64)
    L = lib()
    for i in range(s.shape[0]):
        L.oracle_control(KINDS[kind], C.byref(p), _dptr(s[i]), _dptr(out[i]))
    return out


def philox(ctr, key) -> np.ndarray:
    c = (C.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (C.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(c, k, o)
    return np.array(list(o), dtype=np.uint32)


def reset_state(kind: str, seed: int, env_id: int, episode: int) -> np.ndarray:
    out = np.empty(STATE_DIM[kind], dtype=np.float32)
    lib().oracle_reset_state(KINDS[kind], seed, env_id, episode, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def reset_states(kind: str, seed: int, env_ids, episodes) -> np.ndarray:
    env_ids = np.asarray(env_ids)
    episodes = np.broadcast_to(np.asarray(episodes), env_ids.shape)
    return np.stack([reset_state(kind, seed, int(e), int(k)) for e, k in zip(env_ids, episodes)])


def random_action(kind: str, seed: int, env_id: int, t: int, lo: float, hi: float) -> np.ndarray:
    out = np.empty(ACTION_DIM[kind], dtype=np.float32)
    lib().oracle_random_action(KINDS[kind], seed, env_id, t, lo, hi, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def random_actions(kind: str, seed: int, env_ids, t: int, lo: float, hi: float) -> np.ndarray:
    return np.stack([random_action(kind, seed, int(e), t, lo, hi) for e in env_ids])


def rollout_random(kind: str, state: np.ndarray, sbd: np.ndarray, episode: np.ndarray, steps: int, seed: int,
                   env_id_base: int, lo: float, hi: float, t0: int = 0, params: Params | None = None):
    """CPU-baseline workload; mutates state/sbd/episode in place.  Returns (env_steps, reward_sum, n_done)."""
    p = params or default_params(kind)
    assert state.dtype == np.float32 and state.flags.c_contiguous
    assert sbd.dtype == np.int32 and episode.dtype == np.uint32
    n = state.shape[0]
    ret = C.c_double(0.0)
    nd = C.c_int64(0)
    k = lib().oracle_rollout_random(KINDS[kind], C.byref(p), n, steps, seed, env_id_base, lo, hi,
                                    state.ctypes.data_as(C.POINTER(C.c_float)),
                                    sbd.ctypes.data_as(C.POINTER(C.c_int32)),
                                    episode.ctypes.data_as(C.POINTER(C.c_uint32)), t0, C.byref(ret), C.byref(nd))
    return int(k), ret.value, int(nd.value)


# ---- ReinmavEnv (reinmav_env.py) ----------------------------------------------------------------------
class ReinmavParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("arm_length", "mass", "gravity", "min_force", "max_force")] + [
        ("inertia", C.c_double * 9), ("inv_inertia", C.c_double * 9)] + [
        (n, C.c_double) for n in ("dt", "ds", "t_max")] + [(n, C.c_double * 3) for n in ("kp", "kd", "kp_rot", "kd_rot")]


REINMAV_INIT_STATE = np.array([0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64)  # reinmav_env.py:79


def reinmav_params() -> ReinmavParams:
    p = ReinmavParams()
    lib().oracle_reinmav_default_params(C.byref(p))
    return p


def reinmav_step(s, t: float, action=None, params: ReinmavParams | None = None, rk4: bool = False):
    """One env, one step (50/51 Euler sub-steps; rk4=True: RK4 sub-steps, not in the reference).
    Returns (s_next f64[13], t_next, reward, done, n_substeps)."""
    L = lib()
    L.oracle_reinmav_step.restype = C.c_int
    L.oracle_reinmav_step_rk4.restype = C.c_int
    p = params or reinmav_params()
    s = np.array(s, dtype=np.float64, copy=True)
    tt = C.c_double(float(t))
    r, d = C.c_double(), C.c_int()
    a = None if action is None else np.ascontiguousarray(action, dtype=np.float64)
    fn = L.oracle_reinmav_step_rk4 if rk4 else L.oracle_reinmav_step
    n = fn(C.byref(p), _dptr(s), C.byref(tt), None if a is None else _dptr(a), C.byref(r), C.byref(d))
    return s, tt.value, r.value, bool(d.value), n


def reinmav_batch_step(S, T, actions=None, params: ReinmavParams | None = None, rk4: bool = False):
    """S [n,13], T [n] -> (S_next [n,13], T_next [n], n_substeps [n])."""
    p = params or reinmav_params()
    S = np.array(S, dtype=np.float64, copy=True)
    T = np.array(T, dtype=np.float64, copy=True)
    ns = np.zeros(len(S), np.int32)
    for i in range(len(S)):
        S[i], T[i], _, _, ns[i] = reinmav_step(S[i], T[i], None if actions is None else actions[i], p, rk4=rk4)
    return S, T, ns


def reinmav_controller(s, t: float, params: ReinmavParams | None = None) -> np.ndarray:
    p = params or reinmav_params()
    s = np.ascontiguousarray(s, dtype=np.float64)
    fm = np.zeros(4)
    lib().oracle_reinmav_controller(C.byref(p), _dptr(s), C.c_double(float(t)), _dptr(fm))
    return fm
