"""ctypes binding of ``librmav.so`` (the C ABI declared in ``include/rmav.h``).

The library is the only implementation: if it is missing or no GPU is visible, importing succeeds
(so that host-only logic stays testable) but creating an env raises - there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMAV_LIB_PATH") or os.path.join(_HERE, "librmav.so")   # override: A/B builds only

# enums of include/rmav.h
QUAD2D, QUAD2D_SL, QUAD3D, QUAD3D_SL, REINMAV = 0, 1, 2, 3, 4
KIND_NAMES = {QUAD2D: "quad2d", QUAD2D_SL: "quad2d_sl", QUAD3D: "quad3d", QUAD3D_SL: "quad3d_sl", REINMAV: "reinmav"}
KIND_BY_NAME = {v: k for k, v in KIND_NAMES.items()}
STATE_DIM = {QUAD2D: 5, QUAD2D_SL: 9, QUAD3D: 10, QUAD3D_SL: 16, REINMAV: 13}
ACTION_DIM = {QUAD2D: 2, QUAD2D_SL: 2, QUAD3D: 4, QUAD3D_SL: 4, REINMAV: 4}
PARAM_MASS, PARAM_LOAD_MASS, PARAM_TETHER_LENGTH = 0, 1, 2
HOST, DEVICE = 0, 1
SOA, AOS = 0, 1
ACT_BUFFER, ACT_RANDOM, ACT_CONTROLLER, ACT_POLICY, ACT_POLICY_BF16 = 0, 1, 2, 3, 4
POLICY_FP32, POLICY_BF16_MFMA, POLICY_FP32_MFMA, POLICY_F16_MFMA, POLICY_F16_SHARED = 0, 1, 2, 3, 4
INT_EULER, INT_RK4 = 0, 1
F_AUTO_RESET, F_TRACK_EPISODES = 1, 2
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_ALLOC, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5
TUNE = {"split": 0, "slice": 1, "store_policy": 2, "split_group": 3, "block": 4, "step_lazy": 5, "step_store": 6, "policy_pair": 7, "pair_group": 8}
COMM_ID_BYTES = 128


class RmavError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"librmav error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """``rmav_params``."""

    _fields_ = [
        ("mass", C.c_double),
        ("load_mass", C.c_double),
        ("dt", C.c_double),
        ("g", C.c_double),
        ("tether_length", C.c_double),
        ("pos_limit", C.c_double),
        ("vel_limit", C.c_double),
        ("thrust_scale", C.c_double),
        ("clamp_thrust", C.c_int32),
        ("integrator", C.c_int32),
        ("ref_pos", C.c_double * 3),
        ("ref_vel", C.c_double * 3),
        ("kp", C.c_double),
        ("kv", C.c_double),
        ("tau", C.c_double),
        ("act_lo", C.c_double),
        ("act_hi", C.c_double),
        ("g_vec", C.c_double * 3),
    ]


class EpTotals(C.Structure):
    _fields_ = [("episodes", C.c_uint64), ("return_sum", C.c_double), ("length_sum", C.c_uint64)]


# name -> (restype, argtypes); also the list the symbol-export test walks.
_vp, _fp, _u8p = C.c_void_p, C.c_void_p, C.c_void_p  # raw addresses: host arrays or device pointers
PROTOTYPES = {
    "rmav_version": (C.c_int, []),
    "rmav_last_error": (C.c_char_p, []),
    "rmav_device_count": (C.c_int, []),
    "rmav_state_dim": (C.c_int, [C.c_int]),
    "rmav_action_dim": (C.c_int, [C.c_int]),
    "rmav_algorithmic_bytes": (C.c_int, [C.c_int]),
    "rmav_default_params": (C.c_int, [C.c_int, C.c_int, C.POINTER(Params)]),
    "rmav_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_uint64,
                              C.c_uint32, C.POINTER(Params), C.c_void_p]),
    "rmav_destroy": (C.c_int, [C.c_void_p]),
    "rmav_seed": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rmav_get_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "rmav_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "rmav_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rmav_set_env_param": (C.c_int, [C.c_void_p, C.c_int, _fp, C.c_int]),
    "rmav_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "rmav_get_tuning": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "rmav_num_envs": (C.c_int64, [C.c_void_p]),
    "rmav_sync": (C.c_int, [C.c_void_p]),
    "rmav_reset": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "rmav_step": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _u8p, C.c_int, C.c_int]),
    "rmav_control": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "rmav_control_step": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _u8p, C.c_int, C.c_int]),
    "rmav_step_control": (C.c_int, [C.c_void_p, _fp, _fp, _fp, _u8p, _fp, C.c_int, C.c_int]),
    "rmav_rollout": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, _fp, _fp, _fp, _fp, _u8p, C.c_int, C.c_int,
                               C.c_int]),
    "rmav_trajectory_pitch": (C.c_int64, [C.c_void_p]),
    "rmav_rollout_pitched": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, _fp, _fp, _fp, _fp, _u8p, C.c_int64, C.c_int]),
    "rmav_chunk_envs": (C.c_int64, [C.c_void_p]),
    "rmav_rollout_chunked": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, _fp, _fp, _fp, _fp, _u8p, C.c_int64]),
    "rmav_policy_weight_count": (C.c_int64, [C.c_int]),
    "rmav_policy_weight_count_bf16": (C.c_int64, []),
    "rmav_policy_weight_count_f32_mfma": (C.c_int64, []),
    "rmav_policy_weight_count_shared": (C.c_int64, []),
    "rmav_pack_policy": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _vp, _vp, C.c_int64, _fp]),
    "rmav_pack_policy_f16": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), _vp, _vp, C.c_int64, _fp]),
    "rmav_rollout_policy": (C.c_int, [C.c_void_p, C.c_int32, _fp, _fp, _fp, _fp, _u8p, _fp, _fp, C.c_int]),
    "rmav_gae": (C.c_int, [C.c_void_p, C.c_int32, _fp, _u8p, _fp, C.c_float, C.c_float, C.c_float, _fp, _fp, _vp]),
    "rmav_normalize": (C.c_int, [C.c_void_p, _fp, C.c_int64, C.c_float, C.c_float]),
    "rmav_comm_use_library": (C.c_int, [C.c_char_p]),
    "rmav_comm_unique_id": (C.c_int, [C.c_void_p]),
    "rmav_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rmav_comm_destroy": (C.c_int, [C.c_void_p]),
    "rmav_comm_warmup": (C.c_int, [C.c_void_p, C.c_double]),
    "rmav_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rmav_allgather_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _fp, _vp]),
    "rmav_allgather_stats_post": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rmav_allgather_stats_arm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "rmav_allgather_stats_result": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, _fp, _vp]),
    "rmav_allgather_stats_wait": (C.c_int, [C.c_void_p, C.c_double]),
    "rmav_pack_stats": (C.c_int, [C.c_void_p, C.c_int64, _vp]),
    "rmav_get_state": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "rmav_set_state": (C.c_int, [C.c_void_p, _fp, C.c_int, C.c_int]),
    "rmav_get_sbd": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_set_sbd": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_get_reset_counts": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_set_reset_counts": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_get_time": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_set_time": (C.c_int, [C.c_void_p, _vp, C.c_int]),
    "rmav_get_step_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rmav_set_step_count": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rmav_episode_totals": (C.c_int, [C.c_void_p, C.POINTER(EpTotals), C.c_int]),
    "rmav_episode_buffers": (C.c_int, [C.c_void_p, _fp, _vp, _fp, _vp, C.c_int]),
}

_lib = None


def lib():
    """Load librmav.so (once).  Raises if it has not been built - never falls back to anything."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RmavError(ERR_NO_DEVICE, f"{LIB_PATH} not found: build it with `make -C reinmav-gym_amd` "
                                           "(or __graft_entry__.build()); there is no CPU fallback")
        try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for pure-ctypes use
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.rmav_version() != 101:
            raise RmavError(ERR_INVALID, "librmav.so version mismatch; rebuild")
        _lib = L
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise RmavError(rc, lib().rmav_last_error().decode("utf-8", "replace"))
    return rc


def default_params(kind: int, reading_2d: str | None = None) -> Params:
    p = Params()
    check(lib().rmav_default_params(kind, ord(reading_2d) if reading_2d else 0, C.byref(p)))
    return p
