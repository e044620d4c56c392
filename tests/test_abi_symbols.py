"""The C-ABI library loads and exports every symbol include/rmav.h declares; without a GPU it refuses
to create envs (there is no CPU path behind the ABI)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    inc = os.path.join(ROOT, "include")   # rmav.h (the core) + rmav_ppo.h, rmav_comm.h (its two extensions)
    txt = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rmav_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(built):
    from gym_reinmav_amd import _abi as A

    names = _declared()
    assert len(names) >= 25
    L = C.CDLL(A.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/rmav.h but not exported"
    assert sorted(A.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"


def test_tuning_keys_match_the_header():
    """_abi.TUNE (what env.set_tuning / bench.py --tune name) is exactly enum rmav_tuning_key of include/rmav.h - nine keys since
    round 6 (the keys of rejected variants were removed), numbered densely."""
    from gym_reinmav_amd import _abi as A

    txt = open(os.path.join(ROOT, "include", "rmav.h")).read()
    body = txt[txt.index("enum rmav_tuning_key {"):]
    body = body[:body.index("};")]
    keys = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"RMAV_TUNE_([A-Z_]+)\s*=\s*(\d+)", body)}
    count = keys.pop("count")
    assert keys == A.TUNE and sorted(keys.values()) == list(range(count)) and count == 9


def test_library_level_queries(built):
    from gym_reinmav_amd import _abi as A

    L = A.lib()
    assert L.rmav_version() == 101
    assert [L.rmav_state_dim(k) for k in range(5)] == [5, 9, 10, 16, 13]
    assert [L.rmav_action_dim(k) for k in range(5)] == [2, 2, 4, 4, 4]
    assert [L.rmav_algorithmic_bytes(k) for k in range(5)] == [53, 85, 101, 149, 125]
    assert L.rmav_state_dim(7) == -1
    p = A.Params()
    assert L.rmav_default_params(9, 0, C.byref(p)) == A.ERR_INVALID
    assert b"kind" in L.rmav_last_error()
    assert L.rmav_destroy(None) == A.ERR_INVALID


def test_default_params_match_oracle_defaults(built):
    """Same literals on both sides of the parity check."""
    import oracle as O
    from gym_reinmav_amd import _abi as A

    for kind, name in A.KIND_NAMES.items():
        if name == "reinmav":
            q = O.reinmav_params()
            p = A.default_params(kind)
            assert (p.mass, p.g, p.dt) == (q.mass, q.gravity, q.dt)
            continue
        for reading in ("A", "B"):
            p, q = A.default_params(kind, reading), O.default_params(name, reading)
            for f in ("mass", "load_mass", "dt", "g", "tether_length", "pos_limit", "vel_limit", "thrust_scale",
                      "clamp_thrust", "kp", "kv", "tau"):
                assert getattr(p, f) == getattr(q, f), (name, f)
            assert list(p.ref_pos) == list(q.ref_pos) and list(p.ref_vel) == list(q.ref_vel) and list(p.g_vec) == list(q.g_vec)
            assert list(p.g_vec) == ([0.0, -9.8, 0.0] if name.startswith("quad2d") else [0.0, 0.0, -9.8])


def test_no_cpu_fallback(built):
    from gym_reinmav_amd import BatchedQuadrotor, RmavError, _abi as A

    if A.lib().rmav_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RmavError) as e:
        BatchedQuadrotor("quad3d", 16)
    assert e.value.code == A.ERR_NO_DEVICE
