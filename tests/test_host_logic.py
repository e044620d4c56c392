"""Host-side logic that needs neither a GPU nor the oracle."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    from gym_reinmav_amd.distributed import shard_range

    for n, w in [(1048576, 8), (65536, 1), (10, 4), (7, 8), (0, 3)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert sum(c for _, c in spans) == n
        pos = 0
        for s, c in spans:
            assert s == pos
            pos += c
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_range(1048576, 3, 8) == (393216, 131072)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def test_registry_ids_match_reference():
    import gym_reinmav_amd as g

    # the five native ids of gym_reinmav/__init__.py:3-26 (the three MuJoCo ids are out of scope)
    assert sorted(g.ENTRY_POINTS) == ["quadrotor2d-slungload-v0", "quadrotor2d-v0", "quadrotor3d-slungload-v0",
                                      "quadrotor3d-v0", "reinmav-v0"]
    assert g.ENV_IDS["quadrotor3d-v0"] == "quad3d"
    with pytest.raises(KeyError):
        g.make("MujocoQuadForce-v0")  # out of scope, not silently mapped to something else


def test_register_envs_with_a_stub_gym(monkeypatch):
    """register_envs() against a stand-in `gym` (the image has neither gym nor gymnasium): the five ids of the reference's
    gym_reinmav/__init__.py:3-26 are registered, each with an entry point that resolves to the class of that name."""
    import importlib
    import sys
    import types

    import gym_reinmav_amd as g
    from gym_reinmav_amd import registration

    calls = []
    gym = types.ModuleType("gym")
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda id, entry_point, **kw: calls.append((id, entry_point))   # noqa: A002 - gym's own keyword
    gym.envs, envs.registration = envs, reg
    for name, mod in (("gym", gym), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    assert registration.register_envs() is True
    assert dict(calls) == g.ENTRY_POINTS and len(calls) == 5
    want = {"reinmav-v0": "ReinmavEnv", "quadrotor2d-v0": "Quadrotor2D", "quadrotor2d-slungload-v0": "Quadrotor2DSlungload",
            "quadrotor3d-v0": "Quadrotor3D", "quadrotor3d-slungload-v0": "Quadrotor3DSlungload"}
    for env_id, ep in calls:
        mod, cls = ep.split(":")
        assert cls == want[env_id]                                   # the reference's class names
        assert mod.endswith("envs.native")                           # ... in the package path the reference uses (gym_reinmav.envs.native)
        klass = getattr(importlib.import_module(mod), cls)
        assert all(hasattr(klass, m) for m in ("step", "reset", "seed", "close", "render"))
    # a registry that already holds an id (gym raises on duplicates) must not break the others
    def picky(id, entry_point, **kw):   # noqa: A002
        if id == "quadrotor3d-v0":
            raise RuntimeError("Cannot re-register id")
        calls.append((id, entry_point))
    calls.clear()
    reg.register = picky
    assert registration.register_envs() is True and len(calls) == 4


def test_lazy_infos_list_contract():
    """LazyInfos without a GPU: len / index / negative index / slice / iteration / .get('episode'), one fetch of the statistics."""
    from gym_reinmav_amd.vec_env import LazyInfos

    fetched = []

    class FakeBatch:
        def episode_buffers(self):
            fetched.append(1)
            return {"last_return": np.arange(8, dtype=np.float32) * -1.5, "last_length": np.arange(8, dtype=np.int32) + 10}

    class FakeVec:
        env, _info_seq, _tstart = FakeBatch(), 3, 0.0

    done = np.array([0, 1, 0, 0, 0, 0, 1, 0], bool)
    infos = LazyInfos(8, done, FakeVec(), 3)
    assert len(infos) == 8 and not fetched
    assert set(infos[1]["episode"]) == {"r", "l", "t"} and infos[1]["episode"]["t"] > 0          # Monitor's keys
    assert {k: infos[1]["episode"][k] for k in "rl"} == {"r": -1.5, "l": 11} and infos[-2]["episode"]["l"] == 16 and infos[0] == {}
    infos[0]["x"] = 1                       # a write into one env's empty info touches no other env and no later read
    assert infos[0] == {} and infos[2] == {} and all("x" not in i for i in infos)
    assert [bool(i.get("episode")) for i in infos] == list(done) and len(infos[2:5]) == 3 and len(fetched) == 1
    assert sorted(infos.finished()) == [1, 6]
    with pytest.raises(IndexError):
        infos[8]
    stale = LazyInfos(8, done, FakeVec(), 2)
    with pytest.raises(RuntimeError):
        stale[0]


def test_box_space():
    from gym_reinmav_amd.spaces import Box

    b = Box(low=0.0, high=10.0, shape=(4,), dtype=np.float32)
    assert b.shape == (4,) and b.contains(b.sample()) and not b.contains(np.full(4, 11.0, np.float32))


def test_trajectory_schema_roundtrip(tmp_path, golden):
    """The .npz trajectory schema: write a (synthetic) SoA rollout, read it back in the shared [T, N, dim] form;
    the golden closed-loop fixtures map onto the same schema."""
    from gym_reinmav_amd.trajectory import from_golden, load_rollout, save_rollout

    rng = np.random.RandomState(0)
    T, N = 7, 5
    ro = {"actions": rng.normal(size=(T, 4, N)).astype(np.float32), "obs": rng.normal(size=(T, 10, N)).astype(np.float32),
          "rew": rng.normal(size=(T, N)).astype(np.float32), "done": (rng.uniform(size=(T, N)) < 0.2).astype(np.uint8)}
    s0 = rng.normal(size=(10, N)).astype(np.float32)
    p = tmp_path / "ro.npz"
    save_rollout(p, "quad3d", s0, ro, layout="soa", meta={"seed": 3})
    z = load_rollout(p)
    assert z["kind"] == "quad3d" and z["meta"]["seed"] == 3 and z["meta"]["schema"] == 1
    assert z["state"].shape == (T, N, 10) and z["action"].shape == (T, N, 4) and z["done"].dtype == bool
    assert np.array_equal(z["state"][0], s0.T) and np.array_equal(z["state"][1:], z["next_obs"][:-1])
    assert np.array_equal(z["next_obs"][2], ro["obs"][2].T)
    g = from_golden(golden["quad3d"])
    assert g["state"].shape == (400, 4, 10) and g["done"].shape == (400, 4)


def test_bench_byte_accounting():
    """bench.py's roofline bytes for the fused rollout: trajectory out per env-step + state / bookkeeping per launch."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # quadrotor3d: 16 B actions + 40 B obs + 4 B reward + 1 B done = 61 B per env-step; 80 B state in/out + 16 B episode
    # accumulators in/out + 8 B steps_beyond_done / reset counter in = 104 B per launch
    assert b.fused_bytes_per_launch(65536, 64, 10, 4) == 65536 * (64 * 61 + 104)
    assert b.fused_bytes_per_launch(1, 1, 16, 4) == 4 * 21 + 1 + 8 * 16 + 24
    assert b.HBM_PEAK_GBS == 8000.0


def test_bench_device_sampler_summary():
    """bench.py's device_state: only samples taken while the GPU was busy count; unreadable sysfs gives None fields."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    s = b.DeviceSampler(0)                      # no GPU here: nothing to read, nothing started
    with s:
        pass
    assert s.summary()["power_w_mean"] is None and s.summary()["samples"] == 0
    s.samples = [(250.0, 100.0), (1300.0, 2300.0), (1400.0, 2200.0), (260.0, 90.0)]
    s.cap = 1400.0
    out = s.summary()
    assert out["samples"] == 2 and out["power_w_mean"] == 1350.0 and out["power_w_max"] == 1400.0
    assert out["sclk_mhz_mean"] == 2250 and out["power_cap_w"] == 1400.0


def test_public_attributes_write_through_mechanics():
    """The gym-shaped env's vector-valued attributes (ref_pos, ref_vel, g) without a GPU: a fake handle in place of BatchedQuadrotor.
    Whole-array assignment, element writes on the handed-out array (applied to the CURRENT value, also from a stale snapshot),
    arithmetic yields plain ndarrays, g is a free vector, every write drops the cached control() action."""
    from gym_reinmav_amd.envs.native import base as B

    class P:
        def __init__(self):
            self.ref_pos, self.ref_vel, self.g_vec, self.mass, self.load_mass = [0.0, 0.0, 2.0], [0.0, 0.0, 0.0], [0.0, 0.0, -9.8], 1.0, 0.1

    class FakeBatch:
        def __init__(self):
            self._p, self.writes = P(), 0

        @property
        def params(self):
            q = P()
            q.__dict__ = {k: (list(v) if isinstance(v, list) else v) for k, v in self._p.__dict__.items()}
            return q

        @params.setter
        def params(self, p):
            self._p, self.writes = p, self.writes + 1

    e = B.NativeQuadrotorEnv.__new__(B.NativeQuadrotorEnv)
    e._batch, e._dim, e._has_load, e._ctrl_valid = FakeBatch(), 3, False, True
    a = e.ref_pos
    a[2] = 1.0
    assert list(e.ref_pos) == [0.0, 0.0, 1.0] and list(a) == [0.0, 0.0, 1.0] and e._ctrl_valid is False and e._batch.writes == 1
    e.ref_pos = (1, 0.5, 2.5)
    a[0] = 1.25                                  # stale snapshot: only element 0 of the current value changes
    assert list(e.ref_pos) == [1.25, 0.5, 2.5] and list(a) == [1.25, 0.5, 2.5]
    assert type(e.ref_pos * 2) is np.ndarray and type(e.ref_pos + e.ref_vel) is np.ndarray
    e.mass = 1.3
    assert e.mass == 1.3 and list(e.g) == [0.0, 0.0, -9.8]
    e.g[2] = -3.7
    assert list(e.g) == [0.0, 0.0, -3.7]
    e.g[0] = 1.0                                 # any direction, like the reference's self.g
    with pytest.raises(ValueError):
        e.ref_vel = (1.0, 2.0)
    with pytest.raises(ValueError):
        e.g = (0.0, -9.8)
    assert list(e.g) == [1.0, 0.0, -3.7]
    with pytest.raises(AttributeError):
        e.load_mass                              # noqa: B018 - not a slung-load class
    e._has_load = True
    e.load_mass = 0.25
    assert e.load_mass == 0.25


def test_bench_line_stays_compact_whatever_the_legs():
    """The contract that broke in round 4 (a 22 KB line the driver could not recover), without a GPU: the committed FULL record of a
    `bench.py --secondary all` run (profiles/r05/bench_n1_detail.json: every leg, six actors, descriptions) goes through bench.py's own
    row / line builders - the result is one JSON object below 4 KB that keeps the contract keys; and a record far too big for that
    loses its optional rows, never its headline."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(ROOT, "profiles", "r05", "bench_n1_detail.json")))
    assert len(json.dumps(full)) > 15000                                   # the full record is the size that broke the driver's parser
    legs = bench.leg_rows(full["other_modes"])
    for k in ("step", "step_262144", "step_1048576", "c3_shard", "c4", "sustained", "policy_rollout", "gym1", "vecenv", "reinmav"):
        assert k in legs, k
    assert 0.0 < legs["step_1048576"]["frac"] <= 1.0 and legs["policy_rollout"]["f16_mfma"]["value"] > 0
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                 "dtype", "data", "prewarm_ms", "prewarm_launches")}
    line["config"] = {k: (v if not isinstance(v, str) else v[:200]) for k, v in full["config"].items()}
    line["roofline"] = {k: full["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_needed",
                                                         "bytes_per_launch", "launch_ms_hip_events")}
    line["cpu_baseline"] = dict(full["cpu_baseline"], sample=full["cpu_baseline"]["sample"][:160])
    line["legs"] = legs
    text = bench.compact_text(line)
    assert len(text) < 4096 and "\n" not in text
    back = json.loads(text)
    assert back["roofline"]["frac"] == full["roofline"]["frac"] and back["cpu_baseline"]["value"] > 0 and "legs" in back
    huge = dict(line, legs={f"leg{i}": {"value": 1.0, "note": "x" * 200} for i in range(40)})
    t2 = bench.compact_text(huge)
    assert len(t2) < 4096 and "legs" not in json.loads(t2) and json.loads(t2)["value"] == full["value"]
