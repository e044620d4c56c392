"""Gym registry glue: the reference registers its envs in ``gym_reinmav/__init__.py:3-26``; the same
five native ids are registered here (under gym or gymnasium when importable) and are always
available through :func:`make`."""
from __future__ import annotations

ENTRY_POINTS = {
    "reinmav-v0": "gym_reinmav_amd.envs.native:ReinmavEnv",
    "quadrotor2d-v0": "gym_reinmav_amd.envs.native:Quadrotor2D",
    "quadrotor2d-slungload-v0": "gym_reinmav_amd.envs.native:Quadrotor2DSlungload",
    "quadrotor3d-v0": "gym_reinmav_amd.envs.native:Quadrotor3D",
    "quadrotor3d-slungload-v0": "gym_reinmav_amd.envs.native:Quadrotor3DSlungload",
}


def register_envs() -> bool:
    """Register the ids with gym / gymnasium.  Returns False when neither is installed."""
    for modname in ("gym", "gymnasium"):
        try:   # (neither is installed in the build image: tests/test_host_logic.py runs this against a stand-in module)
            reg = __import__(modname + ".envs.registration", fromlist=["register"])
        except Exception:
            continue
        for env_id, ep in ENTRY_POINTS.items():
            try:
                reg.register(id=env_id, entry_point=ep)
            except Exception:   # already registered (gym raises on duplicates): keep going with the other ids
                pass
        return True
    return False


def make(env_id: str, **kwargs):
    """``gym.make`` equivalent that does not need gym: ``make('quadrotor3d-v0')``."""
    import importlib

    if env_id not in ENTRY_POINTS:
        raise KeyError(f"unknown env id {env_id!r}; known: {sorted(ENTRY_POINTS)}")
    mod, cls = ENTRY_POINTS[env_id].split(":")
    return getattr(importlib.import_module(mod), cls)(**kwargs)
