"""On-disk trajectory schema shared by the reference harness, the CPU oracle and the HIP path.

One ``.npz`` file holds one rollout of N envs over T steps, time-major, env-major inside a step
("AoS": ``[T, N, dim]``), which is also the shape of the golden closed-loop fixtures in
``tests/golden/*.npz`` (``traj_*`` keys, with the seed axis in front instead of the env axis):

    kind      str    'quad2d' | 'quad2d_sl' | 'quad3d' | 'quad3d_sl'
    state     f32/f64 [T, N, nS]   state BEFORE step t (what control()/the policy saw)
    action    f32/f64 [T, N, nA]   action applied at step t
    next_obs  f32/f64 [T, N, nS]   observation returned by step t (post auto-reset when ``done``)
    reward    f32/f64 [T, N]
    done      bool    [T, N]
    meta      json str             seed, env_id_base, dt, action source, producer, library version ...

`state[t+1] == next_obs[t]` for a VecEnv-style (auto-reset) rollout, so ``state`` is stored only for t = 0
unless ``full_state=True`` (the golden fixtures store it in full because reset-on-done there draws from the
reference's own MT19937 stream).
"""
from __future__ import annotations

import json

import numpy as np

SCHEMA_VERSION = 1
_DIMS = {"quad2d": (5, 2), "quad2d_sl": (9, 2), "quad3d": (10, 4), "quad3d_sl": (16, 4)}


def _aos(x, layout):
    x = np.asarray(x)
    return x if layout == "aos" else np.swapaxes(x, -1, -2)


def save_rollout(path, kind: str, state0, rollout: dict, layout: str = "soa", meta: dict | None = None,
                 full_state: bool = False):
    """Write a rollout dict as returned by ``BatchedQuadrotor.rollout(..., want=('actions','obs','rew','done'))``.

    ``state0``: state of every env before the first step (``get_state`` in the same ``layout``)."""
    nS, nA = _DIMS[kind]
    act, obs = _aos(rollout["actions"], layout), _aos(rollout["obs"], layout)
    rew, done = np.asarray(rollout["rew"]), np.asarray(rollout["done"]).astype(bool)
    s0 = _aos(state0, layout)
    T, N = rew.shape
    assert act.shape == (T, N, nA) and obs.shape == (T, N, nS) and s0.shape == (N, nS) and done.shape == (T, N)
    state = np.concatenate([s0[None], obs[:-1]], axis=0) if full_state else s0[None]
    m = {"schema": SCHEMA_VERSION, "kind": kind, "layout": "time-major [T, N, dim]"}
    m.update(meta or {})
    np.savez_compressed(path, kind=np.array(kind), state=state, action=act, next_obs=obs, reward=rew, done=done,
                        meta=np.array(json.dumps(m)))


def load_rollout(path) -> dict:
    z = np.load(path, allow_pickle=False)
    out = {k: z[k] for k in ("state", "action", "next_obs", "reward", "done")}
    out["kind"] = str(z["kind"])
    out["meta"] = json.loads(str(z["meta"]))
    if out["state"].shape[0] == 1 and out["next_obs"].shape[0] > 1:  # expand the implicit VecEnv states
        out["state"] = np.concatenate([out["state"], out["next_obs"][:-1]], axis=0)
    return out


def from_golden(golden: dict, sfx: str = "") -> dict:
    """View a golden fixture's closed-loop runs (``[seed, T, dim]``) in this schema (seed axis -> env axis)."""
    sw = lambda k: np.swapaxes(golden["traj_" + k + sfx], 0, 1)  # noqa: E731
    return {"state": sw("s"), "action": sw("a"), "next_obs": sw("s2"), "reward": sw("r"), "done": sw("d")}
