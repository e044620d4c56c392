"""``QuadrotorVecEnv``: the baselines ``VecEnv`` contract on top of the batched HIP path.

What the reference reaches through ``baselines.common.cmd_util.make_vec_env`` in
``gym_reinmav/run.py:89`` - ``SubprocVecEnv`` / ``DummyVecEnv`` of ``Monitor``-wrapped envs - is one
object here: ``reset() -> obs[N,nS]``, ``step_async(actions[N,nA])``, ``step_wait() -> (obs, rews,
dones, infos)`` with auto-reset on done (the returned obs of a finished env is its post-reset obs,
as in ``DummyVecEnv.step_wait``) and ``info['episode'] = {'r', 'l'}`` for finished envs (what
``Monitor`` adds).  Observations stay on the GPU as torch tensors unless ``numpy_io=True``.
"""
from __future__ import annotations

import numpy as np

from .core import BatchedQuadrotor, torch
from .spaces import Box

ENV_IDS = {
    "reinmav-v0": "reinmav",
    "quadrotor2d-v0": "quad2d",
    "quadrotor2d-slungload-v0": "quad2d_sl",
    "quadrotor3d-v0": "quad3d",
    "quadrotor3d-slungload-v0": "quad3d_sl",
}
_ACTION_BOX = {"reinmav": (0.0, 3.5316), "quad2d": (-10.0, 10.0), "quad2d_sl": (-10.0, 10.0), "quad3d": (0.0, 10.0), "quad3d_sl": (-10.0, 10.0)}


class QuadrotorVecEnv:
    def __init__(self, env_id: str, num_envs: int, device: int = 0, seed: int = 0, env_id_base: int = 0,
                 numpy_io: bool = False, dict_infos=None, reading_2d=None, reuse_buffers: bool = False):
        kind = ENV_IDS.get(env_id, env_id)
        self.env = BatchedQuadrotor(kind, num_envs, device=device, seed=seed, env_id_base=env_id_base,
                                    auto_reset=True, track_episodes=True, reading_2d=reading_2d)
        self.num_envs = int(num_envs)
        self.numpy_io = bool(numpy_io)
        self.dict_infos = (num_envs <= 4096) if dict_infos is None else bool(dict_infos)
        lo, hi = _ACTION_BOX[self.env.kind_name]
        self.action_space = Box(low=lo, high=hi, shape=(self.env.nA,), dtype=np.float32)
        self.observation_space = Box(low=-10.0, high=10.0, shape=(self.env.nS,), dtype=np.float32)
        self._pending = None
        dev = not self.numpy_io
        # reuse_buffers=True: step_wait() hands out the env's own output buffers (two sets, alternating), valid
        # until the step after next - no per-step allocation or copy (device tensors only).  The default returns
        # fresh arrays every step like baselines' DummyVecEnv (whose Runner keeps the returned reward arrays).
        self.reuse_buffers = bool(reuse_buffers) and dev
        self._sets = [(self.env._new((self.num_envs, self.env.nS), np.float32, dev),
                       self.env._new((self.num_envs,), np.float32, dev),
                       self.env._new((self.num_envs,), np.uint8, dev)) for _ in range(2 if self.reuse_buffers else 1)]
        self._flip = 0

    def reset(self):
        return self.env.reset(layout="aos", device_out=not self.numpy_io)

    def step_async(self, actions):
        if self.numpy_io:
            actions = np.asarray(actions, dtype=np.float32)
        # enqueue on the env's stream; device outputs are filled asynchronously
        if self.reuse_buffers or self.numpy_io:
            out = self._sets[self._flip]
            if self.reuse_buffers:
                self._flip ^= 1
        else:   # fresh tensors every step (DummyVecEnv semantics): the kernel writes straight into them, no copies
            out = (self.env._new((self.num_envs, self.env.nS), np.float32, True),
                   self.env._new((self.num_envs,), np.float32, True), self.env._new((self.num_envs,), np.uint8, True))
        self._pending = self.env.step(actions, layout="aos", out=out)

    def step_wait(self):
        assert self._pending is not None, "step_async() must precede step_wait()"
        obs, rew, done = self._pending
        self._pending = None
        if self.numpy_io:
            done_b = done.astype(bool)
            obs, rew = obs.copy(), rew.copy()
        else:
            done_b = done.view(torch.bool)      # the kernel writes 0 / 1: a bool view, no conversion launch
        return obs, rew, done_b, self._infos(done_b)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _infos(self, done_b):
        if not self.dict_infos:
            return ()
        infos = [{} for _ in range(self.num_envs)]
        idx = np.nonzero(done_b if self.numpy_io else done_b.cpu().numpy())[0]
        if len(idx):
            buf = self.env.episode_buffers()
            for i in idx:
                infos[int(i)]["episode"] = {"r": float(buf["last_return"][i]), "l": int(buf["last_length"][i])}
        return infos

    def close(self):
        self.env.close()
