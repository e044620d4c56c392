"""``QuadrotorVecEnv``: the baselines ``VecEnv`` contract on top of the batched HIP path.

What the reference reaches through ``baselines.common.cmd_util.make_vec_env`` in
``gym_reinmav/run.py:89`` - ``SubprocVecEnv`` / ``DummyVecEnv`` of ``Monitor``-wrapped envs - is one
object here: ``reset() -> obs[N,nS]``, ``step_async(actions[N,nA])``, ``step_wait() -> (obs, rews,
dones, infos)`` with auto-reset on done (the returned obs of a finished env is its post-reset obs,
as in ``DummyVecEnv.step_wait``) and ``info['episode'] = {'r', 'l', 't'}`` for finished envs (what
``Monitor`` adds: return, length, seconds since the env was created).  Observations stay on the GPU as torch tensors unless ``numpy_io=True``.

The per-step loop of ``gym_reinmav/run.py:190-211`` / ppo2's ``Runner`` calls ``step`` once per env-step, and
at 65 536 envs the kernel behind it runs ~4.5 us - so the wrapper must cost less than that or it, not the GPU,
sets the step rate.  The device-tensor path therefore does no per-call allocation, validation chain or ctypes
boxing: output buffers (and their ``ctypes`` pointers, and the ``bool`` views of ``done``) are prepared ahead
in blocks, an action tensor that already is float32 / contiguous / on the env's device goes straight to
``rmav_step`` by its ``data_ptr``, and ``step`` is one ABI call.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

from . import _abi as A
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
_BLOCK = 16   # steps of fresh output tensors allocated at a time (one allocation each for obs / rew / done; 64 measured the same)


class LazyInfos:
    """``infos`` of one ``step_wait()`` of a big batch: behaves like baselines' ``list[dict]`` - ``len``, indexing, iteration,
    ``info.get('episode')`` with ``{'r', 'l', 't'}`` for the envs that finished an episode in this step (what ``Monitor`` adds,
    what ppo2's ``Runner`` collects into ``epinfos``) - but builds nothing until somebody looks: the done mask and the two
    statistics arrays come off the GPU on first use, and only the finished envs (~1 % per step) get a non-empty dict.
    65 536 fresh dicts per step would cost ~3 ms of host time against a 5 us step.

    One object per step (never recycled): an object kept past the next step raises when first read - the per-env statistics it
    would report have moved on - instead of silently answering for a later step.  An env that did not finish gets a fresh
    empty dict per access (never a shared one: a caller that writes ``infos[i]['x'] = ...`` must not touch other envs or later
    steps); a caller that wants such writes to persist asks for ``dict_infos=True``."""

    __slots__ = ("_n", "_done", "_env", "_eps", "_seq", "_t")

    def __init__(self, n, done, env, seq):
        self._n, self._done, self._env, self._eps, self._seq = n, done, env, None, seq
        self._t = round(time.time() - env._tstart, 6)

    def _episodes(self):
        if self._eps is None:
            if self._env._info_seq != self._seq or self._done is None:
                raise RuntimeError("these infos were not read before the next step: the per-env episode statistics they would "
                                   "report have moved on (read infos right after step_wait(), as baselines' Runner does)")
            d = self._done
            idx = np.nonzero(d if isinstance(d, np.ndarray) else d.cpu().numpy())[0]
            eps = {}
            if len(idx):
                buf = self._env.env.episode_buffers()
                for i in idx:
                    eps[int(i)] = {"episode": {"r": float(buf["last_return"][i]), "l": int(buf["last_length"][i]), "t": self._t}}
            self._eps = eps
        return self._eps

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        e = self._episodes().get(i)
        return {} if e is None else e

    def __iter__(self):
        eps = self._episodes()
        return (eps.get(i) or {} for i in range(self._n))

    def finished(self):
        """{env index: {'episode': {'r', 'l'}}} of the envs that finished in this step (the non-empty infos)."""
        return self._episodes()


class QuadrotorVecEnv:
    def __init__(self, env_id: str, num_envs: int, device: int = 0, seed: int = 0, env_id_base: int = 0,
                 numpy_io: bool = False, dict_infos=None, reading_2d=None, reuse_buffers: bool = False):
        """``dict_infos``: True = a real ``list[dict]`` per step (default up to 4 096 envs), False = a :class:`LazyInfos` (default
        beyond): the same contract - ``len(infos) == num_envs``, ``infos[i].get('episode')`` - materialised on first use."""
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
        self._info_seq = 0   # step counter of the lazily materialised infos
        self._tstart = time.time()   # Monitor's tstart: info['episode']['t'] = seconds since the env was made
        # reuse_buffers=True: step_wait() hands out the env's own output buffers (two sets, alternating), valid
        # until the step after next.  The default returns tensors no later step overwrites, like baselines'
        # DummyVecEnv (whose Runner keeps the returned reward arrays): slices of blocks of _BLOCK steps.
        self.reuse_buffers = bool(reuse_buffers) and not self.numpy_io
        self._step_fn = self.env._lib.rmav_step
        self._hnd = self.env._h
        self._act_shape = (self.num_envs, self.env.nA)
        self._dev = None if self.numpy_io else torch.device("cuda", self.env.device)
        self._slots, self._cursor = [], 0
        if self.numpy_io:
            self._host = (np.empty((self.num_envs, self.env.nS), np.float32), np.empty(self.num_envs, np.float32),
                          np.empty(self.num_envs, np.uint8))
        elif self.reuse_buffers:
            self._slots = self._make_slots(2)

    # ---- output buffers of the device path -------------------------------------------------------------
    def _make_slots(self, k: int):
        """k steps' worth of output tensors from three allocations -> [(obs, rew, done_bool, p_obs, p_rew, p_done)]."""
        n, nS = self.num_envs, self.env.nS
        obs = torch.empty((k, n, nS), dtype=torch.float32, device=self._dev)
        rew = torch.empty((k, n), dtype=torch.float32, device=self._dev)
        done = torch.empty((k, n), dtype=torch.uint8, device=self._dev)
        ts = self.env._tstream
        if ts is not None and torch.cuda.current_stream(self._dev) != ts:
            # the kernels that fill these run on the env's stream: tell the caching allocator, or memory freed on the
            # allocating stream could be handed out again while a step launch is still writing it
            for t in (obs, rew, done):
                t.record_stream(ts)
        p0, p1, p2 = obs.data_ptr(), rew.data_ptr(), done.data_ptr()
        # the kernels store exactly 0 or 1 into `done` (csrc/rmav_kernels.hpp), so the bool view needs no conversion launch
        return [(o, r, d, C.c_void_p(p0 + i * n * nS * 4), C.c_void_p(p1 + i * n * 4), C.c_void_p(p2 + i * n))
                for i, (o, r, d) in enumerate(zip(obs.unbind(0), rew.unbind(0), done.view(torch.bool).unbind(0)))]

    def _next_slot(self):
        if self.reuse_buffers:
            self._cursor ^= 1
            return self._slots[self._cursor]
        if self._cursor == len(self._slots):
            self._slots, self._cursor = self._make_slots(_BLOCK), 0
        s = self._slots[self._cursor]
        self._cursor += 1
        return s

    # ---- VecEnv ---------------------------------------------------------------------------------------
    def reset(self):
        return self.env.reset(layout="aos", device_out=not self.numpy_io)

    def step_async(self, actions):
        if self.numpy_io:
            self._pending = self.env.step(np.asarray(actions, dtype=np.float32), layout="aos", out=self._host)
            return
        if self._hnd is None:
            raise A.RmavError(A.ERR_INVALID, "step on a closed QuadrotorVecEnv")
        if not (type(actions) is torch.Tensor and actions.dtype is torch.float32 and actions.device == self._dev
                and actions.shape == self._act_shape and actions.is_contiguous()):
            actions, _ = self.env._in(actions if torch.is_tensor(actions) else torch.as_tensor(
                np.asarray(actions, dtype=np.float32), device=self._dev), self._act_shape)   # convert / validate / raise
        slot = self._next_slot()
        # enqueued on the env's stream; the outputs are filled asynchronously
        rc = self._step_fn(self._hnd, C.c_void_p(actions.data_ptr()), slot[3], slot[4], slot[5], A.DEVICE, A.AOS)
        if rc < 0:
            A.check(rc)
        self._pending = slot

    def step_wait(self):
        slot = self._pending
        assert slot is not None, "step_async() must precede step_wait()"
        self._pending = None
        if self.numpy_io:
            obs, rew, done = slot
            done_b = done.astype(bool)
            return obs.copy(), rew.copy(), done_b, self._infos(done_b)
        return slot[0], slot[1], slot[2], self._infos(slot[2])

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _infos(self, done_b):
        if not self.dict_infos:
            # a fresh object per step (~0.3 us): one that is read after a later step sees that its sequence number no longer
            # matches and raises (recycled objects, round 4, passed that check two steps later and answered for the wrong step)
            self._info_seq = seq = self._info_seq + 1
            return LazyInfos(self.num_envs, done_b, self, seq)
        infos = [{} for _ in range(self.num_envs)]
        idx = np.nonzero(done_b if self.numpy_io else done_b.cpu().numpy())[0]
        if len(idx):
            buf = self.env.episode_buffers()
            t = round(time.time() - self._tstart, 6)
            for i in idx:
                infos[int(i)]["episode"] = {"r": float(buf["last_return"][i]), "l": int(buf["last_length"][i]), "t": t}
        return infos

    def close(self):
        self._hnd = None
        self.env.close()
