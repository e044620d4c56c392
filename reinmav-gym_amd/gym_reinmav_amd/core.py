"""``BatchedQuadrotor``: N independent envs of one kind resident on one MI355X, behind ``librmav.so``.

Thin, allocation-aware plumbing over the C ABI:

* NumPy arrays in -> ``RMAV_HOST`` calls (library stages + synchronises), NumPy arrays out.
* torch CUDA tensors in -> ``RMAV_DEVICE`` calls: pointers are handed over as-is, work is enqueued on
  the torch stream the env was created on, nothing synchronises, tensors come back.

Layouts: ``"aos"`` = ``[N, dim]`` (what gym / a policy hands over), ``"soa"`` = ``[dim, N]`` (native,
coalesced).  Trajectories are time-major: ``[T, N, dim]`` / ``[T, dim, N]``.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _abi as A

try:  # torch is plumbing (device memory / streams); the host-array API works without it
    import torch
except Exception:  # pragma: no cover
    torch = None

_MODES = {"buffer": A.ACT_BUFFER, "random": A.ACT_RANDOM, "controller": A.ACT_CONTROLLER}


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _layout(layout: str) -> int:
    if layout not in ("aos", "soa"):
        raise ValueError("layout must be 'aos' or 'soa'")
    return A.AOS if layout == "aos" else A.SOA


class BatchedQuadrotor:
    """N envs of ``kind`` ('quad2d' | 'quad2d_sl' | 'quad3d' | 'quad3d_sl') on GPU ``device``."""

    def __init__(self, kind, num_envs: int, device: int = 0, seed: int = 0, env_id_base: int = 0,
                 auto_reset: bool = True, track_episodes: bool = True, params: Optional[A.Params] = None,
                 reading_2d: Optional[str] = None, use_torch_stream: bool = True):
        self.kind = A.KIND_BY_NAME[kind] if isinstance(kind, str) else int(kind)
        self.kind_name = A.KIND_NAMES[self.kind]
        self.num_envs = int(num_envs)
        self.nS, self.nA = A.STATE_DIM[self.kind], A.ACTION_DIM[self.kind]
        self.device = int(device)
        self.auto_reset, self.track_episodes = bool(auto_reset), bool(track_episodes)
        self._lib = A.lib()
        p = params if params is not None else A.default_params(self.kind, reading_2d)
        stream = None
        self._tstream = None
        if use_torch_stream and torch is not None and torch.cuda.is_available():
            self._tstream = torch.cuda.current_stream(self.device)
            # torch's default stream is the legacy NULL stream (handle 0); NULL means "create your own"
            # in rmav_create, so name it explicitly: hipStreamLegacy == (hipStream_t)1.
            stream = C.c_void_p(self._tstream.cuda_stream or 1)
        flags = (A.F_AUTO_RESET if auto_reset else 0) | (A.F_TRACK_EPISODES if track_episodes else 0)
        h = C.c_void_p()
        A.check(self._lib.rmav_create(C.byref(h), self.kind, self.num_envs, self.device, seed & (2**64 - 1),
                                      env_id_base, flags, C.byref(p), stream))
        self._h = h

    # ---- lifetime ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rmav_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed: int):
        A.check(self._lib.rmav_seed(self._h, int(seed) & (2**64 - 1)))

    def use_stream(self, torch_stream=None):
        """Rebind to a torch stream (default: torch's current stream on this device)."""
        st = torch_stream if torch_stream is not None else torch.cuda.current_stream(self.device)
        self._tstream = st
        A.check(self._lib.rmav_set_stream(self._h, C.c_void_p(st.cuda_stream or 1)))

    def sync(self):
        A.check(self._lib.rmav_sync(self._h))

    def set_tuning(self, **kv):
        """Explicit overrides of the launch rules (``rmav_set_tuning``, ``_abi.TUNE``): split, slice, store_policy, split_group,
        block, step_lazy, step_store, policy_pair, pair_group; -1 = automatic.  Results never depend on them."""
        for k, v in kv.items():
            A.check(self._lib.rmav_set_tuning(self._h, A.TUNE[k], int(v)))

    @property
    def params(self) -> A.Params:
        p = A.Params()
        A.check(self._lib.rmav_get_params(self._h, C.byref(p)))
        return p

    @params.setter
    def params(self, p: A.Params):
        A.check(self._lib.rmav_set_params(self._h, C.byref(p)))

    @property
    def step_count(self) -> int:
        t = C.c_uint64()
        A.check(self._lib.rmav_get_step_count(self._h, C.byref(t)))
        return t.value

    @step_count.setter
    def step_count(self, t: int):
        A.check(self._lib.rmav_set_step_count(self._h, int(t)))

    # ---- buffer helpers --------------------------------------------------------------------------------
    def _shape(self, dim: int, layout: str, T: Optional[int] = None):
        core = (self.num_envs, dim) if layout == "aos" else (dim, self.num_envs)
        return core if T is None else (T,) + core

    def _new(self, shape, dtype, like_tensor: bool):
        if like_tensor:
            tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
            return torch.empty(shape, dtype=tdt, device=f"cuda:{self.device}")
        return np.empty(shape, dtype=dtype)

    @staticmethod
    def _ptr(x):
        if x is None:
            return None
        if _is_tensor(x):
            assert x.is_contiguous(), "device buffers must be contiguous"
            return C.c_void_p(x.data_ptr())
        assert x.flags.c_contiguous
        return C.c_void_p(x.ctypes.data)

    def _in(self, x, shape, dtype=np.float32):
        """Validate / convert a caller array; returns (array, mem)."""
        if _is_tensor(x):
            if not x.is_cuda or x.device.index != self.device:
                raise ValueError(f"tensor must live on cuda:{self.device}")
            tdt = {np.float32: torch.float32, np.int32: torch.int32, np.uint32: torch.int32}[dtype]
            if x.dtype != tdt:
                x = x.to(tdt)
            if tuple(x.shape) != tuple(shape):
                raise ValueError(f"expected shape {tuple(shape)}, got {tuple(x.shape)}")
            return x.contiguous(), A.DEVICE
        x = np.ascontiguousarray(x, dtype=dtype)
        if x.shape != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {x.shape}")
        return x, A.HOST

    def _pitch_of(self, out: dict, T: int) -> int:
        """Column pitch P if every tensor of ``out`` is a ``[..., :N]`` view of a feature-major array with pitch P > N (what
        ``rollout(pitched=True)`` returns), 0 if they are plain contiguous arrays; anything else is an error."""
        if all((not _is_tensor(t)) or t.is_contiguous() for t in out.values()):   # the common case, kept cheap: it is on the launch path
            return 0
        pitches = set()
        for k, t in out.items():
            dim = {"actions": self.nA, "obs": self.nS}.get(k)
            shape = (T, self.num_envs) if dim is None else (T, dim, self.num_envs)
            if tuple(t.shape) != shape:
                raise ValueError(f"out[{k!r}]: expected shape {shape}, got {tuple(t.shape)}")
            if t.is_contiguous():
                pitches.add(0)
                continue
            st = t.stride()
            P = st[-2]
            if st[-1] != 1 or P < self.num_envs or (dim is not None and T > 1 and st[0] != dim * P):
                raise ValueError(f"out[{k!r}] is neither contiguous nor a [..., :N] view of a pitched array (strides {st})")
            pitches.add(int(P))
        if len(pitches) > 1:
            raise ValueError(f"`out` buffers mix column pitches {sorted(pitches)}")
        return pitches.pop() if pitches else 0

    # ---- the hot path ----------------------------------------------------------------------------------
    def reset(self, layout: str = "aos", device_out: bool = False):
        obs = self._new(self._shape(self.nS, layout), np.float32, device_out)
        A.check(self._lib.rmav_reset(self._h, self._ptr(obs), A.DEVICE if device_out else A.HOST, _layout(layout)))
        return obs

    def step(self, actions, layout: str = "aos", out=None):
        """actions [N,nA] ('aos') or [nA,N] ('soa') -> (obs, reward f32[N], done u8/bool[N]).

        ``out`` may carry preallocated (obs, rew, done) buffers of the same family as ``actions``."""
        a, mem = self._in(actions, self._shape(self.nA, layout))
        dev = mem == A.DEVICE
        if out is None:
            obs = self._new(self._shape(self.nS, layout), np.float32, dev)
            rew = self._new((self.num_envs,), np.float32, dev)
            done = self._new((self.num_envs,), np.uint8, dev)
        else:
            obs, rew, done = out
        A.check(self._lib.rmav_step(self._h, self._ptr(a), self._ptr(obs), self._ptr(rew), self._ptr(done), mem,
                                    _layout(layout)))
        if not dev and out is None:
            done = done.astype(bool)
        return obs, rew, done

    def control(self, layout: str = "aos", device_out: bool = False):
        act = self._new(self._shape(self.nA, layout), np.float32, device_out)
        A.check(self._lib.rmav_control(self._h, self._ptr(act), A.DEVICE if device_out else A.HOST, _layout(layout)))
        return act

    def control_step(self, layout: str = "aos", out=None, device_out: bool = False):
        """``a = control(); step(a)`` in one launch -> (actions, obs, reward, done)."""
        dev = device_out or (out is not None and _is_tensor(out[0]))
        if out is None:
            act = self._new(self._shape(self.nA, layout), np.float32, dev)
            obs = self._new(self._shape(self.nS, layout), np.float32, dev)
            rew = self._new((self.num_envs,), np.float32, dev)
            done = self._new((self.num_envs,), np.uint8, dev)
        else:
            act, obs, rew, done = out
        A.check(self._lib.rmav_control_step(self._h, self._ptr(act), self._ptr(obs), self._ptr(rew), self._ptr(done),
                                            A.DEVICE if dev else A.HOST, _layout(layout)))
        if not dev and out is None:
            done = done.astype(bool)
        return act, obs, rew, done

    def step_control(self, actions, layout: str = "aos", out=None):
        """``step(actions)`` and, in the same launch, ``control()`` of the new state
        -> (obs, reward, done, next_actions)."""
        a, mem = self._in(actions, self._shape(self.nA, layout))
        dev = mem == A.DEVICE
        if out is None:
            obs = self._new(self._shape(self.nS, layout), np.float32, dev)
            rew = self._new((self.num_envs,), np.float32, dev)
            done = self._new((self.num_envs,), np.uint8, dev)
            nxt = self._new(self._shape(self.nA, layout), np.float32, dev)
        else:
            obs, rew, done, nxt = out
        A.check(self._lib.rmav_step_control(self._h, self._ptr(a), self._ptr(obs), self._ptr(rew), self._ptr(done),
                                            self._ptr(nxt), mem, _layout(layout)))
        if not dev and out is None:
            done = done.astype(bool)
        return obs, rew, done, nxt

    def gae(self, rew, done, values, gamma: float = 0.99, lam: float = 0.95, reward_scale: float = 1.0, out=None,
            sums=None):
        """GAE(lambda) over a time-major device trajectory (``rmav_gae``): rew f32 [T,N], done u8 [T,N],
        values f32 [T+1,N] -> (adv [T,N], returns [T,N]); ``sums`` (f64[2] device tensor, optional) receives
        (sum A, sum A^2)."""
        T = int(rew.shape[0])
        assert tuple(rew.shape) == (T, self.num_envs) and tuple(done.shape) == (T, self.num_envs)
        assert tuple(values.shape) == (T + 1, self.num_envs) and done.dtype == torch.uint8
        adv, ret = out if out is not None else (torch.empty_like(rew), torch.empty_like(rew))
        A.check(self._lib.rmav_gae(self._h, T, self._ptr(rew), self._ptr(done), self._ptr(values), float(gamma), float(lam),
                                   float(reward_scale), self._ptr(adv), self._ptr(ret), self._ptr(sums)))
        return adv, ret

    def normalize_(self, x, mean: float, rstd: float):
        """In place ``x <- (x - mean) * rstd`` on the env's stream (advantage normalisation)."""
        A.check(self._lib.rmav_normalize(self._h, self._ptr(x), x.numel(), float(mean), float(rstd)))
        return x

    def rollout(self, n_steps: int, mode: str = "random", actions=None, layout: str = "soa", fused: bool = True,
                want=("obs", "rew", "done"), device_out: bool = False, out: Optional[dict] = None, pitched: bool = False) -> dict:
        """Run ``n_steps`` steps of every env.  Returns a dict of the requested trajectories
        (subset of 'actions', 'obs', 'rew', 'done').  ``out`` may carry preallocated buffers.

        ``pitched=True`` (device, feature-major, in-kernel actions): the arrays this call allocates get a column pitch of
        ``rmav_trajectory_pitch()`` = N rounded up to 64 and the results are ``[..., :N]`` views of them - same values, but a batch
        size that is not a multiple of 16 keeps the fast store path (include/rmav.h: rmav_rollout_pitched; 65 599 envs: 66.6 -> 46.3 us
        per 64-step launch).  Opt-in, because such views are not contiguous (``.view()`` on them fails).  They can be handed back
        as ``out=`` for the allocate-once-then-reuse idiom: ``out`` tensors that are ``[..., :N]`` views with a common pitch are
        recognised and the call goes through rmav_rollout_pitched again.

        ``layout="chunked"`` (device tensors, fused): the trajectory layout this GPU stores fastest for EVERY kind and batch size -
        chunk-major ``[C, T, dim, chunk]`` / ``[C, T, chunk]`` with ``chunk = rmav_chunk_envs()`` (:meth:`rollout_chunked`): 65 536-env
        chunks for quadrotor3d beyond 65 536 envs (131 072 envs: 0.67 -> 0.78 of the HBM roofline), ONE chunk (C = 1, i.e. the plain
        feature-major array with a leading axis of 1) everywhere else.  A learner that flattens (step, env) samples - PPO2 does -
        consumes it as it is: ``x.permute(0, 1, 3, 2).reshape(-1, dim)``; :meth:`unchunk` gives the plain ``[T, dim, N]`` copy."""
        if layout == "chunked":
            if not fused:
                raise ValueError("layout='chunked' is a fused-rollout layout")
            return self.rollout_chunked(n_steps, mode=mode, actions=actions, want=want, out=out)
        T = int(n_steps)
        m = _MODES[mode]
        a_in, mem = None, (A.DEVICE if device_out else A.HOST)
        if m == A.ACT_BUFFER:
            a_in, mem = self._in(actions, self._shape(self.nA, layout, T))
        if out and any(_is_tensor(v) for v in out.values()):
            mem = A.DEVICE
        dev = mem == A.DEVICE
        P = self._pitch_of(out, T) if (out and dev and layout == "soa" and m != A.ACT_BUFFER) else 0
        if pitched and not P:
            if not (dev and layout == "soa" and m != A.ACT_BUFFER) or out:
                raise ValueError("pitched=True needs device_out=True, layout='soa', in-kernel actions and no plain `out` buffers")
            P = int(self._lib.rmav_trajectory_pitch(self._h))
            if P < self.num_envs:
                A.check(P)
        if P:
            N = self.num_envs
            full = {k: v for k, v in (out or {}).items()}
            for key, shape, dt in (("actions", (T, self.nA, P), np.float32), ("obs", (T, self.nS, P), np.float32),
                                   ("rew", (T, P), np.float32), ("done", (T, P), np.uint8)):
                if key in want and key not in full:
                    full[key] = self._new(shape, dt, True)[..., :N]
            pp = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
            A.check(self._lib.rmav_rollout_pitched(self._h, T, m, None, pp(full.get("actions")), pp(full.get("obs")),
                                                   pp(full.get("rew")), pp(full.get("done")), P, 1 if fused else 0))
            return full
        res = dict(out) if out else {}
        if "actions" in want and m != A.ACT_BUFFER and "actions" not in res:
            res["actions"] = self._new(self._shape(self.nA, layout, T), np.float32, dev)
        if "obs" in want and "obs" not in res:
            res["obs"] = self._new(self._shape(self.nS, layout, T), np.float32, dev)
        if "rew" in want and "rew" not in res:
            res["rew"] = self._new((T, self.num_envs), np.float32, dev)
        if "done" in want and "done" not in res:
            res["done"] = self._new((T, self.num_envs), np.uint8, dev)
        A.check(self._lib.rmav_rollout(self._h, T, m, self._ptr(a_in), self._ptr(res.get("actions")) if m != A.ACT_BUFFER else None,
                                       self._ptr(res.get("obs")), self._ptr(res.get("rew")), self._ptr(res.get("done")),
                                       mem, _layout(layout), 1 if fused else 0))
        if m == A.ACT_BUFFER and "actions" in want:
            res["actions"] = a_in
        return res

    def rollout_chunked(self, n_steps: int, mode: str = "random", actions=None, chunk: Optional[int] = None,
                        want=("obs", "rew", "done"), out: Optional[dict] = None) -> dict:
        """Fused rollout with CHUNK-MAJOR device trajectories (``rmav_rollout_chunked``): 'actions' ``[C, T, nA, chunk]``, 'obs'
        ``[C, T, nS, chunk]``, 'rew' / 'done' ``[C, T, chunk]`` with ``C = ceil(N / chunk)``; env ``i`` is column ``i % chunk`` of chunk
        ``i // chunk`` (columns past N in the last chunk are never written).  Same values as :meth:`rollout`; for big batches of the
        3-D kinds each chunk runs as its own launch at the 65 536-env rate (include/rmav.h says why).  ``chunk`` defaults to
        ``rmav_chunk_envs()``.  :meth:`unchunk` gives the plain ``[T, dim, N]`` view of a result (a copy)."""
        T, m = int(n_steps), _MODES[mode]
        ch = int(self._lib.rmav_chunk_envs(self._h)) if chunk is None else int(chunk)
        if ch <= 0:
            A.check(ch)
        ch = min(ch, (self.num_envs + 63) // 64 * 64)
        nc = -(-self.num_envs // ch)
        shapes = {"actions": (nc, T, self.nA, ch), "obs": (nc, T, self.nS, ch), "rew": (nc, T, ch), "done": (nc, T, ch)}
        a_in = None
        if m == A.ACT_BUFFER:
            a_in, _ = self._in(actions, shapes["actions"])
        res = dict(out) if out else {}
        for key in want:
            if key == "actions" and m == A.ACT_BUFFER:
                continue
            if key not in res:
                res[key] = self._new(shapes[key], np.uint8 if key == "done" else np.float32, True)
            elif tuple(res[key].shape) != shapes[key] or not res[key].is_contiguous():
                raise ValueError(f"out[{key!r}]: expected a contiguous tensor of shape {shapes[key]}")
        A.check(self._lib.rmav_rollout_chunked(self._h, T, m, self._ptr(a_in), self._ptr(res.get("actions")) if m != A.ACT_BUFFER else None,
                                               self._ptr(res.get("obs")), self._ptr(res.get("rew")), self._ptr(res.get("done")), ch))
        if m == A.ACT_BUFFER and "actions" in want:
            res["actions"] = a_in
        return res

    def unchunk(self, x):
        """``[C, T, dim, chunk]`` / ``[C, T, chunk]`` -> ``[T, dim, N]`` / ``[T, N]`` (a copy, for code that wants the plain layout)."""
        if x.dim() == 4:
            c, t, d, ch = x.shape
            return x.permute(1, 2, 0, 3).reshape(t, d, c * ch)[..., :self.num_envs].contiguous()
        c, t, ch = x.shape
        return x.permute(1, 0, 2).reshape(t, c * ch)[..., :self.num_envs].contiguous()

    # ---- state access ----------------------------------------------------------------------------------
    def get_state(self, layout: str = "aos", device_out: bool = False):
        s = self._new(self._shape(self.nS, layout), np.float32, device_out)
        A.check(self._lib.rmav_get_state(self._h, self._ptr(s), A.DEVICE if device_out else A.HOST, _layout(layout)))
        return s

    def set_state(self, s, layout: str = "aos"):
        s, mem = self._in(s, self._shape(self.nS, layout))
        A.check(self._lib.rmav_set_state(self._h, self._ptr(s), mem, _layout(layout)))

    def get_sbd(self) -> np.ndarray:
        out = np.empty(self.num_envs, dtype=np.int32)
        A.check(self._lib.rmav_get_sbd(self._h, self._ptr(out), A.HOST))
        return out

    def set_sbd(self, sbd):
        sbd = np.ascontiguousarray(sbd, dtype=np.int32)
        assert sbd.shape == (self.num_envs,)
        A.check(self._lib.rmav_set_sbd(self._h, self._ptr(sbd), A.HOST))

    def get_reset_counts(self) -> np.ndarray:
        out = np.empty(self.num_envs, dtype=np.uint32)
        A.check(self._lib.rmav_get_reset_counts(self._h, self._ptr(out), A.HOST))
        return out

    def set_reset_counts(self, rc):
        rc = np.ascontiguousarray(rc, dtype=np.uint32)
        assert rc.shape == (self.num_envs,)
        A.check(self._lib.rmav_set_reset_counts(self._h, self._ptr(rc), A.HOST))

    def set_env_param(self, name: str, values):
        """Per-env (domain-randomised) constant: name in {'mass', 'load_mass', 'tether_length'}; ``values`` is one
        float per env (NumPy array or CUDA tensor) or None to return to the shared ``params`` value."""
        which = {"mass": A.PARAM_MASS, "load_mass": A.PARAM_LOAD_MASS, "tether_length": A.PARAM_TETHER_LENGTH}[name]
        if values is None:
            A.check(self._lib.rmav_set_env_param(self._h, which, None, A.HOST))
            return
        v, mem = self._in(values, (self.num_envs,))
        A.check(self._lib.rmav_set_env_param(self._h, which, self._ptr(v), mem))

    def get_time(self) -> np.ndarray:
        """'reinmav' envs only: each env's own clock t (float64)."""
        out = np.empty(self.num_envs, dtype=np.float64)
        A.check(self._lib.rmav_get_time(self._h, self._ptr(out), A.HOST))
        return out

    def set_time(self, t):
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype=np.float64), (self.num_envs,)))
        A.check(self._lib.rmav_set_time(self._h, self._ptr(t), A.HOST))

    # ---- episode statistics ------------------------------------------------------------------------------
    def episode_totals(self, clear: bool = False) -> dict:
        t = A.EpTotals()
        A.check(self._lib.rmav_episode_totals(self._h, C.byref(t), 1 if clear else 0))
        return {"episodes": int(t.episodes), "return_sum": float(t.return_sum), "length_sum": int(t.length_sum)}

    def pack_stats(self, send):
        """send i32[2 * cmax] (device tensor) <- (last returns as bits | last lengths), zero padded: the payload of
        the per-rollout all-gather, written by one small launch on the env's stream."""
        A.check(self._lib.rmav_pack_stats(self._h, send.numel() // 2, self._ptr(send)))
        return send

    def episode_buffers(self, device_out: bool = False) -> dict:
        n = (self.num_envs,)
        lr, ll = self._new(n, np.float32, device_out), self._new(n, np.int32, device_out)
        cr, cl = self._new(n, np.float32, device_out), self._new(n, np.int32, device_out)
        A.check(self._lib.rmav_episode_buffers(self._h, self._ptr(lr), self._ptr(ll), self._ptr(cr), self._ptr(cl),
                                               A.DEVICE if device_out else A.HOST))
        return {"last_return": lr, "last_length": ll, "cur_return": cr, "cur_length": cl}
