"""Sharding of the env batch over the GPUs of one node, and the one collective the path has.

Envs are independent, so the data path needs no communication: rank ``r`` of ``R`` owns a contiguous
range of GLOBAL env ids and keys its RNG by the global id, which makes results independent of ``R``.
The only exchange is one all-gather of per-env episode statistics per rollout (RCCL over xGMI when
the tensors are on GPUs and the process group backend is ``nccl``; ``gloo`` on CPU tensors in tests).
"""
from __future__ import annotations

from typing import Optional, Tuple


try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """(first global env id, count) of ``rank``; the remainder goes to the lowest ranks."""
    if world <= 0 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("need 0 <= rank < world and n_total >= 0")
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def make_sharded(kind, n_total: int, rank: int, world: int, device: Optional[int] = None, seed: int = 0, **kw):
    """This rank's ``BatchedQuadrotor`` shard of an ``n_total``-env batch (same ``seed`` on every rank)."""
    from .core import BatchedQuadrotor

    start, count = shard_range(n_total, rank, world)
    if count == 0:
        raise ValueError("empty shard: fewer envs than ranks")
    return BatchedQuadrotor(kind, count, device=rank if device is None else device, seed=seed, env_id_base=start, **kw)


def all_gather_episode_stats(last_return, last_length, n_total: int, group=None):
    """All-gather per-env (return, length) of the most recently finished episodes.

    ``last_return`` f32[count], ``last_length`` i32[count] are this rank's shard (torch tensors, on
    the GPU for the nccl backend).  Returns (returns f32[n_total], lengths i32[n_total]) in global env
    order on every rank.  Shards may differ by one env, so each rank pads to the largest shard."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    _, count = shard_range(n_total, rank, world)
    assert last_return.shape[0] == count and last_length.shape[0] == count
    cmax = -(-n_total // world)
    pad = cmax - count
    # one message: returns and lengths packed into a single int32 buffer (bit-cast), 8*cmax bytes/rank
    send = torch.empty(2 * cmax, dtype=torch.int32, device=last_return.device)
    send[:count] = last_return.contiguous().view(torch.int32)
    send[cmax:cmax + count] = last_length
    if pad:
        send[count:cmax] = 0
        send[cmax + count:] = 0
    recv = torch.empty(world * 2 * cmax, dtype=torch.int32, device=last_return.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, 2, cmax)
    rets, lens = [], []
    for r in range(world):
        _, c = shard_range(n_total, r, world)
        rets.append(recv[r, 0, :c].view(torch.float32))
        lens.append(recv[r, 1, :c])
    return torch.cat(rets), torch.cat(lens)


class EpisodeStatsExchange:
    """The per-rollout all-gather, overlapped with the next rollout.

    ``post(env)`` packs this rank's (last_return, last_length) into one of two send buffers on the env's
    stream (a stream-ordered snapshot: the next rollout may overwrite the per-env arrays at once) and issues
    the RCCL all-gather on a second stream that waits only for that pack; ``result()`` makes the current
    stream wait for the most recent gather and returns (returns f32[n_total], lengths i32[n_total]) in global
    env order.  Equal-sized shards take the zero-slicing path; ragged ones pad to the largest shard."""

    def __init__(self, n_total: int, device, group=None):
        self.n_total, self.group = int(n_total), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        _, self.count = shard_range(self.n_total, self.rank, self.world)
        self.cmax = -(-self.n_total // self.world)
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.send = [torch.zeros(2 * self.cmax, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.recv = [torch.empty(self.world * 2 * self.cmax, dtype=torch.int32, device=self.device) for _ in range(2)]
        # high priority: a hardware queue of its own (streams of one priority share a few queues round-robin, and a
        # collective's stream that lands on the compute stream's queue serialises with it)
        self.comm_stream = torch.cuda.Stream(device=self.device, priority=-1) if self.on_gpu else None
        self.done_ev = [None, None]
        self._ready = [torch.cuda.Event(), torch.cuda.Event()] if self.on_gpu else None   # reused: no per-post allocation
        self._done = [torch.cuda.Event(), torch.cuda.Event()] if self.on_gpu else None
        self.i = 0
        self.last = None

    def arm(self, env=None):
        """Interface parity with NativeStatsExchange (nothing to prepare on this path)."""

    def post(self, last_return=None, last_length=None, env=None):
        """Either the two per-env tensors, or ``env=`` a BatchedQuadrotor shard (packed by one launch of its own)."""
        k = self.i & 1
        self.i += 1
        cur = torch.cuda.current_stream(self.device) if self.on_gpu else None
        if self.on_gpu and self.done_ev[k] is not None:
            cur.wait_event(self.done_ev[k])               # the gather that last used this pair of buffers
        s = self.send[k]
        if env is not None:
            assert env.num_envs == self.count
            env.pack_stats(s)
        else:
            s[:self.count].copy_(last_return.view(torch.int32))
            s[self.cmax:self.cmax + self.count].copy_(last_length)
        if self.on_gpu:
            ready = self._ready[k]
            ready.record(cur)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                dist.all_gather_into_tensor(self.recv[k], s, group=self.group)
                ev = self._done[k]
                ev.record(self.comm_stream)
            self.done_ev[k] = ev
        else:
            dist.all_gather_into_tensor(self.recv[k], s, group=self.group)
        self.last = k

    def result(self):
        assert self.last is not None, "post() first"
        k = self.last
        if self.on_gpu:
            torch.cuda.current_stream(self.device).wait_event(self.done_ev[k])
        recv = self.recv[k].view(self.world, 2, self.cmax)
        if self.n_total % self.world == 0:
            return recv[:, 0, :].reshape(-1).view(torch.float32), recv[:, 1, :].reshape(-1)
        rets, lens = [], []
        for r in range(self.world):
            _, c = shard_range(self.n_total, r, self.world)
            rets.append(recv[r, 0, :c].view(torch.float32))
            lens.append(recv[r, 1, :c])
        return torch.cat(rets), torch.cat(lens)


class NativeStatsExchange:
    """The same overlapped exchange with the collective issued by ``librmav.so`` itself (``rmav_comm_*`` +
    ``rmav_allgather_stats_post / _result``: RCCL ``ncclAllGather`` on the communicator's own HIP stream, double
    buffered).  ``torch.distributed`` only carries the 128-byte RCCL unique id from rank 0 to the others.  About
    15 us of host time per post against ~100 us for ``all_gather_into_tensor`` behind Python, which matters when one
    exchange follows every ~100 us rollout launch.

    ``connect_timeout_s``: bound the set-up.  ``rmav_comm_create`` (RCCL's rendezvous) AND ``rmav_comm_warmup`` (a tiny
    all-gather on the communicator's own stream: RCCL connects its transports inside the FIRST collective's enqueue, a
    host-side exchange that blocks when a peer has died after the rendezvous) run on a helper thread that touches
    nothing but the communicator and is joined with the deadline; then ONE complete exchange is posted and awaited on the
    HOST with the rest of it (``rmav_allgather_stats_wait`` polls the gather's completion event).  On ``TimeoutError`` the env and its stream are therefore exactly as before - a caller
    can fall back to ``EpisodeStatsExchange`` on the same env - and the communicator is abandoned (never destroyed: a
    thread or a collective may still be inside RCCL)."""

    def __init__(self, env, n_total: int, group=None, connect_timeout_s: Optional[float] = None):
        import ctypes as C

        from . import _abi as A

        self._A, self._C, self.env = A, C, env
        self.n_total, self.group = int(n_total), group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        L = A.lib()
        dev = torch.device("cuda", env.device)
        on_gpu = dist.get_backend(group) == "nccl"
        uid = torch.zeros(A.COMM_ID_BYTES, dtype=torch.uint8)
        err = None
        if self.rank == 0:
            try:   # a failure here must still reach the broadcast below, or the other ranks would wait for ever
                buf = (C.c_char * A.COMM_ID_BYTES)()
                A.check(L.rmav_comm_unique_id(buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            except Exception as e:  # pragma: no cover
                err = e
        uid = uid.to(dev) if on_gpu else uid
        dist.broadcast(uid, src=0, group=group)
        raw = bytes(uid.cpu().numpy().tobytes())
        if err is not None or not any(raw):   # all-zero id = rank 0 could not create one: every rank raises
            raise RuntimeError(f"no RCCL unique id from rank 0 ({err!r})")
        self._comm = C.c_void_p()
        self._abandoned = False
        self.ret = torch.empty(self.n_total, dtype=torch.float32, device=dev)
        self.len = torch.empty(self.n_total, dtype=torch.int32, device=dev)

        def create():
            A.check(L.rmav_comm_create(C.byref(self._comm), raw, self.rank, self.world, env.device))

        if connect_timeout_s is None:
            create()
            return
        import threading
        import time

        t0 = time.monotonic()
        box = []

        def target():
            try:
                with torch.cuda.device(dev):
                    create()
                    A.check(L.rmav_comm_warmup(self._comm, -1.0))   # the first collective: transports connect here
            except BaseException as e:  # noqa: BLE001 - handed to the caller's thread
                box.append(e)

        th = threading.Thread(target=target, daemon=True)
        th.start()
        th.join(connect_timeout_s)
        if th.is_alive():
            self._abandoned = True
            raise TimeoutError(f"rmav_comm_create + warm-up collective (RCCL rendezvous and transport set-up) did not complete in {connect_timeout_s} s")
        if box:
            raise box[0]
        # one whole exchange before anybody relies on it: post (a pack + signal on the env's stream, the gather on the
        # communicator's own stream), then a HOST-side wait with the rest of the deadline
        self.post()
        rc = L.rmav_allgather_stats_wait(self._comm, max(0.5, connect_timeout_s - (time.monotonic() - t0)))
        if rc == A.ERR_TIMEOUT:
            self._abandoned = True
            raise TimeoutError(f"the first exchange did not complete in {connect_timeout_s} s")
        A.check(rc)
        self.result()

    def info(self) -> dict:
        """{'rank', 'world'} as created and {'lib_rank', 'lib_world'} = what the collective library reports for its communicator
        (ncclCommUserRank / ncclCommCount; -1 if it does not export them)."""
        C, A = self._C, self._A
        v = [C.c_int(-1) for _ in range(4)]
        A.check(A.lib().rmav_comm_info(self._comm, *[C.byref(x) for x in v]))
        return dict(zip(("rank", "world", "lib_rank", "lib_world"), (x.value for x in v)))

    def arm(self, env=None):
        """Call BEFORE the rollout whose statistics the next post() exchanges: that launch then writes the snapshot itself
        and post() adds nothing to the env's stream (rmav_allgather_stats_arm)."""
        e = env if env is not None else self.env
        self._A.check(self._A.lib().rmav_allgather_stats_arm(e._h, self._comm, self.n_total))

    def post(self, env=None, **_):
        e = env if env is not None else self.env
        self._A.check(self._A.lib().rmav_allgather_stats_post(e._h, self._comm, self.n_total))

    def wait(self, timeout_s: float = -1.0) -> bool:
        """Host-side bounded wait for the most recent post (touches no stream); False on timeout."""
        rc = self._A.lib().rmav_allgather_stats_wait(self._comm, float(timeout_s))
        if rc == self._A.ERR_TIMEOUT:
            return False
        self._A.check(rc)
        return True

    def result(self):
        C = self._C
        self._A.check(self._A.lib().rmav_allgather_stats_result(self.env._h, self._comm, self.n_total,
                                                                 C.c_void_p(self.ret.data_ptr()), C.c_void_p(self.len.data_ptr())))
        return self.ret, self.len

    def close(self):
        if self._comm and not self._abandoned:
            self._A.lib().rmav_comm_destroy(self._comm)
        self._comm = None


def all_reduce_totals(totals: dict, device=None, group=None) -> dict:
    """Sum {'episodes','return_sum','length_sum'} over ranks (one 3-element all-reduce)."""
    t = torch.tensor([float(totals["episodes"]), float(totals["return_sum"]), float(totals["length_sum"])],
                     dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    e, r, l = t.tolist()
    return {"episodes": int(round(e)), "return_sum": r, "length_sum": int(round(l))}
