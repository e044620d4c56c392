"""The N>1 path on CPU: world_size-2 gloo run of the per-rollout episode-statistics all-gather and of
the shard-count invariance of the RNG keying (global env ids)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, out_dir):
    for p in (os.path.join(ROOT, "reinmav-gym_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import oracle as O
    from gym_reinmav_amd.distributed import EpisodeStatsExchange, all_gather_episode_stats, all_reduce_totals, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = shard_range(n_total, rank, world)
    # each rank produces its shard's stats with the CPU oracle standing in for the GPU shard
    ids = np.arange(start, start + count)
    state = O.reset_states("quad3d", 7, ids, 0)
    sbd = np.full(count, -1, np.int32)
    epi = np.ones(count, np.uint32)
    ret = np.zeros(count, np.float32)
    ln = np.zeros(count, np.int32)
    for e in range(count):  # 20 steps per env, record the running return as "last_return"
        st = state[e:e + 1].copy()
        k, rs, nd = O.rollout_random("quad3d", st, sbd[e:e + 1], epi[e:e + 1], 20, 7, int(ids[e]), 0.0, 10.0)
        ret[e], ln[e] = rs, 20
    g_ret, g_len = all_gather_episode_stats(torch.from_numpy(ret), torch.from_numpy(ln), n_total)
    # the overlapped exchange object bench.py uses (CPU tensors: same code path minus the streams); three posts
    # cycle both send / receive buffer pairs
    ex = EpisodeStatsExchange(n_total, "cpu")
    for k in range(3):
        ex.post(torch.from_numpy(ret + k), torch.from_numpy(ln + k))
        r2, l2 = ex.result()
        assert torch.equal(r2, g_ret + k) and torch.equal(l2, g_len + k)
    tot = all_reduce_totals({"episodes": count, "return_sum": float(ret.sum()), "length_sum": int(ln.sum())})
    np.save(os.path.join(out_dir, f"ret_{rank}.npy"), g_ret.numpy())
    np.save(os.path.join(out_dir, f"len_{rank}.npy"), g_len.numpy())
    np.save(os.path.join(out_dir, f"tot_{rank}.npy"), np.array([tot["episodes"], tot["return_sum"], tot["length_sum"]]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_total", [37, 36])   # odd: shards of 19 and 18 exercise the padding; even: the no-slicing path
def test_all_gather_episode_stats_world2(tmp_path, built, n_total):
    import oracle as O

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    # single-process result over the whole batch (1 "GPU") must equal the gathered shards
    ids = np.arange(n_total)
    state = O.reset_states("quad3d", 7, ids, 0)
    exp = np.zeros(n_total, np.float32)
    for e in range(n_total):
        st = state[e:e + 1].copy()
        _, rs, _ = O.rollout_random("quad3d", st, np.full(1, -1, np.int32), np.ones(1, np.uint32), 20, 7, e, 0.0, 10.0)
        exp[e] = rs
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"ret_{r}.npy"), exp)
        assert np.array_equal(np.load(tmp_path / f"len_{r}.npy"), np.full(n_total, 20, np.int32))
        tot = np.load(tmp_path / f"tot_{r}.npy")
        assert tot[0] == n_total and tot[2] == 20 * n_total and abs(tot[1] - exp.astype(np.float64).sum()) < 1e-3
