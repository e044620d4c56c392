"""Two OS processes, one GPU: the multi-rank path end to end (sharding by global env id, per-rank HIP work,
the per-rollout all-gather and the totals all-reduce) with a gloo group carrying the statistics.  The node's
RCCL path itself is covered with one rank in test_gpu_api.py::test_rccl_all_gather_single_rank (RCCL refuses
two ranks on one device)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_total, T, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    import torch
    import torch.distributed as dist

    from gym_reinmav_amd.distributed import all_gather_episode_stats, all_reduce_totals, make_sharded, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = make_sharded("quad3d", n_total, rank, world, device=0, seed=3)   # both ranks share cuda:0 here
    start, count = shard_range(n_total, rank, world)
    assert env.num_envs == count
    tr = env.rollout(T, mode="random", layout="soa", want=("obs", "rew", "done"))
    eb = env.episode_buffers()
    rets, lens = all_gather_episode_stats(torch.from_numpy(eb["last_return"]), torch.from_numpy(eb["last_length"]), n_total)
    tot = all_reduce_totals(env.episode_totals())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), obs=tr["obs"], rew=tr["rew"], done=tr["done"], rets=rets.numpy(),
             lens=lens.numpy(), tot=np.array([tot["episodes"], tot["return_sum"], tot["length_sum"]]), start=start)
    env.close()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_equal_one(tmp_path, built):
    import torch
    import torch.multiprocessing as mp

    assert torch.cuda.is_available()
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    import gym_reinmav_amd as g

    n_total, T, world = 20001, 96, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_total, T, str(tmp_path)), nprocs=world, join=True)
    full = g.BatchedQuadrotor("quad3d", n_total, seed=3)
    tr = full.rollout(T, mode="random", layout="soa", want=("obs", "rew", "done"))
    eb, tot = full.episode_buffers(), full.episode_totals()
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert [int(p["start"]) for p in parts] == [0, 10001]
    for key in ("obs", "rew", "done"):
        assert np.array_equal(np.concatenate([p[key] for p in parts], axis=-1), tr[key]), key
    for r, p in enumerate(parts):   # every rank holds the same gathered statistics, equal to the unsharded run
        bad = np.flatnonzero(p["rets"] != eb["last_return"])
        assert bad.size == 0, f"rank {r}: last_return differs at envs {bad[:8]}: {p['rets'][bad[:8]]} vs {eb['last_return'][bad[:8]]}"
        bad = np.flatnonzero(p["lens"] != eb["last_length"])
        assert bad.size == 0, f"rank {r}: last_length differs at envs {bad[:8]}: {p['lens'][bad[:8]]} vs {eb['last_length'][bad[:8]]}"
        assert p["tot"][0] == tot["episodes"] and p["tot"][2] == tot["length_sum"], f"rank {r}: totals {p['tot']} vs {tot}"
        assert abs(p["tot"][1] - tot["return_sum"]) <= 1e-6 * abs(tot["return_sum"]) + 1e-3, f"rank {r}: totals {p['tot']} vs {tot}"
    assert tot["episodes"] > 0, tot
    assert tot["episodes"] == int(tr["done"].sum()), (tot, int(tr["done"].sum()))
    full.close()
