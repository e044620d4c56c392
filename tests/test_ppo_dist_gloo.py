"""BASELINE configs[4] (C5) splits the PPO2 learner over GPUs data-parallel: every rank collects its env shard's
rollout and the minibatch gradients are averaged (the reference's counterpart is baselines' MPI mode,
gym_reinmav/run.py:18-21,177-182).  World-size-2 gloo run on CPU tensors: two ranks that start from DIFFERENT
parameters and see DIFFERENT rollouts must, after sync_parameters() + PPO.update(), hold identical parameters,
equal to what one process gets from the average of the two shards' gradients."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NS, NA, T, N = 10, 4, 6, 48


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_rollout(rank):
    """A synthetic rollout of rank `rank`'s env shard with the RolloutCollector's buffer shapes (CPU tensors)."""
    g = torch.Generator().manual_seed(100 + rank)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    ro = types.SimpleNamespace()
    ro.obs, ro.act = r(T + 1, NS, N), r(T, NA, N)
    ro.logp, ro.val, ro.rew = -4.0 + 0.3 * r(T, N), r(T + 1, N), r(T, N)
    ro.done = (torch.rand(T, N, generator=g) < 0.1).to(torch.uint8)
    ro.env = None
    return ro


def _flat(policy):
    return torch.cat([p.detach().reshape(-1) for p in policy.parameters()])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    import torch.distributed as dist

    from gym_reinmav_amd.ppo import PPO, MlpPolicy, sync_parameters

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)                       # different initial parameters per rank ...
    pol = MlpPolicy(NS, NA)
    before = _flat(pol).clone()
    sync_parameters(pol, src=0)                   # ... until rank 0's are broadcast
    synced = _flat(pol).clone()
    ppo = PPO(pol, epochs=2, minibatches=1)       # one minibatch = the whole shard: the permutation does not matter
    stats = ppo.update(_shard_rollout(rank))
    np.save(os.path.join(out_dir, f"before_{rank}.npy"), before.numpy())
    np.save(os.path.join(out_dir, f"synced_{rank}.npy"), synced.numpy())
    np.save(os.path.join(out_dir, f"after_{rank}.npy"), _flat(pol).numpy())
    assert all(np.isfinite(v) for v in stats.values())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ppo_data_parallel_world2(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
    from gym_reinmav_amd.ppo import PPO, MlpPolicy, gae

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ld = lambda n: np.load(tmp_path / n)  # noqa: E731
    assert not np.array_equal(ld("before_0.npy"), ld("before_1.npy"))
    assert np.array_equal(ld("synced_0.npy"), ld("before_0.npy")) and np.array_equal(ld("synced_1.npy"), ld("before_0.npy"))
    assert np.array_equal(ld("after_0.npy"), ld("after_1.npy")), "ranks diverged after the averaged update"
    assert not np.array_equal(ld("after_0.npy"), ld("synced_0.npy"))

    # one process, same start, gradient = mean of the two shards' gradients per minibatch
    torch.manual_seed(0)
    pol = MlpPolicy(NS, NA)
    assert np.array_equal(_flat(pol).numpy(), ld("before_0.npy"))
    ppo = PPO(pol, epochs=2, minibatches=1)
    shards = []
    for r in range(world):
        ro = _shard_rollout(r)
        adv, ret = gae(ro.rew, ro.val, ro.done, ppo.gamma, ppo.lam)
        shards.append((ro.obs[:T].permute(1, 0, 2).reshape(NS, T * N), ro.act.permute(1, 0, 2).reshape(NA, T * N),
                       ro.logp.reshape(-1), ro.val[:T].reshape(-1), adv.reshape(-1), ret.reshape(-1)))
    params = list(pol.parameters())
    for _ in range(ppo.epochs):
        grads = []
        for sh in shards:
            ppo.opt.zero_grad(set_to_none=True)
            ppo.loss(*sh)[0].backward()
            grads.append([p.grad.clone() for p in params])
        for p, g0, g1 in zip(params, *grads):
            p.grad = (g0 + g1) / world
        torch.nn.utils.clip_grad_norm_(params, ppo.max_grad_norm)
        ppo.opt.step()
    assert np.allclose(_flat(pol).numpy(), ld("after_0.npy"), rtol=0, atol=2e-6), \
        float(np.abs(_flat(pol).numpy() - ld("after_0.npy")).max())
