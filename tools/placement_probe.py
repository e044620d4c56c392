#!/usr/bin/env python3
"""Is the +-10 % spread of the 131 072-env rollout between runs a property of WHERE the trajectory buffers landed?
One process: T trials, each with a freshly allocated ring of 6 buffer sets (addresses shifted by a retained spacer of
random size), 300 cold-ring launches timed with HIP events; then every ring is timed a second time."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
n, T, K, R = int(os.environ.get("N", "131072")), 64, 300, 6
dev = torch.device("cuda", 0)
env = g.BatchedQuadrotor("quad3d", n, seed=0)
want = ("actions", "obs", "rew", "done")
random.seed(int(os.environ.get("SEED", "0")))
def time_ring(ring):
    for i in range(60):
        env.rollout(T, mode="random", want=want, device_out=True, out=ring[i % R])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(K):
        env.rollout(T, mode="random", want=want, device_out=True, out=ring[i % R])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
rings, spacers, first = [], [], []
for t in range(int(os.environ.get("TRIALS", "6"))):
    spacers.append(torch.empty(random.randrange(1, 4096) * 4096 + random.randrange(0, 64) * 64, dtype=torch.uint8, device=dev))
    ring = [dict(actions=torch.zeros((T, 4, n), device=dev), obs=torch.zeros((T, 10, n), device=dev),
                 rew=torch.zeros((T, n), device=dev), done=torch.zeros((T, n), dtype=torch.uint8, device=dev)) for _ in range(R)]
    rings.append(ring)
    first.append(time_ring(ring))
second = [time_ring(r) for r in rings]
third = [time_ring(r) for r in rings]
print("trial  obs[0] address      first  second  third  (us per launch)")
for t, r in enumerate(rings):
    print(f"{t:5d}  {r[0]['obs'].data_ptr():#016x}  {first[t]:6.1f}  {second[t]:6.1f}  {third[t]:6.1f}")
