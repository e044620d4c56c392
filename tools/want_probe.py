import os, sys, time
sys.path.insert(0, "/root/repo/reinmav-gym_amd")
import torch
import gym_reinmav_amd as g
for n in (65536, 131072, 262144):
    for mode in ("random", "controller"):
        for want in (("actions", "obs", "rew", "done"), ("obs", "rew", "done"), ("obs",), ("rew", "done"), ("actions",), ()):
            env = g.BatchedQuadrotor("quad3d", n, seed=0)
            ring = [dict((k, t) for k, t in dict(actions=torch.zeros((64, 4, n), device="cuda"), obs=torch.zeros((64, 10, n), device="cuda"),
                         rew=torch.zeros((64, n), device="cuda"), done=torch.zeros((64, n), dtype=torch.uint8, device="cuda")).items() if k in want) for _ in range(5)]
            for i in range(150): env.rollout(64, mode=mode, want=want, device_out=True, out=ring[i % 5] if want else None)
            torch.cuda.synchronize(); t0 = time.perf_counter(); K = 400
            for i in range(K): env.rollout(64, mode=mode, want=want, device_out=True, out=ring[i % 5] if want else None)
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / K * 1e6
            b = n * 64 * sum(dict(actions=16, obs=40, rew=4, done=1)[k] for k in want)
            print(f"n={n} {mode:10s} want={'+'.join(want) or '-':22s} {us:7.1f} us  {b / us / 1e6:5.2f} TB/s", flush=True)
            env.close()
