#!/usr/bin/env python3
"""A/B of QuadrotorVecEnv.step (device tensors, 65 536 envs) under rmav_set_tuning overrides: python tools/vecenv_ab.py key=value ..."""
import sys, time
sys.path.insert(0, "reinmav-gym_amd")
import torch
import gym_reinmav_amd as g
n = 65536
variants = [dict()] + [dict([kv.split("=")[0], int(kv.split("=")[1])] for kv in [a]) for a in sys.argv[1:]]
for rep in range(2):
    for tune in variants:
        for reuse in (True, False):
            venv = g.QuadrotorVecEnv("quadrotor3d-v0", n, seed=0, reuse_buffers=reuse)
            venv.env.set_tuning(**tune)
            venv.reset()
            act = torch.empty((n, 4), device="cuda").uniform_(0, 10)
            for _ in range(300): venv.step(act)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3000): venv.step(act)
            torch.cuda.synchronize()
            print(f"tune={tune} reuse_buffers={reuse}: {1e6 * (time.perf_counter() - t0) / 3000:.2f} us per step", flush=True)
            venv.close()
