#!/usr/bin/env python3
"""How much of a fused env-step is the trajectory stores?  Same kernel, different sets of outputs (1 GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

dev = torch.device("cuda", 0)
kind, n, T = os.environ.get("KIND", "quad3d"), int(os.environ.get("N", "65536")), 64
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    for want in [(), ("rew", "done"), ("obs",), ("actions",), ("obs", "rew", "done"), ("actions", "obs", "rew", "done")]:
        env = g.BatchedQuadrotor(kind, n, seed=0, auto_reset=True, track_episodes=os.environ.get("TRACK", "0") == "1")
        out = env.rollout(T, mode=os.environ.get("MODE", "random"), want=want, device_out=True)
        def run():
            env.rollout(T, mode=os.environ.get("MODE", "random"), want=want, device_out=True, out=out)
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 100
        e0.record(stream)
        for _ in range(K): run()
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        print(f"{kind} n={n} T={T} outputs={'+'.join(want) or 'none':28s} {us:8.2f} us/launch  {us / T:6.3f} us/env-step batch", flush=True)
        env.close()
