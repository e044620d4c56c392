#!/usr/bin/env python3
"""Fused rollouts with caller-provided actions (mode='buffer': replaying / evaluating recorded action sequences) next to
random-action ones: us per 64-step launch on cold trajectory rings."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
T = 64
for kind in ("quad3d", "quad3d_sl"):
    for n in (16384, 65536, 131072, 262144):
        env = g.BatchedQuadrotor(kind, n, seed=0)
        nS, nA = env.nS, env.nA
        acts = [torch.empty((T, nA, n), device="cuda").uniform_(0, 10) for _ in range(5)]
        ring = [dict(obs=torch.zeros((T, nS, n), device="cuda"), rew=torch.zeros((T, n), device="cuda"),
                     done=torch.zeros((T, n), dtype=torch.uint8, device="cuda")) for _ in range(5)]
        res = {}
        for mode in ("buffer", "random"):
            kw = dict(actions=None)
            for i in range(100):
                env.rollout(T, mode=mode, actions=acts[i % 5] if mode == "buffer" else None, want=("obs", "rew", "done"), device_out=True, out=ring[i % 5])
            torch.cuda.synchronize(); t0 = time.perf_counter(); K = 300
            for i in range(K):
                env.rollout(T, mode=mode, actions=acts[i % 5] if mode == "buffer" else None, want=("obs", "rew", "done"), device_out=True, out=ring[i % 5])
            torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / K * 1e6
        print(f"{kind} n={n}: caller actions {res['buffer']:.1f} us, random actions (no action output) {res['random']:.1f} us", flush=True)
        env.close()
