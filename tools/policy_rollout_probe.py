#!/usr/bin/env python3
"""Rollout-only timing of the in-kernel policy collectors (no learner): us per env-step batch, per kind."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

N = int(os.environ.get("N", 65536)); T = int(os.environ.get("T", 64)); iters = int(os.environ.get("ITERS", 40))
for kind in os.environ.get("KINDS", "quad3d quad2d quad2d_sl quad3d_sl").split():
    for bf16 in (True, False) if os.environ.get("FP32", "0") == "1" else (True,):
        torch.manual_seed(0)
        env = g.BatchedQuadrotor(kind, N, seed=0)
        pol = MlpPolicy(env.nS, env.nA).cuda()
        ro = FusedPolicyCollector(env, pol, T, bf16_mfma=bf16)
        for _ in range(5):
            ro.collect(); ro.roll_over()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            ro.collect(); ro.roll_over()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{kind:10s} {'bf16 MFMA' if bf16 else 'fp32':9s} n={N} T={T}: {dt / iters / T * 1e6:7.2f} us per env-step batch, {N * T * iters / dt / 1e9:6.2f} G env-steps/s", flush=True)
        env.close()
