#!/usr/bin/env python3
"""Rollouts with the fp32-MFMA actor only (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy
torch.manual_seed(0)
N, T = int(os.environ.get("N", 65536)), 32
env = g.BatchedQuadrotor("quad3d", N, seed=0)
pol = MlpPolicy(env.nS, env.nA).cuda()
ro = FusedPolicyCollector(env, pol, T, f32_mfma=True)
for _ in range(int(os.environ.get("ITERS", 12))):
    ro.collect(); ro.roll_over()
torch.cuda.synchronize()
