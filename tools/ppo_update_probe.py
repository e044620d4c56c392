#!/usr/bin/env python3
"""Where does PPO.update spend its time at C5's per-GPU shape (65 536 envs x 32 steps)?  torch.profiler kernel table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy
N, T = int(os.environ.get("N", 65536)), 32
torch.manual_seed(0)
env = g.BatchedQuadrotor("quad3d", N, seed=0)
pol = MlpPolicy(env.nS, env.nA).cuda()
ro = FusedPolicyCollector(env, pol, T)
ppo = PPO(pol)
for _ in range(2):
    ro.collect(); ppo.update(ro); ro.roll_over()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ro.collect(); st = ppo.update(ro); ro.roll_over()
torch.cuda.synchronize(); print(f"full iteration {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms", st)
if os.environ.get("PROFILE", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        ro.collect(); ppo.update(ro); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=70))
