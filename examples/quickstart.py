#!/usr/bin/env python3
"""Quick tour of gym_reinmav_amd on one MI355X (run from the repo root after `python __graft_entry__.py`).

1. the reference's test loop (test/test_quadrotor3d.py) on a gym-shaped env,
2. 65 536 envs behind the baselines-VecEnv contract with torch tensors,
3. whole rollouts in one launch: random actions, the geometric controller, an in-kernel MLP policy."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinmav-gym_amd"))
import torch

import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

# 1. drop-in single env (old-gym 4-tuple API, .control() like the reference)
env = g.make("quadrotor3d-v0", seed=0)
env.reset()
for i in range(400):
    obs, reward, done, _ = env.step(env.control())
    if done:
        env.reset()
print("quadrotor3d-v0 under its geometric controller, final |p - ref| =", float(((obs[:3] - env.ref_pos) ** 2).sum() ** 0.5))
env.close()

# 2. VecEnv: device tensors in, device tensors out
venv = g.QuadrotorVecEnv("quadrotor3d-v0", 65536, seed=0)
obs = venv.reset()
actions = torch.rand((65536, 4), device="cuda") * 10
obs, rew, done, infos = venv.step(actions)
print("VecEnv step:", tuple(obs.shape), float(rew.mean()), int(done.sum()), "done")
venv.close()

# 3. fused rollouts
batch = g.BatchedQuadrotor("quad3d", 65536, seed=0)
for mode in ("random", "controller"):
    want = ("actions", "obs", "rew", "done")
    tr = batch.rollout(64, mode=mode, layout="soa", want=want, device_out=True)   # allocates the trajectory tensors
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr = batch.rollout(64, mode=mode, layout="soa", want=want, device_out=True, out=tr)   # reuses them
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"64-step rollout, {mode:10s}: {65536 * 64 / dt / 1e9:6.1f} G env-steps/s, mean reward {float(tr['rew'].mean()):.3f}")
policy = MlpPolicy(batch.nS, batch.nA).cuda()
for bf16 in (False, True):
    ro = FusedPolicyCollector(batch, policy, 32, bf16_mfma=bf16)
    ro.collect(); torch.cuda.synchronize(); t0 = time.perf_counter()
    ro.collect(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"32-step PPO rollout, in-kernel policy ({'bf16 MFMA' if bf16 else 'fp32'}): {65536 * 32 / dt / 1e9:5.2f} G env-steps/s")
print("episodes finished:", batch.episode_totals())
batch.close()
