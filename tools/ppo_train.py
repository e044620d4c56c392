#!/usr/bin/env python3
"""End-to-end sanity run: PPO2-style training of quadrotor3d-v0 with the fused in-kernel rollout.
Prints a learning curve (mean reward per env-step, mean episode length) as JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy

N = int(os.environ.get("N", 8192)); T = int(os.environ.get("T", 64)); iters = int(os.environ.get("ITERS", 150))
bf16 = os.environ.get("BF16", "0") == "1"
torch.manual_seed(0)
env = g.BatchedQuadrotor("quad3d", N, seed=0)
pol = MlpPolicy(env.nS, env.nA, init_logstd=0.0).cuda()
with torch.no_grad():
    pol.pi[2].bias[0] = 9.8   # start around hover thrust (the action Box is [0, 10])
ro = FusedPolicyCollector(env, pol, T, bf16_mfma=bf16)
ppo = PPO(pol, lr=float(os.environ.get('LR', 1e-3)), epochs=4, minibatches=4, reward_scale=float(os.environ.get('RS', 0.05)))
t0 = time.perf_counter()
for it in range(iters):
    env.episode_totals(clear=True)
    ro.collect()
    st = ppo.update(ro)
    ro.roll_over()
    if it % 10 == 0 or it == iters - 1:
        tot = env.episode_totals()
        rec = {"iter": it, "env_steps": (it + 1) * N * T, "mean_reward_per_step": float(ro.rew.mean()),
               "episodes_finished": tot["episodes"], "mean_episode_length": tot["length_sum"] / max(1, tot["episodes"]),
               "mean_episode_return": tot["return_sum"] / max(1, tot["episodes"]), "explained_variance": st["explained_variance"],
               "wall_s": time.perf_counter() - t0}
        print(json.dumps(rec), flush=True)
env.close()
