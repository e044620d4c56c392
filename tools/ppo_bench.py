#!/usr/bin/env python3
"""Config-5-style measurement on one GPU: PPO2 rollout collection (policy forward + env.step per env-step)
eager vs hipGraph-captured, and a full iteration including the learner.  One JSON line per variant."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy, RolloutCollector

N = int(os.environ.get("N", 65536)); T = int(os.environ.get("T", 32)); iters = int(os.environ.get("ITERS", 20))
for graph in (False, True, "fused", "fused-bf16"):
    torch.manual_seed(0)
    env = g.BatchedQuadrotor("quad3d", N, seed=0)
    pol = MlpPolicy(env.nS, env.nA).cuda()
    ro = (FusedPolicyCollector(env, pol, T, bf16_mfma=(graph == "fused-bf16")) if isinstance(graph, str)
          else RolloutCollector(env, pol, T, graph=graph))
    for _ in range(3):
        ro.collect(); ro.roll_over()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        ro.collect(); ro.roll_over()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    line = {"what": "ppo rollout collection", "collector": {False: "torch eager", True: "torch + hipGraph", "fused": "in-kernel policy fp32 (rmav_rollout_policy)", "fused-bf16": "in-kernel policy bf16 MFMA"}[graph], "envs": N, "nsteps": T, "us_per_env_step_batch": dt / iters / T * 1e6,
            "env_steps_per_s": N * T * iters / dt}
    ppo = PPO(pol)
    ppo.update(ro); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ro.collect(); st = ppo.update(ro); ro.roll_over()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    line["full_iteration_ms"] = dt / 5 * 1e3
    line["env_steps_per_s_with_update"] = N * T * 5 / dt
    line["stats"] = st
    print(json.dumps(line), flush=True)
    env.close()
