#!/usr/bin/env python3
"""Kernel-only time of the in-kernel-policy rollout (rmav_rollout_policy) by actor and pairs per workgroup.
N, T, KIND, ITERS as environment variables; prints one line per variant (ms per rollout, G env-steps/s)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

kind, T, iters = os.environ.get("KIND", "quad3d"), int(os.environ.get("T", "32")), int(os.environ.get("ITERS", "60"))
variants = [("fp32_valu", {}), ("fp32_mfma", {}), ("bf16_1w", {"policy_pair": 0})] + [(a, {"pair_group": G}) for a in ("bf16", "f16", "f16_shared") for G in (1, 2, 4)]
if os.environ.get("VARIANTS"):
    variants = [v for v in variants if v[0] in os.environ["VARIANTS"].split(",")]
for n in [int(x) for x in os.environ.get("N", "65536").split(",")]:
    for actor, tune in variants:
        torch.manual_seed(0)
        env = g.BatchedQuadrotor(kind, n, seed=0, auto_reset=True, track_episodes=True)
        if tune:
            env.set_tuning(**tune)
        pol = MlpPolicy(env.nS, env.nA, value_network=("shared" if actor == "f16_shared" else "copy")).cuda()
        ro = FusedPolicyCollector(env, pol, T, bf16_mfma=actor.startswith("bf16"), f16_mfma=actor.startswith("f16"), f32_mfma=(actor == "fp32_mfma"))
        for _ in range(5):
            ro.collect()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ro._A.check(ro._call[0](env._h, *ro._call[1:]))
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
        print(f"{kind} n={n} T={T} {actor:10s} {str(tune):22s} {best:8.4f} ms/rollout  {n * T / best / 1e6:7.2f} G env-steps/s", flush=True)
        env.close()
