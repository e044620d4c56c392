#!/usr/bin/env python3
"""Train / save / load / play with the batched envs - the flow of the reference's entry point
(`python -m gym_reinmav.run --alg=ppo2 --env=quadrotor3d-v0 --num_timesteps=... --save_path=... [--load_path=...] [--play]`,
gym_reinmav/run.py:186-211) on top of this library, with the same flag names.  baselines / TensorFlow are third party and
absent; the learner is gym_reinmav_amd.ppo.PPO (baselines' ppo2 defaults), the model file is a torch state_dict.

    python examples/train_ppo2.py --env quadrotor3d-v0 --num_env 8192 --num_timesteps 2e7 --save_path /tmp/quad3d.pt
    python examples/train_ppo2.py --env quadrotor3d-v0 --load_path /tmp/quad3d.pt --num_timesteps 0 --play
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinmav-gym_amd"))
import torch

import gym_reinmav_amd as g
from gym_reinmav_amd.ppo import PPO, FusedPolicyCollector, MlpPolicy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="quadrotor3d-v0", choices=sorted(g.ENV_IDS))
    ap.add_argument("--num_env", type=int, default=8192, help="envs on this GPU (the reference: SubprocVecEnv workers)")
    ap.add_argument("--num_timesteps", type=float, default=2e7)
    ap.add_argument("--nsteps", type=int, default=64, help="env-steps per env and rollout")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reward_scale", type=float, default=0.05)
    ap.add_argument("--actor", default="f16", choices=["fp32", "bf16", "f16"], help="arithmetic of the in-kernel actor")
    ap.add_argument("--save_path", default=None)
    ap.add_argument("--load_path", default=None)
    ap.add_argument("--play", action="store_true", help="after training: run the policy (mean action) on one env and print its path")
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    kind = g.ENV_IDS[args.env]
    env = g.BatchedQuadrotor(kind, args.num_env, seed=args.seed)
    policy = MlpPolicy(env.nS, env.nA).cuda()
    if args.load_path:                                      # run.py:188 model.load(load_path)
        policy.load_state_dict(torch.load(args.load_path, map_location="cuda"))
    elif kind in ("quad3d", "quad3d_sl"):
        with torch.no_grad():
            policy.pi[2].bias[0] = 9.8                      # start around hover thrust
    collector = FusedPolicyCollector(env, policy, args.nsteps, bf16_mfma=(args.actor == "bf16"), f16_mfma=(args.actor == "f16"))
    learner = PPO(policy, lr=1e-3, reward_scale=args.reward_scale)
    iters = int(args.num_timesteps // (args.num_env * args.nsteps))
    t0 = time.perf_counter()
    for it in range(iters):
        env.episode_totals(clear=True)
        collector.collect()                                 # one kernel launch: policy + env for nsteps steps of every env
        stats = learner.update(collector)                   # GAE + clipped-surrogate epochs
        collector.roll_over()
        if it % 10 == 0 or it == iters - 1:
            tot = env.episode_totals()
            print(f"iter {it:4d}  timesteps {(it + 1) * args.num_env * args.nsteps:.3g}  eprewmean {tot['return_sum'] / max(1, tot['episodes']):8.2f}  "
                  f"eplenmean {tot['length_sum'] / max(1, tot['episodes']):7.1f}  explained_variance {stats['explained_variance']:.3f}  "
                  f"{(it + 1) * args.num_env * args.nsteps / (time.perf_counter() - t0):.3g} steps/s", flush=True)
    env.close()
    if args.save_path:                                      # run.py:186 model.save(save_path)
        torch.save(policy.state_dict(), args.save_path)
        print("saved", args.save_path)
    if args.play:                                           # run.py:190-211: obs = env.reset(); loop model.step / env.step
        venv = g.QuadrotorVecEnv(args.env, 1, seed=args.seed)
        obs = venv.reset()
        ep_rew = 0.0
        for k in range(400):
            with torch.no_grad():
                mean, _ = policy(obs.t().contiguous())      # the policy is feature-major: obs [nS, N]
            obs, rew, done, _ = venv.step(mean.t().contiguous())
            ep_rew += float(rew[0])
            if bool(done[0]):
                print(f"episode_rew={ep_rew:.2f} after {k + 1} steps")
                ep_rew = 0.0
        print("final obs", [round(float(x), 3) for x in obs[0]])
        venv.close()


if __name__ == "__main__":
    main()
