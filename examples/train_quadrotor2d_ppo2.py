#!/usr/bin/env python3
"""Counterpart of the reference's examples/train_quadrotor2d_ppo2.py (stable-baselines PPO2 on quadrotor2d-v0,
300 k timesteps, MlpPolicy): the same task through this package's CLI."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinmav-gym_amd"))
from gym_reinmav_amd.run import main

if __name__ == "__main__":
    main(["train_quadrotor2d_ppo2", "--alg=ppo2", "--env=quadrotor2d-v0", "--network=mlp", "--num_env=1024",
          "--num_timesteps=3e6", "--nsteps=32", "--reward_scale=0.1", "--lr=1e-3", "--save_path=/tmp/ppo2_quadrotor2d.pt",
          "--play", "--play_episodes=3"] + sys.argv[1:])
