#!/usr/bin/env python3
"""Run-to-run determinism probe: same seed, fresh process -> identical bits?  Prints checksums."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import numpy as np
import gym_reinmav_amd as g
for kind in ("quad3d", "quad3d_sl"):
    env = g.BatchedQuadrotor(kind, 65536, seed=0)
    for it in range(20):
        env.rollout(32, mode="random", want=())
    for it in range(300):
        env.rollout(1, mode="random", want=(), fused=False)
    s = env.get_state(layout="soa")
    print(kind, hashlib.sha1(s.tobytes()).hexdigest()[:16], env.episode_totals(), int(env.get_reset_counts().sum()), int(env.get_sbd().sum()))
    env.close()
