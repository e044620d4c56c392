#!/usr/bin/env python3
"""How much of the 1e-6 parity bar does the device path use?  Teacher-forced fused rollouts (random actions and
the controller) vs the fp64 oracle, many seeds; prints the worst scaled error |d| / max(1, |y_ref|) per kind.
Test infrastructure (uses oracle/), not part of the product.  Run on a GPU box: python tools/parity_margin.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("reinmav-gym_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np

import gym_reinmav_amd as g
import oracle as O
from util import KINDS, near_threshold, scaled_err

SEEDS, N, T = int(os.environ.get("SEEDS", "12")), 8192, 32
print("| kind | action source | env-steps compared | worst obs error | 99.99 % obs error | worst reward error | worst action error | done mismatches away from a limit |")
print("|---|---|---|---|---|---|---|---|")
for kind in KINDS:
    for mode in ("random", "controller"):
        worst_o, worst_r, worst_a, cnt, bad_done, samples = 0.0, 0.0, 0.0, 0, 0, []
        for seed in range(SEEDS):
            env = g.BatchedQuadrotor(kind, N, seed=seed, auto_reset=True, track_episodes=True)
            prev = env.get_state()
            tr = env.rollout(T, mode=mode, layout="aos", want=("actions", "obs", "rew", "done"))
            sbd = None
            for k in range(T):
                if mode == "controller":
                    a = O.batch_control(kind, prev.astype(np.float64))
                    worst_a = max(worst_a, float(scaled_err(tr["actions"][k], a).max()))
                o2, r, d, sbd_o = O.batch_step(kind, prev.astype(np.float64), tr["actions"][k].astype(np.float64), sbd)
                dk = tr["done"][k].astype(bool)
                ok = near_threshold(kind, o2)
                bad_done += int(((dk != d) & ~ok).sum())
                alive = ~dk & ~d
                e = scaled_err(tr["obs"][k][alive], o2[alive])
                worst_o = max(worst_o, float(e.max()))
                samples.append(e.max(axis=1))
                worst_r = max(worst_r, float(scaled_err(tr["rew"][k][alive], r[alive]).max()))
                cnt += int(alive.sum())
                sbd = None   # the reward machine is covered by the tests; here only alive steps are compared
                prev = tr["obs"][k]
            env.close()
        q = float(np.quantile(np.concatenate(samples), 0.9999))
        print(f"| {kind} | {mode} | {cnt} | {worst_o:.2e} | {q:.2e} | {worst_r:.2e} | {worst_a:.2e} | {bad_done} |", flush=True)
