#!/usr/bin/env python3
"""k_step with feature-major [nA][N] vs batch-major [N][nA] caller actions, warm (2 buffers) and cold (64 buffers) action rings:
HIP-event time per launch.  N as argv (default 262144 1048576)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

for n in [int(x) for x in (sys.argv[1:] or ["262144", "1048576"])]:
    for layout in ("soa", "aos"):
        for ring_n in (2, 64):
            env = g.BatchedQuadrotor("quad3d", n, seed=0, auto_reset=True, track_episodes=True)
            shape = (ring_n, 4, n) if layout == "soa" else (ring_n, n, 4)
            ring = torch.empty(shape, device="cuda").uniform_(0.0, 10.0)
            rew = torch.empty((ring_n, n), device="cuda")
            done = torch.empty((ring_n, n), dtype=torch.uint8, device="cuda")
            K = max(200, 2000 * 65536 // n)

            def run(k):
                while k > 0:
                    m = min(k, ring_n)
                    env.rollout(m, mode="buffer", actions=ring[:m], layout=layout, fused=False, want=("rew", "done"), out={"rew": rew[:m], "done": done[:m]})
                    k -= m
            run(100)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(K)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / K)
            print(f"| {n} | {layout} | {ring_n} | {best * 1e3:.2f} | {101 * n / best / 1e6 / 8000:.3f} |", flush=True)
            env.close()
            del ring, rew, done
            torch.cuda.empty_cache()
