"""Does the COLUMN PITCH of the trajectory arrays matter for the fused rollout's store stream?  (power-of-two pitches put the same env of
every component row on the same HBM channel.)  quadrotor3d, random actions, 64-step launches into a cold ring; HIP events.
python tools/pitch_probe.py  ->  us per launch by (envs, pitch)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

dev = torch.device("cuda", 0)
T = 64
print("| kind | envs | column pitch | us per launch | frac of 8 TB/s |\n|---|---|---|---|---|")
for kind, n in (("quad3d", 65536), ("quad3d", 131072), ("quad3d", 262144), ("quad3d_sl", 262144), ("quad2d", 131072)):
    for extra in (0, 64, 192, 1024, 4096 + 64, n // 2 + 64, n):
        P = n + extra
        env = g.BatchedQuadrotor(kind, n, seed=0)
        nS, nA = env.nS, env.nA
        per_set = P * T * (4 * (nS + nA + 1) + 1)
        R = max(5, -(-int(1.5e9) // per_set))
        ring = []
        for _ in range(R):
            full = {"actions": torch.zeros((T, nA, P), device=dev), "obs": torch.zeros((T, nS, P), device=dev),
                    "rew": torch.zeros((T, P), device=dev), "done": torch.zeros((T, P), dtype=torch.uint8, device=dev)}
            ring.append({k: (v[..., :n] if extra else v) for k, v in full.items()})
        K, W = max(100, 65536 * 600 // n), max(30, 65536 * 150 // n)
        for i in range(W):
            env.rollout(T, mode="random", layout="soa", want=("actions", "obs", "rew", "done"), device_out=True, out=ring[i % R])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            env.rollout(T, mode="random", layout="soa", want=("actions", "obs", "rew", "done"), device_out=True, out=ring[(W + i) % R])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        b = n * (T * (4 * (nS + nA + 1) + 1) + 8 * nS + 24)
        print(f"| {kind} | {n} | N + {extra} | {us:.2f} | {b / us / 1e6 / 8:.3f} |", flush=True)
        env.close(); del ring; torch.cuda.empty_cache()
