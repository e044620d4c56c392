"""Chunk-major vs plain trajectory layout for big batches (rmav_rollout_chunked): us per 64-step launch, cold ring, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

T, want = 64, ("actions", "obs", "rew", "done")
print("| kind | envs | actions | layout | us per launch | frac of 8 TB/s |\n|---|---|---|---|---|---|")
for kind, n, mode in (("quad3d", 131072, "random"), ("quad3d", 262144, "random"), ("quad3d", 1048576, "random"), ("quad3d_sl", 262144, "random"),
                      ("quad3d_sl", 131072, "random"), ("quad3d", 131072, "controller"), ("quad3d", 262144, "controller"), ("quad2d", 262144, "random")):
    for chunk in (0, 65536, 32768, 131072):
        if chunk >= n: continue
        env = g.BatchedQuadrotor(kind, n, seed=0)
        nS, nA = env.nS, env.nA
        per_set = n * T * (4 * (nS + nA + 1) + 1)
        R = max(5, -(-int(1.5e9) // per_set))
        try:
            if chunk:
                ring = [env.rollout_chunked(T, mode=mode, chunk=chunk, want=want) for _ in range(R)]
                run = lambda i: env.rollout_chunked(T, mode=mode, chunk=chunk, want=want, out=ring[i % R])
            else:
                ring = [env.rollout(T, mode=mode, layout="soa", want=want, device_out=True) for _ in range(R)]
                run = lambda i: env.rollout(T, mode=mode, layout="soa", want=want, device_out=True, out=ring[i % R])
        except Exception as e:
            print(f"| {kind} | {n} | {mode} | chunk {chunk} | {e!r} | |"); env.close(); continue
        K, W = max(60, 65536 * 500 // n), max(20, 65536 * 120 // n)
        for i in range(W): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K): run(W + i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / K * 1e3
        b = n * (T * (4 * (nS + nA + 1) + 1) + 8 * nS + 24)
        print(f"| {kind} | {n} | {mode} | {'chunks of ' + str(chunk) if chunk else 'plain'} | {us:.2f} | {b / us / 1e6 / 8:.3f} |", flush=True)
        env.close(); del ring; torch.cuda.empty_cache()
