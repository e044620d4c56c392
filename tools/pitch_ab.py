"""Pitched vs plain trajectory layout at batch sizes that are not multiples of 16 (us per 64-step launch, HIP events)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

def run(kind, n, pitched, reps=60, T=64):
    env = g.BatchedQuadrotor(kind, n, seed=0, auto_reset=True, track_episodes=True)
    for _ in range(10):
        env.rollout(T, mode="random", device_out=True, pitched=pitched, want=("actions", "obs", "rew", "done"))
    torch.cuda.synchronize()
    ts = env._tstream or torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    keep = []
    with torch.cuda.stream(ts):
        e0.record(ts)
        for _ in range(reps):
            keep.append(env.rollout(T, mode="random", device_out=True, pitched=pitched, want=("actions", "obs", "rew", "done")))
            if len(keep) > 6:
                keep.pop(0)
        e1.record(ts)
    torch.cuda.synchronize()
    env.close()
    return e0.elapsed_time(e1) / reps * 1e3

for kind in ("quad3d", "quad3d_sl"):
    for n in (65536, 65599, 131071, 262143, 1048575):
        reps = max(8, 60 * 65536 // n)
        a, b = run(kind, n, False, reps), run(kind, n, True, reps)
        print(f"{kind} n={n}: plain {a:8.1f} us   pitched {b:8.1f} us", flush=True)
