#!/usr/bin/env python3
"""Does the rollout's speed depend on how long the GPU has been busy?  One cold ring, launches timed in slices of 100,
PHASES busy phases of SECS seconds separated by idle gaps of GAP seconds; rocm-smi sampled once per phase end."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g
n, T, R = int(os.environ.get("N", "131072")), 64, 6
dev = torch.device("cuda", 0)
env = g.BatchedQuadrotor("quad3d", n, seed=0)
want = ("actions", "obs", "rew", "done")
ring = [dict(actions=torch.zeros((T, 4, n), device=dev), obs=torch.zeros((T, 10, n), device=dev),
             rew=torch.zeros((T, n), device=dev), done=torch.zeros((T, n), dtype=torch.uint8, device=dev)) for _ in range(R)]
MEMSET = os.environ.get("MEMSET") == "1"   # the same bytes per launch as one fill kernel (the write ceiling over time)
flat = [torch.empty(n * (T * 61 + 104), dtype=torch.uint8, device=dev) for _ in range(R)] if MEMSET else None
torch.cuda.synchronize()
def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.split(":", 1)[-1].strip() for l in o.splitlines() if any(k in l for k in ("ower", "sclk", "fclk", "mclk", "unction", "emory"))]
        return " | ".join(keep)
    except Exception as e:
        return repr(e)
secs, gap = float(os.environ.get("SECS", "1.5")), float(os.environ.get("GAP", "3"))
for ph in range(int(os.environ.get("PHASES", "3"))):
    t_end = time.perf_counter() + secs
    rows, i = [], 0
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            if MEMSET:
                flat[i % R].zero_()
            else:
                env.rollout(T, mode="random", want=want, device_out=True, out=ring[i % R])
            i += 1
        e1.record(); torch.cuda.synchronize()
        rows.append(e0.elapsed_time(e1) * 10)
    print(f"phase {ph}: {len(rows)} slices of 100 launches, us per launch:", " ".join(f"{r:.1f}" for r in rows[:12]), "...", " ".join(f"{r:.1f}" for r in rows[-4:]), flush=True)
    print("   smi right after:", smi(), flush=True)
    time.sleep(gap)
    print("   smi after the idle gap:", smi(), flush=True)
