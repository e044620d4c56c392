#!/usr/bin/env python3
"""Two side measurements of the single-step boundary (1 GPU):

1. hipGraph replay of a chain of single-step launches vs the plain C launch loop in rmav_rollout(fused=0)
   (is the 4.7 us per launch a stream-launch cost that a graph removes?).
2. PCIe-inclusive rate of the RMAV_HOST path: rmav_step with NumPy (pageable) buffers, one env-step per call.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import numpy as np
import torch

import gym_reinmav_amd as g

dev = torch.device("cuda", 0)
kind = "quad3d"
nS, nA = 10, 4
out_rows = []
for n in [int(x) for x in os.environ.get("NS", "65536,262144").split(",")]:
    CH = 64
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor(kind, n, seed=0)
        ring = torch.empty((CH, nA, n), dtype=torch.float32, device=dev).uniform_(0, 10)
        rew = torch.empty((CH, n), device=dev)
        done = torch.empty((CH, n), dtype=torch.uint8, device=dev)

        def chain():
            env.rollout(CH, mode="buffer", actions=ring, layout="soa", fused=False, want=("rew", "done"),
                        device_out=True, out={"rew": rew, "done": done})

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / (reps * CH) * 1e3

        us_loop = timed(chain, 32)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            chain()
        us_graph = timed(gr.replay, 32)
        env.close()
    out_rows.append({"n_envs": n, "us_per_step_launch_loop": round(us_loop, 3), "us_per_step_graph_replay": round(us_graph, 3)})
    print(out_rows[-1], flush=True)

# host path: pageable NumPy buffers in and out, one rmav_step per call
for n in (65536,):
    env = g.BatchedQuadrotor(kind, n, seed=0)
    rng = np.random.default_rng(0)
    act = rng.uniform(0, 10, size=(nA, n)).astype(np.float32)
    for _ in range(5):
        env.step(act, layout="soa")
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        obs, r, d = env.step(act, layout="soa")
    el = time.perf_counter() - t0
    row = {"n_envs": n, "host_path_us_per_step": round(el / K * 1e6, 1), "host_path_env_steps_per_s": round(n * K / el),
           "bytes_over_pcie_per_env_step": 4 * nA + 4 * nS + 4 + 1}
    out_rows.append(row)
    print(row, flush=True)
    env.close()

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "step_graph_probe.json"), "w") as f:
    json.dump(out_rows, f, indent=1)
