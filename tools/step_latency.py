#!/usr/bin/env python3
"""Launch-latency probe of the single-step kernel: HIP-event time per launch for feature subsets."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))
import torch
import gym_reinmav_amd as g

dev = torch.device("cuda", 0)
kind = sys.argv[1] if len(sys.argv) > 1 else "quad3d"
nS, nA = {"quad3d": (10, 4), "quad3d_sl": (16, 4), "quad2d": (5, 2), "quad2d_sl": (9, 2)}[kind]
RING = 256
rows = []
for n in [int(x) for x in os.environ.get("NS", "65536").split(",")]:
    for track in (True, False):
        for auto in (True, False):
            for outs in ("rew+done", "none", "obs+rew+done"):
                for mode in ("buffer", "random"):
                    stream = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(stream):
                        env = g.BatchedQuadrotor(kind, n, seed=0, auto_reset=auto, track_episodes=track)
                        ring = torch.empty((RING, nA, n), dtype=torch.float32, device=dev).uniform_(0, 10)
                        out = {}
                        want = ()
                        if "rew" in outs:
                            out = {"rew": torch.empty((RING, n), device=dev), "done": torch.empty((RING, n), dtype=torch.uint8, device=dev)}
                            want = ("rew", "done")
                        if "obs" in outs:
                            out["obs"] = torch.empty((RING, nS, n), device=dev)
                            want = ("obs", "rew", "done")
                        def run(k):
                            while k > 0:
                                m = min(k, RING)
                                env.rollout(m, mode=mode, actions=ring[:m] if mode == "buffer" else None, layout="soa", fused=False,
                                            want=want, device_out=True, out={k_: v[:m] for k_, v in out.items()})
                                k -= m
                        run(100)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        K = 1024
                        e0.record(stream); run(K); e1.record(stream); torch.cuda.synchronize()
                        us = e0.elapsed_time(e1) / K * 1e3
                        rows.append((n, track, auto, outs, mode, us))
                        print(f"n={n:8d} track={track!s:5} auto_reset={auto!s:5} outs={outs:13s} mode={mode:6s} {us:7.2f} us/launch", flush=True)
                        env.close()
