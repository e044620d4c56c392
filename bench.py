#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched quadrotor3d-v0 hot path on N MI355X (one process per GPU).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched under
``python -m torch.distributed.run --nproc-per-node N``.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1], per GPU): quadrotor3d-v0, 65 536 envs, random actions
T~U[0,10), w~U[0,10)^3 drawn in-kernel from the counter RNG, auto-reset on done, episode tracking on.
One bench "step" = ONE launch of the hot-path kernel over the whole batch:

  --mode rollout (default): the fused rollout kernel advances every env ``--chunk`` (64) env-steps with the
      state held in registers and writes the full trajectory (actions, obs, reward, done per env-step)
      to HBM - the unit an RL learner consumes.
  --mode step: the same kernel at chunk = 1, one launch per env-step: actions read from a device buffer
      (what a policy would have written), state updated in place (obs == state), reward/done written.

value = (envs on all ranks) * chunk * K / max-over-ranks wall time of the K timed launches (inputs
already resident in HBM; barrier + synchronize on both sides).  For N > 1 the env batch is sharded by
global env id (weak scaling: 65 536 envs per GPU) and the timed region ends with the one collective the
path has: the RCCL all-gather of per-env episode returns/lengths.

roofline: algorithmic bytes per env-step (read state + read action + write state + write
reward + write done = 101 B for quadrotor3d) * env-steps per launch / average launch duration measured
with HIP events on the launch stream, against 8 TB/s.  cpu_baseline: the C oracle (oracle/, a port of
the reference's NumPy step) timed on one host core over a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "reinmav-gym_amd"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s HBM3E


def _omp_set_threads(k: int):
    import ctypes

    for name in ("libgomp.so.1", "libomp.so"):
        try:
            ctypes.CDLL(name).omp_set_num_threads(int(k))
            return True
        except Exception:
            continue
    return False


def cpu_baseline(kind: str, n: int, chunk: int, lo: float, hi: float, budget_s: float, threads: int = 1):
    """Time the CPU oracle (test infrastructure, used here only as the reported baseline)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import oracle as O

    O.lib()
    if not _omp_set_threads(threads):
        threads = 1

    nS = O.STATE_DIM[kind]
    state = np.random.RandomState(0).uniform(-1, 1, (n, nS)).astype(np.float32)
    sbd = np.full(n, -1, np.int32)
    epi = np.ones(n, np.uint32)
    O.rollout_random(kind, state[:1024].copy(), sbd[:1024].copy(), epi[:1024].copy(), 4, 0, 0, lo, hi)  # warm
    done_steps, t, t0 = 0, 0, time.perf_counter()
    while True:
        k, _, _ = O.rollout_random(kind, state, sbd, epi, chunk, 0, 0, lo, hi, t0=t)
        done_steps += k
        t += chunk
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {
        "value": done_steps / el,
        "unit": "env-steps/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{kind} C oracle (fp64, scalar code, {threads} thread{'s' if threads > 1 else ''}), {n} envs x {t} "
                  f"env-steps, random actions + auto-reset, {el:.1f} s on the GPU box's host CPU "
                  f"({os.cpu_count()} logical cores present)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed launches (SURVEY 8d asks for >= 1000 after >= 100 warm-up; 2000 launches ~ 0.1 s)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--kind", default="quad3d", choices=["quad2d", "quad2d_sl", "quad3d", "quad3d_sl", "reinmav"])
    ap.add_argument("--actions", default="random", choices=["random", "controller"],
                    help="action source of the fused rollout (controller = the reference's built-in / geometric controller)")
    ap.add_argument("--envs-per-gpu", type=int, default=65536)
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--chunk", type=int, default=64,
                    help="env-steps per launch in rollout mode (64 amortises the ~4.5 us fixed cost of a launch; "
                         "see profiles/*/sweep_kinds_sizes.md for 8..128)")
    ap.add_argument("--layout", default="soa", choices=["soa", "aos"], help="trajectory layout in rollout mode")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other mode's short measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gym_reinmav_amd as g
    from gym_reinmav_amd.distributed import all_gather_episode_stats, all_reduce_totals

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK is set) the distributed path is exercised even for one rank
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    torch.manual_seed(0)  # the step mode's action ring is filled by torch's generator
    kind = args.kind
    n = args.envs_per_gpu
    n_total = n * world
    A = g._abi
    nS, nA = A.STATE_DIM[A.KIND_BY_NAME[kind]], A.ACTION_DIM[A.KIND_BY_NAME[kind]]
    algo_bytes = A.lib().rmav_algorithmic_bytes(A.KIND_BY_NAME[kind])
    p = A.default_params(A.KIND_BY_NAME[kind])
    lo, hi = float(p.act_lo), float(p.act_hi)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor(kind, n, device=local_rank, seed=0, env_id_base=rank * n, auto_reset=True,
                                 track_episodes=True)

        RING = 512  # step mode: ring of pre-generated action buffers (fresh random actions every launch)

        def make_runner(mode, chunk):
            """Returns (run(k): enqueue k launches, env-steps per env per launch)."""
            if mode == "rollout":
                shp = (lambda d: (chunk, d, n)) if args.layout == "soa" else (lambda d: (chunk, n, d))
                bufs = {
                    "actions": torch.empty(shp(nA), dtype=torch.float32, device=dev),
                    "obs": torch.empty(shp(nS), dtype=torch.float32, device=dev),
                    "rew": torch.empty((chunk, n), dtype=torch.float32, device=dev),
                    "done": torch.empty((chunk, n), dtype=torch.uint8, device=dev),
                }

                def run(k):
                    for _ in range(k):
                        env.rollout(chunk, mode=args.actions, layout=args.layout, fused=True,
                                    want=("actions", "obs", "rew", "done"), device_out=True, out=bufs)
                return run, chunk
            # step mode: one launch per env-step; the launch loop runs inside librmav (rmav_rollout with
            # fused=0), so Python/ctypes overhead is paid once per RING launches.  State is updated in
            # place (obs == state, as SURVEY.md 8d defines the 101 algorithmic bytes); reward and done
            # are written per step.
            ring = torch.empty((RING, nA, n), dtype=torch.float32, device=dev).uniform_(lo, hi)
            bufs = {"rew": torch.empty((RING, n), dtype=torch.float32, device=dev),
                    "done": torch.empty((RING, n), dtype=torch.uint8, device=dev)}

            def run(k):
                while k > 0:
                    m = min(k, RING)
                    env.rollout(m, mode="buffer", actions=ring[:m], layout="soa", fused=False, want=("rew", "done"),
                                out={"rew": bufs["rew"][:m], "done": bufs["done"][:m]})
                    k -= m
            return run, 1

        def measure(mode, chunk, K, W):
            run, per_launch = make_runner(mode, chunk)
            run(W)
            if use_dist:   # warm the collective too (RCCL connects rings lazily on the first call of each kind)
                eb = env.episode_buffers(device_out=True)
                all_gather_episode_stats(eb["last_return"], eb["last_length"], n_total)
            stream.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            run(K)
            e1.record(stream)
            gathered = None
            if use_dist:  # the path's one exchange: per-rollout all-gather of episode statistics (RCCL / xGMI)
                eb = env.episode_buffers(device_out=True)
                gathered = all_gather_episode_stats(eb["last_return"], eb["last_length"], n_total)
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            wall = time.perf_counter() - t0
            kernel_ms = e0.elapsed_time(e1) / K  # HIP events on the launch stream
            if use_dist:
                w = torch.tensor([wall], dtype=torch.float64, device=dev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall = float(w.item())
                assert gathered[0].numel() == n_total
            return wall, kernel_ms, per_launch

        secondary = None
        if not args.no_secondary:   # the other mode, for the record (every rank runs it: it contains collectives)
            other = "step" if args.mode == "rollout" else "rollout"
            K2 = 4000 if other == "step" else 500
            w2, k2, pl2 = measure(other, args.chunk, K2, 200 if other == "step" else 50)
            secondary = {"mode": other, "launches": K2, "env_steps_per_launch": n * pl2,
                         "value": n_total * pl2 * K2 / w2, "unit": "env-steps/s", "ms_per_launch_wall": 1e3 * w2 / K2,
                         "ms_per_launch_hip_events": k2,
                         "roofline_frac": algo_bytes * n * pl2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # the headline measurement: W untimed launches, then exactly K timed ones.  (The first ~5 ms of GPU work
        # after idle run ~15 % slower on these boxes - 50.9 vs 43.1 us per launch measured with K=100 / K=1000 -
        # so the defaults are sized well past that.)
        wall, kernel_ms, per_launch = measure(args.mode, args.chunk, args.steps, args.warmup)
        totals = env.episode_totals()
        if use_dist:
            totals = all_reduce_totals(totals, device=dev)

    value = n_total * per_launch * args.steps / wall
    achieved = algo_bytes * n * per_launch / (kernel_ms * 1e-3) / 1e9  # GB/s, one GPU's dominant kernel
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes/launch (see profiles/README.md)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"{kind}:{args.mode}:{args.chunk if args.mode == 'rollout' else 1}:{n}")
        except Exception:
            traffic = None

    if rank == 0:
        line = {
            "metric": "env-steps/sec (batched quadrotor3d-v0)" if kind == "quad3d" else f"env-steps/sec (batched {kind})",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if kind in ("quad2d", "quad3d") else "f64 arithmetic on f32 storage",
            "data": "synthetic",
            "config": {
                "workload": (f"{ {'quad3d': 'quadrotor3d-v0', 'quad3d_sl': 'quadrotor3d-slungload-v0', 'quad2d': 'quadrotor2d-v0', 'quad2d_sl': 'quadrotor2d-slungload-v0', 'reinmav': 'reinmav-v0'}[kind]}"
                             f", {n} envs per GPU, random actions U[{lo:g},{hi:g})^{nA}, auto-reset, episode tracking; "
                             + (f"one step = one fused rollout launch = {per_launch} env-steps per env, in-kernel action source '{args.actions}', "
                                "trajectory (actions, obs, reward, done) written to HBM" if args.mode == "rollout"
                                else "one step = one launch = 1 env-step per env, actions read from a device buffer, "
                                     "obs/reward/done written")),
                "envs_per_gpu": n,
                "env_steps_per_launch_per_env": per_launch,
                "mode": args.mode,
                "trajectory_layout": args.layout if args.mode == "rollout" else "soa",
                "parallelism": f"env-shard x{world} (global env ids; one RCCL all-gather of episode stats per timed region)"
                if world > 1 else "single GPU",
                "finished_episodes": totals["episodes"],
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # the physical side of the same launch: PMC bytes / measured duration (the fused kernel keeps the state
                # in registers, so it moves fewer bytes than the algorithmic definition counts and `frac` can exceed 1)
                "traffic_achieved": (traffic / (kernel_ms * 1e-3) / 1e9) if traffic else None,
                "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_env_step": algo_bytes,
                "env_steps_per_launch": n * per_launch,
                "launch_ms_hip_events": kernel_ms,
            },
        }
        if secondary:
            line["other_mode"] = secondary
        if world == 1 and args.cpu_seconds > 0 and kind != "reinmav":
            line["cpu_baseline"] = cpu_baseline(kind, n, args.chunk, lo, hi, args.cpu_seconds, threads=1)
            ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            if ncpu > 1:
                # the same port with OpenMP over envs.  Containers often expose more logical CPUs than their
                # quota lets them use, so a few thread counts are tried briefly and the best one is reported
                # (cores = the threads actually used for that number).
                best = None
                for k in sorted({c for c in (4, 16, 64, ncpu) if c <= ncpu}):
                    r = cpu_baseline(kind, n, args.chunk, lo, hi, min(1.5, args.cpu_seconds), threads=k)
                    if best is None or r["value"] > best["value"]:
                        best = r
                line["cpu_baseline_multithread"] = best
            # interpreter-bound stand-in for the reference's own Python (which cannot travel to this box):
            # a per-env NumPy restatement in the reference's style (oracle/numpy_ref.py), ~2 s sample
            try:
                import numpy_ref

                v, k, el = numpy_ref.time_steps(min(2.0, args.cpu_seconds))
                line["cpu_baseline_python"] = {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                               "sample": f"per-env NumPy restatement of Quadrotor3D.step, {k} env-steps, "
                                                         f"{el:.1f} s, 1 process (the reference itself ran at 16.1 k "
                                                         "env-steps/s in the authoring container, BASELINE.md section 2)"}
            except Exception as e:  # pragma: no cover
                line["cpu_baseline_python"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    env.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
