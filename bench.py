#!/usr/bin/env python3
"""bench.py - env-steps/sec of the batched quadrotor3d-v0 hot path on N MI355X (one process per GPU).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it is launched under
``python -m torch.distributed.run --nproc-per-node N``.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs): quadrotor3d-v0, random actions T~U[0,10), w~U[0,10)^3 drawn in-kernel from the
counter RNG, auto-reset on done, episode tracking on.
  N = 1 : configs[1] (C2) - 65 536 envs on the GPU.
  N > 1 : the same 65 536 envs on EVERY GPU, sharded by GLOBAL env id - weak scaling in the strict sense (per-GPU work fixed as N
          grows; through round 3 the default for N > 1 was 131 072 per GPU, which compared a different per-GPU workload with N = 1's).
          configs[2] (C3: 1 048 576 envs over 8 GPUs) is measured beside it in the same run: ``other_modes.c3`` = 131 072 envs on
          every rank (N = 8: exactly C3), barrier-bracketed, slowest rank counts; ``--gpus 8 --envs-per-gpu 131072`` makes it the
          headline instead, and its per-GPU kernel is also the ``c3_shard`` leg of every N = 1 line.
One bench "step" = ONE launch of the hot-path kernel over the rank's whole shard:

  --mode rollout (default): the fused rollout kernel advances every env ``--chunk`` (64) env-steps with the
      state held in registers and writes the full trajectory (actions, obs, reward, done per env-step) to HBM -
      the unit an RL learner consumes.  Every launch writes the NEXT buffer set of a ring (>= 5 sets, > 1.5 GB:
      what a learner double-buffering real rollouts does), so no store is absorbed by rewriting lines that
      still sit in the 256 MiB Infinity Cache.  ``--in-place`` rewrites one set every launch instead.
  --mode step: the same kernel at chunk = 1, one launch per env-step: actions read from a device buffer
      (what a policy would have written), state updated in place (obs == state), reward/done written.

value = (envs on all ranks) * chunk * K / max-over-ranks wall time of the K timed launches (inputs already
resident in HBM; barrier + synchronize on both sides; the W warm-up launches are preceded by ``--prewarm-ms`` (40 ms) of
the same untimed launches, because the first ~5 ms of GPU work after idle run ~15 % slow on these boxes and a short
--steps / --warmup would otherwise measure the clock ramp).  For N > 1 every rollout launch is followed by the one
collective the path has: the RCCL all-gather of per-env episode returns/lengths (packed by one small launch,
gathered on a second stream so that it overlaps the next rollout).

roofline (dominant kernel = the timed launch; duration from HIP events on the launch stream):
  rollout: bytes the fused kernel must move per launch = N * (chunk * (4 (nS + nA + 1) + 1) + 8 nS + 24)
           (trajectory out: obs + actions + reward f32, done u8; per launch: state in/out, episode accumulators
           in/out, steps_beyond_done + reset counter in) - SURVEY 8d's "fused-rollout variant".  The state
           stays in registers between steps, so the 101 B/env-step of the single-step definition would count
           80 B that never move; that figure is reported beside it as ``algorithmic_equiv_frac``.
  step:    SURVEY 8d's 4 (2 nS + nA + 1) + 1 = 101 B per env-step.
cpu_baseline: the C oracle (oracle/, a port of the reference's NumPy step) timed on host cores over a bounded
sample of the same workload.

Output: the LAST stdout line is ONE compact JSON object (< 4 KB: the contract keys, ``roofline``, ``cpu_baseline``, and
``legs`` = one short row per secondary measurement).  Everything longer - workload descriptions, byte definitions, every
secondary leg in full, device state, the CPU thread scan, calibration - goes to ``--detail`` (default
``gpurun_out/bench_detail.json``), named by the line's ``detail`` key; nothing else is printed to stdout, and stderr
stays quiet unless something fails.  (Round 4's single 22 KB line could not be recovered from the driver's 8 KB tail.)
Default ``--secondary`` = the per-step kernel at three batch sizes, BASELINE configs[2]'s per-GPU shard (plain and chunk-major
trajectories), configs[3] and configs[4]'s per-GPU shard with the policy in-kernel (~1 s each), and the sustained stretch (11 s of the headline launches, so that an outside GPU-busy
sampler cannot miss it); with the 10 s + 3 s CPU baselines the default run takes ~30 s.  ``--secondary all`` adds the
policy-in-kernel rollouts, ReinmavEnv, the gym-shaped env and the VecEnv (~25 s more).
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
ENV_ID = {"quad3d": "quadrotor3d-v0", "quad3d_sl": "quadrotor3d-slungload-v0", "quad2d": "quadrotor2d-v0",
          "quad2d_sl": "quadrotor2d-slungload-v0", "reinmav": "reinmav-v0"}


def _omp_set_threads(k: int):
    import ctypes

    for name in ("libgomp.so.1", "libomp.so"):
        try:
            ctypes.CDLL(name).omp_set_num_threads(int(k))
            return True
        except Exception:
            continue
    return False


def host_cpus():
    """What this process may use of the box's CPU: logical CPUs present, the affinity mask, physical cores (distinct
    (package, core) pairs of the CPUs in the mask), and the cgroup CPU quota (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us`` /
    ``cpu.cfs_period_us``; None = unlimited).  ``usable`` = min(affinity, quota rounded up) - the thread count the OpenMP
    baseline runs at; ``usable_physical`` = the same bounded by the physical cores (SMT siblings share one FPU)."""
    import math

    present = os.cpu_count() or 1
    aff = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(present))
    cores = set()
    for c in aff:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except Exception:
            cores.add(("?", str(c)))
    quota, src = None, None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        src = "/sys/fs/cgroup/cpu.max = " + q + " " + per
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            src = f"/sys/fs/cgroup/cpu/cpu.cfs_quota_us = {q}, cpu.cfs_period_us = {per}"
            quota = None if q <= 0 else q / per
        except Exception:
            src = "no cgroup cpu controller file readable"
    usable = len(aff) if quota is None else max(1, min(len(aff), int(math.ceil(quota))))
    model = None
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    return {"logical_present": present, "affinity": len(aff), "physical_cores_in_affinity": len(cores), "cgroup_quota_cpus": quota,
            "cgroup_source": src, "usable": usable, "usable_physical": min(usable, len(cores)), "model": model}


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


class DeviceSampler:
    """Socket power and shader clock of THIS rank's GPU while the headline workload runs (hwmon / sysfs of the device's
    PCI function, read every 10 ms from a thread).  Context for ``roofline``: the fused rollout runs the package at its
    power cap (1.3-1.4 kW of 1.4 kW; a fill kernel of the same bytes draws 0.93 kW), and how a box's power management
    reacts is the +-10 % spread between boxes (profiles/r02/power.md).  Every field is None when sysfs is unreadable."""

    def __init__(self, device_index: int):
        import glob
        import threading

        import torch

        self.samples, self._stop, self.cap = [], False, None
        self._pow = self._clk = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            d = f"/sys/bus/pci/devices/{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            hw = sorted(glob.glob(d + "/hwmon/hwmon*"))[0]
            self._pow = next(f for f in (hw + "/power1_input", hw + "/power1_average") if os.path.exists(f))
            self._clk = hw + "/freq1_input"
            self.cap = int(open(hw + "/power1_cap").read()) / 1e6
        except Exception:
            pass
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop:
            try:
                self.samples.append((int(open(self._pow).read()) / 1e6, int(open(self._clk).read()) / 1e6))
            except Exception:
                pass
            time.sleep(0.01)

    def __enter__(self):
        if self._pow:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._pow:
            self._thread.join()

    def summary(self):
        busy = [x for x in self.samples if x[0] > 0.5 * max(p for p, _ in self.samples)] if self.samples else []
        if not busy:
            return {"power_w_mean": None, "power_w_max": None, "power_cap_w": self.cap, "sclk_mhz_mean": None, "samples": 0}
        return {"power_w_mean": round(sum(p for p, _ in busy) / len(busy), 1), "power_w_max": round(max(p for p, _ in busy), 1),
                "power_cap_w": self.cap, "sclk_mhz_mean": round(sum(c for _, c in busy) / len(busy)), "samples": len(busy),
                "note": "hwmon power1 / freq1 of this GPU every 10 ms over the prewarm + warm-up + timed launches of the headline workload; the power sensor is a moving average, so a run shorter than ~1 s under-reads (steady state: 1.32-1.40 kW, profiles/r02/power.md)"}


def fused_bytes_per_launch(n: int, chunk: int, nS: int, nA: int) -> int:
    """What one fused rollout launch must move (see the module docstring)."""
    return n * (chunk * (4 * (nS + nA + 1) + 1) + 8 * nS + 24)


def lookup_traffic(key: str):
    """HBM-side bytes per launch of this exact workload from the round's rocprofv3 --pmc passes (profiles/traffic.json, written
    by tools/parse_rocprof.py from separate FETCH_SIZE / WRITE_SIZE runs of the same command line; counters cannot be
    collected inside this process).  -> (bytes | None, source | None); a source whose file is not in the repo is dropped."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        ent = json.load(open(tpath)).get(key)
        if ent:
            src = ent.get("source") or ""
            if not os.path.exists(os.path.join(ROOT, src.split(" ")[0])):
                src = "profiles/traffic.json (summary file not in the repo)"
            return float(ent["bytes"]), src
    except Exception:
        pass
    return None, None


def roofline_obj(bytes_launch: float, launch_ms: float, traffic, traffic_src, definition: str, extra=None):
    achieved = bytes_launch / (launch_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "traffic_source": traffic_src,
         "traffic_over_needed": (traffic / bytes_launch) if traffic else None,
         "bytes_per_launch": bytes_launch, "bytes_definition": definition, "launch_ms_hip_events": launch_ms}
    if extra:
        r.update(extra)
    return r


def rollout_leg(g, torch, dev, kind: str, n: int, chunk: int, K: int, W: int, label: str, tune=None, per_env_params=False,
                cpu_seconds: float = 0.0, env_id_base: int = 0, before_timed=None, chunk_major=False):
    """One more single-GPU BASELINE config as its own short measurement: fused random-action rollouts of `kind` over `n` envs
    into a cold ring of trajectory buffer sets, timed with HIP events on the launch stream (same method as the headline).
    per_env_params: every env gets its own mass / load mass / tether length (rmav_set_env_param: SURVEY 8f-4, the constants
    the reference hard-codes in quadrotor3d_slungload.py:45-59), +-10 % around the defaults."""
    A = g._abi
    K_ = A.KIND_BY_NAME[kind]
    nS, nA = A.STATE_DIM[K_], A.ACTION_DIM[K_]
    per_set = n * chunk * (4 * (nS + nA + 1) + 1)
    R = max(5, -(-int(1.5e9) // per_set))
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor(kind, n, device=dev.index, seed=0, env_id_base=env_id_base, auto_reset=True, track_episodes=True)
        if tune:
            env.set_tuning(**tune)
        if per_env_params:
            gen = torch.Generator(device="cpu").manual_seed(1)
            pr = env.params
            for name, base in (("mass", pr.mass), ("load_mass", pr.load_mass), ("tether_length", pr.tether_length)):
                env.set_env_param(name, (base * (0.9 + 0.2 * torch.rand(n, generator=gen))).to(torch.float32).numpy())
        ce = int(env._lib.rmav_chunk_envs(env._h)) if chunk_major else n     # chunk-major trajectories: rmav_rollout_chunked (include/rmav.h)
        nc = -(-n // ce)
        shp = (lambda d: (nc, chunk, d, ce)) if chunk_major else (lambda d: (chunk, d, n))
        ring = [{"actions": torch.zeros(shp(nA), dtype=torch.float32, device=dev),
                 "obs": torch.zeros(shp(nS), dtype=torch.float32, device=dev),
                 "rew": torch.zeros((nc, chunk, ce) if chunk_major else (chunk, n), dtype=torch.float32, device=dev),
                 "done": torch.zeros((nc, chunk, ce) if chunk_major else (chunk, n), dtype=torch.uint8, device=dev)} for _ in range(R)]
        it = 0
        for phase, count in (("warm", W), ("timed", K)):
            if phase == "timed":
                stream.synchronize()
                if before_timed:   # (multi-GPU leg: the ranks' barrier)
                    before_timed()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(stream)
            for _ in range(count):
                if chunk_major:
                    env.rollout_chunked(chunk, mode="random", chunk=ce, want=("actions", "obs", "rew", "done"), out=ring[it % R])
                else:
                    env.rollout(chunk, mode="random", layout="soa", fused=True, want=("actions", "obs", "rew", "done"),
                                device_out=True, out=ring[it % R])
                it += 1
        e1.record(stream)
        stream.synchronize()
        wall = time.perf_counter() - t0
        ms = e0.elapsed_time(e1) / K
        fin = env.episode_totals()["episodes"]
        env.close()
    del ring
    torch.cuda.empty_cache()
    b = fused_bytes_per_launch(n, chunk, nS, nA) + (12 * n if per_env_params else 0)
    tr, src = lookup_traffic(f"{kind}:rollout:{chunk}:{n}:ring:random:" + ("chunked" if chunk_major else "soa") + (":pe" if per_env_params else ""))
    out = {"workload": f"{label}: {ENV_ID[kind]}, {n} envs, random actions, auto-reset, episode tracking; {chunk}-step fused launches "
                       f"into a ring of {R} trajectory buffer sets ({R * per_set / 1e9:.2f} GB: cold stores)" +
                       (f"; CHUNK-MAJOR trajectory arrays [{nc}][{chunk}][dim][{ce}], one launch per chunk (rmav_rollout_chunked)" if chunk_major else "") +
                       ("; per-env mass, load mass and tether length (3 x 4 B per env and launch read, constants re-derived per lane)" if per_env_params else ""),
           "launches": K, "warmup": W, "value": n * chunk * K / wall, "unit": "env-steps/s", "ms_per_launch_wall": 1e3 * wall / K,
           "finished_episodes": fin,
           "roofline": roofline_obj(b, ms, tr, src, f"{n} envs x ({chunk} env-steps x {4 * (nS + nA + 1) + 1} B trajectory out + "
                                                    f"{8 * nS + 24 + (12 if per_env_params else 0)} B state / episode bookkeeping per launch)")}
    if cpu_seconds > 0:   # the same workload on the host cores (C oracle, one thread): the side-by-side of BASELINE.md section 4
        p = g._abi.default_params(K_)
        out["cpu_baseline"] = cpu_baseline(kind, min(n, 65536), chunk, float(p.act_lo), float(p.act_hi), cpu_seconds, threads=1)
    return out


def step_leg(g, torch, dev, kind: str, n: int, K: int, W: int, tune=None):
    """The per-step kernel (``rmav_step``'s ``k_step``: what baselines' one-``step()``-per-call loop issues, gym_reinmav/run.py:89 ->
    quadrotor3d.py:81-124) as its own measurement at ``n`` envs: one launch per env-step (the loop runs inside
    ``rmav_rollout(fused = 0)``), actions read from a ring of device buffers, state updated in place, reward + done written,
    auto-reset + episode tracking on; HIP events on the launch stream.  At 65 536 envs a launch lasts about one launch latency;
    262 144 and 1 048 576 envs are where the kernel, not the launch, is what is timed."""
    A = g._abi
    K_ = A.KIND_BY_NAME[kind]
    nA = A.ACTION_DIM[K_]
    algo = A.lib().rmav_algorithmic_bytes(K_)
    p = A.default_params(K_)
    ring_n = max(4, min(64, int(256e6) // (4 * nA * n)))
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor(kind, n, device=dev.index, seed=0, auto_reset=True, track_episodes=True)
        if tune:
            env.set_tuning(**tune)
        ring = torch.empty((ring_n, nA, n), dtype=torch.float32, device=dev).uniform_(float(p.act_lo), float(p.act_hi))
        bufs = {"rew": torch.empty((ring_n, n), dtype=torch.float32, device=dev), "done": torch.empty((ring_n, n), dtype=torch.uint8, device=dev)}

        def run(k):
            while k > 0:
                m = min(k, ring_n)
                env.rollout(m, mode="buffer", actions=ring[:m], layout="soa", fused=False, want=("rew", "done"),
                            out={"rew": bufs["rew"][:m], "done": bufs["done"][:m]})
                k -= m
        run(W)
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        run(K)
        e1.record(stream)
        stream.synchronize()
        wall = time.perf_counter() - t0
        ms = e0.elapsed_time(e1) / K
        env.close()
    del ring, bufs
    torch.cuda.empty_cache()
    tr, src = lookup_traffic(f"{kind}:step:1:{n}:inplace:random:soa" + ("".join(f":{k}={v}" for k, v in (tune or {}).items())))
    return {"workload": f"one launch of k_step per env-step over {n} envs of {ENV_ID[kind]} (the loop runs inside rmav_rollout(fused = 0)): actions read "
                        f"from a ring of {ring_n} device buffers, state updated in place, reward + done written, auto-reset + episode tracking on",
            "launches": K, "warmup": W, "env_steps_per_launch": n, "value": n * K / wall, "unit": "env-steps/s", "ms_per_launch_wall": 1e3 * wall / K,
            "roofline": roofline_obj(algo * n, ms, tr, src,
                                     f"{n} envs x {algo} B (SURVEY 8d: state in/out, action in, reward + done out); the episode bookkeeping the "
                                     "VecEnv contract adds (running return / length in/out, steps_beyond_done + reset counter in) is traffic "
                                     "beyond this definition",
                                     {"launch_floor_note": "an EMPTY 65 536-thread kernel chain runs 2.8-2.9 us per launch on this GPU "
                                                           "(tools/micro/launch_floor.hip): a bound on any one-launch-per-step kernel at small batches"})}


def leg_rows(other: dict) -> dict:
    """One short row per secondary leg for the compact line (the full objects go to the detail file)."""
    legs = {}
    for k, v in other.items():
        if not isinstance(v, dict):
            legs[k] = str(v)[:120]
            continue
        row = {}
        if "value" in v:
            row["value"] = float(f"{v['value']:.4g}")
        r_ = v.get("roofline")
        if isinstance(r_, dict) and "frac" in r_:
            row["frac"] = round(r_["frac"], 4)
            row["bound"] = r_.get("bound", "hbm")
            ms_ = r_.get("launch_ms_hip_events", v.get("ms_per_launch_hip_events"))
            if ms_:
                row["us"] = round(1e3 * ms_, 3)
            if r_.get("traffic_over_needed"):
                row["traffic_over_needed"] = round(r_["traffic_over_needed"], 3)
        elif "roofline_frac" in v:
            row["frac"] = round(v["roofline_frac"], 4)
            row["us"] = round(1e3 * v.get("ms_per_launch_hip_events", 0.0), 3)
        elif "roofline_frac_slowest_rank" in v:
            row["frac"] = round(v["roofline_frac_slowest_rank"], 4)
            row["envs_total"] = v.get("envs_total")
            if "chunk_major" in v:
                row["chunk_major"] = {"value": float(f"{v['chunk_major']['value']:.4g}"), "frac": round(v["chunk_major"]["roofline_frac_slowest_rank"], 4)}
        if k == "policy_rollout":
            row = {a: {"value": float(f"{x['kernel_env_steps_per_s']:.4g}"), "bound": x["bound"], "frac": round(x["bound_frac"], 3)}
                   for a, x in v.items() if isinstance(x, dict) and "bound" in x}
        if k == "gym1":
            row = {"us_control_plus_step": round(v["us_per_iteration_control_plus_step"], 2), "reference_us_per_step": v["reference_us_per_step"]}
        if k == "vecenv":
            row = {kk: round(vv["us_per_step"], 2) for kk, vv in v.items() if isinstance(vv, dict)}
        legs[k] = row
    return legs


def compact_text(line: dict) -> str:
    """The final stdout line: < 4 KB whatever the legs (the driver keeps an 8 KB tail; round 4's 22 KB line was lost to it) - optional
    rows are dropped before that could happen."""
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > 4000:
        line = {k: v for k, v in line.items() if k not in ("legs", "cpu_mt")}
        text = json.dumps(line, separators=(",", ":"))
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed launches (SURVEY 8d asks for >= 1000 after >= 100 warm-up; 2000 launches ~ 0.1-0.2 s)")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--kind", default="quad3d", choices=["quad2d", "quad2d_sl", "quad3d", "quad3d_sl", "reinmav"])
    ap.add_argument("--actions", default="random", choices=["random", "controller", "buffer"],
                    help="action source of the fused rollout (controller = the reference's built-in / geometric controller; "
                         "buffer = caller-provided actions, read from a ring of [chunk][nA][N] device buffers)")
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help="default: 65536 per GPU (BASELINE C2 on one GPU; the same shard on every GPU of a multi-GPU run); 131072 = C3's shard")
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"])
    ap.add_argument("--chunk", type=int, default=64,
                    help="env-steps per launch in rollout mode (64 amortises the ~4.5 us fixed cost of a launch; "
                         "see profiles/*/sweep_kinds_sizes.md for 8..128)")
    ap.add_argument("--layout", default="auto", choices=["auto", "soa", "aos", "chunked"],
                    help="trajectory layout in rollout mode; auto = the library's recommendation (rmav_chunk_envs): chunk-major [C][T][dim][65536] "
                         "for quadrotor3d beyond 65 536 envs (a learner flattens the samples anyway), plain feature-major [T][dim][N] otherwise")
    ap.add_argument("--in-place", action="store_true", help="rollout mode: rewrite ONE trajectory buffer set (cache-assisted)")
    ap.add_argument("--ring", type=int, default=0, help="rollout mode: number of trajectory buffer sets (0 = >= 5 and > 1.5 GB)")
    ap.add_argument("--prewarm-ms", type=float, default=40.0,
                    help="untimed stretch of the headline's launches before the --warmup launches (GPU clock ramp; 0 = none)")
    ap.add_argument("--exchange-every", type=int, default=1,
                    help="multi-rank runs: post the episode-stats all-gather after every k-th rollout launch (default 1 = every launch)")
    ap.add_argument("--action-ring", type=int, default=64, help="step mode: number of pre-generated action buffers")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other modes' short measurements")
    ap.add_argument("--tune", default="", help="comma list of key=value overrides of the launch heuristics (rmav_set_tuning), "
                                               "e.g. split=0,store_policy=2")
    ap.add_argument("--sustained-seconds", type=float, default=11.0,
                    help="length of the `sustained` leg: the headline launches back to back, long enough for an outside GPU-busy sampler "
                         "with a 5 s period to see at least two busy samples")
    ap.add_argument("--secondary", default="step,sustained,cpu_mt,c3_shard,c4,c5",
                    help="comma list of the other measurements (single process only): step (k_step at 65 536 / 262 144 / 1 048 576 envs), "
                         "sustained, in_place, c3_shard, c4, c4_pe, reinmav, gym1, vecenv, policy, cpu_mt (OpenMP CPU baseline), cpu_py; "
                         "'all' = every one of them (~25 s more)")
    ap.add_argument("--detail", default=None,
                    help="file that receives the full record (every leg, descriptions, device state); default gpurun_out/bench_detail.json "
                         "(bench_detail_n<N>.json for N > 1); '-' = none")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import gym_reinmav_amd as g
    from gym_reinmav_amd.distributed import EpisodeStatsExchange, NativeStatsExchange, all_reduce_totals

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
    # several ranks on one GPU (the two-process test on a 1-GPU box) share device 0
    local_dev = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RMAV_BENCH_BACKEND", "nccl")   # "gloo": ranks sharing one GPU (RCCL refuses duplicates)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    torch.manual_seed(0)  # the step mode's action ring is filled by torch's generator
    kind = args.kind
    n = args.envs_per_gpu if args.envs_per_gpu else 65536
    n_total = n * world
    A = g._abi
    K_ = A.KIND_BY_NAME[kind]
    nS, nA = A.STATE_DIM[K_], A.ACTION_DIM[K_]
    algo_bytes = A.lib().rmav_algorithmic_bytes(K_)
    p = A.default_params(K_)
    lo, hi = float(p.act_lo), float(p.act_hi)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor(kind, n, device=local_dev, seed=0, env_id_base=rank * n, auto_reset=True,
                                 track_episodes=True)
        tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune.split(",") if kv}
        if tune:
            env.set_tuning(**tune)
        chunk_envs = int(A.lib().rmav_chunk_envs(env._h))
        if args.layout == "auto":   # what BatchedQuadrotor.rollout(layout="chunked") gives: one chunk = the plain layout
            args.layout = "chunked" if (chunk_envs < n and args.actions != "buffer" and args.mode == "rollout") else "soa"
        gloo = use_dist and dist.get_backend() == "gloo"
        exchange, exchange_kind, native_abandoned, comm_info = None, None, False, None
        if use_dist:
            # the collective behind the C ABI (RCCL from librmav's own stream, ~15 us of host time per post); every rank must
            # take the same path, so fall back together to the torch.distributed exchange if any rank cannot set it up
            ok = 0
            # (test hook: RMAV_BENCH_RCCL_LIB = a stand-in for librccl.so.1 that lets rank processes share ONE GPU - tests/stub_rccl -
            #  handed to the library with rmav_comm_use_library; the native exchange then runs under the gloo process group too)
            rccl_lib = os.environ.get("RMAV_BENCH_RCCL_LIB", "")
            if rccl_lib:
                A.check(A.lib().rmav_comm_use_library(rccl_lib.encode()))
            if (not gloo or rccl_lib) and os.environ.get("RMAV_BENCH_EXCHANGE", "native") == "native":
                try:   # communicator + one whole exchange under a watchdog: a rank that is stuck falls back with the others
                    exchange = NativeStatsExchange(env, n_total, connect_timeout_s=float(os.environ.get("RMAV_BENCH_CONNECT_TIMEOUT", "90")))
                    ok = 1
                except BaseException as e:  # pragma: no cover
                    print(f"[rank {rank}] native exchange unavailable: {e!r}", file=sys.stderr)
                    native_abandoned = isinstance(e, TimeoutError)
                flag = torch.tensor([ok], dtype=torch.int32, device="cpu" if gloo else dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            if ok:
                exchange_kind = "rmav_allgather_stats_post/_result (RCCL from librmav.so, own stream)"
                try:   # what the collective library itself says about the communicator (ncclCommCount / ncclCommUserRank)
                    comm_info = exchange.info()
                except Exception as e:  # pragma: no cover
                    comm_info = {"error": repr(e)}
            else:
                if exchange is not None:
                    exchange.close()
                exchange = EpisodeStatsExchange(n_total, "cpu" if gloo else dev)
                exchange_kind = "torch.distributed all_gather_into_tensor (" + dist.get_backend() + "), second stream"
        # the exchange takes host tensors only when it is torch.distributed's over gloo (ranks sharing a GPU without the stand-in)
        host_xchg = gloo and not isinstance(exchange, NativeStatsExchange)

        # step mode: ring of pre-generated action buffers (fresh random actions every launch).  64 buffers = 67 MB at
        # 65 536 envs: like the actions a policy kernel has just written, they are still on chip (L2 / Infinity Cache)
        # when the step launch reads them.  (512 buffers - every action read a cold HBM miss - cost +0.45 us per launch.)
        RING = args.action_ring

        def traj_bytes(chunk):
            return n * chunk * (4 * (nS + nA + 1) + 1)

        def ring_size(chunk, in_place):
            if in_place:
                return 1
            if args.ring > 0:
                return args.ring
            r = max(5, -(-int(1.5e9) // traj_bytes(chunk)))
            while r > 2 and r * traj_bytes(chunk) > 64e9:   # stay far below 288 GB for multi-million-env sweeps
                r -= 1
            return r

        def make_runner(mode, chunk, in_place=False):
            """Returns (run(k): enqueue k launches, env-steps per env per launch, buffer sets)."""
            if mode == "rollout":
                chunked = args.layout == "chunked"
                ce = min(chunk_envs, (n + 63) // 64 * 64)
                ncz = -(-n // ce)
                shp = ((lambda d: (ncz, chunk, d, ce)) if chunked else (lambda d: (chunk, d, n)) if args.layout == "soa" else (lambda d: (chunk, n, d)))
                shp1 = (ncz, chunk, ce) if chunked else (chunk, n)
                R = ring_size(chunk, in_place)
                # zero-filled at set-up so that every buffer set is mapped before the warm-up (a short --warmup would
                # otherwise first-touch part of the ring inside the timed region)
                ring = [{
                    "actions": torch.zeros(shp(nA), dtype=torch.float32, device=dev),
                    "obs": torch.zeros(shp(nS), dtype=torch.float32, device=dev),
                    "rew": torch.zeros(shp1, dtype=torch.float32, device=dev),
                    "done": torch.zeros(shp1, dtype=torch.uint8, device=dev),
                } for _ in range(R)]
                it = [0]
                if args.actions == "buffer":   # the ring's action buffers are the INPUT: filled once, read every launch
                    for b in ring:
                        b["actions"].uniform_(lo, hi)

                arm = exchange is not None and not host_xchg and os.environ.get("RMAV_BENCH_ARM", "1") == "1"

                def run(k):
                    for _ in range(k):
                        if arm and (it[0] + 1) % args.exchange_every == 0:   # this launch writes the exchange's snapshot itself
                            exchange.arm(env)
                        if args.actions == "buffer":
                            b = ring[it[0] % R]
                            env.rollout(chunk, mode="buffer", actions=b["actions"], layout=args.layout, fused=True,
                                        want=("obs", "rew", "done"), device_out=True, out={k: b[k] for k in ("obs", "rew", "done")})
                        else:
                            env.rollout(chunk, mode=args.actions, layout=args.layout, fused=True,
                                        want=("actions", "obs", "rew", "done"), device_out=True, out=ring[it[0] % R])
                        it[0] += 1
                        if exchange is not None and it[0] % args.exchange_every == 0:   # the path's one exchange, once per rollout
                            if host_xchg:
                                eb = env.episode_buffers()
                                exchange.post(torch.from_numpy(eb["last_return"]), torch.from_numpy(eb["last_length"]))
                            else:
                                exchange.post(env=env)
                return run, chunk, R
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
            return run, 1, 1

        prewarm_launches = [0]

        def measure(mode, chunk, K, W, in_place=False, prewarm_ms=0.0):
            run, per_launch, R = make_runner(mode, chunk, in_place)
            if prewarm_ms > 0:
                # The first ~5 ms of GPU work after idle run ~15 % slow on these boxes (clock ramp).  A short --steps /
                # --warmup (the driver runs 20 / 5) would sit entirely inside that ramp, so the headline is preceded by an
                # untimed stretch of the same launches; then come the W warm-up launches and the K timed ones.
                # (a launch COUNT derived from the problem size, not a wall-clock loop: every rank must issue the same
                # number of launches - each one posts a collective)
                est_us = (0.72 * per_launch if mode == "rollout" else 5.0) * max(1.0, n / 65536.0)
                prewarm_launches[0] = max(1, int(prewarm_ms * 1e3 / est_us))
                run(prewarm_launches[0])
                stream.synchronize()
            run(W)
            stream.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            run(K)
            e1.record(stream)
            gathered = exchange.result() if (exchange is not None and mode == "rollout") else None
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            wall = time.perf_counter() - t0
            kernel_ms = e0.elapsed_time(e1) / K  # HIP events on the launch stream
            if use_dist:
                w = torch.tensor([wall], dtype=torch.float64, device="cpu" if gloo else dev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall = float(w.item())
                if gathered is not None:
                    assert gathered[0].numel() == n_total and gathered[1].numel() == n_total
            return wall, kernel_ms, per_launch, R, gathered

        other = {}
        single = not use_dist and not args.no_secondary
        sec = set(args.secondary.split(",")) if single else set()
        all_legs = "all" in sec
        if "all" in sec:
            sec = {"in_place", "step", "c3_shard", "c4", "c4_pe", "reinmav", "gym1", "vecenv", "policy", "sustained", "cpu_mt", "cpu_py"}
        # the headline measurement: W untimed launches, then exactly K timed ones.  (The first ~5 ms of GPU work
        # after idle run ~15 % slower on these boxes, so the defaults are sized well past that.)
        with DeviceSampler(dev.index or 0) as sampler:
            wall, kernel_ms, per_launch, R, gathered = measure(args.mode, args.chunk, args.steps, args.warmup, args.in_place,
                                                               prewarm_ms=args.prewarm_ms)
        device_state = sampler.summary()
        # (round 6: the headline is measured FIRST, the long sustained stretch LAST - the other configs' legs used to run ahead of the
        #  headline and left the package at its power limit: the 20 timed launches of the driver's command came out 3 - 8 % below the
        #  sustained rate; and C4's fp64 kernel measured right after 11 s at the power limit loses 15 %)
        if "step" in sec or "in_place" in sec:
            if args.mode == "rollout" and "step" in sec:
                # the per-step kernel at the headline's batch, and where it is not launch-bound (262 144 / 1 048 576 envs)
                for n_s in sorted({n, 262144, 1048576}):
                    key = "step" if n_s == n else f"step_{n_s}"
                    other[key] = step_leg(g, torch, dev, kind, n_s, max(400, 4000 * 65536 // max(n_s, 65536)), 200, tune=tune)
            if args.mode == "rollout" and "in_place" in sec and not args.in_place:
                w2, k2, pl2, _, _ = measure("rollout", args.chunk, 500, 100, in_place=True)
                b2 = fused_bytes_per_launch(n, args.chunk, nS, nA)
                other["rollout_in_place"] = {
                    "note": "one trajectory buffer set rewritten every launch: at 65 536 envs its 256 MB sit in the "
                            "256 MiB Infinity Cache, so the stores are cache-assisted (round 1's headline)",
                    "launches": 500, "value": n * pl2 * 500 / w2, "unit": "env-steps/s", "ms_per_launch_hip_events": k2,
                    "roofline_frac": b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if single and kind == "quad3d" and args.mode == "rollout":
            try:   # the other single-GPU BASELINE configs, each a short leg with its own roofline object
                if "c3_shard" in sec and n != 131072:
                    other["c3_shard"] = rollout_leg(g, torch, dev, "quad3d", 131072, args.chunk, 600, 150, "BASELINE configs[2]'s per-GPU shard")
                    other["c3_shard_chunked"] = rollout_leg(g, torch, dev, "quad3d", 131072, args.chunk, 600, 150,
                                                            "BASELINE configs[2]'s per-GPU shard, chunk-major trajectories", chunk_major=True)
                if "c4" in sec:
                    other["c4"] = rollout_leg(g, torch, dev, "quad3d_sl", 262144, args.chunk, 300, 80, "BASELINE configs[3] (C4)",
                                              cpu_seconds=min(3.0, args.cpu_seconds) if all_legs else 0.0)   # (its own CPU side-by-side: --secondary all)
                if "c4_pe" in sec:
                    other["c4_per_env_params"] = rollout_leg(g, torch, dev, "quad3d_sl", 262144, args.chunk, 300, 80,
                                                             "BASELINE configs[3] (C4) with per-env constants", per_env_params=True)
                    if "c4" in other:
                        a_, b_ = other["c4"]["roofline"]["launch_ms_hip_events"], other["c4_per_env_params"]["roofline"]["launch_ms_hip_events"]
                        other["c4_per_env_params"]["vs_shared_constants"] = {"us_per_launch": 1e3 * b_, "us_per_launch_shared": 1e3 * a_,
                                                                             "extra_us": 1e3 * (b_ - a_), "extra_bytes_per_launch": 12 * 262144}
                if "reinmav" in sec:
                    other["reinmav"] = bench_reinmav(g, torch, dev, cpu_seconds=min(3.0, args.cpu_seconds))
            except Exception as e:  # pragma: no cover
                other["legs_error"] = repr(e)
        if single and "sustained" in sec and args.mode == "rollout":
            # --sustained-seconds (11 s) of the headline launches back to back (outside the K timed ones): a stretch long enough for
            # clocks and the package power limit to settle, and for an outside busy sampler with a 5 s period to see the GPU at work
            # in at least two samples - the K = 20 launches of the driver's default command are 0.9 ms of a ~30 s process (round 5's
            # 2.5 s stretch was missed by all four of the driver's samples)
            k_s = max(args.steps, int(args.sustained_seconds / max(1e-6, kernel_ms * 1e-3)))
            with DeviceSampler(dev.index or 0) as sampler2:
                w_s, kms_s, pl_s, _, _ = measure(args.mode, args.chunk, k_s, 0, args.in_place, prewarm_ms=0.0)
            other["sustained"] = {"workload": f"the headline launches, back to back for >= {args.sustained_seconds:g} s", "seconds": w_s, "launches": k_s,
                                  "value": n_total * pl_s * k_s / w_s, "unit": "env-steps/s", "ms_per_launch_hip_events": kms_s,
                                  "roofline_frac": fused_bytes_per_launch(n, pl_s, nS, nA) / (kms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "device_state": sampler2.summary()}
        totals = env.episode_totals()
        exchange_check = None
        if use_dist and exchange is not None and args.mode == "rollout":
            # outside the timed region: one more exchange of the final per-env statistics, checked element by element
            # against a plain torch.distributed all-gather of the same buffers (global env order = rank order here)
            during = [t.clone() for t in gathered] if gathered is not None else None   # the timed region's last exchange
            eb = env.episode_buffers(device_out=not gloo)
            mine = [torch.as_tensor(eb["last_return"]), torch.as_tensor(eb["last_length"])]
            if host_xchg:
                exchange.post(mine[0], mine[1])
            else:
                exchange.post(env=env)
            got = [t.clone() for t in exchange.result()]
            torch.cuda.synchronize()
            ref = [torch.empty(n_total, dtype=t.dtype, device=t.device) for t in mine]
            for r_, m_ in zip(ref, mine):
                dist.all_gather_into_tensor(r_, m_.contiguous())
            same = lambda g_: bool(torch.equal(g_[0].to(ref[0].device).view(torch.int32), ref[0].view(torch.int32))  # noqa: E731
                                   and torch.equal(g_[1].to(ref[1].device), ref[1]))
            exchange_check = same(got)
            if during is not None and args.exchange_every == 1:   # posted after the last launch: the same statistics
                exchange_check = exchange_check and same(during)
            okf = torch.tensor([int(exchange_check)], dtype=torch.int32, device="cpu" if gloo else dev)
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)   # rank 0 reports for every rank
            if not exchange_check:
                print(f"[rank {rank}] the gathered episode statistics differ from torch.distributed's all-gather", file=sys.stderr)
            exchange_check = bool(okf.item())
        if use_dist:
            totals = all_reduce_totals(totals, device="cpu" if gloo else dev)
        if use_dist and kind == "quad3d" and args.mode == "rollout" and n != 131072 and not args.no_secondary:
            # BASELINE configs[2]'s shape beside the weak-scaling series: 131 072 envs on EVERY rank (world = 8: exactly C3's 1 048 576),
            # sharded by global env id, no exchange in this leg; barrier on both sides, slowest rank counts
            try:
                leg = rollout_leg(g, torch, dev, "quad3d", 131072, args.chunk, 300, 80, f"BASELINE configs[2]'s shard on each of {world} ranks",
                                  env_id_base=rank * 131072, before_timed=dist.barrier)
                tmax = torch.tensor([leg["ms_per_launch_wall"], leg["roofline"]["launch_ms_hip_events"]], dtype=torch.float64, device="cpu" if gloo else dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                other["c3"] = {"workload": leg["workload"], "envs_total": 131072 * world, "n_gpus": world, "launches": 300, "warmup": 80,
                               "ms_per_launch_wall_max_over_ranks": float(tmax[0]), "launch_ms_hip_events_max_over_ranks": float(tmax[1]),
                               "value": 131072 * world * args.chunk / (float(tmax[0]) * 1e-3), "unit": "env-steps/s",
                               "roofline_frac_slowest_rank": leg["roofline"]["bytes_per_launch"] / (float(tmax[1]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "is_baseline_config_2": world == 8}
                # the same shard with chunk-major trajectory arrays (rmav_rollout_chunked: one 65 536-env launch per chunk)
                leg2 = rollout_leg(g, torch, dev, "quad3d", 131072, args.chunk, 300, 80, f"BASELINE configs[2]'s shard on each of {world} ranks, chunk-major",
                                   env_id_base=rank * 131072, before_timed=dist.barrier, chunk_major=True)
                t2 = torch.tensor([leg2["ms_per_launch_wall"], leg2["roofline"]["launch_ms_hip_events"]], dtype=torch.float64, device="cpu" if gloo else dev)
                dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                other["c3"]["chunk_major"] = {"value": 131072 * world * args.chunk / (float(t2[0]) * 1e-3), "launch_ms_hip_events_max_over_ranks": float(t2[1]),
                                              "roofline_frac_slowest_rank": leg2["roofline"]["bytes_per_launch"] / (float(t2[1]) * 1e-3) / 1e9 / HBM_PEAK_GBS}
            except Exception as e:  # pragma: no cover
                other["c3_error"] = repr(e)
        if gathered is not None:
            gathered_finished = int((gathered[1] > 0).sum().item())
        else:   # single process: the same statistic from the local per-env buffers
            gathered_finished = int((env.episode_buffers()["last_length"] > 0).sum())

    if single:
        try:
            if "gym1" in sec:
                other["gym1"] = bench_gym1()
            if "vecenv" in sec:
                other["vecenv"] = bench_vecenv(dev, n)
            if "policy" in sec and kind != "reinmav":
                other["policy_rollout"] = bench_policy(dev, kind, n)
            elif "c5" in sec and kind == "quad3d":   # configs[4]'s per-GPU shard with the two default actors only (~1 s): the driver's line carries C5 too
                other["policy_rollout"] = bench_policy(dev, kind, 65536, actors=("f16_mfma", "f16_shared"))
                if args.cpu_seconds > 0:
                    other["policy_rollout"]["cpu_baseline"] = cpu_policy_baseline(kind, min(3.0, args.cpu_seconds))
        except Exception as e:  # pragma: no cover - never lose the headline line to a secondary leg
            other["error"] = repr(e)

    value = n_total * per_launch * args.steps / wall
    if args.mode == "rollout":
        bytes_launch = fused_bytes_per_launch(n, per_launch, nS, nA)
        bytes_def = (f"{n} envs x ({per_launch} env-steps x {4 * (nS + nA + 1) + 1} B trajectory out + {8 * nS + 24} B "
                     "state / episode bookkeeping per launch)")
    else:
        bytes_launch = algo_bytes * n
        bytes_def = f"{n} envs x {algo_bytes} B (SURVEY 8d: state in/out, action in, reward + done out)"
    achieved = bytes_launch / (kernel_ms * 1e-3) / 1e9  # GB/s, one GPU's dominant kernel
    tkey = f"{kind}:{args.mode}:{per_launch}:{n}:{'inplace' if (args.in_place or args.mode == 'step') else 'ring'}:{args.actions}:{args.layout}" + (f":{args.tune}" if args.tune else "")
    traffic, traffic_src = lookup_traffic(tkey)   # rocprofv3 --pmc bytes per launch of this very command line

    if rank == 0:
        cfg_name = ("BASELINE configs[1] (C2)" if (world == 1 and n == 65536 and kind == "quad3d") else
                    f"BASELINE configs[1]'s 65 536 envs on each of {world} GPUs (weak scaling)" if (n == 65536 and kind == "quad3d") else
                    f"BASELINE configs[2] (C3: {n_total} envs over {world} GPUs)" if (n == 131072 and kind == "quad3d" and world == 8) else
                    "BASELINE configs[2]'s per-GPU shard (131 072 envs per GPU)" if (n == 131072 and kind == "quad3d") else
                    "BASELINE configs[3] (C4)" if (world == 1 and n == 262144 and kind == "quad3d_sl") else "custom")
        workload_long = (f"{cfg_name}: {ENV_ID[kind]}, {n} envs per GPU ({n_total} total, global env ids {rank * n}.. per rank), "
                         f"random actions U[{lo:g},{hi:g})^{nA}, auto-reset, episode tracking; "
                         + (f"one step = one fused rollout launch = {per_launch} env-steps per env, action source '{args.actions}', "
                            f"trajectory (actions, obs, reward, done) written to HBM into "
                            + ("ONE buffer set rewritten in place" if R == 1 else f"a ring of {R} buffer sets ({R * traj_bytes(per_launch) / 1e9:.2f} GB: cold stores)")
                            if args.mode == "rollout"
                            else "one step = one launch = 1 env-step per env, actions read from a device buffer, "
                                 "obs/reward/done written"))
        parallelism_long = (f"env-shard x{world} (contiguous global env ids, seed 0 on every rank; one all-gather of "
                            f"per-env episode stats after every {'rollout launch' if args.exchange_every == 1 else str(args.exchange_every) + ' rollout launches'}, overlapped with the next launch on a second stream: "
                            f"{exchange_kind})") if use_dist else "single GPU"
        roof = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_over_needed": (traffic / bytes_launch) if traffic else None,
            "bytes_per_launch": bytes_launch, "launch_ms_hip_events": kernel_ms,
        }
        roof_long = dict(roof, **{
            "traffic_source": traffic_src, "traffic_frac": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "bytes_definition": bytes_def, "env_steps_per_launch": n * per_launch,
            # the single-step definition applied to the fused launch (it counts 8 nS bytes of state in/out per
            # env-step that the fused kernel keeps in registers): a speed-up-equivalent, NOT an HBM fraction
            "algorithmic_equiv_bytes_per_env_step": algo_bytes,
            "algorithmic_equiv_frac": algo_bytes * n * per_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        line = {
            "metric": "env-steps/sec (batched quadrotor3d-v0)" if kind == "quad3d" else f"env-steps/sec (batched {kind})",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * wall / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if kind in ("quad2d", "quad3d") else "f64 arithmetic on f32 storage",
            "data": "synthetic",
            "prewarm_ms": args.prewarm_ms,            # untimed stretch of the same launches AHEAD of the warm-up (GPU clock ramp) ...
            "prewarm_launches": prewarm_launches[0],  # ... and the number of launches it was
            "config": {
                "workload": (f"{cfg_name}: {ENV_ID[kind]}, {n} envs/GPU, random actions in-kernel, auto-reset; "
                             + (f"1 step = 1 fused {per_launch}-env-step rollout launch, trajectory written to a ring of {R} buffer sets"
                                if args.mode == "rollout" else "1 step = 1 k_step launch (1 env-step per env)")),
                "config_id": ("C2" if (n == 65536 and kind == "quad3d" and world == 1) else f"C2x{world}" if (n == 65536 and kind == "quad3d") else
                              "C3" if (n == 131072 and kind == "quad3d" and world == 8) else f"C3shard x{world}" if (n == 131072 and kind == "quad3d") else
                              "C4" if (n == 262144 and kind == "quad3d_sl" and world == 1) else "custom"),
                "kind": kind, "actions": args.actions, "envs_per_gpu": n, "envs_total": n_total,
                "env_steps_per_launch_per_env": per_launch, "mode": args.mode,
                "trajectory_layout": args.layout if args.mode == "rollout" else "soa", "trajectory_ring": R, "tune": args.tune,
                "parallelism": (f"env-shard x{world}, one all-gather of episode stats per "
                                + ("rollout" if args.exchange_every == 1 else f"{args.exchange_every} rollouts") + ": "
                                + ("rmav_allgather_stats_post (RCCL from librmav.so)" if (exchange_kind or "").startswith("rmav_") else
                                   "torch.distributed all_gather_into_tensor (" + (dist.get_backend() if use_dist else "") + ")"))
                if use_dist else "single GPU",
                "finished_episodes": totals["episodes"],
                "gathered_envs_with_a_finished_episode": gathered_finished,
                "exchange_equals_plain_all_gather": exchange_check,
                "rccl_ranks": (comm_info or {}).get("lib_world"),   # ncclCommCount of the library's own communicator (None: torch's exchange)
            },
            "roofline": roof,
        }
        detail = dict(line)
        detail["config"] = dict(line["config"], workload=workload_long, parallelism=parallelism_long)
        detail["roofline"] = roof_long
        detail["device_state"] = device_state
        detail["host_cpus"] = hc = host_cpus()
        if other:
            detail["other_modes"] = other
        calib = os.path.join(ROOT, "profiles", "cpu_calibration.json")
        if os.path.exists(calib):   # build-CPU / reference ratio per kind, measured in the authoring container (oracle/calibrate.py)
            detail["calibration"] = json.load(open(calib))
        if world == 1 and not use_dist and args.cpu_seconds > 0 and kind != "reinmav":
            cb = cpu_baseline(kind, n, args.chunk, lo, hi, args.cpu_seconds, threads=1)
            detail["cpu_baseline"] = cb
            line["cpu_baseline"] = dict(cb, sample=f"{kind} C oracle (fp64 scalar port of the reference step), {n} envs, random actions + auto-reset, "
                                                   f"{args.cpu_seconds:g} s, 1 thread; host has {hc['usable']} usable CPUs")
            if "cpu_mt" in sec and hc["usable"] > 1:
                # the same port with OpenMP over envs (static schedule) at the thread count this process may actually use:
                # min(affinity mask, cgroup quota), and at the physical-core count when SMT doubles it; >= 3 s each
                scan = {}
                for k in sorted({hc["usable"], hc["usable_physical"]}):
                    scan[k] = cpu_baseline(kind, n, args.chunk, lo, hi, 3.0, threads=k)
                best = max(scan.values(), key=lambda r: r["value"])
                best = dict(best, cores_available=hc["usable"], physical_cores=hc["physical_cores_in_affinity"],
                            cgroup_quota_cpus=hc["cgroup_quota_cpus"], speedup_over_1_thread=best["value"] / cb["value"],
                            parallel_efficiency=best["value"] / cb["value"] / best["cores"],
                            scan={str(k): r["value"] for k, r in scan.items()})
                detail["cpu_baseline_multithread"] = best
                line["cpu_mt"] = {"value": best["value"], "cores": best["cores"], "cores_available": hc["usable"],
                                  "efficiency": round(best["parallel_efficiency"], 3)}
            if "cpu_py" in sec:
                # interpreter-bound stand-in for the reference's own Python (which cannot travel to this box):
                # a per-env NumPy restatement in the reference's style (oracle/numpy_ref.py), ~2 s sample
                try:
                    import numpy_ref

                    v, k, el = numpy_ref.time_steps(min(2.0, args.cpu_seconds))
                    detail["cpu_baseline_python"] = {"value": v, "unit": "env-steps/s", "cores": 1, "kind": "port",
                                                     "sample": f"per-env NumPy restatement of Quadrotor3D.step, {k} env-steps, "
                                                               f"{el:.1f} s, 1 process (the reference itself ran at 16.1 k "
                                                               "env-steps/s in the authoring container, BASELINE.md section 2)"}
                except Exception as e:  # pragma: no cover
                    detail["cpu_baseline_python"] = {"error": repr(e)}
        legs = leg_rows(other)
        if legs:
            line["legs"] = legs
        if "sustained" in other:   # the robust figure beside the K timed launches: the same launches over --sustained-seconds
            line["value_sustained"] = float(f"{other['sustained']['value']:.5g}")
        dpath = args.detail
        if dpath is None:
            dpath = os.path.join(ROOT, "gpurun_out", "bench_detail.json" if world == 1 else f"bench_detail_n{world}.json")
        if dpath != "-":
            try:
                os.makedirs(os.path.dirname(os.path.abspath(dpath)), exist_ok=True)
                with open(dpath, "w") as f:
                    json.dump(detail, f, indent=1)
                line["detail"] = os.path.relpath(dpath, ROOT) if os.path.abspath(dpath).startswith(ROOT) else dpath
            except Exception as e:  # pragma: no cover - the line matters more than the file
                line["detail"] = "not written: " + repr(e)[:80]
        text = compact_text(line)
        sys.stdout.flush()
        print(text, flush=True)
    if native_abandoned:   # a thread of this process is still inside RCCL: do not wait for it in any destructor
        sys.stdout.flush()
        sys.stderr.flush()
        if use_dist:
            dist.barrier()
        os._exit(0)
    if exchange is not None and hasattr(exchange, "close"):
        torch.cuda.synchronize()
        exchange.close()
    env.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_reinmav(g, torch, dev, n: int = 65536, chunk: int = 4, K: int = 30, W: int = 8, cpu_seconds: float = 0.0):
    """ReinmavEnv (reinmav_env.py:90-126,188-264) batched - 13-state rigid body, 50-or-51 Euler sub-steps of 1/5000 s
    per env-step with the built-in PD controller evaluated every sub-step, all in fp64.  The one compute-bound kind: its roofline
    is the fp64 vector ALU (78.6 TFLOP/s dense, MI355X_MICROARCH.md), flops counted from the kernel's ISA
    (k_rollout<REINMAV, ACT_CONTROLLER>: ~850 fp64 flops per sub-step with fma = 2, x 50.47 sub-steps per step on average)."""
    stream = torch.cuda.Stream(device=dev)
    flops_per_step = 850.0 * 50.47
    with torch.cuda.stream(stream):
        env = g.BatchedQuadrotor("reinmav", n, device=dev.index, seed=0, auto_reset=True, track_episodes=True)
        s0 = env.get_state(layout="aos")
        gen = torch.Generator(device="cpu").manual_seed(2)
        env.set_state(s0 + 0.05 * torch.randn(s0.shape, generator=gen).numpy().astype(s0.dtype))   # (they all start from one state)
        out = {"obs": torch.zeros((chunk, env.nS, n), dtype=torch.float32, device=dev), "rew": torch.zeros((chunk, n), dtype=torch.float32, device=dev),
               "done": torch.zeros((chunk, n), dtype=torch.uint8, device=dev)}
        for phase, count in (("warm", W), ("timed", K)):
            if phase == "timed":
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(stream)
            for _ in range(count):
                env.rollout(chunk, mode="controller", layout="soa", fused=True, want=("obs", "rew", "done"), device_out=True, out=out)
        e1.record(stream)
        stream.synchronize()
        wall = time.perf_counter() - t0
        ms = e0.elapsed_time(e1) / K
        env.close()
    ach = flops_per_step * n * chunk / (ms * 1e-3) / 1e12
    res = {"workload": f"reinmav-v0 batched: {n} envs, {chunk} env-steps per launch = ~{chunk * 50.47:.0f} Euler sub-steps with the built-in controller "
                       "in every sub-step (the reference's ReinmavEnv.step), obs / reward / done written per env-step",
           "launches": K, "value": n * chunk * K / wall, "unit": "env-steps/s", "ms_per_launch_hip_events": ms,
           "sub_steps_per_s": 50.47 * n * chunk * K / wall,
           "roofline": {"bound": "valu_fp64", "achieved": ach, "peak": 78.6, "unit": "TFLOP/s", "frac": ach / 78.6,
                        "flops_per_env_step": flops_per_step,
                        "flops_definition": "~850 fp64 flops per sub-step (v_fma_f64 = 2, v_mul / v_add_f64 = 1, counted in the ISA of the sub-step "
                                            "loop incl. the controller's sin / cos / atan2 / asin expansions) x 50.47 sub-steps per step",
                        "note": "~1 070 instructions per sub-step, 780 of them fp64 at half rate, one dependent chain per env: issue- and "
                                "latency-bound well below the flop peak at 1 - 2 wavefronts per SIMD"}}
    if cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np

        import oracle as O

        S = np.tile(np.asarray(s0[:1], np.float64), (256, 1)) + 0.05 * np.random.RandomState(0).randn(256, 13)
        Tt = np.zeros(256)
        done_steps, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < cpu_seconds:
            S, Tt, _ = O.reinmav_batch_step(S, Tt)
            done_steps += 256
        el = time.perf_counter() - t1
        res["cpu_baseline"] = {"value": done_steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": f"reinmav C oracle (fp64, scalar code, 1 thread), 256 envs x {done_steps // 256} env-steps with the built-in "
                                         f"controller, {el:.1f} s on the GPU box's host CPU (the reference's own step: BASELINE.md section 2)"}
    return res


def cpu_policy_baseline(kind: str, budget_s: float, n: int = 4096):
    """C5's per-env work on ONE host core: the fp32 2x64 tanh MLP policy + value net as NumPy matrix products (feature-major, like
    the learner's torch form) + the C oracle's env step, for a batch of n envs - what the rollout of gym_reinmav/run.py:63-68 costs
    per env-step without a GPU, less baselines' own Python."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np

    import oracle as O

    _omp_set_threads(1)
    try:   # one BLAS thread: the number is per core
        from threadpoolctl import threadpool_limits

        limit, blas = threadpool_limits(limits=1), "one BLAS thread"
    except Exception:  # pragma: no cover
        limit, blas = None, f"BLAS threads not limited, {os.cpu_count()} logical cores present"
    nS, nA = O.STATE_DIM[kind], O.ACTION_DIM[kind]
    rng = np.random.RandomState(0)
    W = [[(rng.randn(o, i) * 0.1).astype(np.float32) for i, o in ((nS, 64), (64, 64), (64, k))] for k in (nA, 1)]
    s = rng.uniform(-1, 1, (n, nS)).astype(np.float32)
    sbd = np.full(n, -1, np.int32)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        x = s.T
        outs = []
        for net in W:
            h = x
            for li, w in enumerate(net):
                h = w @ h
                if li < 2:
                    h = np.tanh(h)
            outs.append(h)
        act = (outs[0] + rng.standard_normal(outs[0].shape).astype(np.float32)).T
        s2, _, d, sbd = O.batch_step(kind, s.astype(np.float64), act.astype(np.float64), sbd)
        s = np.where(d[:, None], rng.uniform(-1, 1, (n, nS)), s2).astype(np.float32)
        steps += n
    el = time.perf_counter() - t0
    if limit is not None:
        limit.restore_original_limits()
    return {"value": steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{kind}: NumPy fp32 2x64 tanh MLP policy + value net ({blas}) + C oracle step, {n} envs x {steps // n} env-steps, {el:.1f} s"}


# ---- the boundaries the reference's callers use (reported under other_modes) ---------------------------------------
def bench_gym1(episodes: int = 12):
    """BASELINE configs[0] (C1): quadrotor2d-v0, ONE env behind the gym.Env interface, the loop of the reference's
    test/test_quadrotor2d.py:11-24 - 400 x (action = env.control(); env.step(action); reset on done) - timed with a
    monotonic clock around the loop exactly as the reference does.  Reference (authoring container):
    59 us per step() alone."""
    import gym_reinmav_amd as g

    env = g.make("quadrotor2d-v0")
    env.reset()
    for _ in range(400):   # warm-up: first launches, pinned block
        a = env.control()
        _, _, d, _ = env.step(a)
        if d:
            env.reset()
    best, tot, its = None, 0.0, 0
    for _ in range(episodes):
        env.reset()
        t0 = time.perf_counter()
        for _ in range(400):
            action = env.control()
            _, reward, done, _ = env.step(action)
            if done:
                env.reset()
        el = time.perf_counter() - t0
        tot += el
        its += 400
        best = el if best is None or el < best else best
    # step() alone with a constant action (what BASELINE.md section 2 timed for the reference)
    a = env.control()
    env.reset()
    t0 = time.perf_counter()
    for _ in range(2000):
        _, _, d, _ = env.step(a)
        if d:
            env.reset()
    st = (time.perf_counter() - t0) / 2000
    env.close()
    return {"workload": "quadrotor2d-v0, batch = 1, gym.Env API, 400 x (control(); step(); reset on done) per episode "
                        "(test/test_quadrotor2d.py:11-24), host NumPy in / out",
            "us_per_iteration_control_plus_step": 1e6 * tot / its, "us_per_iteration_best_episode": 1e6 * best / 400,
            "us_per_step_alone": 1e6 * st, "episodes": episodes,
            "reference_us_per_step": 59.0, "reference_source": "BASELINE.md section 2 (reference NumPy step(), authoring container)"}


def bench_vecenv(dev, n: int, iters: int = 3000):
    """The VecEnv contract baselines reaches through make_vec_env (gym_reinmav/run.py:89): QuadrotorVecEnv.step with
    device tensors, one env-step per call, Python in the loop."""
    import torch

    import gym_reinmav_amd as g

    out = {"workload": f"QuadrotorVecEnv('quadrotor3d-v0', {n}).step(actions[N,4] device tensor) -> (obs, rew, done, infos), "
                       "auto-reset, one launch per call"}
    for reuse in (False, True):
        venv = g.QuadrotorVecEnv("quadrotor3d-v0", n, device=dev.index, seed=0, reuse_buffers=reuse)
        venv.reset()
        act = torch.empty((n, 4), dtype=torch.float32, device=dev).uniform_(0.0, 10.0)
        for _ in range(200):
            venv.step(act)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            venv.step(act)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["reuse_buffers" if reuse else "fresh_tensors_per_step"] = {"us_per_step": 1e6 * el / iters,
                                                                      "env_steps_per_s": n * iters / el}
        venv.close()
    return out


def bench_policy(dev, kind: str, n: int, T: int = 32, iters: int = 60, actors=None):
    """BASELINE configs[4] (C5)'s per-GPU shape: 65 536 envs x 32-step PPO2-style rollouts with the Gaussian MLP policy
    and value net evaluated inside the rollout kernel (rmav_rollout_policy), plus the GAE pass over the result."""
    import torch

    import gym_reinmav_amd as g
    from gym_reinmav_amd.ppo import FusedPolicyCollector, MlpPolicy

    out = {"workload": f"{ENV_ID[kind]}, {n} envs x {T}-step rollouts, 2x64 tanh MLP policy + value net in-kernel, "
                       "trajectory + logp + values written to HBM; then rmav_gae over the [T][N] result"}
    # instructions per 64 envs and env-step by class, measured with SQ counters (tools/profile_actors.sh -> profiles/actor_instr_mix.json),
    # priced with the per-class vector-pipe cycles of tools/micro/issue_rate.hip (profiles/r04/issue_rate.md): the roofline that binds
    mix_path = os.path.join(ROOT, "profiles", "actor_instr_mix.json")
    mix = json.load(open(mix_path)) if os.path.exists(mix_path) else {}
    # bf16_1w: round 3's kernel (one wavefront per 64 envs); bf16_mfma / f16_mfma: (actor, critic) wavefront pairs (rmav_policy_pair.hpp)
    # f16_shared: a DIFFERENT architecture - one 2x64 trunk with a mean head and a value head (baselines' value_network = 'shared': what
    # ppo2 builds for gym_reinmav's native env type) - half the activations of the two-net policy the other rows evaluate
    for actor in (actors or ("fp32_valu", "fp32_mfma", "bf16_1w", "bf16_mfma", "f16_mfma", "f16_shared")):
        torch.manual_seed(0)
        env = g.BatchedQuadrotor(kind, n, device=dev.index, seed=0, auto_reset=True, track_episodes=True)
        if actor == "bf16_1w":
            env.set_tuning(policy_pair=0)
        pol = MlpPolicy(env.nS, env.nA, value_network=("shared" if actor == "f16_shared" else "copy")).to(dev)
        ro = FusedPolicyCollector(env, pol, T, bf16_mfma=actor.startswith("bf16"), f32_mfma=(actor == "fp32_mfma"), f16_mfma=actor.startswith("f16"))
        adv, ret = torch.empty_like(ro.rew), torch.empty_like(ro.rew)
        sums = torch.zeros(2, dtype=torch.float64, device=dev)
        for _ in range(5):
            ro.collect()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(iters):
            ro.collect()
        e1.record()
        for _ in range(iters):
            env.gae(ro.rew, ro.done, ro.val, out=(adv, ret), sums=sums)
        e2.record()
        torch.cuda.synchronize()
        ms_ro, ms_gae = e0.elapsed_time(e1) / iters, e1.elapsed_time(e2) / iters
        # the rollout kernel alone (weights already packed): what its two rooflines are priced on
        e3, e4 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        C_, A_ = ro._C, ro._A
        pp = lambda t: C_.c_void_p(t.data_ptr())  # noqa: E731
        prec = ro._call[-1]
        e3.record()
        for _ in range(iters):
            A_.check(A_.lib().rmav_rollout_policy(env._h, T, pp(ro.weights), pp(ro.act), pp(ro.obs[1:]), pp(ro.rew), pp(ro.done),
                                                  pp(ro.logp), pp(ro.val), prec))
        e4.record()
        torch.cuda.synchronize()
        ms_k = e3.elapsed_time(e4) / iters
        nS, nA = env.nS, env.nA
        hbm_b = n * (T * (4 * (nS + nA + 1) + 1 + 8) + 8 * nS + 24 + 4)   # trajectory + logp + value per step; state etc. per launch
        useful = 2 * ((nS * 64 + 64 * 64 + 64 * nA) + (nS * 64 + 64 * 64 + 64))          # policy net + value net, per env-step
        if actor == "f16_shared":
            useful = 2 * (nS * 64 + 64 * 64 + 64 * (nA + 1))                              # one trunk, two heads
        roof = {"hbm": roofline_obj(hbm_b, ms_k, None, None,
                                    f"{n} envs x ({T} env-steps x {4 * (nS + nA + 1) + 1 + 8} B (actions, obs, reward, done, logp, value) + "
                                    f"{8 * nS + 28} B per launch)")}
        if actor != "fp32_valu":
            # matrix-pipe work incl. tile padding: bf16 = 56 v_mfma_f32_32x32x16_bf16 per 64 envs and step (inputs padded 10 -> 16,
            # outputs 4 / 1 -> 32); fp32 = 2 nets x (2 ceil(nS / 2) + 64) v_mfma_f32_32x32x2_f32 per 32 envs (layer 3 on the vector ALU)
            half = actor != "fp32_mfma"                          # bf16 / f16 operands: the same instruction count and peak
            padded = (28 if actor == "f16_shared" else 56) * 2 * 32 * 32 * 16 / 64 if half else 2 * (2 * ((nS + 1) // 2) + 64) * 2 * 32 * 32 * 2 / 32
            peak = 2500.0 if half else 157.3   # dense TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
            ach = padded * n * T / (ms_k * 1e-3) / 1e12
            roof["mfma"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            "flops_per_env_step_incl_tile_padding": padded, "useful_flops_per_env_step": useful,
                            "useful_frac": useful * n * T / (ms_k * 1e-3) / 1e12 / peak}
        else:
            ach = useful * n * T / (ms_k * 1e-3) / 1e12
            roof["valu"] = {"bound": "valu_fp32", "achieved": ach, "peak": 157.3, "unit": "TFLOP/s", "frac": ach / 157.3,
                            "useful_flops_per_env_step": useful}
        m = mix.get(actor)
        if m:   # SIMD-cycles the launch had per 64 envs and env-step vs the cycles its measured instruction mix needs on the vector pipe
            sclk_ghz = float(m.get("clock_ghz", 2.2))   # GRBM_GUI_ACTIVE / kernel time of that actor's profiled launches
            avail = ms_k * 1e-3 * sclk_ghz * 1e9 * 1024 / (n / 64 * T)
            other = m["valu"] - m["trans"] - m["cvt"] - m["mfma"] - m["packed_static"]
            need = ((m["trans"] * 8.8 + (m["valu"] - m["trans"] - m["mfma"]) * 5.3) if m["lone_wavefront"] else
                    (m["trans"] * 8.4 + m["cvt"] * 4.45 + m["packed_static"] * 5.0 + other * 2.8))
            roof["valu_pipe"] = {"bound": "valu_issue" if m["lone_wavefront"] else "valu_pipe", "frac": need / avail,
                                 "needed_simd_cycles_per_64_env_steps": need, "available_simd_cycles_per_64_env_steps": avail,
                                 "shader_clock_GHz": sclk_ghz,
                                 "instructions_per_64_env_steps": {k: m[k] for k in ("valu", "trans", "cvt", "mfma", "salu", "lds", "vmem_wr", "packed_static")},
                                 "cycles_per_instruction": ({"transcendental": 8.8, "other": 5.3, "note": "ONE wavefront per SIMD: its issue rate"}
                                                            if m["lone_wavefront"] else
                                                            {"transcendental": 8.4, "convert": 4.45, "packed_f32": 5.0, "other": 2.8,
                                                             "note": "two wavefronts per SIMD: the vector pipe's occupancy"}),
                                 "source": "SQ counters (profiles/r04/actors_sq.md) x tools/micro/issue_rate.hip (profiles/r04/issue_rate.md)"}
        binding = max(roof, key=lambda k: roof[k]["frac"])
        out[actor] = {"bound": roof[binding]["bound"], "bound_frac": roof[binding]["frac"],
                      "ms_per_rollout_incl_weight_pack": ms_ro, "ms_per_rollout_kernel": ms_k,
                      "env_steps_per_s": n * T / (ms_ro * 1e-3), "kernel_env_steps_per_s": n * T / (ms_k * 1e-3),
                      "gae_ms": ms_gae, "gae_GBps": 17.0 * n * T / (ms_gae * 1e-3) / 1e9, "roofline": roof}
        env.close()
    return out


if __name__ == "__main__":
    main()
