#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories written by tools/profile_r02.sh into markdown, and emit the
profiles/traffic.json entries bench.py echoes as roofline.traffic (keyed by the exact workload)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(root).replace("prof_", "")   # directory under profiles/ the summary is committed to
KERNELS = ("k_rollout", "k_step")


def find(d, suffix):
    hits = glob.glob(os.path.join(root, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "")
    for a, b in (("rmav::", ""), ("ParamsT<", "P<"), ("RolloutArgs", "Args")):
        name = name.replace(a, b)
    return name[:70]


def bench_line(d):
    f = os.path.join(root, d + ".log")
    if not os.path.exists(f):
        return None
    for l in open(f, errors="replace"):
        if l.startswith("{"):
            try:
                return json.loads(l)
            except Exception:
                pass
    return None


print(f"# rocprofv3 summary ({os.path.basename(root)})\n")
traces = sorted(d for d in os.listdir(root) if d.startswith("trace_") and os.path.isdir(os.path.join(root, d)))
avg_us = {}
for d in traces:
    f = find(d, "kernel_stats.csv")
    if not f:
        print(f"## {d}: no kernel_stats.csv found\n")
        continue
    j = bench_line(d)
    print(f"## {d}: kernel stats (rocprofv3 --kernel-trace --stats)\n")
    if j:
        r = j["roofline"]
        print(f"bench line of the same run: {j['value'] / 1e9:.2f} G env-steps/s, launch {r['launch_ms_hip_events'] * 1e3:.2f} us (HIP events), "
              f"{r['bytes_per_launch'] / 1e6:.1f} MB per launch, roofline frac {r['frac']:.3f}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print("| %s | %s | %.3f | %.3f | %.3f | %.3f | %s |" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
        if any(k in r["Name"] for k in KERNELS) and d not in avg_us:
            avg_us[d] = float(r["AverageNs"]) / 1e3
    print()

print("## PMC passes (per-dispatch values of the hot-path kernel, warm-up dispatches dropped)\n")
print("| run | counter | dispatches | mean KiB per dispatch |")
print("|---|---|---|---|")
vals = {}
cover = {}
for d in sorted(os.listdir(root)):
    if not d.startswith("pmc_") or not os.path.isdir(os.path.join(root, d)):
        continue
    f = find(d, "counter_collection.csv")
    if not f:
        print(f"| {d} | - | 0 | no counter_collection.csv |")
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if not any(k in r.get("Kernel_Name", "") for k in KERNELS):
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        # envs one dispatch covers: the two-wavefront kernels (k_rollout<K, 5 | 6, *>) run 128 threads per 64 envs
        two = any(t in r["Kernel_Name"] for t in (", 5, ", ", 6, "))
        cover[d[4:].rsplit("_", 2)[0]] = int(r["Grid_Size"]) // (2 if two else 1)
    for c, v in acc.items():
        v = v[len(v) // 3:]   # drop the warm-up third (first-touch traffic)
        m = sum(v) / max(1, len(v))
        vals[(d, c)] = m
        print(f"| {d} | {c} | {len(v)} | {m:.1f} |")
print()

# calibration: 16 M envs x 1 step of quadrotor3d: reads 4 (nS + nA) + 16 = 72 B/env, writes 4 nS + 5 + 8 = 53 B/env
# (state out, reward, done, episode accumulators; conditional stores of the ~1.3 % finishing lanes not counted)
n_cal = 16777216
known_r, known_w = 72.0 * n_cal / 1024, 53.0 * n_cal / 1024
fr, wr = vals.get(("pmc_calib_FETCH_SIZE", "FETCH_SIZE")), vals.get(("pmc_calib_WRITE_SIZE", "WRITE_SIZE"))
fcorr, wcorr = 2.0, 1.0
if fr and wr:
    print(f"calibration (16 777 216 envs, one step, far beyond the Infinity Cache): FETCH_SIZE reports {fr:.0f} KiB for "
          f"{known_r:.0f} KiB read (x{known_r / fr:.2f}); WRITE_SIZE reports {wr:.0f} KiB for {known_w:.0f} KiB written (x{known_w / wr:.2f})\n")
print("## HBM-side bytes per launch (FETCH_SIZE x 2 [gfx950 half-count, guide + calibration] + WRITE_SIZE)\n")
print("| case | bytes per launch | needed bytes (bench) | ratio | rocprofv3 avg us | physical TB/s |")
print("|---|---|---|---|---|---|")
traffic = {}
for d in traces:
    name = d[len("trace_"):]
    f_, w_ = vals.get((f"pmc_{name}_FETCH_SIZE", "FETCH_SIZE")), vals.get((f"pmc_{name}_WRITE_SIZE", "WRITE_SIZE"))
    j = bench_line(d)
    if f_ is None or w_ is None or not j:
        continue
    # a batch beyond the two-wavefront kernel's capacity may run as several dispatches per bench launch (two rounds)
    per_launch = max(1, round(j["config"]["envs_per_gpu"] / cover[name])) if cover.get(name) else 1
    b = (fcorr * f_ + wcorr * w_) * 1024.0 * per_launch
    need = j["roofline"]["bytes_per_launch"]
    us = avg_us.get(d)
    if us and per_launch > 1:
        us *= per_launch
        name_note = f" ({per_launch} dispatches per launch)"
    else:
        name_note = ""
    print(f"| {name}{name_note} | {b / 1e6:.1f} MB | {need / 1e6:.1f} MB | {b / need:.3f} | {us:.2f} | {b / us / 1e6:.2f} |" if us else f"| {name} | {b/1e6:.1f} MB | {need/1e6:.1f} MB | {b/need:.3f} | - | - |")
    c = j["config"]
    key = (f"{c['kind']}:{c['mode']}:{c['env_steps_per_launch_per_env']}:{c['envs_per_gpu']}:"
           f"{'inplace' if (c['trajectory_ring'] == 1 or c['mode'] == 'step') else 'ring'}:{c['actions']}:{c['trajectory_layout']}"
           + (f":{c['tune']}" if c.get("tune") else ""))
    traffic[key] = {"bytes": b, "source": f"profiles/{ROUND}/rocprofv3_summary.md ({name}: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"}
print("\n```json")
print(json.dumps(traffic, indent=1))
print("```")
json.dump(traffic, open(os.path.join(root, "traffic.json"), "w"), indent=1)
