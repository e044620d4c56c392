#!/usr/bin/env python3
"""Summarise the rocprofv3 output directories written by tools/profile_round.sh into markdown."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(d, suffix):
    hits = glob.glob(os.path.join(root, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("void ", "")
    for a, b in (("rmav::", ""), ("ParamsT<", "P<"), ("RolloutArgs", "Args")):
        name = name.replace(a, b)
    return name[:70]


print(f"# rocprofv3 summary ({os.path.basename(root)})\n")
for d in ("trace_rollout", "trace_step"):
    f = find(d, "kernel_stats.csv")
    if not f:
        print(f"## {d}: no kernel_stats.csv found\n")
        continue
    print(f"## {d}: kernel stats (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print("| %s | %s | %.3f | %.3f | %.3f | %.3f | %s |" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    print()

print("## PMC passes (per-dispatch values of the k_rollout kernel, averaged)\n")
print("| run | counter | dispatches | mean value | unit note |")
print("|---|---|---|---|---|")
vals = {}
for d in sorted(os.listdir(root)):
    if not d.startswith("pmc_") or not os.path.isdir(os.path.join(root, d)):
        continue
    f = find(d, "counter_collection.csv")
    if not f:
        print(f"| {d} | - | 0 | - | no counter_collection.csv |")
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_rollout" not in r.get("Kernel_Name", ""):
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        # drop the warm-up dispatches (first 25 %) so that first-touch traffic does not bias the mean
        v = v[len(v) // 4:]
        m = sum(v) / max(1, len(v))
        vals[(d, c)] = m
        print(f"| {d} | {c} | {len(v)} | {m:.1f} | KiB per dispatch (rocprofv3 derives it from TCC_EA requests) |")
print()
import json
print("```json")
print(json.dumps({f"{k[0]}:{k[1]}": v for k, v in vals.items()}, indent=1))
print("```")
