#!/usr/bin/env python3
import json, sys
print("| kind | envs | mode | env-steps/launch/env | us/launch | G env-steps/s | algorithmic GB/s | frac |")
print("|---|---|---|---|---|---|---|---|")
for l in open(sys.argv[1]):
    d = json.loads(l)
    c, r = d["config"], d["roofline"]
    kind = c["workload"].split(",")[0]
    print(f"| {kind} | {c['envs_per_gpu']} | {c['mode']} | {c['env_steps_per_launch_per_env']} | {r['launch_ms_hip_events']*1e3:.2f} | "
          f"{d['value']/1e9:.2f} | {r['achieved']:.0f} | {r['frac']:.3f} |")
