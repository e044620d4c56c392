#!/bin/bash
# fused rollout, SoA vs AoS trajectory layout, per kind / batch size (1 GPU).  Output: gpurun_out/layout_sweep.jsonl
mkdir -p gpurun_out
: > gpurun_out/layout_sweep.jsonl
for kind in quad3d quad3d_sl; do
  for n in 65536 262144 1048576; do
    for layout in soa aos; do
      timeout 120 python bench.py --kind $kind --envs-per-gpu $n --layout $layout --steps $(( 65536 * 1500 / n + 100 )) --warmup $(( 65536 * 300 / n + 20 )) \
        --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 >> gpurun_out/layout_sweep.jsonl
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/layout_sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    c = d["config"]
    print(c.get("workload", "")[:40], c.get("trajectory_layout"), c.get("envs_per_gpu"), round(d["ms_per_step"] * 1e3, 1), "us/launch", round(d["value"] / 1e9, 2), "G/s", d["roofline"]["frac"])
PY
