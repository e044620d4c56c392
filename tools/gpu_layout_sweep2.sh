#!/bin/bash
# AoS trajectory layout with the LDS-transposed obs path forced on/off (RMAV_STORE_POLICY=3 / 0) vs SoA
mkdir -p gpurun_out
: > gpurun_out/layout_sweep2.jsonl
for kind in quad3d quad3d_sl; do
  for n in 65536 131072 262144 1048576; do
    for cfg in "soa -1" "aos 0" "aos 3"; do
      set -- $cfg
      if [ "$2" = "-1" ]; then unset RMAV_STORE_POLICY; else export RMAV_STORE_POLICY=$2; fi
      timeout 120 python bench.py --kind $kind --envs-per-gpu $n --layout $1 --steps $(( 65536 * 1500 / n + 100 )) --warmup $(( 65536 * 300 / n + 20 )) \
        --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | sed "s/^{/{\"policy\": \"$2\", /" >> gpurun_out/layout_sweep2.jsonl
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/layout_sweep2.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    c = d["config"]
    print(c.get("workload", "")[:28], c.get("trajectory_layout"), "policy", d["policy"], c.get("envs_per_gpu"), round(d["ms_per_step"] * 1e3, 1), "us/launch", round(d["value"] / 1e9, 2), "G/s")
PY
