#!/bin/bash
# A/B of the producer/consumer split of random-action rollouts on one box.  Output: gpurun_out/split_ab.jsonl
mkdir -p gpurun_out
: > gpurun_out/split_ab.jsonl
for kind in ${KINDS:-quad3d quad3d_sl quad2d quad2d_sl}; do
  for n in ${NS:-65536 131072 262144 1048576}; do
    for sp in 0 1; do
      S=$(( 65536 * 1500 / n + 100 ))   # >= ~50 ms timed after >= ~10 ms of warm-up
      RMAV_SPLIT=$sp timeout 120 python bench.py --kind $kind --envs-per-gpu $n --steps $S --warmup $((S / 5)) \
        --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | sed "s/^{/{\"split\": $sp, /" >> gpurun_out/split_ab.jsonl
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/split_ab.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    c = d["config"]
    print(c["workload"].split(",")[0], c["envs_per_gpu"], "split", d["split"], round(d["ms_per_step"] * 1e3, 1), "us/launch", round(d["value"] / 1e9, 2), "G/s", round(d["roofline"]["frac"], 3))
PY
