#!/bin/bash
# pairs per workgroup chosen as ceil(N / 16384) (one workgroup per CU) vs fixed 4 vs the one-wavefront kernel (cold ring)
OUT=gpurun_out/r02_o; mkdir -p $OUT
: > $OUT/autog.jsonl
for ACT in random controller; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 16384 32768 49152 65536 81920 98304 114688 131072 163840; do
    S=$(( 65536 * 500 / N + 30 ))
    for V in "single|RMAV_SPLIT=0" "split_g4|RMAV_SPLIT=1 RMAV_SPLIT_GROUP=4" "split_auto|RMAV_SPLIT=1"; do
      name=${V%%|*}; envs=${V#*|}
      env $envs timeout 300 python bench.py --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'actions': '$ACT', 'kind': '$K', 'n': $N, 'variant': '$name', 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/autog.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/autog.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r['actions'], r['kind'], r['n'])][r['variant']] = r
vs = ["single", "split_g4", "split_auto"]
print("| actions | kind | envs | " + " | ".join(vs) + " |")
for k, v in t.items():
    print(f"| {k[0]} | {k[1]} | {k[2]} | " + " | ".join((f"{v[x]['us']:.1f} ({v[x]['TBps']:.2f})" if x in v else "-") for x in vs) + " |")
PY
