#!/bin/bash
# round 2, GPU session I: split crossover per kind and action source with 4 pairs per workgroup (cold ring), then
# the rocprofv3 evidence of the round
TAG=${1:-r02_i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
: > $OUT/split_cross.jsonl
for ACT in random controller; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 49152 65536 81920 98304 114688 131072 163840 196608 262144; do
    S=$(( 65536 * 500 / N + 40 ))
    for SP in 0 1; do
      RMAV_SPLIT=$SP timeout 300 python bench.py --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'actions': '$ACT', 'n': $N, 'split': $SP, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/split_cross.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/split_cross.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r['actions'], r['kind'], r['n'])][r['split']] = r
print("| actions | kind | envs | one wavefront us (TB/s) | two wavefronts us (TB/s) | two-wavefront gain |")
print("|---|---|---|---|---|---|")
for k, v in t.items():
    a, b = v.get(0), v.get(1)
    if a and b:
        print(f"| {k[0]} | {k[1]} | {k[2]} | {a['us']:.1f} ({a['TBps']:.2f}) | {b['us']:.1f} ({b['TBps']:.2f}) | {100 * (a['us'] / b['us'] - 1):+.1f} % |")
PY
bash tools/profile_r02.sh r02
