#!/bin/bash
# round 2, third GPU session: sliced store-pattern microbench; finer split crossover sweep (cold ring)
TAG=${1:-r02_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
SLICED_ONLY=1 tools/micro/_build/sp 65536 131072 262144 1048576 > $OUT/store_sliced.txt 2>&1
cat $OUT/store_sliced.txt
: > $OUT/split_cross.jsonl
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 49152 65536 81920 98304 114688 131072 163840 196608; do
    S=$(( 65536 * 600 / N + 40 ))
    for SP in 0 1; do
      POL=$(( SP == 1 ? 1 : 2 ))
      RMAV_SPLIT=$SP RMAV_STORE_POLICY=$POL timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'split': $SP, 'policy': $POL, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/split_cross.jsonl
    done
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/split_cross.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r['kind'], r['n'])][r['split']] = r['us']
print("| kind | envs | single (nt) us | split (wt) us | split gain |")
print("|---|---|---|---|---|")
for k, v in t.items():
    print(f"| {k[0]} | {k[1]} | {v.get(0, 0):.1f} | {v.get(1, 0):.1f} | {100 * (v.get(0, 0) / v.get(1, 1) - 1):+.1f} % |")
PY
