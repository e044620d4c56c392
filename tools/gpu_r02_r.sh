#!/bin/bash
# sliced two-wavefront launches beyond the one-workgroup-per-CU capacity vs the one-wavefront kernel (cold ring);
# slung-load kinds now hand over every env-step (8 pairs fit)
OUT=gpurun_out/r02_r; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity with slicing forced"; RMAV_SLICE=1 timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
echo "== parity default"; timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -2
: > $OUT/slice.jsonl
for REP in 1 2; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 98304 131072 163840 196608 262144 393216 524288 1048576; do
    S=$(( 65536 * 400 / N + 25 ))
    for SL in 0 1; do
      RMAV_SLICE=$SL timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'slice': $SL, 'rep': $REP, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/slice.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/slice.jsonl")]
t = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: t[(r['kind'], r['n'])][r['slice']].append(r)
print("| kind | envs | default us (TB/s) | sliced two-wavefront us (TB/s) | gain |")
for k, v in t.items():
    a = min(x['us'] for x in v[0]); b = min(x['us'] for x in v[1])
    ta = max(x['TBps'] for x in v[0]); tb = max(x['TBps'] for x in v[1])
    print(f"| {k[0]} | {k[1]} | " + "/".join(f"{x['us']:.1f}" for x in v[0]) + f" ({ta:.2f}) | " + "/".join(f"{x['us']:.1f}" for x in v[1]) + f" ({tb:.2f}) | {100 * (a / b - 1):+.1f} % |")
PY
