#!/bin/bash
# after restoring the one-wavefront kernels' register budget: one wavefront vs two (one workgroup per CU), cold ring
OUT=gpurun_out/r02_u; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"; timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
: > $OUT/occ.jsonl
for ACT in random controller; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 65536 98304 131072 163840 262144 524288 1048576; do
    S=$(( 65536 * 500 / N + 30 ))
    for SP in 0 1; do
      [ $N -gt 131072 ] && [ $SP = 1 ] && continue
      RMAV_SPLIT=$SP timeout 300 python bench.py --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'actions': '$ACT', 'kind': '$K', 'n': $N, 'split': $SP, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3, 'frac': r['frac']}))" >> $OUT/occ.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/occ.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r['actions'], r['kind'], r['n'])][r['split']] = r
print("| actions | kind | envs | one wavefront us (TB/s, frac) | two wavefronts us (TB/s, frac) |")
for k, v in t.items():
    f = lambda x: f"{x['us']:.1f} ({x['TBps']:.2f}, {x['frac']:.3f})" if x else "-"
    print(f"| {k[0]} | {k[1]} | {k[2]} | {f(v.get(0))} | {f(v.get(1))} |")
PY
