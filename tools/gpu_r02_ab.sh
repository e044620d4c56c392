#!/bin/bash
# batches beyond the two-wavefront kernel's capacity: one one-wavefront launch vs two-wavefront launches over balanced slices
# (RMAV_SLICE=1) vs full-capacity slices + remainder (RMAV_SLICE=2); cold ring, final kernels
OUT=gpurun_out/r02_ab; mkdir -p $OUT; : > $OUT/slice.jsonl
for ACT in random controller; do
for K in quad3d quad3d_sl; do
  for N in 163840 196608 229376 262144 327680 393216 524288 1048576; do
    S=$(( 65536 * 400 / N + 20 ))
    for SL in 0 1 2; do
      RMAV_SLICE=$SL timeout 300 python bench.py --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'actions': '$ACT', 'kind': '$K', 'n': $N, 'slice': $SL, 'us': j['ms_per_step'] * 1e3, 'frac': r['bytes_per_launch'] / (j['ms_per_step'] * 1e-3) / 8e12}))" >> $OUT/slice.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/slice.jsonl")]
t = collections.OrderedDict()
for r in rows: t.setdefault((r['actions'], r['kind'], r['n']), {})[r['slice']] = r
print("| actions | kind | envs | one launch (one wavefront) us (frac) | balanced slices | full slices + remainder |")
print("|---|---|---|---|---|---|")
for k, v in t.items():
    f = lambda x: f"{x['us']:.1f} ({x['frac']:.3f})" if x else "-"
    print(f"| {k[0]} | {k[1]} | {k[2]} | {f(v.get(0))} | {f(v.get(1))} | {f(v.get(2))} |")
PY
