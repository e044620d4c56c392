#!/bin/bash
# round 2, second GPU session: full GPU suite with the lean step kernel, step-kernel A/B, cold-ring sweep of
# store policy x split over kinds and sizes (crossover table)
TAG=${1:-r02_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.log
echo "== step kernel A/B"
: > $OUT/step_ab.jsonl
for N in 65536 262144 1048576; do
  S=$(( 65536 * 8000 / N + 400 ))
  for SK in 0 1; do
    for K in quad3d quad3d_sl; do
      RMAV_STEP_KERNEL=$SK timeout 300 python bench.py --kind $K --mode step --envs-per-gpu $N --steps $S --warmup $((S/5)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(json.dumps({'kind': '$K', 'n': $N, 'step_kernel': $SK, 'us': j['roofline']['launch_ms_hip_events'] * 1e3, 'frac': j['roofline']['frac']}))" >> $OUT/step_ab.jsonl
    done
  done
done
cat $OUT/step_ab.jsonl
echo "== cold sweep: policy x split"
: > $OUT/policy_split.jsonl
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 65536 131072 262144 524288 1048576; do
    S=$(( 65536 * 600 / N + 40 ))
    for SP in 0 1; do
      for POL in 0 1 2; do
        RMAV_SPLIT=$SP RMAV_STORE_POLICY=$POL timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'split': $SP, 'policy': $POL, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3, 'frac': r['frac'], 'ring': j['config']['trajectory_ring']}))" >> $OUT/policy_split.jsonl
      done
    done
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/policy_split.jsonl")]
best = collections.defaultdict(list)
for r in rows: best[(r['kind'], r['n'])].append(r)
print("| kind | envs | best (split, policy) | us | TB/s | all: (split,policy) us |")
print("|---|---|---|---|---|---|")
for k, v in best.items():
    b = min(v, key=lambda r: r['us'])
    print(f"| {k[0]} | {k[1]} | ({b['split']},{b['policy']}) | {b['us']:.1f} | {b['TBps']:.2f} | " + " ".join(f"({r['split']},{r['policy']}) {r['us']:.1f}" for r in v) + " |")
PY
