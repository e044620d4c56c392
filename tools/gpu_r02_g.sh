#!/bin/bash
# round 2, GPU session G: cooperative reset in k_step (tests + latency), split-group A/B (1 / 2 / 4 pairs per workgroup)
TAG=${1:-r02_g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest (default build)"
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
echo "== pytest parity with the 4-pair build"
RMAV_LIB_PATH=$PWD/reinmav-gym_amd/build/librmav_g4.so timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/pytest_g4.log 2>&1; echo "pytest g4 rc=$?"; tail -3 $OUT/pytest_g4.log
echo "== step latency"
timeout 600 python tools/step_latency.py quad3d 2>&1 | grep -v amdgpu.ids | grep "mode=buffer" > $OUT/step_latency_kstep.txt; cat $OUT/step_latency_kstep.txt
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  timeout 300 python bench.py --kind $K --mode step --steps 8000 --warmup 1000 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$K step', round(j['roofline']['launch_ms_hip_events'] * 1e3, 3), 'us', round(j['roofline']['frac'], 3))"
done
echo "== split group A/B (cold ring, split forced, write-through)"
: > $OUT/group_ab.jsonl
for REP in 1 2; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 49152 65536 98304 131072; do
    S=$(( 65536 * 800 / N + 40 ))
    for G in 1 2 4; do
      LIB=$PWD/reinmav-gym_amd/build/librmav_g$G.so
      [ $G = 1 ] && LIB=$PWD/reinmav-gym_amd/gym_reinmav_amd/librmav.so
      RMAV_LIB_PATH=$LIB RMAV_SPLIT=1 RMAV_STORE_POLICY=1 timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'group': $G, 'rep': $REP, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/group_ab.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/group_ab.jsonl")]
t = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: t[(r['kind'], r['n'])][r['group']].append(r['us'])
print("| kind | envs | 1 pair/WG us | 2 pairs | 4 pairs |")
print("|---|---|---|---|---|")
for k, v in t.items():
    print(f"| {k[0]} | {k[1]} | " + " | ".join("/".join(f"{x:.1f}" for x in v[g]) for g in (1, 2, 4)) + " |")
PY
