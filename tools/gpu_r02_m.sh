#!/bin/bash
OUT=gpurun_out/r02_m; mkdir -p $OUT
: > $OUT/g8_ab.jsonl
for REP in 1 2; do
for K in quad3d quad2d; do
  for N in 32768 49152 65536 98304 131072; do
    S=$(( 65536 * 800 / N + 40 ))
    for G in 4 8; do
      LIB=$PWD/reinmav-gym_amd/build/librmav_g$G.so
      [ $G = 4 ] && LIB=$PWD/reinmav-gym_amd/gym_reinmav_amd/librmav.so
      RMAV_LIB_PATH=$LIB RMAV_SPLIT=1 timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'group': $G, 'rep': $REP, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/g8_ab.jsonl
    done
  done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/g8_ab.jsonl")]
t = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: t[(r['kind'], r['n'])][r['group']].append(r['us'])
for k, v in t.items():
    print(k, " | ".join(f"G={g}: " + "/".join(f"{x:.1f}" for x in v[g]) for g in (4, 8)))
PY
