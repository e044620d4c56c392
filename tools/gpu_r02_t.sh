#!/bin/bash
# wide (ds_read_b128 + buffer_store_dwordx4) vs dword drain of the hand-over tiles
OUT=gpurun_out/r02_t; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (wide drain, default build)"; timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q 2>&1 | tail -2
: > $OUT/wide_ab.jsonl
for REP in 1 2 3; do
for ACT in random controller; do
for K in quad3d quad3d_sl quad2d quad2d_sl; do
  for N in 32768 65536 98304 131072; do
    S=$(( 65536 * 800 / N + 40 ))
    for V in wide narrow; do
      LIB=$PWD/reinmav-gym_amd/gym_reinmav_amd/librmav.so
      [ $V = narrow ] && LIB=$PWD/reinmav-gym_amd/build/librmav_narrow.so
      RMAV_LIB_PATH=$LIB timeout 300 python bench.py --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'actions': '$ACT', 'kind': '$K', 'n': $N, 'variant': '$V', 'rep': $REP, 'us': r['launch_ms_hip_events'] * 1e3}))" >> $OUT/wide_ab.jsonl
    done
  done
done
done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/wide_ab.jsonl")]
t = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: t[(r['actions'], r['kind'], r['n'])][r['variant']].append(r['us'])
print("| actions | kind | envs | dword drain us | wide drain us | gain (medians) |")
import statistics as st
for k, v in t.items():
    a, b = st.median(v['narrow']), st.median(v['wide'])
    print(f"| {k[0]} | {k[1]} | {k[2]} | " + "/".join(f"{x:.1f}" for x in v['narrow']) + " | " + "/".join(f"{x:.1f}" for x in v['wide']) + f" | {100 * (a / b - 1):+.1f} % |")
PY
