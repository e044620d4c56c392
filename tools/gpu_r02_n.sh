#!/bin/bash
# workgroup-size scan of the one-wavefront kernel and 8-pair two-wavefront kernel beyond 65 536 envs (cold ring)
OUT=gpurun_out/r02_n; mkdir -p $OUT
: > $OUT/wg_scan.jsonl
B=$PWD/reinmav-gym_amd/build
for K in quad3d quad2d quad3d_sl; do
  for N in 81920 98304 131072 163840 196608 262144 524288 1048576; do
    S=$(( 65536 * 600 / N + 30 ))
    for V in "single256|$B/librmav_kb1024.so|RMAV_SPLIT=0 RMAV_BLOCK=256" "single512|$B/librmav_kb1024.so|RMAV_SPLIT=0 RMAV_BLOCK=512" "single1024|$B/librmav_kb1024.so|RMAV_SPLIT=0 RMAV_BLOCK=1024" "split4|$PWD/reinmav-gym_amd/gym_reinmav_amd/librmav.so|RMAV_SPLIT=1" "split8|$B/librmav_g8.so|RMAV_SPLIT=1"; do
      name=${V%%|*}; rest=${V#*|}; lib=${rest%%|*}; envs=${rest#*|}
      [ $K = quad3d_sl ] && [ $name = split8 ] && continue
      [ $N -gt 262144 ] && [ ${name:0:5} = split ] && continue
      env $envs RMAV_LIB_PATH=$lib timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'kind': '$K', 'n': $N, 'variant': '$name', 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3}))" >> $OUT/wg_scan.jsonl
    done
  done
done
python - <<PY
import json, collections
rows = [json.loads(l) for l in open("$OUT/wg_scan.jsonl")]
t = collections.defaultdict(dict)
for r in rows: t[(r['kind'], r['n'])][r['variant']] = r
vs = ["single256", "single512", "single1024", "split4", "split8"]
print("| kind | envs | " + " | ".join(vs) + " |")
for k, v in t.items():
    print(f"| {k[0]} | {k[1]} | " + " | ".join((f"{v[x]['us']:.1f} ({v[x]['TBps']:.2f})" if x in v else "-") for x in vs) + " |")
PY
