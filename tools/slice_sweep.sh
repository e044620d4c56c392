#!/bin/bash
# Big batches as sequences of two-wavefront launches over slices (RMAV_TUNE_SLICE_ENVS) x store policy: gpurun_out/<tag>/slice.md
TAG=${1:-slice}; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "| envs | kind | tuning | us per launch | frac |" > $OUT/slice.md; echo "|---|---|---|---|---|" >> $OUT/slice.md
for rep in 1 2; do
for CASE in ${CASES:-"quad3d:131072" "quad3d:262144"}; do K=${CASE%%:*}; N=${CASE##*:}
  for T in ${TUNES:-"" store_policy=1 store_policy=2 slice_envs=65536 slice_envs=65536,store_policy=1 slice_envs=65536,store_policy=2 slice_envs=65536,store_policy=0}; do
    S=$(( 65536 * 700 / N + 60 ))
    timeout 300 python bench.py --kind $K --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary --detail - ${T:+--tune $T} 2>/dev/null | grep '^{' | \
      python -c "import json,sys; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('| $N | $K | ${T:-default} | %.2f | %.3f |' % (r['launch_ms_hip_events']*1e3, r['frac']))" >> $OUT/slice.md
  done
done
done
cat $OUT/slice.md
