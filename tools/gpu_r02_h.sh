#!/bin/bash
# which part of the episode tracking costs what in the single-step kernel? (ablation builds)
OUT=gpurun_out/r02_h; mkdir -p $OUT
for REP in 1 2; do
for V in full NO_TOTALS NO_LAST NO_EP; do
  LIB=$PWD/reinmav-gym_amd/build/librmav_$V.so
  [ $V = full ] && LIB=$PWD/reinmav-gym_amd/gym_reinmav_amd/librmav.so
  RMAV_LIB_PATH=$LIB timeout 300 python bench.py --mode step --steps 8000 --warmup 1000 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$V', round(j['roofline']['launch_ms_hip_events'] * 1e3, 3), 'us')"
done
done | tee $OUT/track_ablation.txt
