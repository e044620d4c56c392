#!/bin/bash
# hand-over every env-step (8.7 KB of LDS per pair) vs every 2 (17 KB): alone on cold rings, and next to a stand-in collective
OUT=gpurun_out/r02_v; mkdir -p $OUT
CH1=$PWD/reinmav-gym_amd/build/librmav_ch1.so
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["launch_ms_hip_events"]*1e3,2), "us", round(d["roofline"]["frac"],3))'
for args in "--kind quad3d" "--kind quad3d --envs-per-gpu 131072" "--kind quad2d" "--kind quad2d --envs-per-gpu 131072" "--kind quad3d --actions controller" "--kind quad3d --actions controller --envs-per-gpu 131072"; do
  for rep in 1 2 3; do
    echo "[$args] CH=2: $(python bench.py $args --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")   CH=1: $(RMAV_LIB_PATH=$CH1 python bench.py $args --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")"
  done
done | tee $OUT/alone.txt
echo "== CH=2" | tee $OUT/contention.md
PROBE_N=131072 PROBE_SPLIT=1 python tools/contention_probe.py 2>&1 | grep -v amdgpu | tee -a $OUT/contention.md
echo "== CH=1" | tee -a $OUT/contention.md
RMAV_LIB_PATH=$CH1 PROBE_N=131072 PROBE_SPLIT=1 python tools/contention_probe.py 2>&1 | grep -v amdgpu | tee -a $OUT/contention.md
