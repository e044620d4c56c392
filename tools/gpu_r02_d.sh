#!/bin/bash
# round 2, GPU session D: single-step latency decomposition with the arena allocation + k_step
TAG=${1:-r02_d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/step_latency.py quad3d 2>&1 | grep -v amdgpu.ids > $OUT/step_latency_kstep.txt
RMAV_STEP_KERNEL=0 timeout 600 python tools/step_latency.py quad3d 2>&1 | grep -v amdgpu.ids > $OUT/step_latency_krollout.txt
paste $OUT/step_latency_kstep.txt $OUT/step_latency_krollout.txt | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9, "| k_rollout:", $(NF-1)}'
for R in 64 512; do
  for K in quad3d quad3d_sl; do
    timeout 300 python bench.py --kind $K --mode step --steps 8000 --warmup 1000 --cpu-seconds 0 --no-secondary --action-ring $R 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$K ring $R', j['roofline']['launch_ms_hip_events'] * 1e3, 'us', j['roofline']['frac'])"
  done
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/lf tools/micro/launch_floor.hip && /tmp/lf | tee $OUT/launch_floor.txt
