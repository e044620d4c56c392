#!/bin/bash
# usage: bash tools/run_probe.sh <tag> -- runs the latency probe for block sizes 64/128/256
TAG=${1:-probe}
mkdir -p gpurun_out/$TAG
for B in 256 128 64; do
  echo "== RMAV_BLOCK=$B"
  RMAV_BLOCK=$B timeout 900 python tools/step_latency.py quad3d 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/step_latency_b$B.txt
done
