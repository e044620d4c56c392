#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, bench (both modes), rocprofv3 kernel trace.
# Usage (from the repo root, via gpurun): bash tools/gpu_round.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -5 $OUT/bench.log
