#!/bin/bash
# Bench sweep over kinds / batch sizes / chunk lengths (one JSON line each) -> gpurun_out/<tag>/sweep.jsonl
TAG=${1:-sweep}
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/sweep.jsonl
: > $OUT
b() { timeout 300 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | grep '^{' >> $OUT; }
for N in 65536 131072 262144 1048576 4194304; do
  b --kind quad3d --envs-per-gpu $N --mode rollout --chunk 32 --steps 100 --warmup 10
  b --kind quad3d --envs-per-gpu $N --mode step --steps 1000 --warmup 50
done
for T in 8 16 64 128; do b --kind quad3d --envs-per-gpu 65536 --mode rollout --chunk $T --steps 100 --warmup 10; done
for K in quad3d_sl quad2d quad2d_sl; do
  for N in 65536 262144 1048576; do
    b --kind $K --envs-per-gpu $N --mode rollout --chunk 32 --steps 100 --warmup 10
    b --kind $K --envs-per-gpu $N --mode step --steps 1000 --warmup 50
  done
done
python tools/print_sweep.py $OUT
