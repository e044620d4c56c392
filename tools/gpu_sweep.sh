#!/bin/bash
# Bench sweep over kinds / batch sizes / chunk lengths (one JSON line each) -> gpurun_out/<tag>/sweep.jsonl
TAG=${1:-sweep}
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/sweep.jsonl
: > $OUT
b() { timeout 300 python bench.py --cpu-seconds 0 --no-secondary "$@" 2>/dev/null | grep '^{' >> $OUT; }
# launch counts sized so every line times >= ~50 ms of GPU work after >= ~10 ms of warm-up (clocks ramp for ~5 ms)
steps_for() { echo $(( 65536 * 1500 / $1 + 100 )); }
for N in 65536 131072 262144 1048576 4194304; do
  S=$(steps_for $N)
  b --kind quad3d --envs-per-gpu $N --mode rollout --chunk 64 --steps $S --warmup $((S / 5))
  b --kind quad3d --envs-per-gpu $N --mode step --steps $((S * 4)) --warmup $S
done
for T in 8 16 32 128; do b --kind quad3d --envs-per-gpu 65536 --mode rollout --chunk $T --steps $((96000 / T)) --warmup $((19200 / T)); done
for K in quad3d_sl quad2d quad2d_sl; do
  for N in 65536 262144 1048576; do
    S=$(steps_for $N)
    b --kind $K --envs-per-gpu $N --mode rollout --chunk 64 --steps $S --warmup $((S / 5))
    b --kind $K --envs-per-gpu $N --mode step --steps $((S * 4)) --warmup $S
  done
done
python tools/print_sweep.py $OUT
