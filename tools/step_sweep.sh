#!/bin/bash
# k_step (one env-step per launch) by batch size and launch option: HIP-event launch time, fraction of the 8 TB/s roofline on the
# 101 algorithmic bytes.  Usage (GPU box): bash tools/step_sweep.sh <tag>   -> gpurun_out/<tag>/step_sweep.md
TAG=${1:-step_sweep}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
echo "| envs | tuning | us per launch | G env-steps/s | frac of 8 TB/s (101 B) |" > $OUT/step_sweep.md
echo "|---|---|---|---|---|" >> $OUT/step_sweep.md
for N in ${SWEEP_N:-65536 131072 262144 1048576 4194304}; do
  for T in ${SWEEP_T:-"" step_lazy=1 block=64 block=128 step_store=1 step_store=2 step_lazy=1,step_store=2 step_lazy=1,block=128}; do
    S=$(( 4000 * 65536 / N + 300 ))
    timeout 300 python bench.py --mode step --kind ${KIND:-quad3d} --envs-per-gpu $N --steps $S --warmup 200 --cpu-seconds 0 --no-secondary --detail - ${T:+--tune $T} 2>/dev/null | grep '^{' | \
      python -c "import json,sys; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('| $N | ${T:-default} | %.2f | %.2f | %.3f |' % (r['launch_ms_hip_events']*1e3, j['value']/1e9, r['frac']))" >> $OUT/step_sweep.md
  done
done
cat $OUT/step_sweep.md
