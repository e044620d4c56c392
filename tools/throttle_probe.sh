#!/bin/bash
# Which limiter holds the 131 072-env rollout launch in its "decayed" state?  
# For each workload: amd-smi's throttle-residency accumulators (PPT / socket thermal / VR thermal / HBM thermal / PROCHOT) before and
# after ~6 s of back-to-back launches, with the launch-time / power / clock time series of tools/power_probe.py in between.
# Usage (GPU box): bash tools/throttle_probe.sh <tag>  ->  gpurun_out/<tag>/throttle/
TAG=${1:-r04_throttle}; OUT=$PWD/gpurun_out/$TAG/throttle; mkdir -p $OUT
SMI=/opt/rocm/bin/amd-smi
snap() { $SMI metric --throttle --power --clock --temperature --json > $OUT/smi_$1.json 2> $OUT/smi_$1.err; }
$SMI metric --help > $OUT/smi_help.txt 2>&1
snap idle
i=0
for W in "rollout 131072 quad3d" "rollout 65536 quad3d" "memset 131072 quad3d" "compute 131072 quad3d" "rollout 262144 quad3d_sl" "rollout 131072 quad3d"; do
  set -- $W; i=$((i+1))
  snap before_$i
  MODE=$1 N=$2 KIND=$3 SECS=${SECS:-6} SERIES=$OUT/series_${i}_$1_$2_$3.txt SHOW_CAP=$([ $i = 1 ] && echo 1) timeout 300 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/power_probe.txt
  snap after_$i
  sleep 2
done
python tools/parse_throttle.py $OUT | tee $OUT/throttle.md
