#!/bin/bash
# Runs tools/micro/issue_rate on the GPU box while sampling the shader clock (hwmon freq1_input); output -> gpurun_out/<tag>/issue_rate.txt
TAG=${1:-issue_rate}; OUT=gpurun_out/$TAG; mkdir -p $OUT
BIN=tools/micro/_build/issue_rate
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/micro/issue_rate.hip
F=$(ls /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
( while true; do [ -n "$F" ] && cat $F; sleep 0.05; done ) > $OUT/sclk_samples.txt &
SAMPLER=$!
timeout 600 $BIN ${GHZ:-2.4} > $OUT/issue_rate.txt 2>&1
kill $SAMPLER
sort -n $OUT/sclk_samples.txt | awk '{a[NR]=$1} END {if (NR) printf "sclk Hz: min %d median %d max %d (%d samples)\n", a[1], a[int((NR+1)/2)], a[NR], NR}' >> $OUT/issue_rate.txt
cat $OUT/issue_rate.txt
