#!/bin/bash
# Runs tools/hazard/probe.py on every variant library under tools/hazard/_build (built by build_variant.sh) -> gpurun_out/<tag>/hazard.txt
TAG=${1:-hz}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
: > $OUT/hazard.txt
RMAV_LIB_PATH= python tools/hazard/probe.py >> $OUT/hazard.txt 2>$OUT/hazard.err
for d in ${VARIANTS:-$(ls tools/hazard/_build)}; do
  RMAV_LIB_PATH=$PWD/tools/hazard/_build/$d/librmav.so timeout 300 python tools/hazard/probe.py >> $OUT/hazard.txt 2>>$OUT/hazard.err || echo "$d FAILED rc=$?" >> $OUT/hazard.txt
done
cat $OUT/hazard.txt
