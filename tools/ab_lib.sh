#!/bin/bash
# A/B of two builds of librmav.so on the same box, alternating: bash tools/ab_lib.sh <tag> <old.so> "<bench args>" ["<bench args>" ...]
# -> gpurun_out/<tag>/ab.md (HIP-event launch time and roofline fraction per run; A = old library, B = the tree's)
TAG=$1; OLD=$2; shift 2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
echo "| bench arguments | build | us per launch (3 alternating runs) | frac |" > $OUT/ab.md; echo "|---|---|---|---|" >> $OUT/ab.md
for ARGS in "$@"; do
  for W in A B; do US=""; FR=""
    for i in 1 2 3; do
      L=$( [ $W = A ] && echo $OLD || echo "" )
      R=$(RMAV_LIB_PATH=$L timeout 300 python bench.py $ARGS --cpu-seconds 0 --no-secondary --detail - 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('%.2f %.3f' % (r['launch_ms_hip_events']*1e3, r['frac']))")
      US="$US ${R% *}"; FR="$FR ${R#* }"
    done
    echo "| $ARGS | $W | $US | $FR |" >> $OUT/ab.md
  done
done
cat $OUT/ab.md
