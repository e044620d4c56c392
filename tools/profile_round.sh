#!/bin/bash
# rocprofv3 evidence for one round: kernel trace + stats of the bench command, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots, they need 3 + 2).
# Usage (via gpurun, from the repo root): bash tools/profile_round.sh r01
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
B="python $REPO/bench.py --no-secondary --cpu-seconds 0"
run() {  # name, rocprof args..., -- cmd
  local name=$1; shift
  echo "== $name"
  timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1
  echo "rc=$? $(tail -1 $OUT/$name.log | cut -c1-300)"
}
# 1. kernel trace + stats of the exact default bench command (and of the step mode)
run trace_rollout --kernel-trace --stats --output-format csv -d $OUT/trace_rollout -- $B
run trace_step    --kernel-trace --stats --output-format csv -d $OUT/trace_step    -- $B --mode step --steps 2000 --warmup 50
# 2. PMC passes: HBM-side bytes of the dominant kernel (per dispatch)
for C in FETCH_SIZE WRITE_SIZE; do
  run pmc_rollout_$C --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_rollout_$C -- $B --steps 40 --warmup 20
  run pmc_step_$C    --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_step_$C    -- $B --mode step --steps 40 --warmup 4
  # calibration: same kernel, working set far beyond the 256 MiB Infinity Cache, known byte count
  run pmc_calib_$C   --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_calib_$C   -- $B --mode step --envs-per-gpu 16777216 --steps 6 --warmup 2
done
cd $REPO
python tools/parse_rocprof.py $OUT > $OUT/summary.md 2>&1
tail -60 $OUT/summary.md
# keep the merge small: drop raw per-dispatch traces beyond the stats/counter CSVs
find $OUT -name "*_agent_info.csv" -delete
du -sh $OUT
