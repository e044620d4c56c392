#!/bin/bash
# rocprofv3 evidence of one round: kernel trace + stats of the bench commands, then separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950: TCC has 4 slots, they need 3 + 2; no --stats with --pmc).
# Usage (via gpurun, from the repo root): bash tools/profile_round.sh <tag> [round]     e.g.  r03_p r03
# Writes gpurun_out/<tag>/prof/{summary.md, traffic.json, trace_*, pmc_*}; tools/collect_profiles.sh copies the judged files.
TAG=${1:-r06_p}; ROUND=${2:-r06}
OUT=$PWD/gpurun_out/$TAG/prof
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
B="python $REPO/bench.py --no-secondary --cpu-seconds 0 --detail -"
run() {  # name, rocprof args..., -- cmd
  local name=$1; shift
  echo "== $name"
  timeout 600 rocprofv3 "$@" > $OUT/$name.log 2>&1
  echo "rc=$? $(grep '^{' $OUT/$name.log | tail -1 | cut -c1-160)"
}
# name | bench arguments
CASES=(
  "c2_ring|"
  "c2_inplace|--in-place"
  "c2_step|--mode step --steps 3000 --warmup 300"
  "c2_step_lazy|--mode step --steps 3000 --warmup 300 --tune step_lazy=1"
  "step_262144|--mode step --envs-per-gpu 262144 --steps 2000 --warmup 300"
  "step_1048576|--mode step --envs-per-gpu 1048576 --steps 800 --warmup 200"
  "c3shard_ring|--envs-per-gpu 131072 --layout soa --steps 1000 --warmup 100"
  "c3shard_chunked|--envs-per-gpu 131072 --layout chunked --steps 1000 --warmup 100"
  "c4_ring|--kind quad3d_sl --envs-per-gpu 262144 --steps 400 --warmup 50"
)
# the secondary legs whose kernels have no bench case of their own: C4 with per-env constants (k_rollout<3, 5, 1> of THIS run is that leg only),
# ReinmavEnv (k_rollout<4, 2, 0>) and the policy-in-kernel rollouts of C5's per-GPU shape (k_rollout<2, 3|8|4, 0>, k_rollout_pair<2, *>,
# k_rollout_pair_shared<2>) - traced under the driver's own command line, so that the averages are comparable with other_modes.*
if [ -z "$ONLY" ] || [[ " $ONLY " =~ " legs " ]]; then
  run trace_legs --kernel-trace --stats --output-format csv -d $OUT/trace_legs -- python $REPO/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary c4_pe,reinmav,policy --detail -
fi
for c in "${CASES[@]}"; do
  name=${c%%|*}; args=${c#*|}
  [ -n "$ONLY" ] && [[ ! " $ONLY " =~ " $name " ]] && continue
  run trace_$name --kernel-trace --stats --output-format csv -d $OUT/trace_$name -- $B $args
  for C in FETCH_SIZE WRITE_SIZE; do
    # fewer launches under the counters; the warm-up still cycles the whole ring once
    run pmc_${name}_$C --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${name}_$C -- $B $args --steps 60 --warmup 30
  done
done
# calibration of the two counters on a known byte count far beyond the Infinity Cache (16 M envs, single step)
for C in FETCH_SIZE WRITE_SIZE; do
  run pmc_calib_$C --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_calib_$C -- $B --mode step --envs-per-gpu 16777216 --steps 6 --warmup 2
done
cd $REPO
python tools/parse_rocprof.py $OUT $ROUND > $OUT/summary.md 2>&1
tail -80 $OUT/summary.md
find $OUT -name "*_agent_info.csv" -delete
find $OUT -name "*kernel_trace.csv" -size +2M -delete
du -sh $OUT
