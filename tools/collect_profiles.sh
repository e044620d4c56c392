#!/bin/bash
# Copy the judged evidence of a GPU session into profiles/<round>/ (tracked).
# Usage: bash tools/collect_profiles.sh <session tag> <round>      e.g.  r03_p r03
# Expects gpurun_out/<tag>/prof/ from `gpu_session.sh <tag> prof` and, optionally, bench_*.json / device.txt / sweep*.md /
# sq_counters.txt / step_latency.txt from the other stages of the same tag.
TAG=${1:?session tag}; ROUND=${2:?round}; F=gpurun_out/$TAG; P=$F/prof; D=profiles/$ROUND
mkdir -p $D/pmc
if [ -d $P ]; then
  for d in $P/trace_*; do
    [ -d "$d" ] || continue; c=${d##*/trace_}
    f=$(find $d -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d' ' -f2); [ -n "$f" ] && cp "$f" $D/trace_${c}_kernel_stats.csv
  done
  for d in $P/pmc_*; do
    [ -d "$d" ] || continue; c=${d##*/}
    best=""; for g in $(find $d -name "*counter_collection.csv"); do grep -q "k_rollout\|k_step" "$g" && best=$g; done
    [ -n "$best" ] && { head -1 "$best"; grep "k_rollout\|k_step" "$best"; } > $D/pmc/$c.csv
  done
  cp $P/summary.md $D/rocprofv3_summary.md
  cp $P/traffic.json profiles/traffic.json
fi
for f in $F/bench_*.json; do [ -f "$f" ] && grep '^{' $f > $D/$(basename $f).log; done
for f in device.txt sweep.md sweep.jsonl sq_counters.txt step_latency.txt; do [ -f $F/$f ] && cp $F/$f $D/$f; done
ls $D
