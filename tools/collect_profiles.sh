#!/bin/bash
# copy the judged evidence of tools/gpu_r02_final.sh <tag> (gpurun_out/<tag>, gpurun_out/prof_r02c) into profiles/r02
TAG=${1:-r02_final}; P=gpurun_out/prof_r02c; F=gpurun_out/$TAG; D=profiles/r02
for c in c2_inplace c2_ring c2_step c3shard_ring c4_ring; do
  f=$(find $P/trace_$c -name "*kernel_stats.csv" -printf "%s %p\n" | sort -n | tail -1 | cut -d' ' -f2); cp "$f" $D/trace_${c}_kernel_stats.csv
done
for c in c2_inplace c2_ring c2_step c3shard_ring c4_ring calib; do for C in FETCH_SIZE WRITE_SIZE; do
  best=""; for g in $(find $P/pmc_${c}_$C -name "*counter_collection.csv"); do grep -q "k_rollout\|k_step" "$g" && best=$g; done
  [ -n "$best" ] && { head -1 "$best"; grep "k_rollout\|k_step" "$best"; } > $D/pmc/pmc_${c}_$C.csv
done; done
cp $P/summary.md $D/rocprofv3_summary.md; cp $P/traffic.json profiles/traffic.json
for f in bench_n1 bench_n1_k20 bench_c3shard bench_c4 bench_torchrun1_c3shard; do grep '^{' $F/$f.json > $D/$f.json.log; done
cp $F/device.txt $D/device.txt
