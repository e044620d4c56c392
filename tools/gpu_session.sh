#!/bin/bash
# One GPU-box session made of named stages (round 3 on; replaces the per-session gpu_r02_*.sh scripts).
# Usage (from the repo root, via gpurun):  bash tools/gpu_session.sh <tag> <stage> [<stage> ...]
# Everything is written under gpurun_out/<tag>/; copy what is cited into profiles/.
#   base      device.txt, smoke(), pytest -m gpu
#   bench     bench.py with its defaults, and with the driver's --steps 20 --warmup 5
#   legs      the other single-GPU BASELINE configs as their own bench lines (C3's shard, C4)
#   sweep     kernel sweep: kinds x {random, controller} at SWEEP_N (default "65536 131072") -> sweep.md
#             (SWEEP_TUNE="split=0" adds rmav_set_tuning overrides; SWEEP_TAG names the output: sweep$SWEEP_TAG.md)
#   steplat   tools/step_latency.py (single-step launch latency by feature subset)
#   vecenv    bench.py's vecenv / gym1 legs only
#   sq        SQ instruction / wait counters of the default bench command (EXTRA= adds bench arguments)
#   prof      rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the bench cases -> summary.md, traffic.json
#   actors    rocprofv3 trace + SQ counters of the policy-in-kernel rollouts (tools/profile_actors.sh)
#   sqlegs    SQ counters of C3's shard and C4
#   throttle  amd-smi throttle residencies around 6 s of each workload (tools/throttle_probe.sh)
#   dist1     bench.py under torch.distributed.run with one rank (native exchange)
TAG=${1:?tag}; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
B="python $REPO/bench.py"

line() {  # print the headline of a bench JSON file
  python - "$1" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = j["roofline"]
    print("  ", sys.argv[1].split("/")[-1], round(j["value"] / 1e9, 2), "G/s", round(r["launch_ms_hip_events"] * 1e3, 2), "us/launch frac", round(r["frac"], 3))
    for k, v in j.get("legs", {}).items():
        print("     ", k, json.dumps(v)[:400])
    print("      line bytes:", len([l for l in open(sys.argv[1]) if l.startswith("{")][-1]), " cpu:", json.dumps(j.get("cpu_baseline", {}))[:200], json.dumps(j.get("cpu_mt", {})))
except Exception as e:
    print("  ", sys.argv[1], "ERR", e)
PY
}

for STAGE in "$@"; do
echo "==== stage $STAGE"
case $STAGE in
base)
  rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > $OUT/device.txt
  lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/device.txt
  timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
  timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
  ;;
bench)
  timeout 900 $B --secondary all --detail $OUT/bench_n1_detail.json > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "rc=$?"; line $OUT/bench_n1.json
  ( time timeout 900 $B --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_n1_k20_detail.json > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err ) 2>&1 | grep real; line $OUT/bench_n1_k20.json
  ;;
legs)
  timeout 600 $B --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary > $OUT/bench_c3shard.json 2>/dev/null; line $OUT/bench_c3shard.json
  timeout 600 $B --envs-per-gpu 131072 --layout soa --cpu-seconds 0 --no-secondary > $OUT/bench_c3shard_plain.json 2>/dev/null; line $OUT/bench_c3shard_plain.json
  timeout 600 $B --kind quad3d_sl --envs-per-gpu 262144 --steps 500 --warmup 100 --cpu-seconds 0 --no-secondary > $OUT/bench_c4.json 2>/dev/null; line $OUT/bench_c4.json
  ;;
sweep)
  SW=$OUT/sweep$SWEEP_TAG
  : > $SW.jsonl
  for ACT in ${SWEEP_ACT:-random controller}; do for K in ${SWEEP_K:-quad3d quad3d_sl quad2d quad2d_sl}; do for N in ${SWEEP_N:-65536 131072}; do
    S=$(( 65536 * 600 / N + 40 ))
    timeout 300 $B --kind $K --actions $ACT --envs-per-gpu $N --steps $S --warmup $((S/4)) --cpu-seconds 0 --no-secondary ${SWEEP_TUNE:+--tune $SWEEP_TUNE} 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(json.dumps({'actions': '$ACT', 'kind': '$K', 'n': $N, 'us': r['launch_ms_hip_events'] * 1e3, 'TBps': r['achieved'] / 1e3, 'frac': r['frac']}))" >> $SW.jsonl
  done; done; done
  python - $SW.jsonl > $SW.md <<'PY'
import json, sys
print("| actions | kind | envs | us per 64-step launch | TB/s | frac of 8 TB/s |\n|---|---|---|---|---|---|")
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(f"| {r['actions']} | {r['kind']} | {r['n']} | {r['us']:.1f} | {r['TBps']:.2f} | {r['frac']:.3f} |")
PY
  cat $SW.md
  ;;
steplat)
  timeout 900 python tools/step_latency.py quad3d > $OUT/step_latency.txt 2>&1; tail -30 $OUT/step_latency.txt
  ;;
vecenv)
  timeout 600 $B --steps 200 --warmup 50 --cpu-seconds 0 --secondary gym1,vecenv,step > $OUT/bench_vecenv.json 2> $OUT/bench_vecenv.err; line $OUT/bench_vecenv.json
  ;;
sq)
  EXTRA="$EXTRA" bash tools/pmc_sq.sh $TAG/sq > $OUT/sq_counters.txt 2>&1; tail -40 $OUT/sq_counters.txt
  ;;
prof)
  bash tools/profile_round.sh $TAG ${ROUND:-r06} 2>&1 | tail -60
  ;;
actors)
  bash tools/profile_actors.sh $TAG 2>&1 | tail -14
  ;;
sqlegs)   # SQ counters of C3's shard and of C4 (the default `sq` stage covers C2)
  EXTRA="--envs-per-gpu 131072" bash tools/pmc_sq.sh $TAG/sq_c3shard > $OUT/sq_counters_c3shard.txt 2>&1; tail -32 $OUT/sq_counters_c3shard.txt
  EXTRA="--kind quad3d_sl --envs-per-gpu 262144" bash tools/pmc_sq.sh $TAG/sq_c4 > $OUT/sq_counters_c4.txt 2>&1; tail -32 $OUT/sq_counters_c4.txt
  ;;
throttle)
  bash tools/throttle_probe.sh $TAG 2>&1 | grep -E "^(rollout|memset|compute)" 
  ;;
dist1)
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --envs-per-gpu 131072 > $OUT/bench_torchrun1_c3shard.json 2> $OUT/bench_dist1.err; echo "rc=$?"; line $OUT/bench_torchrun1_c3shard.json
  ;;
*) echo "unknown stage $STAGE";;
esac
done
du -sh $OUT
