#!/bin/bash
# armed statistics exchange (the rollout launch writes the snapshot + arrival words): tests, then its cost inside bench.py
OUT=gpurun_out/r02_aa; mkdir -p $OUT
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_bench.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
P='import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us/step", d["config"].get("exchange_equals_plain_all_gather"))'
run() { echo "$1: $(env $2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary $3 2>/dev/null | grep '^{' | python -c "$P")"; }
for rep in 1 2; do
echo "alone: $(python bench.py --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "$P")"
run "armed (snapshot + arrival words in the rollout kernel)" "RMAV_BENCH_ARM=1"
run "pack + signal kernels" "RMAV_BENCH_ARM=0"
run "armed, 80 us stand-in collective" "RMAV_BENCH_ARM=1 RMAV_DBG_EXCHANGE=3"
run "pack + signal, 80 us stand-in collective" "RMAV_BENCH_ARM=0 RMAV_DBG_EXCHANGE=3"
done | tee $OUT/arm_ab.txt
echo "alone C2: $(python bench.py --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "$P")" | tee -a $OUT/arm_ab.txt
