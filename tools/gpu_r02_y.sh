#!/bin/bash
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"]*1e3, "us/step")'
run() { echo "$1: $(env $2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary $3 2>/dev/null | grep '^{' | python -c "$P")"; }
echo "alone: $(python bench.py --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "$P")"
for Q in 4 8 2 16; do
run "signal word, high-priority comm stream, GPU_MAX_HW_QUEUES=$Q" "GPU_MAX_HW_QUEUES=$Q"
run "events,      high-priority comm stream, GPU_MAX_HW_QUEUES=$Q" "GPU_MAX_HW_QUEUES=$Q RMAV_EXCHANGE_EVENTS=1"
done
run "signal word, plain comm stream" "RMAV_COMM_STREAM_PRIORITY=0"
run "torch.distributed exchange" "RMAV_BENCH_EXCHANGE=torch"
