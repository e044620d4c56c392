#!/bin/bash
# cost of the per-launch statistics exchange inside bench.py (one rank under torch.distributed.run, 131 072 envs)
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"]*1e3, "us/step")'
run() { echo "$1: $(env $2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary $3 2>/dev/null | grep '^{' | python -c "$P")"; }
for rep in 1 2; do
echo "alone: $(python bench.py --envs-per-gpu 131072 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "$P")"
run "exchange every launch" "A=1"
run "no collective (pack + signal + waits only)" "RMAV_DBG_EXCHANGE=1"
run "device-side buffer wait" "RMAV_EXCHANGE_DEVICE_WAIT=1"
run "exchange every 1000000th launch (torchrun, no posts)" "A=1" "--exchange-every 1000000"
run "exchange every 4th launch" "A=1" "--exchange-every 4"
done
