#!/bin/bash
OUT=gpurun_out/r02_p; mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
for N in 65536 131072; do
  timeout 300 python bench.py --envs-per-gpu $N --cpu-seconds 0 --no-secondary | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print($N, round(r['launch_ms_hip_events'] * 1e3, 2), 'us', 'frac', round(r['frac'], 3))"
done
RMAV_BENCH_EXCHANGE=native timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --envs-per-gpu 131072 --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print('torchrun 1 rank, 131072, native exchange', round(r['launch_ms_hip_events'] * 1e3, 2), 'us', 'frac', round(r['frac'], 3))"
