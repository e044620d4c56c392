#!/bin/bash
OUT=gpurun_out/r02_k; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_bench.py tests/test_gpu_boundary.py -x -q 2>&1 | tail -5
for EX in native torch; do
  RMAV_BENCH_EXCHANGE=$EX timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --envs-per-gpu 131072 --steps 1000 --warmup 100 > $OUT/bench_dist1_$EX.json 2> $OUT/bench_dist1_$EX.err
  python - <<PY
import json
j = json.loads([l for l in open("$OUT/bench_dist1_$EX.json") if l.startswith("{")][0]); r = j["roofline"]
print("$EX", round(j["value"] / 1e9, 2), "G/s", round(j["ms_per_step"] * 1e3, 2), "us/step wall", round(r["launch_ms_hip_events"] * 1e3, 2), "us (events)", "frac", round(r["frac"], 3))
PY
done
timeout 300 python bench.py --envs-per-gpu 131072 --steps 1000 --warmup 100 --no-secondary --cpu-seconds 0 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('no exchange', round(j['ms_per_step'] * 1e3, 2), 'us/step')"
