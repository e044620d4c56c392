#!/bin/bash
# Exercise bench.py's distributed path (RCCL init, barrier, all-gather, all-reduce) with one rank.
mkdir -p gpurun_out/dist1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 50 --warmup 5 --cpu-seconds 0 > gpurun_out/dist1/bench_torchrun.log 2>&1
echo "rc=$?"; tail -4 gpurun_out/dist1/bench_torchrun.log | cut -c1-1500
timeout 300 python bench.py --steps 100 --warmup 10 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | cut -c1-1600
