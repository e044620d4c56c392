#!/bin/bash
mkdir -p gpurun_out/ppo
timeout 900 python -m pytest tests/test_gpu_ppo.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15
timeout 600 python tools/ppo_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ppo/ppo_bench.jsonl
