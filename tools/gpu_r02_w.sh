#!/bin/bash
# hand-over every env-step + signal folded into the pack kernel: tests that touch them, the exchange's cost, the headline
OUT=gpurun_out/r02_w; mkdir -p $OUT
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us/step", round(d["roofline"]["launch_ms_hip_events"]*1e3,2), "us kernel", round(d["roofline"]["frac"],3))'
for rep in 1 2; do
echo "alone:    $(python bench.py --envs-per-gpu 131072 --steps 1000 --warmup 100 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | python -c "$P")"
for SK in 0 1; do
echo "exchange, signal kernel $SK: $(RMAV_EXCHANGE_SIGNAL_KERNEL=$SK timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$rep bench.py --gpus 1 --envs-per-gpu 131072 --steps 1000 --warmup 100 --cpu-seconds 0 --no-secondary 2>/dev/null | grep '^{' | tee $OUT/bench_torchrun1_$rep.json | python -c "$P")"
done
done | tee $OUT/exchange_cost.txt
for SK in 0 1; do RMAV_EXCHANGE_SIGNAL_KERNEL=$SK PROBE_SPLIT=1 PROBE_N=131072 python tools/contention_probe.py 2>&1 | grep -v amdgpu | tee $OUT/contention_$SK.md; done
echo "== tests"; timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_bench.py tests/test_gpu_multiprocess.py -x -q 2>&1 | tail -3
