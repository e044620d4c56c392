#!/bin/bash
# three-input xor (v_bitop3_b32) in the Philox rounds: parity, then same-box A/B against the previous build
OUT=gpurun_out/r02_z; mkdir -p $OUT
PREV=$PWD/reinmav-gym_amd/build/librmav_prev.so
echo "== parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_ppo.py -x -q 2>&1 | tail -2
for cfg in "quad3d 65536" "quad3d 131072" "quad3d_sl 262144" "quad2d 65536"; do set -- $cfg
for M in rollout compute; do for rep in 1 2; do
echo "prev: $(KIND=$1 N=$2 MODE=$M SECS=2 RMAV_LIB_PATH=$PREV python tools/power_probe.py 2>&1 | tail -1)"
echo "new:  $(KIND=$1 N=$2 MODE=$M SECS=2 python tools/power_probe.py 2>&1 | tail -1)"
done; done; done | tee $OUT/ab_xor3.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", round(d["roofline"]["frac"],3))'
for rep in 1 2 3; do
echo "step mode  prev: $(RMAV_LIB_PATH=$PREV python bench.py --mode step --steps 4000 --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")  new: $(python bench.py --mode step --steps 4000 --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")"
echo "C2 bench   prev: $(RMAV_LIB_PATH=$PREV python bench.py --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")  new: $(python bench.py --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")"
done | tee -a $OUT/ab_xor3.txt
