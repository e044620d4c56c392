#!/bin/bash
# Same-box A/B of two builds of librmav.so: the in-tree one vs $1 (default reinmav-gym_amd/build/librmav_prev.so).
PREV=${1:-$PWD/reinmav-gym_amd/build/librmav_prev.so}
P='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"]*1e3,2), "us", round(d["value"]/1e9,2), "G/s", round(d["roofline"]["frac"],3))'
for args in "${@:2}" "--kind quad3d" "--kind quad2d" "--kind quad3d --envs-per-gpu 131072" "--kind quad3d --envs-per-gpu 1048576" "--kind quad3d --mode step --steps 2000" "--kind quad3d --actions controller"; do
  for rep in 1 2; do
    echo "[$args] prev: $(RMAV_LIB_PATH=$PREV python bench.py $args --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")   new: $(python bench.py $args --cpu-seconds 0 --no-secondary 2>/dev/null | tail -1 | python -c "$P")"
  done
done
