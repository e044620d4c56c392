#!/bin/bash
for N in 65536 131072 262144 1048576; do
  for M in rollout step; do
    python bench.py --no-secondary --cpu-seconds 0 --envs-per-gpu $N --mode $M --steps $([ $M = step ] && echo 1000 || echo 60) --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N=$N $M', round(d['roofline']['launch_ms_hip_events']*1e3,2), 'us', round(d['roofline']['frac'],3), round(d['value']/1e9,2),'G/s')"
  done
done
