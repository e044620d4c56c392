#!/bin/bash
OUT=$PWD/gpurun_out/ppo_prof; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD; cd /tmp
ITERS=10 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python $REPO/tools/ppo_bench.py > $OUT/log.txt 2>&1
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-90s calls %6s avg %10.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
