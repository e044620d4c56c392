#!/bin/bash
# kernel trace of bench.py with the per-launch statistics exchange (one rank under torch.distributed.run): which kernels the
# exchange adds, how long they run, and how much of the collective's kernel overlaps the next rollout
OUT=$PWD/gpurun_out/prof_exchange; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 $REPO/bench.py --gpus 1 --envs-per-gpu 131072 --steps 400 --warmup 50 --cpu-seconds 0 --no-secondary > $OUT/run.log 2>&1
echo "rc=$? $(grep '^{' $OUT/run.log | tail -1 | cut -c1-160)"
cd $REPO
python - <<PY
import csv, glob, os
best = max(glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = list(csv.DictReader(open(best)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
names = {}
for s, e, n, q in ev:
    k = n.split("(")[0][:70]
    d = names.setdefault((k, q), [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
print("| kernel | queue | calls | avg us |\n|---|---|---|---|")
for (k, q), (c, t) in sorted(names.items(), key=lambda x: -x[1][1])[:10]:
    print(f"| {k} | {q} | {c} | {t / c:.2f} |")
roll = [(s, e) for s, e, n, q in ev if "k_rollout" in n]
coll = [(s, e) for s, e, n, q in ev if "nccl" in n.lower() or "rccl" in n.lower() or "AllGather" in n]
if coll:
    import bisect
    starts = [s for s, e in roll]
    ov = tot = 0
    for s, e in coll:
        tot += e - s
        i = max(0, bisect.bisect_right(starts, s) - 1)
        for rs, re_ in roll[i:i + 3]:
            ov += max(0, min(e, re_) - max(s, rs))
    print(f"\ncollective kernels: {len(coll)}, mean {tot / len(coll) / 1e3:.2f} us, {100 * ov / tot:.0f} % of their time inside a rollout kernel's interval")
gaps = [roll[i + 1][0] - roll[i][1] for i in range(len(roll) - 1)]
gaps.sort()
print(f"rollout kernels: {len(roll)}, mean {sum(e - s for s, e in roll) / len(roll) / 1e3:.2f} us; gap between consecutive rollouts: median {gaps[len(gaps) // 2] / 1e3:.2f} us, p90 {gaps[int(len(gaps) * .9)] / 1e3:.2f} us")
PY
