#!/bin/bash
# SQ-side PMC passes for the default bench command (where do the wave cycles go?)
TAG=${1:-sq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|GRBM_[A-Z_]+|TCP_[A-Z0-9_]+|TCC_[A-Z0-9_]+)\b" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
B="python $REPO/bench.py --no-secondary --cpu-seconds 0 --steps 60 --warmup 40 $EXTRA"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- $B > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $REPO
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"] or "k_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k][len(acc[k]) // 4:]
    print(f"{k:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
