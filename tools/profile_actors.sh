#!/bin/bash
# rocprofv3 evidence for the policy-in-kernel rollouts (C5's per-GPU shape: quadrotor3d, 65 536 envs x 32 steps):
#   kernel trace + stats of tools/actor_bench.py (every actor), then separate --pmc passes of SQ instruction / cycle counters.
# Usage (on the GPU box, via gpurun):  bash tools/profile_actors.sh <tag>     -> gpurun_out/<tag>/actors/{actors_sq.md, actor_instr_mix.json, *.csv}
TAG=${1:-r04_actors}; OUT=$PWD/gpurun_out/$TAG/actors; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
CMD="python $REPO/tools/actor_bench.py"
export ITERS=20
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
cd $REPO
python tools/parse_actor_profiles.py $OUT > $OUT/actors_sq.md; cat $OUT/actors_sq.md
# keep the summaries, drop the bulky raw traces
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
