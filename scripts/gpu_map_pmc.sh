#!/bin/bash
# Instruction mix of map_sample_kernel inside Simulation(map=...).run() (scripts/frontend_trace.py map): per-type VALU counts,
# issue and wait cycles, per launch.  Separate --pmc passes, --kernel-trace only.  <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-mappmc}
cd $ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  echo "== $set"
  bash scripts/pmc_kernel.sh $TAG "$set" frontend_trace.py map | grep "map_sample\|atm_tod" || { tail -5 gpurun_out/$TAG/log.txt; exit 1; }
done
