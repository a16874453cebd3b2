#!/bin/bash
# SQ / TA / TCP / TCC counters of the one-launch synthesis, the stand-alone writer and sampler (exp_synth.py, SYNTH_ONLY).  <tag> [config] [lib]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-synthpmc}; CFG=${2:-atlast_10k}; LIB=${3:-maria_amd/libmrx.so}
cd $ROOT
# (round 6: at most FOUR counters of the TA / TCP / TD / TCC blocks a pass -- round 5's passes of eight "did not collect")
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_INT32" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TC_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum"; do
  echo "== $set"
  SYNTH_ONLY=1 MRX_LIB_PATH=$ROOT/$LIB bash scripts/pmc_kernel.sh $TAG "$set" exp_synth.py $CFG 0 | grep "atm_tod\|spline_upsample_fused\|atm_sample_px" || { tail -5 gpurun_out/$TAG/log.txt; }
done
