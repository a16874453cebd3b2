#!/bin/bash
# SQ counters of the one-launch synthesis for every build under scripts/ab/ (exp_synth.py, SYNTH_ONLY).  <tag> [config]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05pmc}; CFG=${2:-atlast_10k}
cd $ROOT
for lib in scripts/ab/libmrx_*.so; do
  n=$(basename $lib .so)
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
    echo "== $n: $set"
    SYNTH_ONLY=1 MRX_LIB_PATH=$ROOT/$lib bash scripts/pmc_kernel.sh $TAG/$n "$set" exp_synth.py $CFG 0 | grep "atm_tod\|spline_upsample_fused\|atm_sample_px" || { tail -5 gpurun_out/$TAG/$n/log.txt; }
  done
done
