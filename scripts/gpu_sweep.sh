#!/bin/bash
# Pipelined-run sweep (blocks x resident sampler workgroups per CU x steps per thread) on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sweep}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for KT in 1 2; do
  echo "== steps per thread $KT" | tee -a $OUT/sweep.log
  MRX_SAMPLE_TIMES=$KT MRX_BLOCKS=${MRX_BLOCKS:-4,8,12} timeout -k 10 300 python3 scripts/exp_pipe_sweep.py 2>&1 | tee -a $OUT/sweep.log
done
