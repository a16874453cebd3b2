#!/bin/bash
# every build under scripts/ab/ through scripts/exp_wgs_blocks.py, both configurations
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for lib in scripts/ab/libmrx_*.so; do
  MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_wgs_blocks.py atlast_10k 3,4 4 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/log.txt || exit 1
  MRX_LIB_PATH=$lib timeout -k 10 400 python3 scripts/exp_wgs_blocks.py atlast_50k 3,4,5 8,12 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $OUT/log.txt || exit 1
done
