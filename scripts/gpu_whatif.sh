#!/bin/bash
# A/B of the builds under scripts/ab/ through scripts/exp_whatif.py, both configurations
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for lib in scripts/ab/libmrx_*.so; do
  MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_whatif.py atlast_10k 1 4 2>&1 | grep -v Warning | tee -a $OUT/whatif.log || exit 1
  MRX_LIB_PATH=$lib timeout -k 10 400 python3 scripts/exp_whatif.py atlast_50k 1 4 8 2>&1 | grep -v Warning | tee -a $OUT/whatif.log || exit 1
done
