#!/bin/bash
# A/B of the builds under scripts/ab/ through scripts/exp_synth.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do
  MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_synth.py "$@" 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s|^|$lib |" | tee -a $OUT/synth_ab.log || exit 1
done
done
