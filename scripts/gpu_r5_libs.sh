#!/bin/bash
# A/B of the builds under scripts/ab/ through scripts/exp_synth.py (one-launch form only), two rounds, alternating.
#   scripts/gpu_r5_libs.sh <tag> [config...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05c}; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cfg in ${@:-atlast_10k atlast_50k}; do
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do
  SYNTH_ONLY=1 MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_synth.py $cfg 0 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s|^|$(basename $lib) |" | tee -a $OUT/libs.log || exit 1
done
done
done
