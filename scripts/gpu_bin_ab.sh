#!/bin/bash
# scripts/bin_bench.py through every build under scripts/ab/, twice, on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do
  for n in "$@"; do
    MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/bin_bench.py $n 1 3 2>&1 | grep "^bin" | sed "s|^|$lib |" | cut -c1-130 | tee -a $OUT/bin_ab.log || exit 1
  done
done
done
