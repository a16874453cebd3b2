#!/bin/bash
# kernel times of the bucketed binning with what-if builds (scripts/ab/libmrx_*.so): bash scripts/exp_bin_variants.sh <map n> <lib>...
N=$1; shift
for v in "$@"; do
  export MRX_LIB_PATH=${GRAFT_REPO_ROOT:-$(pwd)}/$v
  bash scripts/prof_kbench.sh binv bin_bench.py $N 1 2 | grep "bin_" | sed "s|^|$v: |"
done
