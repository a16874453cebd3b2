#!/bin/bash
# A/B of scripts/ab/libmrx_*.so through scripts/krj_bench.py (TOD synthesis in K_RJ), alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for i in 1 2 3; do
  for lib in scripts/ab/libmrx_*.so; do
    echo "$lib $(MRX_LIB_PATH=$ROOT/$lib timeout -k 10 300 python3 scripts/krj_bench.py 2>&1 | grep 'TOD synthesis')"
  done
done
