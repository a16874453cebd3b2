#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05gauss
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 scripts/gauss_bench.py 10 2>&1 | grep -v "Warn\|amdgpu.ids" > $OUT/gauss_bench.txt || { tail -5 $OUT/gauss_bench.txt; exit 1; }
grep -v "^\[{" $OUT/gauss_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o run -- python3 $ROOT/scripts/gauss_bench.py 3 > $OUT/prof.log 2>&1
S=$(ls $OUT/prof/*/run_kernel_stats.csv $OUT/prof/run_kernel_stats.csv 2>/dev/null | head -1)
cp $S $OUT/gauss_kernel_stats.csv; head -8 $OUT/gauss_kernel_stats.csv | cut -c1-200
