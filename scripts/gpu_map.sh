#!/bin/bash
# Map sampling / binning: tests, then timings.   scripts/gpu_map.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-map}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_map.py -x -q -m gpu > $OUT/pytest_map.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_map.log
timeout -k 10 300 python3 scripts/map_bench.py 10000 240000 3 2>&1 | grep -v amdgpu.ids | tee $OUT/map_bench.txt
for n in 1024; do
  timeout -k 10 300 python3 scripts/bin_bench.py $n 1 3 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bin_bench.txt
done
