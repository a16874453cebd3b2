#!/bin/bash
# bench.py twice per variant on ONE box: the default form against the arguments given after the output name
#   bash scripts/gpu_bench_ab.sh <out> "<args A>" "<args B>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  i=0
  for a in "$@"; do
    i=$((i+1))
    f=$OUT/v${i}_$rep.json
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-frontend --steps 10 --warmup 3 $a 2>/dev/null | tail -1 > $f || exit 1
    echo "[$a]" $(python3 scripts/show_bench.py $f) | tee -a $OUT/ab.log
  done
done
