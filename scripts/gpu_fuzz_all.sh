#!/bin/bash
# Every randomised sweep with fresh seeds, one after the other (development aid; under gpurun).
# Usage: bash scripts/gpu_fuzz_all.sh <first seed> [trials]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-100}; N=${2:-25}
OUT=$ROOT/gpurun_out/fuzz_all_$S
mkdir -p $OUT
cd $ROOT
for f in fuzz_frontend fuzz_map_frontend fuzz_noise fuzz_small_kernels fuzz_round3 fuzz_new_kernels; do
  for s in $S $((S+1)); do
    timeout -k 10 600 python3 scripts/$f.py $s $N > $OUT/${f}_$s.log 2>&1
    echo "$f seed $s: rc=$? $(tail -1 $OUT/${f}_$s.log)"
    grep -A2 "BAD" $OUT/${f}_$s.log | grep -v "^all ok" | head -12
  done
done
