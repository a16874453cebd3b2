#!/bin/bash
# A/B on one box: geometric block shares of the pipelined step (a short first block, then growing), alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do
  for sh in ${SHARES:-1,1,1,1 6,9,13,19,26,27 5,8,12,18,27,30 10,15,22,26,27 8,16,24,26,26 10,20,30,40}; do
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline --block-shares $sh > $OUT/b.json 2> $OUT/b.err || exit 1
    python3 -c "import json,sys; r=json.load(open('$OUT/b.json')); print('$sh', round(r['ms_per_step'],4), round(r['stage_ms']['tod_synthesis_pipelined'],4), round(r['roofline']['frac'],4))"
  done
done
