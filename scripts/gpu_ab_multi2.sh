#!/bin/bash
# A/B of several library builds x resident workgroups per CU of the sampler beside the writer, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for i in 1 2; do
  for lib in scripts/ab/libmrx_*.so; do
    for w in ${WGS:-2 3 4 5}; do
      MRX_AB_RESIDENT_WGS=$w MRX_LIB_PATH=$lib timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline "$@" > $OUT/b.json 2> $OUT/b.err || exit 1
      python3 -c "import json,sys; r=json.load(open('$OUT/b.json')); print('$lib wgs $w', round(r['ms_per_step'],4), round(r['stage_ms']['tod_synthesis_pipelined'],4), 'writer', round(r['roofline']['ms_per_launch'],4))"
    done
  done
done
