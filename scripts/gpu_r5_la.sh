#!/bin/bash
# bench.py with and without --lookahead, 4 and 5 resident workgroups per CU of the one-launch synthesis
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05la}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cfg in atlast_10k atlast_50k; do
for rep in 1 2; do
for v in "" "--lookahead" "--lookahead --synth-wgs-per-cu 4" "--synth-wgs-per-cu 4"; do
  timeout -k 10 300 python3 bench.py --config $cfg --no-cpu-baseline --no-frontend --steps 20 --warmup 5 $v > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python3 -c "
import json; r = json.load(open('$OUT/b.json')); print('$cfg', '$v'.ljust(40), 'ms_per_step %.3f' % r['ms_per_step'], 'kernel %.3f' % r['roofline']['ms_per_launch'])" | tee -a $OUT/la.log
done
done
done
