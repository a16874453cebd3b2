#!/bin/bash
# three bench lines in a row on one box (run-to-run scatter of the step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-frontend "$@" > $OUT/bench$i.json 2> $OUT/bench$i.err || exit 1
  python3 -c "import json,sys; r=json.load(open('$OUT/bench$i.json')); print(round(r['ms_per_step'],4), {k:(round(v,4) if isinstance(v,float) else v) for k,v in r['stage_ms'].items() if k!='serial_breakdown'}, round(r['roofline']['frac'],4), round(r['roofline']['frac_alone'],4))"
done
