#!/bin/bash
# bench.py of the round-4 tree (build/r4tree: git archive of 2b903d5, built in place) against this tree's, alternating on one box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r05_vs_r4}
mkdir -p $OUT
for cfg in atlast_10k atlast_50k; do
for rep in 1 2 3; do
for tree in build/r4tree .; do
  cd $ROOT/$tree
  timeout -k 10 300 python3 bench.py --config $cfg --no-cpu-baseline --no-frontend --steps 20 --warmup 5 > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
  python3 -c "
import json; r = json.load(open('$OUT/b.json')); print('$cfg', '$tree'.ljust(14), 'ms_per_step %.3f' % r['ms_per_step'], 'kernel %.3f' % r['roofline']['ms_per_launch'], 'frac %.3f' % r['roofline']['frac'], 'writer alone %.3f' % r['roofline'].get('frac_alone', 0), r['stage_ms'].get('form', '')[:40])" | tee -a $OUT/vs.log
done
done
done
