#!/bin/bash
# bench.py of every tree under scripts/ab/trees/ (scripts/ab_tree.sh) and of this tree, alternating on ONE box:
#   bash scripts/gpu_ab_trees.sh <tag> <reps> [bench.py args ...]     -> gpurun_out/<tag>/trees.log
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; REPS=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for rep in $(seq 1 $REPS); do
  for tree in $(ls -d $ROOT/scripts/ab/trees/*/ 2>/dev/null) $ROOT/; do
    cd $tree
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-frontend --steps 20 --warmup 5 "$@" > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
    python3 -c "
import json; r = json.load(open('$OUT/b.json')); sb = r['stage_ms']['serial_breakdown']
print('$(basename $tree)'.ljust(10), 'ms_per_step %.3f' % r['ms_per_step'], 'kernel %.3f' % r['roofline']['ms_per_launch'], 'frac %.3f' % r['roofline']['frac'],
      'writer behind the sampler %.3f ms = %.3f' % (sb['upsample_with_spline_solve'], r['roofline'].get('frac_alone', 0)), 'sampler %.3f' % sb['sample'])" | tee -a $OUT/trees.log
  done
done
