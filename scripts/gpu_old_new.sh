#!/bin/bash
# the tree of an earlier commit (scripts/ab/old, built by hand) against this one, bench.py twice each on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for a in "$@"; do
    for tree in old new; do
      dir=$ROOT; [ $tree = old ] && dir=$ROOT/scripts/ab/old
      f=$OUT/${tree}_$(echo $a | tr -c 'a-z0-9_\n' '_')_$rep.json
      (cd $dir && timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-frontend --steps 10 --warmup 3 $a 2>/dev/null | tail -1 > $f) || exit 1
      echo "[$tree $a]" $(cd $ROOT && python3 scripts/show_bench.py $f) | tee -a $OUT/ab.log
    done
  done
done
