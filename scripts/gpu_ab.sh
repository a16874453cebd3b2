#!/bin/bash
# ONE parametrised A/B runner (round 6; it replaces the gpu_r5_*_ab.sh one-offs): every library under scripts/ab/
# (scripts/ab.sh builds libmrx_a.so from a git ref and libmrx_b.so from the working tree; any further libmrx_<x>.so
# dropped there joins in) runs the same command in turn, <reps> times over, on ONE box -- the pool's boxes differ by
# +-5 %, so only alternating runs on one box compare builds.  Every output line is prefixed with the library's name
# and appended to gpurun_out/<tag>/ab.log.
#
#   scripts/gpu_ab.sh <tag> <reps> [--grep REGEX] -- <command ...>
#   e.g. gpurun -- 'bash scripts/gpu_ab.sh r06map 3 --grep map_sample -- python3 scripts/map_bench.py 10000 240000 3'
#
# The command sees MRX_LIB_PATH (maria_amd/_lib.py loads that library).  A failing run ends the script (no retries on a
# GPU box).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; REPS=$2; shift 2
GREP=.
if [ "$1" == "--grep" ]; then GREP=$2; shift 2; fi
[ "$1" == "--" ] && shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
echo "# $*" >> $OUT/ab.log
for rep in $(seq 1 $REPS); do
  for lib in scripts/ab/libmrx_*.so; do
    name=$(basename $lib .so)
    MRX_LIB_PATH=$ROOT/$lib timeout -k 10 600 "$@" > $OUT/run.out 2> $OUT/run.err || { tail -20 $OUT/run.err; echo "FAILED: $name rep $rep"; exit 1; }
    grep -E "$GREP" $OUT/run.out | sed "s|^|$name |" | tee -a $OUT/ab.log
  done
done
