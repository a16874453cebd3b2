#!/bin/bash
# Whole-tree A/B (a bisect over commits whose C ABI differs from today's Python): unpack each <ref> into
# scripts/ab/trees/<ref>/ (bench.py, the package, the header, the oracle bench.py imports) and build its library in place.
#   bash scripts/ab_tree.sh <ref> [<ref> ...]      then, under gpurun:  bash scripts/gpu_ab_trees.sh <tag> <reps> [bench.py args]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for REF in "$@"; do
  D=$ROOT/scripts/ab/trees/$REF
  rm -rf $D && mkdir -p $D
  git -C $ROOT archive $REF bench.py maria_amd include oracle scripts/__init__.py scripts/kbench.py profiles/README.md | tar -x -C $D
  make -s -C $D/maria_amd/csrc -j8 ROOT=$D OUT=$D/maria_amd/libmrx.so OBJDIR=/tmp/mrx_tree_$REF
  ls -la $D/maria_amd/libmrx.so
done
