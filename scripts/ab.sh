#!/bin/bash
# A/B timing of two builds on ONE box (boxes differ by 5-10 % on HBM-bound kernels): build the
# working tree as variant B into build/ab/, check out <ref> (default HEAD) into a scratch
# worktree as variant A, and print the commands to run both under gpurun.
#   bash scripts/ab.sh [ref]    -> scripts/ab/libmrx_a.so (ref), scripts/ab/libmrx_b.so (working tree)
set -e
REF=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scripts/ab
rm -rf /tmp/mrx_ab_a && mkdir -p /tmp/mrx_ab_a
git -C $ROOT archive $REF maria_amd/csrc include | tar -x -C /tmp/mrx_ab_a
make -s -C /tmp/mrx_ab_a/maria_amd/csrc -j4 OUT=$ROOT/scripts/ab/libmrx_a.so OBJDIR=/tmp/mrx_ab_a/obj ROOT=/tmp/mrx_ab_a
make -s -C $ROOT/maria_amd/csrc -j4 OUT=$ROOT/scripts/ab/libmrx_b.so OBJDIR=$ROOT/build/ab_obj_b
ls -la $ROOT/scripts/ab/*.so
echo "run: MRX_LIB_PATH=scripts/ab/libmrx_a.so python3 <script> ; MRX_LIB_PATH=scripts/ab/libmrx_b.so python3 <script>"
