#!/bin/bash
# Round 5: the one-launch form's tests, then sweeps of its knobs (scripts/exp_synth.py) on atlast_10k and atlast_50k's share.
#   scripts/gpu_r5_synth.sh <tag> [tests|notest] [lib]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05b}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$2" != "notest" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_synthesize.py -x -q > $OUT/pytest_synth.log 2>&1
  rc=$?; tail -15 $OUT/pytest_synth.log
  [ $rc -eq 0 ] || exit $rc
fi
SYNTH_WGS=${SYNTH_WGS:-0,1,2,3,8} SYNTH_CHUNK=${SYNTH_CHUNK:-16,32,64} timeout -k 10 300 python3 scripts/exp_synth.py atlast_10k 0 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $OUT/synth.log || exit 1
SYNTH_WGS=${SYNTH_WGS:-0,1,2,3,8} SYNTH_CHUNK=${SYNTH_CHUNK50:-16,32} timeout -k 10 400 python3 scripts/exp_synth.py atlast_50k 0 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $OUT/synth.log || exit 1
