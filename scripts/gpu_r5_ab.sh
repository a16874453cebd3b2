#!/bin/bash
# Round 5: the GPU tier, then an A/B of the builds under scripts/ab/ (exp_synth.py: one-launch, pipelined, serial) on
# atlast_10k and the per-GPU share of atlast_50k.   scripts/gpu_r5_ab.sh <tag> [notest]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ "$2" != "notest" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
  rc=$?; tail -3 $OUT/pytest.log
  [ $rc -eq 0 ] || exit $rc
fi
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do
  SYNTH_HEADS=6 SYNTH_WGS=2 MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_synth.py atlast_10k 512 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s|^|$lib |" | tee -a $OUT/synth_ab.log || exit 1
done
done
for lib in scripts/ab/libmrx_*.so; do
  SYNTH_HEADS=0 SYNTH_WGS=3 MRX_LIB_PATH=$lib timeout -k 10 300 python3 scripts/exp_synth.py atlast_50k 512 2>&1 | grep -v "Warn\|amdgpu.ids" | sed "s|^|$lib |" | tee -a $OUT/synth_ab.log || exit 1
done
