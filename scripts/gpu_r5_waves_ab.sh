#!/bin/bash
# A/B of the libraries under scripts/ab/ on the one launch (scripts/exp_synth.py, SYNTH_ONLY): atlast_10k, the 8-GPU shard of it,
# atlast_50k's share; two alternations.   scripts/gpu_r5_waves_ab.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05waves}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
run() { SYNTH_ONLY=1 SYNTH_CHUNK=0 SYNTH_WGS=2 MRX_LIB_PATH=$ROOT/$1 timeout -k 10 300 python3 scripts/exp_synth.py "${@:2}" 2>&1 | grep "median" | sed "s|^|$(basename $1) |; s|identical True (differing 0) flags 0 ||" | tee -a $OUT/ab.log; }
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do run $lib atlast_10k 0 || exit 1; done
done
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do SYNTH_DETS=1264 run $lib atlast_10k 0 || exit 1; done
done
for rep in 1 2; do
for lib in scripts/ab/libmrx_*.so; do run $lib atlast_50k 0 || exit 1; done
done
