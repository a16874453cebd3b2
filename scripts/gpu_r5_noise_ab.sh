#!/bin/bash
# A/B of the libraries under scripts/ab/ on the noise generator (scripts/noise_bench.py: 5 modes, no modes, white only), four lanes
# and one, alternating three times on one box.   scripts/gpu_r5_noise_ab.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05noiseab}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
for lib in scripts/ab/libmrx_*.so; do
for lanes in 4 1; do
  MRX_NOISE_LANES=$lanes MRX_LIB_PATH=$ROOT/$lib timeout -k 10 300 python3 scripts/noise_bench.py 10000 240000 5 2>&1 | grep "^noise" | sed "s|^|$(basename $lib) lanes $lanes: |; s| -> .*||" | tee -a $OUT/noise_ab.log || exit 1
done
done
done
