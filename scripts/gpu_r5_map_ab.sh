#!/bin/bash
# A/B of the libraries under scripts/ab/ on the map sampler: scripts/map_bench.py (pW: az/el, ra/dec, + atmospheric calibration),
# alternating three times on one box.   scripts/gpu_r5_map_ab.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05mapab}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for rep in 1 2 3; do
for lib in scripts/ab/libmrx_*.so; do
  MRX_LIB_PATH=$ROOT/$lib timeout -k 10 300 python3 scripts/map_bench.py 10000 240000 3 2>&1 | grep "map_sample" | sed "s|^|$(basename $lib) |; s| (includes host staging of inputs)||" | tee -a $OUT/map_ab.log || exit 1
done
done
