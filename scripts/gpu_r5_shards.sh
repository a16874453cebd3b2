#!/bin/bash
# The one-launch form on the detector shards of an N-GPU run of atlast_10k (5008 / 2512 / 1264 rows) against the stages back to back.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05g}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for d in 5008 2512 1264 640; do
  SYNTH_DETS=$d SYNTH_WGS=${SYNTH_WGS:-1,2,3,8} SYNTH_CHUNK=${SYNTH_CHUNK:-16,32} timeout -k 10 300 python3 scripts/exp_synth.py atlast_10k 0 2>&1 | grep -v "Warn\|amdgpu.ids" | tee -a $OUT/shards.log || exit 1
done
