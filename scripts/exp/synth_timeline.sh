#!/bin/bash
# The timeline of the one-launch synthesis for every trace build under scripts/ab_trace/ (see synth_timeline.py):
#   scripts/exp/synth_timeline.sh [config] [bin_us]   -> gpurun_out/timeline/<lib>_<config>.txt, summary lines on stdout
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
CFG=${1:-atlast_10k}; BIN=${2:-100}
mkdir -p $ROOT/gpurun_out/timeline; cd $ROOT
for lib in scripts/ab_trace/libmrx_trace*.so; do
  n=$(basename $lib .so)
  MRX_LIB_PATH=$ROOT/$lib timeout -k 10 300 python3 scripts/exp/synth_timeline.py $CFG $BIN > gpurun_out/timeline/${n}_$CFG.txt 2>&1 || { tail -5 gpurun_out/timeline/${n}_$CFG.txt; echo "FAILED $n"; exit 1; }
  echo "== $n"; grep -E "^launch span|^tile:|^item:|^gap|^first tile" gpurun_out/timeline/${n}_$CFG.txt
done
