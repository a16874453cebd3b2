#!/bin/bash
# Copy the summaries of gpurun_out/<tag> (scripts/profile_round.sh) into profiles/<tag>_*.
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
cp $SRC/stats/run_kernel_stats.csv $ROOT/profiles/${TAG}_kernel_stats.csv
cp $SRC/bench_under_rocprof.json $ROOT/profiles/${TAG}_bench_under_rocprof.json
cp $SRC/bench.json $ROOT/profiles/${TAG}_bench.json
python3 $ROOT/scripts/pmc_summary.py $SRC $ROOT/profiles/$TAG 4   # 4 detector blocks per step: one writer launch each
for k in krj noise map; do [ -f $SRC/$k/run_kernel_stats.csv ] && cp $SRC/$k/run_kernel_stats.csv $ROOT/profiles/${TAG}_${k}_kernel_stats.csv; done
grep -h "pW\|groups" $SRC/krj_bench.log > $ROOT/profiles/${TAG}_krj_bench.txt
grep -h "^noise" $SRC/noise_bench.log > $ROOT/profiles/${TAG}_noise_bench.txt
grep -h "^map_sample" $SRC/map_bench.log > $ROOT/profiles/${TAG}_map_bench.txt
ls -la $ROOT/profiles | grep $TAG
