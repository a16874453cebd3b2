#!/bin/bash
# Collect the round's evidence on the GPU box: kernel stats, PMC traffic passes (each in its own
# run, --kernel-trace only), the plain bench line, and the follow-on rows' kernel stats.
# Usage (from the repo root, under gpurun): bash scripts/profile_round.sh <tag>
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/noise -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 3 > $OUT/noise_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/map -o run -- python3 $ROOT/scripts/map_bench.py 10000 240000 2 > $OUT/map_bench.log 2>&1
cd $ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
tail -c 600 $OUT/bench.json
