#!/bin/bash
# rocprofv3 kernel stats of the binning bench: bash scripts/prof_bin.sh <tag>
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp MRX_BENCH_BIN=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/map_bench.py > $OUT/log.txt 2>&1
cd $ROOT
grep "^bin_map" $OUT/log.txt
python3 - $OUT/run_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{n:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
