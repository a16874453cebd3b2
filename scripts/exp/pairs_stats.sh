#!/bin/bash
# kernel-trace stats of the bench at one config (arguments after the tag go to bench.py):  scripts/exp/pairs_stats.sh <tag> [--config atlast_50k]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-frontend "$@" > $OUT/bench.json 2> $OUT/stats.log || { tail -5 $OUT/stats.log; exit 1; }
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/stats/**/run_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-90s calls %5s avg_us %10.1f pct %5s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
