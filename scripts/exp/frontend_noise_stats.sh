#!/bin/bash
# kernels of Simulation(noise=True).run() under rocprofv3 --kernel-trace --stats
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/fe_noise; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/scripts/frontend_trace.py noise > $OUT/run.log 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
grep "^run" $OUT/run.log
python3 - $OUT <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/stats/**/run_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-84s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
