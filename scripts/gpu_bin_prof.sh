#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-binprof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/bin_bench.py 1024 1 3 > $OUT/log.txt 2>&1
grep "^bin" $OUT/log.txt
python3 - $OUT/run_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print(f"{n:44s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
