#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-fe}; MODE=${2:-atm}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fe_$MODE -o run -- python3 $ROOT/scripts/frontend_trace.py $MODE > $OUT/fe_$MODE.log 2>&1
echo rc=$?
grep -v "^W\|amdgpu.ids" $OUT/fe_$MODE.log | head -45
S=$(ls $OUT/fe_$MODE/*/run_kernel_stats.csv $OUT/fe_$MODE/run_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
