#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/${1:-fe}
for MODE in atm noise; do
  echo "== $MODE"
  timeout -k 10 300 python3 scripts/frontend_trace.py $MODE 2>&1 | grep -v "^W\|amdgpu.ids" | head -40
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${1:-fe}/fe_noise -o run -- python3 $ROOT/scripts/frontend_trace.py noise > $ROOT/gpurun_out/${1:-fe}/fe_noise.log 2>&1
S=$(ls $ROOT/gpurun_out/${1:-fe}/fe_noise/*/run_kernel_stats.csv $ROOT/gpurun_out/${1:-fe}/fe_noise/run_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$S" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
