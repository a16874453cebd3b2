#!/bin/bash
# Kernel-trace timeline of the last bench step + per-kernel stats.   scripts/gpu_trace.sh <tag> [bench args]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $OUT/trace_bench.json 2> $OUT/trace.log
echo "trace rc=$?"
F=$(ls $OUT/trace/*/run_kernel_trace.csv $OUT/trace/run_kernel_trace.csv 2>/dev/null | head -1)
S=$(ls $OUT/trace/*/run_kernel_stats.csv $OUT/trace/run_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed steps come before the 3 serial passes: find the last "screen" kernel group of the timed region
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "screen_half_spectrum" in n]
start = idx[-1] if idx else max(0, len(rows) - 40)
sel = rows[start:start + 22]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:34]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:34s} q{r.get('Queue_Id','?'):>3s} start {s:8.1f} end {e:8.1f} dur {e-s:7.1f}")
PY
head -12 "$S" | cut -c1-170
