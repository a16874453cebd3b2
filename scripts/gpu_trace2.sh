#!/bin/bash
# Kernel-trace timeline of one timed bench step (the 4th): every kernel's queue, start, end.   scripts/gpu_trace2.sh <tag> [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --no-cpu-baseline --no-frontend --steps 6 --warmup 2 "$@" > $OUT/trace_bench.json 2> $OUT/trace.log
F=$(ls $OUT/trace/*/run_kernel_trace.csv $OUT/trace/run_kernel_trace.csv 2>/dev/null | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "screen_half_spectrum" in n]
# steps: warmup 2 + 1 initial + 6 timed ... take the 6th occurrence from the start of the timed region
start = idx[5] if len(idx) > 6 else idx[-1]
end = idx[6] if len(idx) > 6 else len(rows)
sel = rows[start:end]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:38]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:38s} q{r.get('Queue_Id','?'):>3s} start {s:8.1f} end {e:8.1f} dur {e-s:7.1f}")
print("next step starts at", (int(rows[end]["Start_Timestamp"]) - t0) / 1e3 if end < len(rows) else None)
PY
