#!/bin/bash
# Kernel timeline of one mrx_noise_generate call (scripts/noise_bench.py, first variant: knee 1 Hz, 5 modes): per kernel the
# launches, the sum of their durations, and how many kernels run at once.   scripts/exp/noise_timeline.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-noisetl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 1 > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, os, sys, collections
root = sys.argv[1]
f = [os.path.join(dp, n) for dp, _, ns in os.walk(root) for n in ns if n.endswith("kernel_trace.csv")][0]
rows = [r for r in csv.DictReader(open(f)) if "noise" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the first call (warm-up) of the first variant: the launches up to the first gap of more than 2 ms
calls, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 1_000_000:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
for ci, call in enumerate(calls[:4]):
    t0 = min(int(r["Start_Timestamp"]) for r in call); t1 = max(int(r["End_Timestamp"]) for r in call)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in call:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:36]
        agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    busy = sum(v[1] for v in agg.values())
    print(f"call {ci}: wall {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {busy:.2f} ms (concurrency {busy / ((t1 - t0) / 1e6):.2f}); " +
          "; ".join(f"{k} x{v[0]} {v[1]:.2f}" for k, v in sorted(agg.items())))
PY
