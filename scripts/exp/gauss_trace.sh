cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6j -o run -- python3 $GRAFT_REPO_ROOT/scripts/gauss_bench.py 5 > $GRAFT_REPO_ROOT/gpurun_out/r6j.log 2>&1
python3 - <<'PY'
import csv, os, collections
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6j"
f = [os.path.join(dp, n) for dp, _, ns in os.walk(root) for n in ns if n.endswith("kernel_trace.csv")][0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gauss" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-32:], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort(); print(k, "n", len(v), "median_us %.1f" % v[len(v)//2])
PY
