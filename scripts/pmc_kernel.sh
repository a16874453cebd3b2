#!/bin/bash
# PMC counters per kernel for a script: bash scripts/pmc_kernel.sh <tag> "<counters>" <script> [args...]
set -o pipefail
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -o run -- python3 $ROOT/scripts/$1 "${@:2}" > $OUT/log.txt 2>&1
cd $ROOT
python3 - $OUT/run_counter_collection.csv <<'PY'
import csv, sys, collections, re
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    k = re.split(r"\(", k)[0][:40]
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(tot):
    print(f"{k:40s} " + " ".join(f"{c}={v / cnt[(k, c)]:.4g}" for c, v in sorted(tot[k].items())))
PY
