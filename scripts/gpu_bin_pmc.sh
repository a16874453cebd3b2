#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-binpmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o run -- python3 $ROOT/scripts/bin_bench.py 1024 1 1 > $OUT/$tag.log 2>&1
done
python3 - $OUT <<'PY'
import csv, sys, collections, glob, re
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/run_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.split(r"\(", k)[0][:40]
        if not k.startswith("bin_"): continue
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in sorted(tot):
    print(k)
    for c, v in sorted(tot[k].items()):
        print(f"   {c:28s} {v / cnt[(k, c)]:.4g}")
PY
