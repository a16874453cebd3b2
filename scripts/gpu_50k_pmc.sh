#!/bin/bash
# PMC counters of the kernels of BASELINE config 5's per-GPU share (bench.py --config atlast_50k), one counter set per run.
#   bash scripts/gpu_50k_pmc.sh <tag> <blocks>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmc50k}; BLOCKS=${2:-1}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config atlast_50k --no-cpu-baseline --no-frontend --steps 2 --warmup 1 --blocks $BLOCKS"
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$i -o run -- $B > $OUT/set$i.json 2> $OUT/set$i.log || { echo "set $i failed"; tail -3 $OUT/set$i.log; exit 1; }
done
cd $ROOT
python3 - $OUT $BLOCKS <<'PY' | tee $OUT/summary.txt
import csv, sys, collections, glob, re
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + "/set*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.split(r"\(", k)[0][:44]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
print(f"# bench.py --config atlast_50k --blocks {sys.argv[2]} (6250 det x 1 440 000 samples, 16 x 4096^2 screens); per-launch averages; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE x2 for bytes on gfx950)")
for k in sorted(tot):
    print(f"{k:44s} launches {max(cnt[(k, c)] for c in tot[k]):3d}  " + " ".join(f"{c}={v / cnt[(k, c)]:.4g}" for c, v in sorted(tot[k].items())))
PY
