#!/bin/bash
# BASELINE config 5's per-GPU share on the GPU box: kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE in their own
# --pmc passes), SQ / TCP / TCC counters, the plain bench line.  bash scripts/profile_50k.sh <tag>
set -o pipefail
TAG=${1:-r04_50k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --config atlast_50k --no-cpu-baseline --no-frontend"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $B --steps 5 --warmup 2 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $set | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name -o run -- $B --steps 2 --warmup 1 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.log || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; exit 1; }
done
cd $ROOT
python3 bench.py --config atlast_50k --steps 5 --warmup 2 --no-frontend > $OUT/bench.json 2> $OUT/bench.log
python3 scripts/show_bench.py $OUT/bench.json
python3 - $OUT <<'PY' > $OUT/kernel_pmc.txt
import csv, sys, collections, glob, re
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.split(r"\(", k)[0][:44]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
print("# python3 bench.py --config atlast_50k --no-cpu-baseline --no-frontend --steps 2 --warmup 1 under rocprofv3 --kernel-trace --pmc <one set per run>")
print("# 6250 det x 1 440 000 samples, 16 x 4096^2 screens; per-launch averages; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (FETCH_SIZE x2 for bytes on gfx950)")
for k in sorted(tot):
    if k.startswith(("atm_", "spline_", "screen_")):
        print(f"{k:44s} launches {max(cnt[(k, c)] for c in tot[k]):3d}  " + " ".join(f"{c}={v / cnt[(k, c)]:.4g}" for c, v in sorted(tot[k].items())))
PY
cat $OUT/kernel_pmc.txt | cut -c1-400
