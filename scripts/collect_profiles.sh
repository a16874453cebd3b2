#!/bin/bash
# Copy the summaries of gpurun_out/<tag> (scripts/profile_round.sh) into profiles/<tag>_*.
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/gpurun_out/$TAG
cp $SRC/stats/run_kernel_stats.csv $ROOT/profiles/${TAG}_kernel_stats.csv
cp $SRC/bench_under_rocprof.json $ROOT/profiles/${TAG}_bench_under_rocprof.json
cp $SRC/bench.json $ROOT/profiles/${TAG}_bench.json
# writer launches per step = the detector blocks the profiled bench line reports
BLOCKS=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['stage_ms']['detector_blocks'])" $SRC/bench.json)
python3 $ROOT/scripts/pmc_summary.py $SRC $ROOT/profiles/$TAG $BLOCKS
for k in krj noise map bin gauss frontend_map; do [ -f $SRC/$k/run_kernel_stats.csv ] && cp $SRC/$k/run_kernel_stats.csv $ROOT/profiles/${TAG}_${k}_kernel_stats.csv; done
[ -f $SRC/gauss_bench.log ] && grep -h "^#\|^gauss_smooth2d\|^map_smooth" $SRC/gauss_bench.log > $ROOT/profiles/${TAG}_gauss_bench.txt
[ -f $SRC/gauss_bench_exact.log ] && grep -h "^#\|^gauss_smooth2d\|^map_smooth" $SRC/gauss_bench_exact.log > $ROOT/profiles/${TAG}_gauss_bench_exact.txt
grep -h "pW\|groups" $SRC/krj_bench.log > $ROOT/profiles/${TAG}_krj_bench.txt
grep -h "^noise" $SRC/noise_bench.log > $ROOT/profiles/${TAG}_noise_bench.txt
python3 - $SRC/noise_pmc/run_counter_collection.csv > $ROOT/profiles/${TAG}_noise_pmc.txt <<'PY'
import csv, sys, collections, re
tot = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    k = re.split(r"\(", k)[0][:40]
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
print("# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 scripts/noise_bench.py 10000 240000 1")
print("# per-launch averages (a launch = 128 detector pairs of 2^18 cells = 3.36e7 cells: VALU instructions per cell = SQ_INSTS_VALU x 64 / 3.36e7)")
for k in sorted(tot):
    if k.startswith("noise"):
        print(f"{k:40s} launches {max(cnt[(k, c)] for c in tot[k]):4d} " + " ".join(f"{c}={v / cnt[(k, c)]:.4g}" for c, v in sorted(tot[k].items())))
PY
grep -h "^bin" $SRC/bin_bench.log > $ROOT/profiles/${TAG}_bin_bench.txt
grep -h "^map_sample" $SRC/map_bench.log > $ROOT/profiles/${TAG}_map_bench.txt
ls -la $ROOT/profiles | grep $TAG
