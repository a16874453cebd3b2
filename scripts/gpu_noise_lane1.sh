#!/bin/bash
# The noise generator's kernels ALONE (MRX_NOISE_LANES=1: one stream, batches four times as long): durations per call, and
# how busy each keeps the vector ALU / how long its waves wait.   scripts/gpu_noise_lane1.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05lane1}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MRX_NOISE_LANES=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 3 > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 1 > $OUT/pmc.log 2>&1 || exit 1
cd $ROOT
grep "^noise" $OUT/stats.log | cut -c1-110
python3 - $OUT <<'PY'
import csv, sys, collections, re
out = sys.argv[1]
calls = 4  # noise_bench ... 3: one warm-up + three timed calls of every variant
print("kernel durations, per call of its variant (one lane: the kernels run one after the other):")
for r in list(csv.DictReader(open(f"{out}/stats/run_kernel_stats.csv")))[:7]:
    n = int(r["Calls"]); per = calls * (2 if ("fft64_combine" in r["Name"] or "pair_means" in r["Name"]) else 1)
    print(f"  {r['Name'][28:78]:50s} {n / per:5.1f} launches x {float(r['AverageNs']) / 1e3:7.1f} us = {float(r['TotalDurationNs']) / 1e6 / per:6.2f} ms")
tot = collections.defaultdict(collections.Counter)
for row in csv.DictReader(open(f"{out}/pmc/run_counter_collection.csv")):
    k = re.split(r"\(", row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[0][:40]
    tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
print("counters (sums over the run):")
for k in sorted(tot):
    c = tot[k]
    if not k.startswith("noise") or c["GRBM_GUI_ACTIVE"] < 1e6: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"  {k:40s} vector issue {100 * 4 * c['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):4.0f} %   waves waiting {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:3.0f} %   waves per SIMD {4 * c['SQ_WAVE_CYCLES'] / (1024 * cyc):4.1f}")
PY
