#!/bin/bash
# Instruction mix of the noise generator's kernels (scripts/noise_bench.py 10000 240000 1: five modes, no modes, white only; every
# variant is called twice -- one warm-up, one timed), per CALL and per sample of the 10 000 x 240 000 field.  <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-noisepmc}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
k=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  k=$((k+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/set$k -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 1 > $OUT/set$k.log 2>&1 || { tail -3 $OUT/set$k.log; exit 1; }
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, collections, re
out = sys.argv[1]
tot = collections.defaultdict(collections.Counter); launches = collections.Counter()
for k in (1, 2):
    seen = collections.Counter()
    for row in csv.DictReader(open(f"{out}/set{k}/run_counter_collection.csv")):
        name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.split(r"\(", name)[0][:44]
        tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
        seen[(name, row["Counter_Name"])] += 1
    for (name, c), n in seen.items():
        launches[name] = max(launches[name], n)
calls = {"<false, 5>": 2, "<5, 4>": 2, "<false, 0>": 2, "<0, 4>": 2}  # every variant: one warm-up + one timed call
samples = 10000 * 240000
print("# per CALL of mrx_noise_generate (10 000 x 240 000; the bench calls every variant twice); 'per sample' = wave-instructions x 64 / 2.4e9")
for name in sorted(tot):
    if not name.startswith("noise"): continue
    n_calls = 4 if "fft64_combine" in name or "pair_means" in name else 2  # (these run in the 5-mode and the 0-mode variant alike)
    c = tot[name]
    line = f"{name:44s} launches/call {launches[name] / n_calls:5.1f} "
    line += " ".join(f"{k.replace('SQ_INSTS_', '').replace('SQ_', '')}={v / n_calls:.4g}" for k, v in sorted(c.items()))
    print(line)
    if "SQ_INSTS_VALU" in c:
        print(f"{'':44s} per sample: VALU {c['SQ_INSTS_VALU'] / n_calls * 64 / samples:.1f}  (int32 {c['SQ_INSTS_VALU_INT32'] / n_calls * 64 / samples:.1f}, fma {c['SQ_INSTS_VALU_FMA_F32'] / n_calls * 64 / samples:.1f}, mul {c['SQ_INSTS_VALU_MUL_F32'] / n_calls * 64 / samples:.1f}, add {c['SQ_INSTS_VALU_ADD_F32'] / n_calls * 64 / samples:.1f}, transcendental {c['SQ_INSTS_VALU_TRANS_F32'] / n_calls * 64 / samples:.2f}, cvt {c['SQ_INSTS_VALU_CVT'] / n_calls * 64 / samples:.1f})")
PY
