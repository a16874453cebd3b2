#!/bin/bash
# Collect the round's evidence on the GPU box: kernel stats, PMC traffic passes (each in its own
# run, --kernel-trace only), the sampler's SQ / TCP counters, the plain bench line, and the
# follow-on rows' kernel stats.
# Usage (from the repo root, under gpurun): bash scripts/profile_round.sh <tag>
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-frontend"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $B --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o run -- $B --steps 3 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o run -- $B --steps 3 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o run -- $B --steps 3 --warmup 1 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $OUT/pmc_tcp -o run -- $B --steps 3 --warmup 1 > $OUT/pmc_tcp.json 2> $OUT/pmc_tcp.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_lds -o run -- $B --steps 3 --warmup 1 > $OUT/pmc_lds.json 2> $OUT/pmc_lds.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/krj -o run -- python3 $ROOT/scripts/krj_bench.py > $OUT/krj_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/noise -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 3 > $OUT/noise_bench.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/noise_pmc -o run -- python3 $ROOT/scripts/noise_bench.py 10000 240000 1 > $OUT/noise_pmc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/map -o run -- python3 $ROOT/scripts/map_bench.py 10000 240000 2 > $OUT/map_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bin -o run -- python3 $ROOT/scripts/bin_bench.py 1024 1 3 > $OUT/bin_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/frontend_map -o run -- python3 $ROOT/scripts/frontend_trace.py map > $OUT/frontend_map.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/gauss -o run -- python3 $ROOT/scripts/gauss_bench.py 10 > $OUT/gauss_bench.log 2>&1
cd $ROOT
MRX_GAUSS_ACCUM=1 python3 scripts/gauss_bench.py 10 > $OUT/gauss_bench_exact.log 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
tail -c 400 $OUT/bench.json
