#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 scripts/exp_sample_whatif.py > $OUT/whatif.log 2>&1 || echo "whatif failed" >> $OUT/whatif.log
timeout -k 10 400 python3 scripts/exp_blockpipe.py > $OUT/blockpipe.log 2>&1 || echo "blockpipe failed" >> $OUT/blockpipe.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr --output-format csv -d $OUT/pmc_tcp -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_tcp.json 2> $OUT/pmc_tcp.log
cat $OUT/whatif.log $OUT/blockpipe.log
