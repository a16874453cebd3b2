#!/bin/bash
# The per-rank step of an N-GPU run of atlast_10k rehearsed on one GPU (bench.py --shard-of N), with and without the next
# step's screens beside this step's launch (--lookahead), alternating twice.   scripts/gpu_r5_shard_la.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05shla}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for n in 8 4 2; do
for rep in 1 2; do
for la in "" "--lookahead"; do
  timeout -k 10 200 python3 bench.py --shard-of $n $la --no-cpu-baseline --no-frontend --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shard of $n'.ljust(11), '${la:-(plain)}'.ljust(12), 'rows', j['config']['n_det_per_gpu'], 'ms_per_step %.3f'%j['ms_per_step'], 'screens %.3f'%j['stage_ms']['screens'], 'synthesis %.3f'%j['stage_ms']['tod_synthesis_pipelined'], '|', j['stage_ms'].get('form','')[:40])" | tee -a $OUT/shard_la.log || exit 1
done
done
done
