#!/bin/bash
# As gpu_r5_shard_la.sh, for the 8- and 4-GPU shards: look-ahead with the one launch's resident grid capped at 3 / 4 workgroups
# per CU, so that the next step's screens find room beside it.   scripts/gpu_r5_shard_la2.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05shla2}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for n in 8 4; do
for rep in 1 2; do
for opt in "" "--synth-wgs-per-cu 4" "--lookahead --synth-wgs-per-cu 4" "--synth-wgs-per-cu 3" "--lookahead --synth-wgs-per-cu 3"; do
  timeout -k 10 200 python3 bench.py --shard-of $n $opt --no-cpu-baseline --no-frontend --steps 40 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('shard of $n'.ljust(11), '${opt:-(plain)}'.ljust(36), 'rows', j['config']['n_det_per_gpu'], 'ms_per_step %.3f'%j['ms_per_step'], 'screens %.3f'%j['stage_ms']['screens'], 'synthesis %.3f'%j['stage_ms']['tod_synthesis_pipelined'])" | tee -a $OUT/shard_la.log || exit 1
done
done
done
