#!/bin/bash
# A/B on one box: bench.py of this tree against a second tree (an older snapshot unpacked into <dir>), alternating.
# Usage (under gpurun): bash scripts/gpu_ab_tree.sh <dir> [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OLD=$ROOT/$1; N=${2:-3}
for i in $(seq $N); do
  for d in $ROOT $OLD; do
    cd $d
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-frontend --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
    python3 -c "import json; r=json.load(open('/tmp/b.json')); print('$d'.split('/')[-1], round(r['ms_per_step'],4), round(r['stage_ms']['screens'],4), round(r['stage_ms']['tod_synthesis_pipelined'],4), round(r['roofline']['ms_per_launch'],4), r['stage_ms']['serial_breakdown']['sample'])"
  done
done
