#!/bin/bash
# The default bench line (CPU baselines + front end) and the 2-rank rehearsal started from a plain shell.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-full}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 600 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
echo "default rc=$?"; tail -3 $OUT/bench_default.time
python3 - $OUT/bench_default.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("ms_per_step", r["ms_per_step"], "frac", r["roofline"]["frac"], r["roofline"]["frac_alone"])
print(json.dumps(r.get("cpu_baseline"), indent=1)[:1500])
print(json.dumps(r.get("frontend"), indent=1))
PY
( time timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --single-device --steps 5 --warmup 2 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err ) 2> $OUT/bench_gloo2.time
echo "gloo2 rc=$?"; tail -3 $OUT/bench_gloo2.time; wc -l $OUT/bench_gloo2.json; tail -c 1500 $OUT/bench_gloo2.json; tail -5 $OUT/bench_gloo2.err
