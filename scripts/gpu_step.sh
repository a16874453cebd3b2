#!/bin/bash
# One GPU-box call of the development loop: selected GPU tests, then the bench line.
#   scripts/gpu_step.sh <tag> "<pytest args>" [bench args...]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
TESTS=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "$TESTS" ]; then
  timeout -k 10 900 python3 -m pytest $TESTS -x -q -m gpu > $OUT/pytest.log 2>&1
  echo "pytest rc=$?"; tail -4 $OUT/pytest.log
fi
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"
python3 - $OUT/bench.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("ms_per_step", round(r["ms_per_step"], 4), "stage", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r["stage_ms"].items() if k != "serial_breakdown"})
print("serial", {k: round(v, 4) for k, v in r["stage_ms"]["serial_breakdown"].items() if isinstance(v, float)})
rf = r["roofline"]
print("writer ms/launch", round(rf["ms_per_launch"], 4), "frac", round(rf["frac"], 4), "alone", round(rf["frac_alone"], 4))
if "cpu_baseline" in r: print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("parity_max_rel_err_vs_gpu"), r["cpu_baseline"].get("parity_fluct_rel_err"))
PY
