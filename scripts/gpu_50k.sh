#!/bin/bash
# BASELINE config 5's per-GPU share (6250 detectors x 1 440 000 samples, 16 x 4096^2 screens) through bench.py
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/b50k
cd $ROOT
for B in 1 4; do
  timeout -k 10 400 python3 bench.py --config atlast_50k --steps 5 --warmup 2 --no-cpu-baseline --no-frontend --blocks $B > gpurun_out/b50k/b$B.json 2> gpurun_out/b50k/b$B.err
  python3 - gpurun_out/b50k/b$B.json $B <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("blocks", sys.argv[2], "ms_per_step", round(r["ms_per_step"], 3), "value", f'{r["value"]:.3e}', {k: round(v, 3) for k, v in r["stage_ms"]["serial_breakdown"].items() if isinstance(v, float)},
      "writer frac", round(r["roofline"]["frac"], 3), r["config"].get("weak_unit"))
PY
done
