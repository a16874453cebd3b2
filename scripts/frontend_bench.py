#!/usr/bin/env python3
"""bench.py's front-end rows alone (Simulation(...).run() in K_RJ: atmosphere, + noise, + map, + map + noise), one line
per row: what scripts/gpu_ab.sh compares between two libraries.   python3 scripts/frontend_bench.py [row ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

rows = bench.frontend_timing(torch.device("cuda:0"))
for name, r in rows.items():
    if isinstance(r, dict) and (len(sys.argv) < 2 or name in sys.argv[1:]):
        print(f"frontend {name}: run_ms {r['run_ms']:.3f} min {r['run_ms_min']:.3f} gpu_span {r['gpu_span_ms']:.3f}"
              + (f" atmosphere_ms {r['atmosphere_ms']:.3f}" if "atmosphere_ms" in r else ""))
