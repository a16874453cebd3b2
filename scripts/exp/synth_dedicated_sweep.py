"""The one launch (mrx_atm_synthesize) against the number of workgroups that only sample while items remain.
    python3 scripts/exp/synth_dedicated_sweep.py [config] [counts ...]      (median of 15 launches each, interleaved x3)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
counts = [int(x) for x in sys.argv[2:]] or [256, 384, 448, 512, 576, 640, 768]
n_det = synthetic.CONFIGS[config]["n_det"]
if config == "atlast_50k":
    n_det //= 8
problem = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(problem, device="cuda:0")
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
path.generate_screens()
for _ in range(3):
    path.synthesize(tod)
torch.cuda.synchronize()
res = {c: [] for c in counts}
for rep in range(3):
    for c in counts:
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            path.synthesize(tod, sampler_wgs=c)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[c] += ts
for c in counts:
    print(f"{config} dedicated samplers {c:5d}: median {np.median(res[c]):.3f} ms  min {min(res[c]):.3f}")
