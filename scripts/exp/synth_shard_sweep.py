"""The one launch at a shard's row count against the chunk length and the dedicated samplers.
    python3 scripts/exp/synth_shard_sweep.py <rows> [chunks ...]      (median of 15 launches each; 0 = the launcher's choice)
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402

rows = int(sys.argv[1])
chunks = [int(x) for x in sys.argv[2:]] or [0, 4, 8, 16, 32]
problem = synthetic.config_problem("atlast_10k", n_det=rows)
path = DevicePath(problem, device="cuda:0")
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
path.generate_screens()
for _ in range(3):
    path.synthesize(tod)
torch.cuda.synchronize()
cases = [(c, s) for c in chunks for s in (0, 256, 768)]
res = {k: [] for k in cases}
for rep in range(3):
    for (c, s) in cases:
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            path.synthesize(tod, chunk=c if c else None, sampler_wgs=s)
            e1.record()
            torch.cuda.synchronize()
            res[(c, s)].append(e0.elapsed_time(e1))
for (c, s) in cases:
    print(f"rows {rows} chunk {c:3d} dedicated {s if s else 'default':>7}: median {np.median(res[(c, s)]):.4f} ms  min {min(res[(c, s)]):.4f}")
