#!/usr/bin/env python3
"""Sweep of the pipelined run: detector blocks x resident sampler workgroups per CU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
p = synthetic.config_problem(sys.argv[1] if len(sys.argv) > 1 else "atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
def wall(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("serial (1 block): %.3f ms" % wall(lambda: path.run(tod, blocks=1)), flush=True)
kt = int(os.environ.get("MRX_SAMPLE_TIMES", "2"))  # time steps interleaved per thread in the co-running sampler
blocks = [int(b) for b in os.environ.get("MRX_BLOCKS", "2,3,4,5,6,8").split(",")]
for B in blocks:
    row = []
    for k in (2, 3, 4, 5, 6):
        row.append("%d/CU %.3f" % (k, wall(lambda: path._run_pipelined(tod, B, resident_wgs_per_cu=k, resident_times=kt))))
    print(f"blocks={B}: " + "  ".join(row), flush=True)
