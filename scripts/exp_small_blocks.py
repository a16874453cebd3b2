#!/usr/bin/env python3
"""Development aid: does the block pipeline pay for the shards of 2, 4 and 8 GPUs (5000, 2500, 1250 rows)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
for n in (5008, 2512, 1264):
    path = DevicePath(p, device="cuda:0", det_slice=slice(0, n))
    path.generate_screens()
    tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
    row = []
    for blocks in (1, 2, 3, 4, 6, 8):
        med, mn = timeit(lambda: path.run(tod, blocks=blocks), 20)
        row.append(f"{blocks}: {med:.3f}")
    print(f"D={n} ms per TOD synthesis by blocks: " + "  ".join(row), flush=True)
    del path, tod
