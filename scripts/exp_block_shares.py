#!/usr/bin/env python3
"""A/B on one box: how the pipelined run cuts the detector rows into blocks (first block shorter: its sampler runs alone)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
for rep in range(2):
    for share in ([1, 2, 2, 2], [1, 1, 1, 1], [1, 2, 2, 2, 2], [1] * 6, [1, 2, 2, 2, 2, 2, 2], [1] * 8, [1] + [2] * 7 + [1], [1] * 10, [1] * 12, [1, 2, 3, 3, 3, 2, 1], [1, 2, 2, 2, 1]):
        path._pipe = None
        path.block_shares = share
        med, mn = timeit(lambda: path.run(tod, blocks=len(share)), 20)
        print(f"rep {rep} shares {share}: median {med:.3f} ms min {mn:.3f} ms", flush=True)
