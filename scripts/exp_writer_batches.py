#!/usr/bin/env python3
"""A/B on one box: row batches per writer workgroup (MRX_OPT_UPSAMPLE_GROUPS) in the pipelined TOD synthesis."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
for rep in range(3):
    for b in (1, 2, 4):
        path.ctx.set_option(4, b)
        med, mn = timeit(lambda: path.run(tod, blocks=4), 20)
        print(f"rep {rep} batches {b}: pipelined TOD synthesis median {med:.3f} ms min {mn:.3f} ms", flush=True)
