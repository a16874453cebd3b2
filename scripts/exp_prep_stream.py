#!/usr/bin/env python3
"""Development aid: the spline solve on the sampler's stream or on the writer's, per block count."""
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
for blocks in (4, 6, 8, 12):
    for flag in (False, True):
        path.prepare_on_writer_stream = flag
        med, mn = timeit(lambda: path.run(tod, blocks=blocks), 15)
        print(f"blocks={blocks} prepare on {'writer' if flag else 'sampler'} stream: median {med:.3f} ms min {mn:.3f} ms", flush=True)
