#!/usr/bin/env python3
"""Experiment: the TOD writer at reduced occupancy (dynamic-LDS padding caps workgroups per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens(); path.sample(); path.prepare()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
for groups in (2, 1, 4):
    path.ctx.set_option(4, groups)
    for pad, label in ((0, "4 WG/CU"), (10000, "3 WG/CU"), (30000, "2 WG/CU"), (60000, "1 WG/CU")):
        path.ctx.set_option(7, pad)
        med, mn = timeit(lambda: path.upsample(tod), 10)
        print(f"groups={groups} writer {label}: median {med:.3f} ms min {mn:.3f}", flush=True)
path.ctx.set_option(7, 0); path.ctx.set_option(4, 0)
print("sample (64-VGPR build):", timeit(path.sample, 10))
