#!/usr/bin/env python3
"""Experiment: the sampler alone on the chip (block 0 of the pipelined step, and every unpipelined run): the plain layer
loop on the full grid against the three-stage ring on resident grids of 4-7 workgroups per CU (MRX_OPT_SAMPLE_WGS_PER_CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
for rep in range(2):
    for wgs in (0, 4, 5, 6, 7):
        path.ctx.set_option(6, wgs)
        med, mn = timeit(path.sample, 10)
        print(f"sampler alone, 10000 rows, workgroups per CU {wgs or 'full grid (plain loop)'}: median {med:.3f} ms min {mn:.3f}", flush=True)
path.ctx.set_option(6, 0)
