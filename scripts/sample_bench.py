#!/usr/bin/env python3
"""atm_sample_kernel only (profiling aid).  Usage: sample_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
for lit in (0, 1):
    path.ctx.set_option(1, lit)
    med, mn = timeit(path.sample, int(sys.argv[1]) if len(sys.argv) > 1 else 10)
    print(f"sample axis_literal={lit}: median {med:.3f} ms min {mn:.3f}")
