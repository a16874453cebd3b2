#!/usr/bin/env python3
"""Screen stage only (generate + smooth, all layers), for profiling.  Usage: screens_bench.py [config] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
cfg = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
p = synthetic.config_problem(cfg, n_det=64)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
med, mn = timeit(path.generate_screens, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
print(f"{cfg}: screens median {med:.3f} ms min {mn:.3f} ms")
med, mn = timeit(lambda: path.generate_screens(smooth=False), 20)
print(f"{cfg}: screens without the beam smoothing median {med:.3f} ms min {mn:.3f} ms")
