#!/usr/bin/env python3
"""Experiment: BASELINE config 5's per-GPU share (or the config named), TOD synthesis by resident sampler workgroups per CU x detector blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_50k"
n_det = synthetic.CONFIGS[config]["n_det"] // (8 if config == "atlast_50k" else 1)
p = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
for times in (1, 2):
    for wgs in (2, 3, 4, 5, 6):
        for b in (4, 8, 12, 16):
            med, mn = timeit(lambda: path._run_pipelined(tod, b, resident_wgs_per_cu=wgs, resident_times=times), 5)
            print(f"{config} steps/thread {times} resident wgs/CU {wgs} blocks {b}: median {med:.3f} ms min {mn:.3f}", flush=True)
