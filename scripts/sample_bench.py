#!/usr/bin/env python3
"""atm_sample kernels only (profiling aid).  Usage: sample_bench.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for cfg in ("atlast_10k", "act_3k"):
    p = synthetic.config_problem(cfg)
    path = DevicePath(p, device="cuda:0")
    path.generate_screens()
    for name, opts in (("pixel, global gathers (default)", {}), ("LDS-tiled", {7: 1}), ("literal jax cells", {1: 1})):
        for k, v in opts.items(): path.ctx.set_option(k, v)
        for chunk in ((0, 16, 32) if 7 in opts else (0,)):
            path.ctx.set_option(3, chunk)
            med, mn = timeit(path.sample, reps)
            print(f"{cfg} sample {name} chunk={chunk or 'auto'}: median {med:.3f} ms min {mn:.3f}", flush=True)
        path.ctx.set_option(3, 0)
        for k in opts: path.ctx.set_option(k, 0)
    del path
