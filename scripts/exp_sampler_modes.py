#!/usr/bin/env python3
"""Development aid: the whole-shard sampler per (resident workgroups per CU, time steps per thread, chunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
for cfg, n_det in (("atlast_10k", None), ("atlast_10k", 1250), ("act_3k", None), ("mustang2_600s", None)):
    p = synthetic.config_problem(cfg)
    path = DevicePath(p, device="cuda:0", det_slice=None if n_det is None else slice(0, n_det))
    path.generate_screens()
    for per_cu, kt in ((5, 2), (8, 1), (5, 2), (8, 1)):
        for chunk in (0,):
            path.ctx.set_option(6, per_cu); path.ctx.set_option(2, kt); path.ctx.set_option(3, chunk)
            med, mn = timeit(path.sample, 15)
            print(f"{cfg} D={path.D}: {per_cu}/CU x {kt} steps, chunk {chunk or 'auto'}: median {med:.3f} ms min {mn:.3f}", flush=True)
    del path
