#!/usr/bin/env python3
"""Development aid: the screen stage with the register transforms and with the LDS Stockham ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic, _lib
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
for cfg in ("atlast_10k", "atlast_50k", "mustang2_600s"):
    p = synthetic.config_problem(cfg, n_det=64)
    path = DevicePath(p, device="cuda:0")
    path.generate_screens()
    for rep in range(2):
        for flag in (1, 0):
            path.ctx.set_option(_lib.OPT_SCREEN_STOCKHAM, flag)
            med, mn = timeit(path.generate_screens, 20)
            print(f"{cfg}: {'Stockham' if flag else 'registers'}: median {med:.3f} ms min {mn:.3f} ms", flush=True)
    del path
