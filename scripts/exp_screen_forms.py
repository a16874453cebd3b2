#!/usr/bin/env python3
"""Development aid: the screen stage with the register transforms and with the LDS Stockham ones, alone and inside
the step (after the previous step's TOD synthesis, as bench.py runs it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from maria_amd import synthetic, _lib
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
for cfg in ("atlast_10k", "atlast_50k"):
    p = synthetic.config_problem(cfg, n_det=10000 if cfg == "atlast_10k" else 1024)
    path = DevicePath(p, device="cuda:0")
    path.generate_screens()
    tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
    for rep in range(2):
        for flag in (1, 0):
            path.ctx.set_option(_lib.OPT_SCREEN_STOCKHAM, flag)
            med, mn = timeit(path.generate_screens, 20)
            ts, steps = [], []
            for k in range(12):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record(); path.generate_screens(); e[1].record(); path.run(tod); e[2].record()
                torch.cuda.synchronize()
                ts.append(e[0].elapsed_time(e[1])); steps.append(e[0].elapsed_time(e[2]))
            print(f"{cfg}: {'Stockham' if flag else 'registers'}: alone {med:.3f} ms; in the step {np.median(ts[2:]):.3f} ms of {np.median(steps[2:]):.3f}", flush=True)
    del path, tod
