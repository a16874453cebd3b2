import os, sys
sys.path.insert(0, os.getcwd())
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
for cfg in ("atlast_10k", "act_3k"):
    p = synthetic.config_problem(cfg)
    path = DevicePath(p, device="cuda:0")
    path.generate_screens()
    for kt in (1, 2):
        for chunk in (8, 16, 32, 64):
            path.ctx.set_option(2, kt); path.ctx.set_option(3, chunk)
            med, mn = timeit(path.sample, 15)
            print(f"{cfg} kT={kt} chunk={chunk}: median {med:.3f} min {mn:.3f}")
    del path
