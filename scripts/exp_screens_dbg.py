import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k", n_det=64)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
for m in (0, 3, 2, 1):
    path.ctx.set_option(7, m)
    med, mn = timeit(path.generate_screens, 10)
    print(f"dbg={m:2d}: screens median {med:.3f} ms", flush=True)
