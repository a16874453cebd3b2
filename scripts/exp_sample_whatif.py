#!/usr/bin/env python3
"""What-if timings of atm_sample_kernel (temporary experiment option 6; results are wrong by design
for mask != 0): bit0 no value gather, bit1 nodes by f32 arithmetic, bit2 f32 affine map, bit3 no
fallback check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
for mask in (0, 8, 1, 2, 4, 3, 6, 7, 15, 0):
    path.ctx.set_option(6, mask)
    med, mn = timeit(path.sample, 12)
    print(f"mask={mask:2d}: median {med:.3f} ms min {mn:.3f} ms", flush=True)
path.ctx.set_option(6, 0)
for ls in (True, False):
    pth = DevicePath(p, device="cuda:0", locality_sort=ls)
    pth.set_screens(path._gen_screens)
    med, mn = timeit(pth.sample, 12)
    print(f"locality_sort={ls}: median {med:.3f} ms", flush=True)
