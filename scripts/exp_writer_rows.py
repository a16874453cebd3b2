#!/usr/bin/env python3
"""Experiment: time per row of the TOD writer as a function of the rows per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd._lib import ptr
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
main = DevicePath(p, device="cuda:0")
main.generate_screens(); main.sample(); main.prepare()
tod = torch.empty((main.D, main.T), dtype=torch.float32, device="cuda:0")
for n in (640, 1280, 2560, 5120, 10000):
    ym = torch.rand((main.Ta, n, 2), dtype=torch.float32, device="cuda:0")
    for rows in (None, main.d_rows):
        fn = lambda: main.ctx.call("mrx_spline_upsample", ptr(ym), n, main.Ta, main.ta0, main.dta, ptr(main.d_t), main.T, None,
                                   None if rows is None else ptr(rows[:n] % n if n < main.D else rows), ptr(tod), tod.stride(0))
        med, mn = timeit(fn, 10)
        print(f"rows={n:6d} scatter={'no' if rows is None else 'morton'}: {med:.3f} ms = {med/n*1e3:.4f} us/row -> {4.0*n*main.T/med/1e6:.0f} GB/s", flush=True)
