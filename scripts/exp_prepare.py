#!/usr/bin/env python3
"""Development aid: the spline solve for the whole shard and for one pipeline block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd._lib import ptr
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens(); path.sample()
med, mn = timeit(path.prepare, 30)
st = path._pipeline_state(8)
lo, hi = st["bounds"][3]
path._run_pipelined(torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0"), 8)
fn = lambda: path.ctx.call("mrx_spline_prepare", ptr(st["loading"][3]), hi - lo, path.Ta, ptr(st["ym"][3]))
med2, mn2 = timeit(fn, 30)
print(f"{os.environ.get('MRX_LIB_PATH', 'product')}: prepare whole shard {1e3*med:.1f} us (min {1e3*mn:.1f}); block of {hi-lo} rows {1e3*med2:.1f} us (min {1e3*mn2:.1f})")
