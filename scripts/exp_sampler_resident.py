#!/usr/bin/env python3
"""Development aid: the sampler of one detector block alone, as a full grid and as a resident grid of k workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic, _lib
from maria_amd._lib import ptr
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
st = path._pipeline_state(8)
lo, hi = st["bounds"][3]
n = hi - lo
sl = lambda t: None if t is None else ptr(t[lo:hi])
for k, kt in ((0, 0), (8, 0), (6, 0), (5, 0), (4, 0), (3, 0), (2, 0), (5, 2), (4, 2), (3, 2), (2, 2), (5, 4), (3, 4)):
    path.ctx.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, k)
    path.ctx.set_option(_lib.OPT_SAMPLE_TIMES, kt)
    fn = lambda: path.ctx.call("mrx_atm_sample", path.plan, ptr(path.d_az), ptr(path.d_el), path.Ta, sl(path.d_dx), sl(path.d_dy), sl(path.d_band), sl(path.d_m00), n,
                               path.pwv0, None, ptr(st["loading"][3]), ptr(path.d_flags))
    med, mn = timeit(fn, 20)
    print(f"block of {n} rows, resident workgroups per CU {k or 'default'}, time steps per thread {kt or 1}: {1e3*med:.1f} us (min {1e3*mn:.1f})", flush=True)
path.ctx.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, 0)
path.ctx.set_option(_lib.OPT_SAMPLE_TIMES, 0)
