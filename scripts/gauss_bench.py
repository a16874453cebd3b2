#!/usr/bin/env python3
"""The stencil the north star names, timed: mrx_gauss_smooth2d (atmosphere/atmosphere.py:341-344 on caller-supplied
screens) and mrx_map_smooth (ProjectionMap.smooth, map/projection.py:485-504; uniform weights and a weight plane) on
1024^2 / 4096^2 / 8192^2 at sigma = 2, 8, 32 pixels.  Prints one line per case with its fraction of 8 TB/s on SURVEY
8(d)'s two-pass accounting 16 N^2 (what the kernels are) and on the fused-tile floor 8 N^2.
Usage: python scripts/gauss_bench.py [reps]      (under rocprofv3 --kernel-trace --stats for profiles/r05_gauss_*)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd._lib import Context, ptr
from scripts.kbench import timeit

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream())
# MRX_GAUSS_ACCUM=1: the exact (float64) mode at every radius; 2: the blocked float32 sums at every radius; default 0
ctx.set_option(12, int(os.environ.get("MRX_GAUSS_ACCUM", "0")))
print(f"# MRX_OPT_GAUSS_ACCUM = {os.environ.get('MRX_GAUSS_ACCUM', '0')} (0: blocked float32 sums from radius 16, exact float64 below; 1: exact; 2: blocked)")
rows = []
for n in [int(v) for v in os.environ.get("MRX_GAUSS_N", "1024,4096,8192").split(",")]:
    x = torch.rand((n, n), dtype=torch.float32, device="cuda:0")
    w = torch.rand((n, n), dtype=torch.float32, device="cuda:0") + 0.5
    out, den = torch.empty_like(x), torch.empty_like(x)
    tmp = torch.empty(2 * n * n, dtype=torch.float32, device="cuda:0")
    for sigma in [float(v) for v in os.environ.get("MRX_GAUSS_SIGMA", "2,8,32").split(",")]:
        cases = {
            "gauss_smooth2d": lambda: ctx.call("mrx_gauss_smooth2d", ptr(x), ptr(out), ptr(tmp), n, n, sigma, sigma, 4.0),
            "map_smooth": lambda: ctx.call("mrx_map_smooth", ptr(x), None, ptr(out), None, ptr(tmp), n, n, sigma, sigma),
            "map_smooth_weighted": lambda: ctx.call("mrx_map_smooth", ptr(x), ptr(w), ptr(out), ptr(den), ptr(tmp), n, n, sigma, sigma),
        }
        for name, fn in cases.items():
            med, mn = timeit(fn, reps)
            # bytes: one plane in, one out per pass; the weighted form filters two planes and writes the quotient + denom
            planes = 1 if name != "map_smooth_weighted" else 2
            two_pass = 16.0 * n * n * planes
            fused = 8.0 * n * n * planes
            row = dict(kernel=name, n=n, sigma_px=sigma, radius=int(4 * sigma + 0.5), ms=med, ms_min=mn,
                       frac_two_pass=two_pass / (med * 1e-3) / 8e12, frac_fused_floor=fused / (med * 1e-3) / 8e12)
            rows.append(row)
            print(f"{name:20s} {n:5d}^2 sigma {sigma:4.0f} px: {med:8.3f} ms (min {mn:.3f})  {two_pass / med / 1e9:7.2f} TB/s on 16 N^2 = {row['frac_two_pass']:.3f} of 8 TB/s; on 8 N^2 {row['frac_fused_floor']:.3f}", flush=True)
    del x, w, out, den, tmp
    torch.cuda.empty_cache()
print(json.dumps(rows))
