#!/usr/bin/env python3
"""Fluctuation parity at full size (atlast_10k, 24 rows): the three sampler rules against the oracle, on the coarse
loading and on the TOD, total and fluctuation-only (per-detector mean removed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from oracle import hotpath

def fluct(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fa, fb = a - a.mean(-1, keepdims=True), b - b.mean(-1, keepdims=True)
    return np.abs(fa - fb).max() / np.abs(fb).max()
def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()

p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
rng = np.random.default_rng(1)
rows = np.sort(rng.choice(path.D, 24, replace=False))
sub = dict(p)
for key in ("offsets", "band_index", "m00"):
    sub[key] = p[key][rows]
sub["layers"] = [dict(l, values=b[0].cpu().numpy()) for l, b in zip(p["layers"], path._layer_bufs)]
ref, inter = hotpath.run_path(sub, return_intermediates=True)
out = {}
for name, opts in (("pixel (default)", {}), ("axis literal", {1: 1}), ("pointing chain", {0: 1})):
    for k, v in opts.items():
        path.ctx.set_option(k, v)
    tod = path.run(blocks=1)
    torch.cuda.synchronize()
    coarse = path.coarse_loading()[rows].cpu().numpy()
    got = tod[rows].cpu().numpy()
    pwv = path.coarse_pwv()[rows].cpu().numpy()
    print(f"{name:18s} pwv rel {rel(pwv, inter['pwv']):.3e} fluct {fluct(pwv, inter['pwv']):.3e}")
    for k in opts:
        path.ctx.set_option(k, 0)
    out[name] = (got, coarse)
    if not opts:
        # the spline alone: scipy's float64 spline of the GPU's OWN coarse loading, in float64 and rounded to float32
        import scipy.interpolate
        own64 = scipy.interpolate.interp1d(p["ta"], coarse.astype(np.float64), kind="cubic", axis=-1, fill_value="extrapolate")(p["t"])
        print(f"{name:18s} GPU TOD vs scipy spline of the GPU's coarse loading: fluct {fluct(got, own64):.3e} (float32 of it: {fluct(got, own64.astype(np.float32)):.3e}; "
              f"float32 rounding alone: {fluct(own64.astype(np.float32), own64):.3e})")
        ref64 = scipy.interpolate.interp1d(p["ta"], inter["loading_a"].astype(np.float64), kind="cubic", axis=-1, fill_value="extrapolate")(p["t"])
        print(f"{name:18s} scipy spline of GPU coarse vs of oracle coarse (float64): fluct {fluct(own64, ref64):.3e}")
    print(f"{name:18s} vs oracle: TOD rel {rel(got, ref):.3e} fluct {fluct(got, ref):.3e} | coarse rel {rel(coarse, inter['loading_a']):.3e} fluct {fluct(coarse, inter['loading_a']):.3e}")
a, b = out["pixel (default)"], out["axis literal"]
print(f"pixel vs axis literal: TOD rel {rel(a[0], b[0]):.3e} fluct {fluct(a[0], b[0]):.3e}")
a, b = out["axis literal"], out["pointing chain"]
print(f"axis literal vs pointing chain: TOD rel {rel(a[0], b[0]):.3e} fluct {fluct(a[0], b[0]):.3e}")
print("fluctuation / mean of the reference TOD:", float(np.abs(ref - ref.mean(-1, keepdims=True)).max() / np.abs(ref).max()))
