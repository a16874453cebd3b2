#!/usr/bin/env python3
"""Development aid: where mrx_map_sample_krj differs from mrx_map_sample + mrx_tod_to_krj (it should not, bit for bit).
Prints, for the first case of tests/test_gpu_map.py::test_map_field_written_in_krj_equals_sampling_then_tod_to_krj, how many
values differ, by how many float32 ulps, in which rows and at which of a thread's four samples -- and the same with a
denominator table that is 1 everywhere (then the K_RJ field IS the pW field: what differs there is the sampler's own output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from maria_amd import map as mmap, synthetic
from maria_amd._lib import Context, ptr
from test_gpu_map import _blob_map, _centre

ctx = Context(0)
dev = "cuda:0"
f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
rate, D, T = 400.0, 37, 3001
rng = np.random.default_rng(int(rate) + D + T)
t = 1.7e9 + np.arange(T) / rate
az, el = synthetic.daisy_scan(t)
az, el = az.astype(np.float32), el.astype(np.float32)
off = synthetic.hex_pack(max(D, 2), np.radians(0.4))[:D]
centre = _centre(az, el, None)
eta, xi = np.linspace(0.02, -0.02, 9), np.linspace(-0.02, 0.02, 9)
values = _blob_map(2, 1, 9, 9, eta, xi, rng)
w = np.ones((D, 1)) * 0.5
axis_pwv, axis_el_s = np.linspace(0.0, 6.0, 13), np.radians(np.linspace(20.0, 90.0, 15))
tabs = np.stack([(1.5e10 + 4e9 * c) * np.exp(-(0.05 + 0.03 * c + 0.04 * axis_pwv[:, None]) / np.sin(axis_el_s)[None, :]) for c in range(2)])
ta = np.arange(t[0], t[-1] + 1.0, 0.5)
coarse = 1.2 + 0.3 * np.cumsum(rng.normal(0, 0.05, (D, len(ta))), axis=1)
kws = {"cal_tables": dict(cal_tables=tabs.astype(np.float32), cal_axis_pwv=axis_pwv, cal_axis_el=axis_el_s, coarse_pwv=coarse.T, ta0=ta[0], dta=0.5, t=t),
       "cal_scalars": dict(cal_scalars=[1.0e10, 1.3e10])}
n_el, n_bands = 29, 2
axis = np.radians(np.linspace(20.0, 90.0, n_el))
den = np.stack([(2.0e-2 + 5e-3 * b) * np.exp(-(0.04 + 0.02 * b) / np.sin(axis)) for b in range(n_bands)])
band = torch.as_tensor(rng.integers(0, n_bands, D).astype(np.int32)).to(dev)
scale = f32(rng.uniform(0.9, 1.1, D))


def ulps(a, b):
    ia, ib = a.view(torch.int32).to(torch.int64), b.view(torch.int32).to(torch.int64)
    return (ia - ib).abs()


for cal_name, kw in kws.items():
    for den_name, dv in (("den table", den), ("den = 1", np.ones_like(den))):
        for sc in (scale, None):
            krj = dict(bore_el=f32(el), dx=f32(off[:, 0]), dy=f32(off[:, 1]), band=band, axis=f32(axis), values=f32(dv))
            pw = mmap.sample_map(ctx, values, eta, xi, centre, az, el, off, w, **kw)
            ref = pw.clone()
            if sc is not None:
                ref *= sc[:, None]
            pre = ref.clone()
            ctx.call("mrx_tod_to_krj", ptr(ref), ref.stride(0), D, T, None, None, ptr(krj["bore_el"]), ptr(krj["dx"]), ptr(krj["dy"]),
                     ptr(krj["band"]), ptr(krj["axis"]), ptr(krj["values"]), n_el, n_bands)
            got = mmap.sample_map(ctx, values, eta, xi, centre, az, el, off, w, krj=krj, scale=sc, **kw)
            torch.cuda.synchronize()
            u = ulps(got, ref)
            bad = u > 0
            n = int(bad.sum())
            line = f"{cal_name:12s} {den_name:10s} scale {'yes' if sc is not None else 'no ':3s}: differing {n:7d} of {got.numel()}  max ulp {int(u.max())}"
            if den_name == "den = 1":
                line += f" | two-pass K_RJ == scaled pW field: {bool(torch.equal(ref, pre))}; fused == scaled pW field: {bool(torch.equal(got, pre))} (max ulp {int(ulps(got, pre).max())})"
            print(line)
            if n:
                rows = bad.any(dim=1).nonzero().flatten().tolist()
                q = torch.arange(T, device=dev) % 4
                per_q = [int((bad & (q == k)[None, :]).sum()) for k in range(4)]
                tiles = sorted(set((bad.any(dim=0).nonzero().flatten() // 1024).tolist()))
                print(f"    rows {rows[:20]}{' ...' if len(rows) > 20 else ''} ({len(rows)} of {D}); by sample index mod 4: {per_q}; sample tiles {tiles}")
                print(f"    histogram of ulps: {torch.bincount(u[bad].clamp(max=8)).tolist()}")
                d0, s0 = [int(x) for x in bad.nonzero()[0]]
                print(f"    first: row {d0} sample {s0}: fused {float(got[d0, s0])!r} two-pass {float(ref[d0, s0])!r} pW*scale {float(pre[d0, s0])!r}")
