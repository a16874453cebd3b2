#!/usr/bin/env python3
"""Development aid: randomised shapes through the kernels added late in round 2 -- the routed binning against the atomic
form, the noise register transforms against the LDS ones, the register screen transforms against the Stockham ones."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from maria_amd import _lib
from maria_amd._lib import Context, MrxSkyMap, ptr

dev = "cuda:0"
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream(torch.device(dev)))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    D, T = int(rng.integers(1, 70)), int(rng.integers(1, 5000))
    S, Cn = int(rng.integers(1, 5)), int(rng.integers(1, 4))
    n_eta, n_xi = int(rng.integers(2, 200)), int(rng.integers(2, 300))
    bil = int(rng.integers(0, 2))
    t = np.arange(T) / 50.0
    az = 0.8 + 0.01 * np.sin(t * rng.uniform(0.1, 2)) + rng.uniform(-1e-3, 1e-3)
    el = 1.0 + 0.01 * np.cos(t * rng.uniform(0.1, 2))
    off = rng.normal(0, 0.004, (D, 2))
    half = rng.uniform(0.005, 0.03)
    sky = MrxSkyMap(None, Cn, S, n_eta, n_xi, half, -2 * half / (n_eta - 1), -half, 2 * half / (n_xi - 1), 0.8, 1.0, bil, 0)
    tod, w = f32(rng.normal(1, 0.5, (D, T))), (f32(rng.uniform(0, 2, (D, T))) if rng.random() < 0.6 else None)
    if w is not None and rng.random() < 0.3:
        w[rng.integers(0, D)] = 0.0  # a dead detector
    sw = torch.as_tensor(rng.normal(0.5, 0.5, (D, S)) * (rng.random((D, S)) > 0.2)).to(dev)
    chan = torch.as_tensor(rng.integers(-1, Cn + 1, D).astype(np.int32)).to(dev) if rng.random() < 0.7 else None
    d_az, d_el, d_dx, d_dy = f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    args = (C.byref(sky), ptr(tod), tod.stride(0), ptr(w), 0 if w is None else w.stride(0), ptr(d_az), ptr(d_el), T, None, ptr(d_dx), ptr(d_dy),
            ptr(sw), ptr(chan), D)
    ref = [torch.zeros((S, Cn, n_eta, n_xi), dtype=torch.float64, device=dev) for _ in range(2)]
    got = [torch.zeros_like(r) for r in ref]
    ctx.call("mrx_bin_map", *args, ptr(ref[0]), ptr(ref[1]))
    lo, full = C.c_size_t(), C.c_size_t()
    assert ctx.lib.mrx_bin_map_work_bytes(C.byref(sky), D, T, C.byref(lo), C.byref(full)) == 0
    size = int(rng.choice([lo.value, full.value, (lo.value + full.value) // 2 // 16 * 16]))
    work = torch.empty(max(size, lo.value), dtype=torch.uint8, device=dev)
    ctx.call("mrx_bin_map_bucketed", *args, ptr(got[0]), ptr(got[1]), ptr(work), work.numel())
    torch.cuda.synchronize()
    for g, r, name in zip(got, ref, ("sum", "wgt")):
        scale = float(r.abs().max()) or 1.0
        err = float((g - r).abs().max()) / scale
        if not err <= 1e-11:
            bad += 1
            print(f"BIN MISMATCH trial {trial} {name}: D={D} T={T} S={S} C={Cn} map {n_eta}x{n_xi} bil={bil} err {err:.3g}")
print("binning trials done, mismatches:", bad)

# noise: register forms vs LDS forms over random T in every period class
for trial in range(12):
    T = int(rng.integers(5000, 1 << int(rng.integers(13, 22))))
    D = int(rng.integers(1, 9))
    modes = int(rng.integers(0, 6))
    need = C.c_size_t()
    ctx.lib.mrx_noise_work_floats(T, modes, D, C.byref(need))
    work = torch.empty(need.value, dtype=torch.float32, device=dev)
    basis = f32(rng.normal(size=(D, max(modes, 1)))) if modes else None
    outs = []
    for opt in (0, 3):
        ctx.set_option(5, opt)
        out = torch.zeros((D, T), dtype=torch.float32, device=dev)
        ctx.call("mrx_noise_generate", 9, D, 0, T, 200.0, 2.0, 0.4, ptr(basis), modes, None, None, 0, 0.0, ptr(out), out.stride(0), 0, ptr(work), need.value)
        outs.append(out)
    ctx.set_option(5, 0)
    torch.cuda.synchronize()
    err = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
    if not err <= 3e-5:
        bad += 1
        print(f"NOISE MISMATCH T={T} D={D} modes={modes}: {err:.3g}")
print("noise trials done, mismatches so far:", bad)
sys.exit(1 if bad else 0)
