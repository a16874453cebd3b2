#!/usr/bin/env python3
"""Development aid: random shapes through the small entry points against scipy / the oracle: mrx_gauss_smooth2d
(scipy.ndimage.gaussian_filter, reflect), mrx_map_smooth (weighted, with holes), mrx_linear_upsample, mrx_pointing_broadcast
(incl. azimuths across the wrap and elevations near the zenith).
Usage: python scripts/fuzz_small_kernels.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage
import torch
from maria_amd._lib import Context, ptr
from oracle import hotpath

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream())
dev = "cuda:0"
bad = 0
f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731


def report(label, msgs):
    global bad
    bad += bool(msgs)
    print(f"{label}: {'ok' if not msgs else 'BAD ' + '; '.join(msgs)}", flush=True)


for trial in range(trials):
    # --- Gaussian smoothing
    ny, nx = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    if rng.random() < 0.2:
        ny, nx = int(rng.integers(1, 12)), int(rng.integers(1, 3000))
    sy, sx = [float(rng.choice([0.0, rng.uniform(0.05, 1.0), rng.uniform(1.0, 12.0), rng.uniform(12.0, 40.0)])) for _ in range(2)]
    a = rng.standard_normal((ny, nx)).astype(np.float32) + np.float32(rng.choice([0.0, 100.0]))
    msgs = []
    try:
        d_in, d_tmp = f32(a), torch.empty((ny, nx), dtype=torch.float32, device=dev)
        d_out = torch.empty_like(d_in)
        ctx.call("mrx_gauss_smooth2d", ptr(d_in), ptr(d_out), ptr(d_tmp), ny, nx, sy, sx, 4.0)
        ref = scipy.ndimage.gaussian_filter(a.astype(np.float64), sigma=(sy, sx), mode="reflect", truncate=4.0)
        e = np.abs(d_out.cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-30)
        if not e <= 3e-6:
            msgs.append(f"smooth2d {e:.2e}")
        ctx.call("mrx_gauss_smooth2d", ptr(d_in), ptr(d_in), ptr(d_tmp), ny, nx, sy, sx, 4.0)  # in place
        if not torch.equal(d_in, d_out):
            msgs.append("in place differs")
    except Exception as exc:  # noqa: BLE001
        msgs.append(f"{type(exc).__name__}: {exc}")
    report(f"trial {trial}: smooth2d {ny}x{nx} sigma ({sy:.2f}, {sx:.2f})", msgs)

    # --- weighted map smoothing
    ny, nx = int(rng.integers(2, 400)), int(rng.integers(2, 400))
    sy, sx = float(rng.uniform(0.3, 8.0)), float(rng.uniform(0.3, 8.0))
    data = rng.standard_normal((ny, nx)).astype(np.float32)
    weight = None
    if rng.random() < 0.7:
        weight = rng.random((ny, nx)).astype(np.float32)
        for _ in range(int(rng.integers(0, 3))):
            y0, x0 = int(rng.integers(0, ny)), int(rng.integers(0, nx))
            weight[y0 : y0 + int(rng.integers(1, 80)), x0 : x0 + int(rng.integers(1, 80))] = 0.0
    msgs = []
    try:
        ref, ref_den = hotpath.map_smooth(data, weight, sy, sx)
        d_data, d_w = f32(data), None if weight is None else f32(weight)
        d_out, d_den = torch.empty_like(d_data), torch.empty_like(d_data)
        d_tmp = torch.empty(2 * ny * nx, dtype=torch.float32, device=dev)
        ctx.call("mrx_map_smooth", ptr(d_data), ptr(d_w), ptr(d_out), ptr(d_den), ptr(d_tmp), ny, nx, sy, sx)
        got, den = d_out.cpu().numpy(), d_den.cpu().numpy()
        # where the denominator is at float32 rounding of zero the quotient is noise in both: compare where it is not
        solid = ref_den > 1e-5
        e = np.abs(got - ref)[solid].max() / max(np.abs(ref).max(), 1e-30) if solid.any() else 0.0
        e2 = np.abs(den - ref_den).max()
        if not (e <= 2e-5 and e2 <= 2e-6 and np.isfinite(got).all()):
            msgs.append(f"map_smooth {e:.2e} denominator {e2:.2e}")
        if not (got[ref_den == 0] == 0).all():
            msgs.append("holes not zero")
    except Exception as exc:  # noqa: BLE001
        msgs.append(f"{type(exc).__name__}: {exc}")
    report(f"trial {trial}: map_smooth {ny}x{nx} sigma ({sy:.2f}, {sx:.2f}) weighted={weight is not None}", msgs)

    # --- linear upsample of the coarse pwv
    D, Ta = int(rng.integers(1, 300)), int(rng.integers(2, 800))
    dta = float(rng.uniform(0.05, 2.0))
    ta0 = float(rng.choice([0.0, 1.7e9]))
    ta = ta0 + dta * np.arange(Ta)
    fs = float(rng.choice([20.0, 50.0, 400.0]))
    t = np.arange(ta[0], ta[-1] + 0.9 * dta, 1.0 / fs)
    pwv = 1 + 0.05 * rng.standard_normal((D, Ta))
    msgs = []
    try:
        ref = hotpath.upsample_linear(ta, pwv, t)
        d_p, d_t = torch.as_tensor(np.ascontiguousarray(pwv.T)).to(dev), torch.as_tensor(t).to(dev)
        ld = len(t) + int(rng.integers(0, 5))
        d_out = torch.full((D, ld), -9.0, dtype=torch.float32, device=dev)
        ctx.call("mrx_linear_upsample", ptr(d_p), D, Ta, ta0, dta, ptr(d_t), len(t), ptr(d_out), ld)
        got = d_out.cpu().numpy()
        e = np.abs(got[:, : len(t)] - ref).max() / np.abs(ref).max()
        # (at unix times a float64 knot time is only good to 2.4e-7 s: the oracle's weights, from the rounded knots as
        # numpy's interp sees them, and the kernel's, from (t - t0) / dt, differ by that over dt of a knot-to-knot step)
        room = 3e-7 + 2 * (np.spacing(ta[-1]) / dta) * np.abs(np.diff(pwv, axis=1)).max() / np.abs(ref).max()
        if not e <= room:
            msgs.append(f"{e:.2e}")
        if not (got[:, len(t) :] == -9.0).all():
            msgs.append("wrote past T")
    except Exception as exc:  # noqa: BLE001
        msgs.append(f"{type(exc).__name__}: {exc}")
    report(f"trial {trial}: linear_upsample D={D} Ta={Ta} T={len(t)} t0={ta0:.0f}", msgs)

    # --- full-rate pointing
    D, T = int(rng.integers(1, 200)), int(rng.integers(1, 9000))
    az0 = float(rng.choice([0.0, 2 * np.pi - 1e-3, rng.uniform(0, 2 * np.pi)]))
    el0 = float(rng.choice([np.radians(89.0), rng.uniform(0.3, 1.5)]))
    az = az0 + np.cumsum(rng.normal(0, 2e-4, T))
    el = np.clip(el0 + np.cumsum(rng.normal(0, 1e-4, T)), 0.05, np.pi / 2 - 1e-3)
    off = rng.normal(0, float(rng.choice([1e-3, 0.01, 0.03])), (D, 2))
    msgs = []
    try:
        ref_az, ref_el = hotpath.broadcast(off, az, el)
        ld = T + int(rng.integers(0, 4))
        out_az = torch.full((D, ld), -9.0, dtype=torch.float32, device=dev)
        out_el = torch.full((D, ld), -9.0, dtype=torch.float32, device=dev)
        d_az, d_el, d_dx, d_dy = f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
        ctx.call("mrx_pointing_broadcast", ptr(d_az), ptr(d_el), T, ptr(d_dx), ptr(d_dy), D, ptr(out_az), ptr(out_el), ld)
        ga, ge = out_az.cpu().numpy()[:, :T], out_el.cpu().numpy()[:, :T]
        # an azimuth is an angle: compare on the circle, scaled by cos(el) (near the zenith it is ill-conditioned in both)
        da = np.abs((ga - ref_az + np.pi) % (2 * np.pi) - np.pi) * np.cos(ref_el)
        # the reference's float32 arcsin is ill-conditioned near the zenith (one ulp of sin(el) is 6e-8 / cos(el) of elevation):
        # a few ulp of sin(el) is what two float32 chains can agree to; against the float64 evaluation of the same formula
        # the kernel must be about as close as the float32 chain is
        room = 8e-7 + 4e-7 / np.cos(ref_el)
        e1, e2 = float(da.max()), float((np.abs(ge - ref_el) / room).max())
        a32, e32, dx32, dy32 = (np.asarray(v, np.float32).astype(np.float64) for v in (az, el, off[:, 0], off[:, 1]))
        r, pang = np.hypot(dx32, dy32)[:, None], np.arctan2(-dx32, -dy32)[:, None]
        w = (np.sin(r) * np.cos(pang) + 1j * np.cos(r)) * np.exp(1j * (e32[None, :] - np.pi / 2))
        exact_el = np.arcsin(np.imag(w))
        mine, theirs = float(np.abs(ge - exact_el).max()), float(np.abs(ref_el - exact_el).max())
        if not (e1 <= 8e-7 and e2 <= 1.0 and mine <= 1.5 * theirs + 2e-7):  # (both are float32 chains)
            msgs.append(f"az {e1:.2e} el {e2:.2f} of the room; against float64: kernel {mine:.2e}, float32 chain {theirs:.2e}")
    except Exception as exc:  # noqa: BLE001
        msgs.append(f"{type(exc).__name__}: {exc}")
    report(f"trial {trial}: pointing D={D} T={T} az0={az0:.3f} el0={el0:.3f}", msgs)
print("BAD" if bad else "all ok", bad)
