#!/usr/bin/env python3
"""Development aid: randomised shapes through the round-3 kernels against their references -- the fused writer
(solve in the tile prologue) against scipy's not-a-knot spline, the float32 anchor+delta sampler against the literal cell
search and the oracle, the amplitude tables against numpy's FFT of the image-summed Matern covariance.
Usage: python scripts/fuzz_round3.py [seed] [trials]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.interpolate
import torch
from maria_amd import _lib, synthetic
from maria_amd._lib import Context, ptr
from maria_amd.pipeline import DevicePath, matern_log_tables
from oracle import hotpath, screens

dev = "cuda:0"
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream(torch.device(dev)))
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
bad = 0


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-300))


# 1. fused writer: random lengths, ratios, phases, gains, row maps, pitches
for trial in range(trials):
    D, Ta = int(rng.integers(1, 90)), int(rng.integers(4, 900))
    ratio = float(np.exp(rng.uniform(0, np.log(500))))
    dta = float(rng.uniform(0.05, 0.3))
    ta = 1.7e9 * (rng.random() < 0.5) + dta * np.arange(Ta)
    lead, tail = rng.uniform(-0.6, 0.4) * dta, rng.uniform(-0.4, 1.6) * dta
    t = np.arange(ta[0] + lead, ta[-1] + tail, dta / ratio)
    if len(t) < 1 or len(t) > 400000:
        continue
    T = len(t)
    y = (rng.uniform(1, 30) + np.cumsum(rng.standard_normal((D, Ta)), axis=1) * rng.uniform(1e-3, 0.1)).astype(np.float32)
    ref = scipy.interpolate.interp1d(ta, y, kind="cubic", bounds_error=False, fill_value="extrapolate", axis=-1)(t)
    scale = rng.uniform(0.5, 2.0, D).astype(np.float32) if rng.random() < 0.5 else None
    rows = rng.permutation(D).astype(np.int32) if rng.random() < 0.5 else None
    ld = T + int(rng.integers(0, 5))
    d_y = torch.as_tensor(np.ascontiguousarray(y.T)).to(dev)
    d_t = torch.as_tensor(t).to(dev)
    d_out = torch.full((D, ld), -7.0, dtype=torch.float32, device=dev)
    d_scale = None if scale is None else torch.as_tensor(scale).to(dev)
    d_rows = None if rows is None else torch.as_tensor(rows).to(dev)
    ctx.call("mrx_spline_upsample_fused", ptr(d_y), D, Ta, float(ta[0]), dta, ptr(d_t), T, ptr(d_scale), ptr(d_rows), ptr(d_out), ld)
    out = d_out.cpu().numpy()
    want = ref if scale is None else ref * scale[:, None]
    if rows is not None:
        full = np.empty_like(want)
        full[rows] = want
        want = full
    err = rel(out[:, :T], want)
    # (one float32 rounding of the value and one of the gain; samples extrapolated past the last knot -- up to 1.6 coarse
    # steps here, a fraction of one in the reference -- amplify the float32 second derivatives: 5e-7 seen)
    ok = err <= 1e-6 and (out[:, T:] == -7.0).all()
    bad += not ok
    if not ok or trial < 3:
        print(f"writer D={D} Ta={Ta} ratio={ratio:.2f} T={T} scale={scale is not None} rows={rows is not None} ld-T={ld - T}: {err:.2e} {'ok' if ok else 'BAD'}", flush=True)

# 2. sampler: random small problems, the default rule against the literal one and the oracle
for trial in range(max(4, trials // 4)):
    n_det, n_layers = int(rng.integers(1, 400)), int(rng.integers(1, 9))
    p = synthetic.make_problem(n_det=n_det, n_bands=int(rng.integers(1, 4)), fov_deg=float(rng.uniform(0.05, 2.0)), fs=float(rng.choice([20.0, 50.0, 100.0])),
                               duration=float(rng.uniform(5.0, 40.0)), n_layers=n_layers, side=int(rng.choice([128, 256, 512])), seed=int(rng.integers(1, 1 << 30)))
    path = DevicePath(p, device=dev, ctx=ctx)
    path.generate_screens()
    a = path.run().cpu().numpy().copy()
    flags = path.check_flags()
    ctx.set_option(_lib.OPT_AXIS_LITERAL, 1)
    b = path.run().cpu().numpy().copy()
    ctx.set_option(_lib.OPT_AXIS_LITERAL, 0)
    for layer, s in zip(p["layers"], path._gen_screens):
        layer["values"] = s.cpu().numpy()
    ref = hotpath.run_path(p)
    e1, e2 = rel(a, ref), rel(a, b)
    ok = e1 <= 1e-5 and e2 <= 1e-5 and flags == 0
    bad += not ok
    print(f"sampler n_det={n_det} layers={n_layers} T={a.shape[1]}: vs oracle {e1:.2e}, vs literal rule {e2:.2e} {'ok' if ok else 'BAD'}", flush=True)

# 2b. sampler variants: stretched axes (array search), cubic emission tables, the float32 pointing chain, gains, row
# order, the pipelined run in 2-5 blocks against the serial one (bit for bit) -- all against the oracle
for trial in range(max(6, trials // 4)):
    n_det = int(rng.integers(600, 2200))
    cubic = bool(rng.random() < 0.35)
    p = synthetic.make_problem(n_det=n_det, n_bands=int(rng.integers(1, 4)), fov_deg=float(rng.uniform(0.1, 1.5)), fs=float(rng.choice([20.0, 50.0])),
                               duration=float(rng.uniform(8.0, 30.0)), n_layers=int(rng.integers(1, 6)), side=int(rng.choice([128, 256])),
                               seed=int(rng.integers(1, 1 << 30)))
    if cubic:
        p["interpolation_method"] = "cubic"
    if rng.random() < 0.5:
        p["gain"] = rng.uniform(0.8, 1.2, n_det)
    path = DevicePath(p, device=dev, ctx=ctx)
    path.generate_screens()
    stretched = False
    if rng.random() < 0.5 and not cubic:  # stretch one axis of one layer smoothly: the plan then searches the axis array
        scr = [s.clone() for s in path._gen_screens]
        l = int(rng.integers(0, len(p["layers"])))
        key = "cross_section" if rng.random() < 0.5 else "extrusion"
        ax = p["layers"][l][key]
        mid = 0.5 * (ax[0] + ax[-1])
        p["layers"][l][key] = mid + (ax - mid) * (1.0 + 0.2 * ((ax - mid) / (ax[-1] - mid)) ** 2)
        for layer, s_ in zip(p["layers"], scr):
            layer["values"] = s_.cpu().numpy()
        path = DevicePath(p, device=dev, ctx=ctx)
        path.set_screens(scr)
        stretched = True
    chain = bool(rng.random() < 0.3)
    ctx.set_option(_lib.OPT_POINTING_CHAIN, 1 if chain else 0)
    serial = path.run(blocks=1).clone()
    flags = path.check_flags()
    blocks = int(rng.integers(2, 6))
    piped = path.run(blocks=blocks).clone()
    ctx.set_option(_lib.OPT_POINTING_CHAIN, 0)
    if not stretched:
        for layer, s_ in zip(p["layers"], path._gen_screens):
            layer["values"] = s_.cpu().numpy()
    ref = hotpath.run_path(p)
    e1 = rel(serial.cpu().numpy(), ref)
    same = bool(torch.equal(serial, piped))
    ok = e1 <= 1e-5 and same and flags == 0
    bad += not ok
    print(f"variants n_det={n_det} cubic={cubic} stretched={stretched} chain={chain} gain={p.get('gain') is not None} blocks={blocks}: vs oracle {e1:.2e}, "
          f"pipelined == serial {same} {'ok' if ok else 'BAD'}", flush=True)

# 3. amplitude tables
for trial in range(max(4, trials // 6)):
    nh = int(rng.choice([0, 0, 8, 16]))
    ny, nx = int(rng.choice([64, 128, 256, 512])), int(rng.choice([64, 128, 256]))
    nu = float(rng.choice([5 / 6, 1 / 3]))
    dh, dy, dx = float(rng.uniform(20, 80)), float(rng.uniform(2, 10)), float(rng.uniform(2, 10))
    r0 = float(rng.uniform(0.08, 0.5) * min(ny * dy, nx * dx, nh * dh if nh else 1e9))
    lf, ls, lc, lsf, xc = matern_log_tables(nu)
    n_t, n_w = C.c_size_t(), C.c_size_t()
    _lib.load().mrx_screen_amp_floats(nh, ny, nx, len(lc), C.byref(n_t), C.byref(n_w))
    table = torch.empty(n_t.value, dtype=torch.float32, device=dev)
    work = torch.empty(n_w.value, dtype=torch.float32, device=dev)
    as_d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    ctx.call("mrx_screen_amplitudes", nh, ny, nx, dh, dy, dx, r0, as_d(lc), as_d(lsf), len(lc), lf, ls, xc, ptr(table), ptr(work), work.numel())
    host = table.cpu().numpy()
    mz, my, mx = (nh // 2 + 1 if nh else 1), ny // 2 + 1, nx // 2 + 1
    got = host[4 : 4 + mx * mz * my].reshape(mx, mz, my).transpose(1, 2, 0)
    shape, steps = ((nh, ny, nx), (dh, dy, dx)) if nh else ((ny, nx), (dy, dx))
    ref, rho0 = screens.covariance_amplitude(shape, steps, r0, nu, x_cut=xc)
    ref = ref.reshape((nh if nh else 1, ny, nx))[:mz, :my, :mx]
    err = float(np.abs(got - ref).max() / ref.max())
    ok = err <= 3e-6
    bad += not ok
    print(f"amplitudes {shape} r0={r0:.0f} nu={nu:.2f}: {err:.2e} {'ok' if ok else 'BAD'}", flush=True)
print("BAD" if bad else "all ok", bad)
