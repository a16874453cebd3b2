#!/usr/bin/env python3
"""Timing of mrx_map_sample on one GPU (development aid).
Usage: python scripts/map_bench.py [n_det] [n_samples] [reps]"""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import map as mmap  # noqa: E402
from maria_amd import synthetic  # noqa: E402
from maria_amd._lib import Context  # noqa: E402
from maria_amd.sim import sky_transform_stack  # noqa: E402
from scripts.kbench import timeit  # noqa: E402


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda:0")
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    t = 1.7e9 + np.arange(T) / 400.0
    az, el = synthetic.daisy_scan(t)
    off = synthetic.hex_pack(D, np.radians(1.0))
    n = 1024
    xi = np.linspace(-0.03, 0.03, n)
    eta = xi[::-1].copy()
    X, Y = np.meshgrid(xi, eta)
    values = np.exp(-(X**2 + Y**2) / 1e-4).astype(np.float32)[None, None]
    w = np.ones((D, 1))
    out = torch.empty((D, T), dtype=torch.float32, device=dev)
    ta = np.arange(t[0], t[-1], 0.1)
    for name, transform, cal in (("az/el, scalar cal", None, False), ("ra/dec, scalar cal", sky_transform_stack(t, -23.0, -67.8), False),
                                 ("ra/dec, atmosphere cal", sky_transform_stack(t, -23.0, -67.8), True)):
        # stage inputs once: time the kernel alone
        kw = dict(cal_scalars=[2e10])
        if cal:
            kw = dict(cal_tables=np.full((1, 24, 20), 2e10, np.float32), cal_axis_pwv=np.linspace(0, 8, 24), cal_axis_el=np.radians(np.linspace(5, 91, 20)),
                      coarse_pwv=torch.ones((len(ta), D), dtype=torch.float64, device=dev), ta0=ta[0], dta=0.1, t=t)
        tr = None if transform is None else torch.as_tensor(transform.reshape(T, 9)).to(dev)
        centre = (float(np.mean(az)), float(np.mean(el))) if transform is None else None
        if centre is None:
            from oracle import mapsample  # geometry helper only (development script)

            phi, theta = mapsample.frame_angles(az[None, ::1000].astype(np.float32), el[None, ::1000].astype(np.float32), transform[::1000])
            centre = (float(np.median(phi)), float(np.median(theta)))
        fn = lambda: mmap.sample_map(ctx, values, eta, xi, centre, az, el, off, w, out=out, transform=tr, **kw)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        med, mn = timeit(fn, reps)
        print(f"map_sample {name}: D={D} T={T}: median {med:.2f} ms (includes host staging of inputs) -> {D*T/med/1e6:.1f} G samples/s; "
              f"nonzero fraction {float((out != 0).float().mean()):.2f}")


if __name__ == "__main__" and not os.environ.get("MRX_BENCH_BIN"):
    main()


def bench_bin():
    """mrx_bin_map at the same scale (nearest pixel, the mapper's default)."""
    import ctypes as C

    from maria_amd._lib import MrxSkyMap, ptr

    D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
    dev = torch.device("cuda:0")
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    t = 1.7e9 + np.arange(T) / 400.0
    az, el = synthetic.daisy_scan(t)
    off = synthetic.hex_pack(D, np.radians(1.0))
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    tod = torch.randn((D, T), dtype=torch.float32, device=dev)
    d_az, d_el, d_dx, d_dy = f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    d_sw = torch.ones((D, 1), dtype=torch.float64, device=dev)
    for n in (256, 1024, 2048):
        for bil in (0, 1):
            msum = torch.zeros((1, 1, n, n), dtype=torch.float64, device=dev)
            mwgt = torch.zeros_like(msum)
            step = 0.05 / n
            sky = MrxSkyMap(None, 1, 1, n, n, 0.025, -step, -0.025, step, float(np.mean(az)), float(np.mean(el)), bil, 0)
            fn = lambda: ctx.call("mrx_bin_map", C.byref(sky), ptr(tod), tod.stride(0), None, 0, ptr(d_az), ptr(d_el), T, None,  # noqa: E731
                                  ptr(d_dx), ptr(d_dy), ptr(d_sw), None, D, ptr(msum), ptr(mwgt))
            med, mn = timeit(fn, 3)
            print(f"bin_map {n}x{n} bilinear={bil}: D={D} T={T}: median {med:.2f} ms -> {D*T/med/1e6:.1f} G samples/s "
                  f"({4.0*D*T/med/1e6:.0f} GB/s of TOD read); hit pixels {int((mwgt > 0).sum())}")
            if True:
                lo, full = C.c_size_t(), C.c_size_t()
                ctx.lib.mrx_bin_map_work_bytes(C.byref(sky), D, T, C.byref(lo), C.byref(full))
                one_atomic = mwgt / 4  # timeit ran the atomic form 4 times into the same maps
                for frac in ((1.0, 0.25) if bil == 0 else (0.125,)):
                    work = torch.empty(int(full.value * frac), dtype=torch.uint8, device=dev)
                    mwgt.zero_()
                    msum.zero_()
                    fn2 = lambda: ctx.call("mrx_bin_map_bucketed", C.byref(sky), ptr(tod), tod.stride(0), None, 0, ptr(d_az), ptr(d_el), T, None,  # noqa: E731
                                           ptr(d_dx), ptr(d_dy), ptr(d_sw), None, D, ptr(msum), ptr(mwgt), ptr(work), work.numel())
                    fn2()
                    torch.cuda.synchronize()
                    same = float((mwgt - one_atomic).abs().max() / one_atomic.abs().max())
                    med, mn = timeit(fn2, 3)
                    print(f"bin_map_bucketed {n}x{n} bilinear={bil}: work {work.numel() / 2**30:.1f} GiB: median {med:.2f} ms -> {D*T/med/1e6:.1f} G samples/s; "
                          f"max |weight map - atomic form's| / max {same:.3g}")
                    del work

if __name__ == "__main__" and os.environ.get("MRX_BENCH_BIN"):
    bench_bin()
