#!/usr/bin/env python3
"""Timing of the two forms of the binning on one GPU (development aid).
Usage: python scripts/bin_bench.py [n_map] [bucketed 0|1] [reps] [n_det] [n_samples]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import synthetic  # noqa: E402
from maria_amd._lib import Context, MrxSkyMap, ptr  # noqa: E402
from scripts.kbench import timeit  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bucketed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    D = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
    T = int(sys.argv[5]) if len(sys.argv) > 5 else 240000
    dev = torch.device("cuda:0")
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    t = 1.7e9 + np.arange(T) / 400.0
    az, el = synthetic.daisy_scan(t)
    off = synthetic.hex_pack(D, np.radians(1.0))
    if os.environ.get("MRX_BIN_MORTON"):  # detectors in Z-order on the focal plane: a tile's 16 detectors are neighbours
        from maria_amd.pipeline import morton_order

        off = off[morton_order(off)]
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    tod = torch.randn((D, T), dtype=torch.float32, device=dev)
    d_az, d_el, d_dx, d_dy = f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    d_sw = torch.ones((D, 1), dtype=torch.float64, device=dev)
    msum = torch.zeros((1, 1, n, n), dtype=torch.float64, device=dev)
    mwgt = torch.zeros_like(msum)
    step = 0.05 / n
    sky = MrxSkyMap(None, 1, 1, n, n, 0.025, -step, -0.025, step, float(np.mean(az)), float(np.mean(el)), int(os.environ.get("MRX_BIN_BILINEAR", "0")), 0)
    args = (C.byref(sky), ptr(tod), tod.stride(0), None, 0, ptr(d_az), ptr(d_el), T, None, ptr(d_dx), ptr(d_dy), ptr(d_sw), None, D, ptr(msum), ptr(mwgt))
    if bucketed:
        lo, full = C.c_size_t(), C.c_size_t()
        ctx.lib.mrx_bin_map_work_bytes(C.byref(sky), D, T, C.byref(lo), C.byref(full))
        work = torch.empty(min(full.value, 40 << 30), dtype=torch.uint8, device=dev)
        fn = lambda: ctx.call("mrx_bin_map_bucketed", *args, ptr(work), work.numel())  # noqa: E731
    else:
        fn = lambda: ctx.call("mrx_bin_map", *args)  # noqa: E731
    med, mn = timeit(fn, reps)
    print(f"bin {n}x{n} bucketed={bucketed}: D={D} T={T}: median {med:.2f} ms min {mn:.2f} ms -> {D*T/med/1e6:.1f} G samples/s; hit pixels {int((mwgt > 0).sum())}")


if __name__ == "__main__":
    main()
