#!/usr/bin/env python3
"""Timing of the TOD pre-processing kernels on one GPU (development aid).
Usage: python scripts/todproc_bench.py [n_det] [n_samples]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import tod_processing as tp  # noqa: E402
from maria_amd._lib import Context, ptr  # noqa: E402
from scripts.kbench import timeit  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
dev = torch.device("cuda:0")
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
x = torch.randn((D, T), dtype=torch.float32, device=dev)
anchors = torch.empty(2 * D + 16, dtype=torch.float64, device=dev)
w = torch.as_tensor(np.hanning(T)).to(dev)
med, _ = timeit(lambda: ctx.call("mrx_tod_detrend_window", ptr(x), x.stride(0), D, T, 1, ptr(w), ptr(anchors)), 5)
print(f"remove_slope + window in place: {med:.2f} ms -> {8.0*D*T/med/1e6:.0f} GB/s (read + write)")
for order, both in ((1, False), (1, True), (3, True)):
    secs = [tp.bessel_sos(0.1, 400.0, order, "high")] + ([tp.bessel_sos(50.0, 400.0, order, "low")] if both else [])
    sos = np.ascontiguousarray(np.concatenate(secs))
    M = torch.as_tensor(tp.chunk_matrix(sos, ctx.lib.mrx_sosfilt_chunk())).to(dev)
    need = C.c_size_t()
    ctx.lib.mrx_sosfilt_work_doubles(D, T, len(sos), C.byref(need))
    work = torch.empty(need.value, dtype=torch.float64, device=dev)
    fn = lambda: ctx.call("mrx_sosfilt", sos.ctypes.data_as(C.POINTER(C.c_double)), len(sos), ptr(M), ptr(x), x.stride(0), D, T, 1,  # noqa: E731
                          ptr(x), x.stride(0), ptr(work))
    med, _ = timeit(fn, 5)
    print(f"sosfilt {len(sos)} sections (order {order}{', low + high' if both else ', high'}): {med:.2f} ms -> {D*T/med/1e6:.1f} G samples/s, "
          f"{12.0*D*T/med/1e6:.0f} GB/s (2 reads + 1 write)")
