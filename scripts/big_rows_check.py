#!/usr/bin/env python3
"""One-off robustness run of the follow-on rows at the size of a config-5 detector shard
(6250 detectors x 1.44 M samples = 9e9 samples, 36 GB per field): indices beyond 2^32."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import map as mmap  # noqa: E402
from maria_amd import noise as mnoise  # noqa: E402
from maria_amd import synthetic  # noqa: E402
from maria_amd import tod_processing as tp  # noqa: E402
from maria_amd._lib import Context, MrxSkyMap, ptr  # noqa: E402

D, T = 6250, 1440000
dev = torch.device("cuda:0")
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream(dev))
out = torch.empty((D, T), dtype=torch.float32, device=dev)
off = synthetic.hex_pack(D, np.radians(1.0))
B = torch.as_tensor(np.ascontiguousarray(mnoise.spatial_basis(off, 5, 16, mnoise.diameter(off)), np.float32)).to(dev)
need = C.c_size_t()
ctx.lib.mrx_noise_work_floats(T, 5, 256, C.byref(need))
work = torch.empty(need.value, dtype=torch.float32, device=dev)
ctx.call("mrx_noise_generate", 7, D, 0, T, 400.0, 1.0, 0.5, ptr(B), 5, None, None, 0, 0.0, ptr(out), out.stride(0), 0, ptr(work), need.value)
torch.cuda.synchronize()
tail = out[-3:, -100000:].double()
print("noise: finite", bool(torch.isfinite(out[::97, ::1013]).all()), "tail std", float(tail.std()), "expected ~", np.sqrt(400 * (1 + 2 / 400 * np.log(T / 2))))
del work
t = 1.7e9 + np.arange(T) / 400.0
az, el = synthetic.daisy_scan(t)
n = 512
xi = np.linspace(-0.03, 0.03, n)
X, Y = np.meshgrid(xi, xi[::-1])
vals = np.exp(-(X**2 + Y**2) / 1e-4).astype(np.float32)[None, None]
mmap.sample_map(ctx, vals, xi[::-1].copy(), xi, (float(np.mean(az)), float(np.mean(el))), az, el, off, np.ones((D, 1)), out=out, cal_scalars=[2e10])
torch.cuda.synchronize()
print("map: finite", bool(torch.isfinite(out[::97, ::1013]).all()), "last row max", float(out[-1].max()), "first row max", float(out[0].max()))
ms = torch.zeros((1, 1, n, n), dtype=torch.float64, device=dev)
mw = torch.zeros_like(ms)
f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
d_az, d_el, d_dx, d_dy = f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
d_sw = torch.ones((D, 1), dtype=torch.float64, device=dev)
sky = MrxSkyMap(None, 1, 1, n, n, 0.03, -0.06 / (n - 1), -0.03, 0.06 / (n - 1), float(np.mean(az)), float(np.mean(el)), 0, 0)
ctx.call("mrx_bin_map", C.byref(sky), ptr(out), out.stride(0), None, 0, ptr(d_az), ptr(d_el), T, None, ptr(d_dx), ptr(d_dy), ptr(d_sw), None, D,
         ptr(ms), ptr(mw))
torch.cuda.synchronize()
print("bin: weight total", float(mw.sum()), "expected", float(D) * T, "peak of binned map", float((ms / mw).nan_to_num().max()))
sos = np.ascontiguousarray(tp.bessel_sos(0.1, 400.0, 1, "high"))
M = torch.as_tensor(tp.chunk_matrix(sos, ctx.lib.mrx_sosfilt_chunk())).to(dev)
ctx.lib.mrx_sosfilt_work_doubles(D, T, len(sos), C.byref(need))
wk = torch.empty(need.value, dtype=torch.float64, device=dev)
ctx.call("mrx_sosfilt", sos.ctypes.data_as(C.POINTER(C.c_double)), len(sos), ptr(M), ptr(out), out.stride(0), D, T, 1, ptr(out), out.stride(0), ptr(wk))
torch.cuda.synchronize()
import scipy.signal  # noqa: E402

print("sosfilt: finite", bool(torch.isfinite(out[::97, ::1013]).all()), "last row abs mean", float(out[-1].abs().mean()))
