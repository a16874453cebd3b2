"""Set-up time of mrx_screen_amplitudes (covariance-matched screen amplitudes) for a few domains."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from maria_amd import _lib  # noqa: E402
from maria_amd._lib import Context, ptr  # noqa: E402
from maria_amd.pipeline import matern_log_tables  # noqa: E402

ctx = Context(0)
for nh, ny, nx, dh, dy, dx, r0, nu in [
    (0, 2048, 2048, 0.0, 5.0, 5.0, 1000.0, 5 / 6),
    (0, 2048, 2048, 0.0, 5.0, 5.0, 1300.0, 5 / 6),
    (0, 8192, 256, 0.0, 5.0, 5.0, 1000.0, 5 / 6),
    (0, 8192, 8192, 0.0, 5.0, 5.0, 1000.0, 5 / 6),
    (16, 4096, 128, 300.0, 10.0, 30.0, 1100.0, 1 / 3),
    (64, 4096, 512, 100.0, 5.0, 10.0, 1100.0, 1 / 3),
]:
    lf, ls, lc, lsf, xc = matern_log_tables(nu)
    n_t, n_w = C.c_size_t(), C.c_size_t()
    _lib.load().mrx_screen_amp_floats(nh, ny, nx, len(lc), C.byref(n_t), C.byref(n_w))
    table = torch.empty(n_t.value, dtype=torch.float32, device="cuda:0")
    work = torch.empty(n_w.value, dtype=torch.float32, device="cuda:0")
    as_d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.call("mrx_screen_amplitudes", nh, ny, nx, dh, dy, dx, r0, as_d(lc), as_d(lsf), len(lc), lf, ls, xc, ptr(table), ptr(work), work.numel())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    zero = float((table[4:] == 0).float().mean())
    print(f"{nh:3d} x {ny} x {nx}  r0 {r0:.0f} nu {nu:.2f}: {dt * 1e3:8.1f} ms, table {n_t.value * 4 / 1e6:.1f} MB, clipped {zero:.2e}", flush=True)
