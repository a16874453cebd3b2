#!/usr/bin/env python3
"""Timing of the noise and calibration rows on one GPU (development aid).
Usage: python scripts/noise_bench.py [n_det] [n_samples] [reps]"""

import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import noise as mnoise  # noqa: E402
from maria_amd import synthetic  # noqa: E402
from maria_amd._lib import Context, ptr  # noqa: E402
from scripts.kbench import timeit  # noqa: E402


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 240000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dev = torch.device("cuda:0")
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    ctx.set_option(5, int(os.environ.get("MRX_NOISE_GENERIC", "0")))
    ctx.set_option(8, int(os.environ.get("MRX_NOISE_LANES", "0")))
    off = synthetic.hex_pack(D, np.radians(1.0))
    B = torch.as_tensor(np.ascontiguousarray(mnoise.spatial_basis(off, 5, 16, mnoise.diameter(off)), np.float32)).to(dev)
    scale = torch.full((D,), 1e-5, dtype=torch.float32, device=dev)
    out = torch.empty((D, T), dtype=torch.float32, device=dev)
    for batch in [int(b) for b in os.environ.get("MRX_NOISE_BATCH", "1024").split(",")]:
        need = C.c_size_t()
        ctx.lib.mrx_noise_work_floats(T, 5, batch, C.byref(need))
        work = torch.empty(need.value, dtype=torch.float32, device=dev)
        for knee, modes in ((1.0, 5), (1.0, 0), (0.0, 0)):
            fn = lambda: ctx.call("mrx_noise_generate", 1, D, 0, T, 400.0, knee, 0.5, ptr(B) if modes else None, modes, ptr(scale),  # noqa: E731
                                  None, 0, 0.0, ptr(out), out.stride(0), 0, ptr(work), need.value)
            med, mn = timeit(fn, reps)
            print(f"noise D={D} T={T} batch={batch} knee={knee} modes={modes}: median {med:.3f} ms min {mn:.3f} ms "
                  f"-> {D*T/med/1e6:.1f} G samples/s, {4.0*D*T/med/1e6:.0f} GB/s of TOD; work {need.value*4/2**30:.2f} GiB")
        del work


if __name__ == "__main__":
    main()
