#!/usr/bin/env python3
"""Experiment: detector noise for two halves of the rows on two streams (pass 1 of one half
beside pass 2 of the other).  Usage: python scripts/exp_noise_streams.py [n_streams...]"""
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


def main():
    D, T = 10000, 240000
    dev = torch.device("cuda:0")
    off = synthetic.hex_pack(D, np.radians(1.0))
    B = torch.as_tensor(np.ascontiguousarray(mnoise.spatial_basis(off, 5, 16, mnoise.diameter(off)), np.float32)).to(dev)
    scale = torch.full((D,), 1e-5, dtype=torch.float32, device=dev)
    out = torch.empty((D, T), dtype=torch.float32, device=dev)
    for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
        for batch in (128, 256):
            streams = [torch.cuda.Stream(dev) for _ in range(ns)]
            ctxs = [Context(0) for _ in range(ns)]
            works = []
            need = C.c_size_t()
            for c, s in zip(ctxs, streams):
                c.set_stream(s)
                c.lib.mrx_noise_work_floats(T, 5, batch, C.byref(need))
                works.append(torch.empty(need.value, dtype=torch.float32, device=dev))
            cuts = [(D * i // ns) // 2 * 2 for i in range(ns)] + [D]

            def run():
                for i, (c, w) in enumerate(zip(ctxs, works)):
                    lo, hi = cuts[i], cuts[i + 1]
                    c.call("mrx_noise_generate", 1, hi - lo, lo, T, 400.0, 1.0, 0.5, B[lo:].data_ptr(), 5, scale[lo:].data_ptr(),
                           None, 0, 0.0, out[lo:].data_ptr(), out.stride(0), 0, ptr(w), need.value)

            ts = []
            for rep in range(6):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream())
                for s in streams:
                    s.wait_event(e0)
                run()
                for s in streams:
                    torch.cuda.current_stream().wait_stream(s)
                e1.record(torch.cuda.current_stream())
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print(f"streams={ns} batch={batch}: median {np.median(ts[1:]):.3f} ms min {min(ts[1:]):.3f} ms", flush=True)
            del works


if __name__ == "__main__":
    main()
