#!/usr/bin/env python3
"""Experiment: the next step's screens generated on a side stream while the TOD of the current
step is synthesised (two DevicePath objects: one only generates, one only runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for n_det in [int(a) for a in sys.argv[1:]] or [10000, 1250]:
        p = synthetic.config_problem("atlast_10k", n_det=10000)
        run = DevicePath(p, device=dev, det_slice=slice(0, n_det))
        gen = DevicePath(p, device=dev, det_slice=slice(0, 256))
        run.generate_screens()
        gen.generate_screens()
        tod = torch.empty((run.D, run.T), dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(dev)
        main_s = torch.cuda.current_stream(dev)

        def serial():
            run.generate_screens()
            run.run(tod)

        def overlapped():
            side.wait_stream(main_s)  # everything so far
            gen.ctx.set_stream(side)
            gen.generate_screens()
            run.run(tod)
            main_s.wait_stream(side)

        for name, fn in (("serial", serial), ("prefetch", overlapped), ("tod only", lambda: run.run(tod)), ("screens only", run.generate_screens)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print(f"D={n_det} {name}: {e0.elapsed_time(e1) / 10:.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
