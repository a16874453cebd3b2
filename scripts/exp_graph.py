#!/usr/bin/env python3
"""Development aid: the whole step (screens + the block-pipelined TOD synthesis on two streams)
replayed as one HIP graph against launching it kernel by kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402


def timeit(fn, reps, stream):
    for _ in range(3):
        fn()
    stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    stream.synchronize()
    return e0.elapsed_time(e1) / reps


for n_det in (10000, 1250):
    p = synthetic.config_problem("atlast_10k")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        path = DevicePath(p, device="cuda:0", det_slice=slice(0, n_det))
        path.ctx.set_stream(s)
        tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")

        def step():
            path.generate_screens()
            path.run(tod)

        step()
        step()
        s.synchronize()
        eager = timeit(step, 10, s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        s.synchronize()
        graph = timeit(g.replay, 10, s)
        print(f"D={n_det}: eager {eager:.3f} ms/step, graph {graph:.3f} ms/step", flush=True)
