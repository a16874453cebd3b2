#!/usr/bin/env python3
"""Development aid: does replaying the screen stage (40 small kernels) as a HIP graph beat
launching it kernel by kernel?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402
from scripts.kbench import timeit  # noqa: E402

p = synthetic.config_problem("atlast_10k")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    path = DevicePath(p, device="cuda:0")
    path.ctx.set_stream(s)
    path.generate_screens()
    path.generate_screens()
    s.synchronize()
    print("eager screens:", timeit(path.generate_screens, 10))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        path.generate_screens()
    s.synchronize()
    print("graph screens:", timeit(g.replay, 10))
    tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")

    def step():
        path.generate_screens()
        path.sample()
        path.prepare()
        path.upsample(tod)

    step()
    print("eager step:", timeit(step, 10))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        step()
    s.synchronize()
    print("graph step:", timeit(g2.replay, 10))
