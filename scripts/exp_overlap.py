#!/usr/bin/env python3
"""Does pipelining independent observations on two HIP streams overlap the VALU-bound
stages (screens, sampling) with the HBM-bound upsample?  Development experiment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maria_amd import synthetic, Context
from maria_amd.pipeline import DevicePath

p = synthetic.config_problem("atlast_10k")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
paths, tods = [], []
for s in streams:
    with torch.cuda.stream(s):
        ctx = Context(0); ctx.set_stream(s)
        path = DevicePath(p, device="cuda:0", ctx=ctx)
        path.ctx.set_stream(s)
        path.generate_screens()
        paths.append(path); tods.append(torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0"))
torch.cuda.synchronize()

def step(i):
    k = i % 2
    with torch.cuda.stream(streams[k]):
        paths[k].ctx.set_stream(streams[k])
        paths[k].generate_screens(); paths[k].sample(); paths[k].prepare(); paths[k].upsample(tods[k])

def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        step(i if two else 0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): step(0); step(1)
print("one stream : %.3f ms/step" % run(20, False))
print("two streams: %.3f ms/step" % run(20, True))
print("one stream : %.3f ms/step" % run(20, False))
print("two streams: %.3f ms/step" % run(20, True))
