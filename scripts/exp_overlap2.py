#!/usr/bin/env python3
"""Two-stage software pipeline: a high-priority stream runs screens+sample+prepare of
observation k+1 while a low-priority stream streams out the TOD of observation k."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maria_amd import synthetic, Context
from maria_amd.pipeline import DevicePath

p = synthetic.config_problem("atlast_10k")
lo_pri, hi_pri = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
front = torch.cuda.Stream(priority=-1)
back = torch.cuda.Stream(priority=0)
ctx_f, ctx_b = Context(0), Context(0)
ctx_f.set_stream(front); ctx_b.set_stream(back)
paths = []
for k in range(2):
    with torch.cuda.stream(front):
        path = DevicePath(p, device="cuda:0", ctx=ctx_f)
        ctx_f.set_stream(front)
        path.generate_screens()
        paths.append(path)
tod = torch.empty((paths[0].D, paths[0].T), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
ready = [torch.cuda.Event(), torch.cuda.Event()]
drained = [torch.cuda.Event(), torch.cuda.Event()]

def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = i % 2
        path = paths[k]
        with torch.cuda.stream(front):
            if i >= 2: front.wait_event(drained[k])      # its knots were consumed
            path.ctx = ctx_f
            path.generate_screens(); path.sample(); path.prepare()
            ready[k].record(front)
        with torch.cuda.stream(back):
            back.wait_event(ready[k])
            path.ctx = ctx_b
            path.upsample(tod)
            drained[k].record(back)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

run(6)
for _ in range(3):
    print("pipelined front(high prio)/back: %.3f ms/step" % run(20))
