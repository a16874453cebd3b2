#!/usr/bin/env python3
"""Experiment: sample block b+1 as a small RESIDENT grid (k workgroups per CU walking the work
items) on a side stream while the HBM-bound writer of block b runs on the main stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maria_amd import synthetic, Context
from maria_amd.dist import shard_slice
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

p = synthetic.config_problem(sys.argv[1] if len(sys.argv) > 1 else "atlast_10k")
main = DevicePath(p, device="cuda:0")
main.generate_screens()
D, T = main.D, main.T
tod = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()

def wall(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for k in (0, 1, 2, 3, 4, 8):
    main.ctx.set_option(6, k)
    print(f"sample alone, {k} WG/CU resident: {timeit(main.sample, 10)[0]:.3f} ms", flush=True)
main.ctx.set_option(6, 0)
def mono():
    main.sample(); main.prepare(); main.upsample(tod)
print(f"monolithic: {wall(mono):.3f} ms", flush=True)

front, back = torch.cuda.Stream(), torch.cuda.Stream()
ctx_f, ctx_b = Context(0), Context(0)
ctx_f.set_stream(front); ctx_b.set_stream(back)
for B in (2, 4, 8):
    blocks = []
    for b in range(B):
        sl = shard_slice(D, B, b)
        bp = DevicePath(p, device="cuda:0", det_slice=sl, ctx=ctx_f)
        ctx_f.set_stream(front)
        bp.set_screens(main._gen_screens)
        blocks.append((sl, bp, torch.cuda.Event()))
    torch.cuda.synchronize()
    for k in (1, 2, 3, 4):
        for first_full in (True, False):
            def piped():
                cur = torch.cuda.current_stream()
                start = torch.cuda.Event(); start.record(cur)
                front.wait_event(start); back.wait_event(start)
                for i, (sl, bp, ev) in enumerate(blocks):
                    bp.ctx = ctx_f
                    ctx_f.set_option(6, 0 if (i == 0 and first_full) else k)
                    bp.sample(); bp.prepare()
                    ev.record(front)
                    back.wait_event(ev)
                    bp.ctx = ctx_b
                    bp.upsample(tod[sl])
                done = torch.cuda.Event(); done.record(back); cur.wait_event(done)
            print(f"B={B} resident {k}/CU first_full={first_full}: {wall(piped):.3f} ms", flush=True)
    del blocks
