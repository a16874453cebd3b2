#!/usr/bin/env python3
"""Sampler with LDS-tiled screens (option 7 = 0) or global gathers (option 7 = 1) beside the TOD
writer: detector blocks pipelined on two streams, the sampler as a resident grid."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maria_amd import synthetic, Context
from maria_amd.dist import shard_slice
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
main = DevicePath(p, device="cuda:0")
main.generate_screens()
D, T = main.D, main.T
tod = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
def wall(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
front, back = torch.cuda.Stream(), torch.cuda.Stream()
ctx_f, ctx_b = Context(0), Context(0)
ctx_f.set_stream(front); ctx_b.set_stream(back)
for xp in (0, 1):  # 1: LDS-tiled, 0: global gathers
    main.ctx.set_option(7, xp); ctx_f.set_option(7, xp)
    for k in (0, 4):
        main.ctx.set_option(6, k)
        print(f"tiles={xp} sample alone {k or 8} WG/CU: {timeit(main.sample, 10)[0]:.3f} ms", flush=True)
    main.ctx.set_option(6, 0)
    def mono():
        main.sample(); main.prepare(); main.upsample(tod)
    print(f"tiles={xp} monolithic: {wall(mono):.3f} ms", flush=True)
    B = 4
    blocks = []
    for b in range(B):
        sl = shard_slice(D, B, b)
        bp = DevicePath(p, device="cuda:0", det_slice=sl, ctx=ctx_f)
        ctx_f.set_stream(front)
        bp.set_screens(main._gen_screens)
        blocks.append((sl, bp, torch.cuda.Event()))
    torch.cuda.synchronize()
    for k in (3, 4, 5, 6):
        def piped():
            cur = torch.cuda.current_stream()
            start = torch.cuda.Event(); start.record(cur)
            front.wait_event(start); back.wait_event(start)
            for i, (sl, bp, ev) in enumerate(blocks):
                bp.ctx = ctx_f
                ctx_f.set_option(6, 0 if i == 0 else k)
                bp.sample(); bp.prepare()
                ev.record(front)
                back.wait_event(ev)
                bp.ctx = ctx_b
                bp.upsample(tod[sl])
            done = torch.cuda.Event(); done.record(back); cur.wait_event(done)
        print(f"tiles={xp} B={B} resident {k}/CU piped: {wall(piped):.3f} ms", flush=True)
    del blocks
