#!/usr/bin/env python3
"""Experiment: pipeline the VALU-bound sampling of detector block b+1 (front stream) against
the HBM-bound TOD writer of block b (back stream) inside ONE observation.  Detector rows are
independent through sample -> prepare -> upsample, so blocks need no halo.
Usage: python scripts/exp_blockpipe.py [config]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from maria_amd import synthetic, Context  # noqa: E402
from maria_amd.dist import shard_slice  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
p = synthetic.config_problem(cfg)
main = DevicePath(p, device="cuda:0")
main.generate_screens()
D, T = main.D, main.T
tod = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()


def wall(fn, n=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def mono():
    main.sample()
    main.prepare()
    main.upsample(tod)


print(f"{cfg}: monolithic sample+prepare+upsample {wall(mono):.3f} ms", flush=True)

for prio in ((0, 0), (-1, 0), (0, -1)):
    front = torch.cuda.Stream(priority=prio[0])
    back = torch.cuda.Stream(priority=prio[1])
    ctx_f, ctx_b = Context(0), Context(0)
    ctx_f.set_stream(front)
    ctx_b.set_stream(back)
    for B in (2, 4, 8, 16):
        blocks = []
        for b in range(B):
            sl = shard_slice(D, B, b)
            bp = DevicePath(p, device="cuda:0", det_slice=sl, ctx=ctx_f)
            ctx_f.set_stream(front)
            bp.set_screens(main._gen_screens)
            blocks.append((sl, bp, torch.cuda.Event()))
        torch.cuda.synchronize()

        def serial():
            for sl, bp, _ in blocks:
                bp.ctx = main.ctx
                bp.sample()
                bp.prepare()
                bp.upsample(tod[sl])

        def piped(prep_front):
            cur = torch.cuda.current_stream()
            start = torch.cuda.Event()
            start.record(cur)
            front.wait_event(start)
            back.wait_event(start)
            for sl, bp, ev in blocks:
                bp.ctx = ctx_f
                bp.sample()
                if prep_front:
                    bp.prepare()
                ev.record(front)
                back.wait_event(ev)
                bp.ctx = ctx_b
                if not prep_front:
                    bp.prepare()
                bp.upsample(tod[sl])
            done = torch.cuda.Event()
            done.record(back)
            cur.wait_event(done)

        print(
            f"prio {prio} B={B:2d}: serial blocks {wall(serial):.3f} ms | piped (prepare on back) {wall(lambda: piped(False)):.3f} ms"
            f" | piped (prepare on front) {wall(lambda: piped(True)):.3f} ms",
            flush=True,
        )
        del blocks
