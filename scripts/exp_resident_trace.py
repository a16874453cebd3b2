#!/usr/bin/env python3
"""One configuration of exp_resident.py for a rocprofv3 --kernel-trace timeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from maria_amd import synthetic, Context
from maria_amd.dist import shard_slice
from maria_amd.pipeline import DevicePath
B, k = int(sys.argv[1]), int(sys.argv[2])
p = synthetic.config_problem("atlast_10k")
main = DevicePath(p, device="cuda:0")
main.generate_screens()
D, T = main.D, main.T
tod = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
front, back = torch.cuda.Stream(), torch.cuda.Stream()
ctx_f, ctx_b = Context(0), Context(0)
ctx_f.set_stream(front); ctx_b.set_stream(back)
blocks = []
for b in range(B):
    sl = shard_slice(D, B, b)
    bp = DevicePath(p, device="cuda:0", det_slice=sl, ctx=ctx_f)
    ctx_f.set_stream(front)
    bp.set_screens(main._gen_screens)
    blocks.append((sl, bp, torch.cuda.Event()))
torch.cuda.synchronize()
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
for _ in range(4):
    piped(); torch.cuda.synchronize()
