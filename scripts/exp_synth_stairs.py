#!/usr/bin/env python3
"""Experiment: the head start of the one-launch synthesis as a staircase (MRX_SYNTH_STAIRS) against one step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
ref = path.run(blocks=1)
tod = torch.empty_like(ref)
for rep in range(2):
    for stairs in (0, 1):
        if stairs:
            os.environ["MRX_SYNTH_STAIRS"] = "1"
        else:
            os.environ.pop("MRX_SYNTH_STAIRS", None)
        for br, head, wgs in ((1024, 3, 2), (1024, 4, 2), (1024, 6, 2), (512, 6, 2), (512, 9, 2), (512, 12, 2), (1024, 4, 3), (512, 8, 1), (512, 12, 1)):
            tod.fill_(float("nan"))
            path.synthesize(tod, block_rows=br, resident_wgs_per_cu=wgs, head_rows=head * br)
            torch.cuda.synchronize()
            same = bool(torch.equal(tod, ref))
            med, mn = timeit(lambda: path.synthesize(tod, block_rows=br, resident_wgs_per_cu=wgs, head_rows=head * br), 8)
            print(f"stairs {stairs} block_rows {br} head {head} wgs {wgs}: identical {same} median {med:.3f} ms min {mn:.3f}", flush=True)
