#!/usr/bin/env python3
"""Experiment: row batches per writer tile (MRX_OPT_UPSAMPLE_GROUPS) in the one-launch synthesis: the sample weights of a
time tile are shared by the batches, and the launch is VALU-busy (65 %: profiles/r04_kernel_pmc.txt)."""
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
    for groups in (1, 2, 4):
        path.ctx.set_option(4, groups)
        tod.fill_(float("nan"))
        path.synthesize(tod)
        torch.cuda.synchronize()
        same = bool(torch.equal(tod, ref))
        med, mn = timeit(lambda: path.synthesize(tod), 8)
        m2, n2 = timeit(lambda: path.upsample_fused(tod), 4)
        print(f"batches {groups}: identical {same} one launch median {med:.3f} ms min {mn:.3f}; writer alone {m2:.3f}", flush=True)
path.ctx.set_option(4, 0)
