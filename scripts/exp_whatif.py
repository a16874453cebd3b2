#!/usr/bin/env python3
"""Experiment: sampler alone (full grid / resident 3 per CU) and the TOD synthesis serial / pipelined, for the library
MRX_LIB_PATH names.  Usage: python scripts/exp_whatif.py <config> [blocks...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
blocks = [int(b) for b in sys.argv[2:]] or [1, 4]
n_det = synthetic.CONFIGS[config]["n_det"] // (8 if config == "atlast_50k" else 1)
p = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
tag = os.environ.get("MRX_LIB_PATH", "default")
for wgs in (0, 3):
    path.ctx.set_option(6, wgs)
    med, mn = timeit(path.sample, 6)
    print(f"{tag} {config} sampler alone wgs/CU {wgs or 'full'}: median {med:.3f} ms min {mn:.3f}", flush=True)
path.ctx.set_option(6, 0)
med, mn = timeit(lambda: path.upsample_fused(tod), 4)
print(f"{tag} {config} writer alone: median {med:.3f} ms min {mn:.3f}", flush=True)
med, mn = timeit(path.generate_screens, 4)
print(f"{tag} {config} screens: median {med:.3f} ms min {mn:.3f}", flush=True)
for b in blocks:
    med, mn = timeit(lambda: path.run(tod, blocks=b), 6)
    print(f"{tag} {config} TOD synthesis blocks {b}: median {med:.3f} ms min {mn:.3f}", flush=True)
