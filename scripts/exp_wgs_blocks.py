#!/usr/bin/env python3
"""Experiment: TOD synthesis by resident sampler workgroups per CU x detector blocks (args: config, wgs list, blocks list)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

config = sys.argv[1]
wgs_list = [int(x) for x in sys.argv[2].split(",")]
blocks_list = [int(x) for x in sys.argv[3].split(",")]
n_det = synthetic.CONFIGS[config]["n_det"] // (8 if config == "atlast_50k" else 1)
p = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
tag = os.environ.get("MRX_LIB_PATH", "default")
for wgs in wgs_list:
    for b in blocks_list:
        med, mn = timeit(lambda: path._run_pipelined(tod, b, resident_wgs_per_cu=wgs), 6)
        print(f"{tag} {config} resident wgs/CU {wgs} blocks {b}: median {med:.3f} ms min {mn:.3f}", flush=True)
