#!/usr/bin/env python3
"""Experiment: the one-launch synthesis (mrx_atm_synthesize) against the pipelined two-stream run: bits and time.
Usage: python scripts/exp_synth.py <config> [block_rows...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
rows = [int(b) for b in sys.argv[2:]] or [256, 512, 1024]
n_det = synthetic.CONFIGS[config]["n_det"] // (8 if config == "atlast_50k" else 1)
if os.environ.get("SYNTH_DETS"):
    n_det = int(os.environ["SYNTH_DETS"])
p = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(p, device="cuda:0")
path.generate_screens()
ref = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
path.run(ref, blocks=1)
torch.cuda.synchronize()
tod = torch.empty_like(ref)
heads = [int(h) for h in os.environ.get("SYNTH_HEADS", "0").split(",")]
for br in rows:
  for head in heads:
    for wgs in [int(w) for w in os.environ.get('SYNTH_WGS', '3,4').split(',')]:
        tod.fill_(float("nan"))
        path.synthesize(tod, block_rows=br, resident_wgs_per_cu=wgs, head_rows=head * br)
        torch.cuda.synchronize()
        flags = int(path.d_flags.item())
        same = bool(torch.equal(tod, ref))
        nbad = 0 if same else int((tod != ref).sum().item())
        med, mn = timeit(lambda: path.synthesize(tod, block_rows=br, resident_wgs_per_cu=wgs, head_rows=head * br), 6)
        print(f"{config} synthesize block_rows {br} head {head} wgs {wgs}: identical {same} (differing {nbad}) flags {flags}  median {med:.3f} ms min {mn:.3f}", flush=True)
med, mn = timeit(lambda: path.run(tod, blocks=path.default_blocks()), 6)
print(f"{config} pipelined run (default blocks {path.default_blocks()}): median {med:.3f} ms min {mn:.3f}", flush=True)
med, mn = timeit(lambda: path.run(tod, blocks=1), 4)
print(f"{config} serial run: median {med:.3f} ms min {mn:.3f}", flush=True)
