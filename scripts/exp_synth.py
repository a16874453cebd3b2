#!/usr/bin/env python3
"""Experiment: the one-launch synthesis (mrx_atm_synthesize) against the pipelined two-stream run: bits and time.
Usage: python scripts/exp_synth.py <config> [block_rows...]
  SYNTH_WGS=0,1,2,3   dedicated sampler workgroups per CU (8: none)     SYNTH_CHUNK=16,32,64   steps per time chunk
  SYNTH_PER_CU=0      resident workgroups per CU (0: as many as fit)    SYNTH_DETS=n           detectors"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd import _lib, synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
rows = [int(b) for b in sys.argv[2:]] or [0]
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
ints = lambda name, default: [int(x) for x in os.environ.get(name, default).split(",")]
for order in ints("SYNTH_ORDER", "0"):
 path.ctx.set_option(_lib.OPT_SYNTH_TILE_ORDER, order)
 for per_cu in ints("SYNTH_PER_CU", "0"):
  path.ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, per_cu)
  for br in rows:
    for chunk in ints("SYNTH_CHUNK", "32"):
      for wgs in ints("SYNTH_WGS", "2"):
        tod.fill_(float("nan"))
        kw = dict(block_rows=br, sampler_wgs_per_cu=wgs, chunk=chunk) if wgs < 16 else dict(block_rows=br, sampler_wgs=wgs, chunk=chunk)
        path.synthesize(tod, **kw)
        torch.cuda.synchronize()
        flags = int(path.d_flags.item())
        same = bool(torch.equal(tod, ref))
        nbad = 0 if same else int((tod != ref).sum().item())
        med, mn = timeit(lambda: path.synthesize(tod, **kw), 8)
        print(f"{config} D {path.D} synthesize order {order} per_cu {per_cu} block_rows {br} chunk {chunk} samplers {wgs}: identical {same} (differing {nbad}) flags {flags}  median {med:.3f} ms min {mn:.3f}", flush=True)
path.ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, 0)
path.ctx.set_option(_lib.OPT_SYNTH_TILE_ORDER, 0)
path.sample()
med, mn = timeit(lambda: path.upsample_fused(tod), 6)
print(f"{config} D {path.D} writer alone (mrx_spline_upsample_fused): median {med:.3f} ms min {mn:.3f}", flush=True)
if not os.environ.get("SYNTH_ONLY"):
    med, mn = timeit(lambda: path._run_pipelined(tod, path.default_blocks()) if path.default_blocks() > 1 else path.run(tod, blocks=1), 6)
    print(f"{config} pipelined run (default blocks {path.default_blocks()}): median {med:.3f} ms min {mn:.3f}", flush=True)
    med, mn = timeit(lambda: path.run(tod, blocks=1), 4)
    print(f"{config} serial run: median {med:.3f} ms min {mn:.3f}", flush=True)
