#!/usr/bin/env python3
"""Development aid: randomised shapes through mrx_atm_synthesize (one launch) against the two calls, every word.
Detectors, layers, bands, sample rate, duration and time step (i.e. Ta, T and their ratio), gain, block size, head start
and resident sampler workgroups per CU are drawn at random; a few trials run the launch several times in a row.
Usage: python scripts/fuzz_synth.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from helpers import attach_numpy_screens
from test_gpu_calibration import _cal_tables
from maria_amd import synthetic
from maria_amd._lib import Context
from maria_amd.pipeline import DevicePath

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
ctx = Context(0)
bad = 0
for trial in range(trials):
    n_det = int(np.exp(rng.uniform(0, np.log(3000))))
    n_layers = int(rng.integers(1, 17))
    n_bands = int(rng.integers(1, 4))
    fs = float(rng.choice([20.0, 50.0, 100.0, 400.0]))
    timestep = float(rng.choice([0.1, 0.2, 0.5]))
    duration = float(rng.uniform(4 * timestep + 0.3, 60.0 if fs < 200 else 25.0))
    p = attach_numpy_screens(synthetic.make_problem(n_det=n_det, n_bands=min(n_bands, n_det), fov_deg=float(rng.uniform(0.05, 1.0)), fs=fs,
                                                    duration=duration, n_layers=n_layers, side=int(rng.choice([64, 128, 256])),
                                                    timestep=timestep, seed=int(rng.integers(1 << 30)), gain=bool(rng.random() < 0.5)))
    path = DevicePath(p, device="cuda:0", ctx=ctx)
    path.clear_flags()
    krj = False
    if rng.random() < 0.4:  # K_RJ on the coarse grid in the same launch (mrx_atm_synthesize_krj), where its bound allows
        _, el_full = synthetic.daisy_scan(p["t"])
        roll = rng.uniform(0, 2 * np.pi)
        R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
        nb = len(p["tables"])
        path.set_calibration(_cal_tables(nb), 273.15, 1.0, el_full, p["offsets"] @ R.T, [bool(rng.random() < 0.5) for _ in range(nb)])
        krj = bool(path.coarse_krj_bound() <= path.COARSE_KRJ_LIMIT)
    want = path.run(blocks=1, krj=krj)
    coarse = None if krj else path.coarse_loading().clone()
    torch.cuda.synchronize()
    f0 = path.check_flags() if False else int(path.d_flags.item())
    block_rows = int(rng.choice([0, 256, 512, 768, 1024, 4096]))
    head_rows = int(rng.choice([0, 1, 256, 600, 100000]))
    wgs = int(rng.integers(1, 8))
    reps = 3 if rng.random() < 0.2 else 1
    ok = True
    for _ in range(reps):
        got = torch.full_like(want, float("nan"))
        path.synthesize(got, block_rows=block_rows, head_rows=head_rows, resident_wgs_per_cu=wgs, krj=krj)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(got, want)) and (krj or bool(torch.equal(path.coarse_loading(), coarse))) and int(path.d_flags.item()) == f0
    tag = "ok " if ok else "BAD"
    bad += not ok
    print(f"{tag} trial {trial}: D {path.D} Ta {path.Ta} T {path.T} layers {n_layers} bands {n_bands} block_rows {block_rows} head {head_rows} wgs {wgs} reps {reps} flags {f0} krj {krj}", flush=True)
print(f"fuzz_synth seed {seed}: {bad} bad of {trials}")
sys.exit(1 if bad else 0)
