#!/usr/bin/env python3
"""Development aid: the two forms of the K_RJ conversion against the numpy oracle on random rows of atlast_10k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from oracle import hotpath
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens()
Tg = np.array([250.0, 270.0, 290.0]); pw = np.linspace(0.0, 10.0, 21)
elg = np.radians(np.linspace(10.0, 90.0, 33)); elg[-1] = np.radians(90.1)
tau = (0.03 + 0.01 * pw[None, :, None]) / np.sin(np.minimum(elg, np.pi / 2))[None, None, :]
tables = [{"T": Tg, "pwv": pw, "el": elg, "values": 20e9 * (Tg[:, None, None] / 270.0) ** 0.1 * np.exp(-tau)}]
az_full, el_full = synthetic.daisy_scan(p["t"])
path.set_calibration(tables, 273.15, 1.0, el_full, p["offsets"])
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
coarse = path.run(torch.empty_like(tod), krj=True)
path.sample(); path.prepare(); path.upsample_krj(tod)
rows = np.sort(np.random.default_rng(1).choice(path.D, 16, replace=False))
sub = dict(p)
for key in ("offsets", "band_index", "m00"):
    sub[key] = p[key][rows]
sub["layers"] = [dict(l, values=b[0].cpu().numpy()) for l, b in zip(p["layers"], path._layer_bufs)]
ref_pw = hotpath.run_path(sub)
_, el_det = hotpath.broadcast(sub["offsets"], az_full, el_full)
ref = hotpath.calibrate_to_krj(ref_pw, sub["band_index"], tables, 273.15, 1.0, el_det, [False])
for name, got in (("per-sample writer", tod), ("coarse form", coarse)):
    g = got[rows].cpu().numpy().astype(np.float64)
    print(f"{name}: max |got - oracle| / max |oracle| = {np.abs(g - ref).max() / np.abs(ref).max():.3g}; max relative per sample = {np.abs(g / ref - 1).max():.3g}")
print("bound", path.coarse_krj_bound())
