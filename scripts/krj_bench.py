#!/usr/bin/env python3
"""Fused K_RJ writer and pW writer timings (A/B aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maria_amd import synthetic
from maria_amd.pipeline import DevicePath
from scripts.kbench import timeit
p = synthetic.config_problem("atlast_10k")
path = DevicePath(p, device="cuda:0")
path.generate_screens(); path.sample(); path.prepare()
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
Tg = np.array([250.0, 270.0, 290.0]); pw = np.linspace(0.0, 10.0, 21)
elg = np.radians(np.linspace(10.0, 90.0, 33)); elg[-1] = np.radians(90.1)
tau = (0.03 + 0.01 * pw[None, :, None]) / np.sin(np.minimum(elg, np.pi / 2))[None, None, :]
tables = [{"T": Tg, "pwv": pw, "el": elg, "values": 20e9 * (Tg[:, None, None] / 270.0) ** 0.1 * np.exp(-tau)}]
_, el_full = synthetic.daisy_scan(p["t"])
path.set_calibration(tables, 273.15, 1.0, el_full, p["offsets"])
for g in (1, 2, 4):
    path.ctx.set_option(4, g)
    print("groups", g, "K_RJ fused %.3f ms" % timeit(lambda: path.upsample_krj(tod), 12)[0], "pW %.3f ms" % timeit(lambda: path.upsample(tod), 12)[0], flush=True)
path.ctx.set_option(4, 0)
for rep in range(1):
    print(os.environ.get("MRX_LIB_PATH", "default"), "pW  %.3f ms" % timeit(lambda: path.upsample(tod), 12)[0],
          "| K_RJ fused %.3f ms" % timeit(lambda: path.upsample_krj(tod), 12)[0],
          "| in place %.3f ms" % timeit(lambda: path.to_krj(tod), 12)[0], flush=True)
# the whole TOD synthesis in the default units (screens excluded): per-sample conversion in the writer
# against the conversion on the coarse grid + pW writer, serial and block-pipelined
def per_sample():
    path.sample(); path.prepare(); path.upsample_krj(tod)
print("coarse K_RJ bound %.3g (limit %.3g)" % (path.coarse_krj_bound(), path.COARSE_KRJ_LIMIT))
print("TOD synthesis in K_RJ: per-sample writer %.3f ms | coarse form, serial %.3f ms | coarse form, two streams %.3f ms | coarse form, one launch %.3f ms | pW two streams %.3f ms | pW one launch %.3f ms" % (
    timeit(per_sample, 10)[0], timeit(lambda: path.run(tod, blocks=1, krj=True), 10)[0],
    timeit(lambda: path.run(tod, blocks=path.default_blocks(), krj=True), 10)[0], timeit(lambda: path.run(tod, krj=True), 10)[0],
    timeit(lambda: path.run(tod, blocks=path.default_blocks()), 10)[0], timeit(lambda: path.run(tod), 10)[0]), flush=True)
a = path.run(torch.empty_like(tod), krj=True)
per_sample()
print("max relative deviation of the coarse form from the per-sample writer: %.3g" % float(((a - tod).abs() / tod.abs()).max()))
