"""Shared builders for the tests: problems with numpy screens attached."""

from __future__ import annotations

import numpy as np

from maria_amd import synthetic
from oracle import hotpath, screens


def attach_numpy_screens(problem, seed=0, smooth=True):
    rng = np.random.default_rng(seed)
    for layer in problem["layers"]:
        ne, nc = len(layer["extrusion"]), len(layer["cross_section"])
        de = layer["extrusion"][1] - layer["extrusion"][0]
        dc = layer["cross_section"][1] - layer["cross_section"][0]
        v = screens.numpy_screen(ne, nc, de, dc, layer["r0"], layer["nu"], rng)
        if smooth:
            v = hotpath.smooth_screen(v, layer["beam_sigma"] / de, layer["beam_sigma"] / dc)
        layer["values"] = np.asarray(v, np.float32)
    return problem


def small_problem(**kw):
    args = dict(n_det=67, n_bands=2, fov_deg=0.5, fs=50.0, duration=20.0, n_layers=3, side=128, t0=1.7e9)
    args.update(kw)
    return attach_numpy_screens(synthetic.make_problem(**args))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / np.abs(b).max()
