"""Host-side set-up of the reference's atmosphere model.  TEST INFRASTRUCTURE ONLY.

Restates, function by function, the geometry that precedes the hot loops
(SURVEY 8(a) rows a1-a3 and a6's linear algebra):

  * ``get_orthogonal_transform`` / ``compute_aligning_transform``
      maria/utils/rotations.py:25-77                       PINNED (golden)
  * ``fast_psd_inverse``      maria/utils/linalg.py:95-102  PINNED (golden)
  * ``generate_layers``       maria/atmosphere/extrusion.py:27-110
  * ``process_geometry``      maria/atmosphere/atmosphere.py:117-257 (one process)

The first two are checked against vectors produced by importing the reference's
own modules (oracle/gen_golden.py -> tests/golden/leaves.json).  The last two
cannot be imported (pandas DataFrame of a Weather object, jax coordinates) and
are restated with plain arrays for the weather profile: parity unpinned.
"""

from __future__ import annotations

import numpy as np
import scipy as sp
import scipy.interpolate
import scipy.linalg
import scipy.optimize
import scipy.spatial

from . import functions


def get_orthogonal_transform(signature, entries):
    """utils/rotations.py:25-42."""
    signature = np.asarray(signature, bool)
    axes = np.where(signature)[0]
    n_dim = len(signature)
    n_axes = int(signature.sum())
    if n_axes * (n_axes - 1) / 2 != len(entries):
        raise ValueError("Bad shape for entries")
    i, j = np.triu_indices(n=n_axes, k=1)
    S = np.zeros((n_dim, n_dim))
    S[axes[i], axes[j]] = entries
    return sp.linalg.expm(S - S.T)


def compute_aligning_transform(points, signature, axes=None, n_init: int = 16):
    """utils/rotations.py:45-77: rotation minimising the log hull volume of all but
    the first axis; 16 random starts from numpy's GLOBAL random state, then SLSQP."""
    *_, n_dim = points.shape
    args = points.reshape(-1, n_dim)

    def loss(entries, *a):
        tp = a[0] @ get_orthogonal_transform(signature=signature, entries=entries)
        if n_dim > 2:
            return np.log(sp.spatial.ConvexHull(tp[..., 1:]).volume)
        return np.log(np.ptp(tp[..., 1:]))

    n_axes = sum(signature)
    n_dof = int(n_axes * (n_axes - 1) / 2)
    x0_samples = np.random.standard_normal(size=(n_init, n_dof))
    best_index = np.argmin([loss(x0, args) for x0 in x0_samples])
    res = sp.optimize.minimize(loss, x0=x0_samples[best_index], args=args, tol=1e-6, method="SLSQP")
    if not res.success:
        raise RuntimeError("Could not find optimal rotation.")
    return get_orthogonal_transform(signature=signature, entries=res.x)


def fast_psd_inverse(M):
    """utils/linalg.py:95-102."""
    cholesky, _ = sp.linalg.lapack.dpotrf(M)
    invM, _ = sp.linalg.lapack.dpotri(cholesky)
    return np.where(invM, invM, invM.T)


H_BOUNDARIES_2D = np.array([0.0, 500.0, 1000.0, 1500.0, 2000.0, 3000.0, 5000.0, 8000.0, 12000.0])


MIN_RES = {"2d": 2, "3d": 15}  # extrusion.py:20-22
MIN_RES_PER_BEAM = {"2d": 0.1, "3d": 0.5}
MIN_RES_PER_FOV = {"2d": 0.02, "3d": 0.1}


def generate_layers(
    field_of_view,
    band_fwhm_args,
    min_el,
    weather,
    site_altitude,
    pwv,
    pwv_rms_frac=3e-2,
    min_res=None,
    min_res_per_beam=None,
    min_res_per_fov=None,
    mode="2d",
    max_height=2e3,
):
    """extrusion.py:27-110 in "2d" mode with ``angular=False``.

    ``band_fwhm_args``: list of (primary_size, band_center_Hz), one per band
    (``one_detector_from_each_band().physical_fwhm``); ``weather``: dict of profile
    arrays ``altitude, absolute_humidity, temperature, wind_east, wind_north,
    divergence`` (what ``Weather.__call__`` interpolates, weather/__init__.py:222).
    Returns a dict of per-layer arrays.
    """
    min_res = min_res or MIN_RES[mode]
    min_res_per_beam = min_res_per_beam or MIN_RES_PER_BEAM[mode]
    min_res_per_fov = min_res_per_fov or MIN_RES_PER_FOV[mode]
    h_samples = np.arange(0.0, 20000.0, 1e0)
    z_samples = h_samples / np.sin(min_el)
    fwhm = np.min(
        [functions.compute_physical_fwhm(ps, z=z_samples + 1e-16, nu=nu) for ps, nu in band_fwhm_args], axis=0
    )
    r1 = min_res * np.ones(len(z_samples))
    r2 = min_res_per_beam * fwhm
    r3 = min_res_per_fov * z_samples * field_of_view
    res_samples = np.minimum(1e3, np.maximum.reduce([r1, r2, r3]))
    res_func = sp.interpolate.interp1d(h_samples, res_samples)

    if mode == "2d":
        h_boundaries = H_BOUNDARIES_2D
        process_index = np.arange(len(h_boundaries) - 1)
    else:  # extrusion.py:69-77
        h_boundaries = [0]
        while True:
            new_h = h_boundaries[-1] + res_func(h_boundaries[-1])
            if new_h > max_height:
                break
            h_boundaries.append(h_boundaries[-1] + res_func(h_boundaries[-1]))
        h_boundaries = np.array(h_boundaries, float)
        process_index = np.zeros(len(h_boundaries) - 1, int)
    h = (h_boundaries[1:] + h_boundaries[:-1]) / 2
    layers = {
        "process_index": process_index,
        "h": h,
        "dh": np.diff(h_boundaries),
        "res": res_func(h),
        "z": h / np.sin(min_el),
    }
    alt = np.asarray(weather["altitude"], float)
    for key in ("absolute_humidity", "temperature", "wind_east", "wind_north", "divergence"):
        layers[key] = np.interp(site_altitude + h, alt, np.asarray(weather[key], float))
    rel_var = (np.exp(-h / 1e3) * h ** (1 / 7)) ** 2  # boundary_layer_profile, :96-98
    layers["pwv_rms"] = np.sqrt((pwv * pwv_rms_frac) ** 2 * rel_var / rel_var.sum())
    return layers


def process_geometry(layer, all_res_min, outer_pp, timestep, n_t):
    """One "2d" process = one layer (atmosphere.py:117-257).

    ``outer_pp``: unit-height projection [n_outer, Ta, 3] of the hull detectors
    (``outer_coords.project(z=1)``, so ``project(z=h) = h * outer_pp``).  Draws from
    numpy's global state in the reference's order: hull jitter (:184), then the 16
    starts of the aligning transform (rotations.py:63).
    """
    w = layer["absolute_humidity"] * layer["temperature"] * layer["divergence"]
    vx = (w * layer["wind_east"] * np.ones(n_t)) / w  # :141-151 with a single layer
    vy = (w * layer["wind_north"] * np.ones(n_t)) / w
    vz = np.zeros(n_t)
    p = layer["h"] * outer_pp + np.cumsum(timestep * np.c_[vx, vy, vz][None], axis=-2)
    pts = p.reshape(-1, 3).copy()
    pts[..., 2] += 1e-6 * np.random.standard_normal(pts[..., 2].shape)
    transform = compute_aligning_transform(pts, signature=(True, True, False))
    tp = pts @ transform
    res = layer["res"]
    n_cross = int(np.maximum(2, (np.ptp(tp[:, 1]) + 2 * res) / res))
    cross_section = np.linspace(tp[:, 1].min() - res, tp[:, 1].max() + res, n_cross)  # :208-219
    min_tx, max_tx = tp[:, 0].min(), tp[:, 0].max()
    extrusion = np.arange(min_tx - 2 * all_res_min, max_tx + 2 * all_res_min, all_res_min)  # :241-245
    outer_scale = np.maximum(1e3, 300 + layer["h"] / 10)  # :247
    return dict(
        vx=vx,
        vy=vy,
        transform=transform,
        cross_section=cross_section,
        extrusion=extrusion,
        r0=float(outer_scale),
        nu=5 / 6,
        extrusion_res=float(np.gradient(extrusion).mean()),
    )


def process_geometry_multi(layers, members, all_res_min, outer_pp, timestep, n_t, model="3d"):
    """One process of several layers (atmosphere.py:117-257, ``model="3d"``): the water-weighted
    wind (:148-151), the hull of the first and the last layer (:153-186), one transform, one
    extrusion grid, a cross-section grid PER LAYER at that layer's resolution (:208-219), the
    outer scale from the mean height (:247) and nu = 1/3 (:249).  ``layers``: dict of per-layer
    arrays from ``generate_layers``; ``members``: indices of this process's layers."""
    members = np.asarray(members)
    w = (layers["absolute_humidity"] * layers["temperature"] * layers["divergence"])[members]
    vx = (w[:, None] * (layers["wind_east"][members][:, None] * np.ones(n_t))).sum(axis=0) / w.sum()
    vy = (w[:, None] * (layers["wind_north"][members][:, None] * np.ones(n_t))).sum(axis=0) / w.sum()
    vz = np.zeros(n_t)
    ends = members[[0, -1]] if len(members) > 1 else members[[0]]
    pts = np.concatenate([(layers["h"][l] * outer_pp + np.cumsum(timestep * np.c_[vx, vy, vz][None], axis=-2))[None] for l in ends], axis=0).reshape(-1, 3)
    pts[..., 2] += 1e-6 * np.random.standard_normal(pts[..., 2].shape)
    transform = compute_aligning_transform(pts, signature=(True, True, False))
    tp = pts @ transform
    cross = {}
    for l in members:
        res = layers["res"][l]
        n_cross = int(np.maximum(2, (np.ptp(tp[:, 1]) + 2 * res) / res))
        cross[int(l)] = np.linspace(tp[:, 1].min() - res, tp[:, 1].max() + res, n_cross)
    extrusion = np.arange(tp[:, 0].min() - 2 * all_res_min, tp[:, 0].max() + 2 * all_res_min, all_res_min)
    outer_scale = np.maximum(1e3, 300 + layers["h"][members].mean() / 10)
    return dict(vx=vx, vy=vy, transform=transform, cross_sections=cross, extrusion=extrusion, r0=float(outer_scale),
                nu=1 / 3 if model == "3d" else 5 / 6, cross_extent=float(np.ptp(tp[:, 1])))
