#!/usr/bin/env python3
"""Generate tests/golden/leaves.json from the reference's own importable modules.

Run in the build container only (``/root/reference`` is read-only and does not
exist on the GPU box):

    python oracle/gen_golden.py

``import maria`` fails here with an ordinary ModuleNotFoundError (jax, dask,
astropy, h5py ... are not installed).  Registering bare parent
packages lets the leaf modules that only need numpy/scipy load unmodified:
``maria.constants``, ``maria.functions``, ``maria.beam``, ``maria.utils.linalg``,
``maria.utils.rotations``, ``maria.utils.signal``, ``maria.plan.patterns``.  No third-party library is stubbed.
Where jax IS installed (not in the build container) the run also writes tests/golden/jax_steps.json: the
reference's float32 pointing chain and jax's RegularGridInterpolator on fixed inputs (``jax_steps`` below) --
the three steps whose parity is otherwise unpinned.  The outputs below
are data (inputs and the reference's answers); no reference source is copied.
"""

from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "leaves.json")


def _load_reference_leaves():
    for pk in ("maria", "maria.utils", "maria.plan"):
        m = types.ModuleType(pk)
        m.__path__ = [os.path.join(REF, *pk.split("."))]
        sys.modules[pk] = m
    import maria.beam as beam
    import maria.constants as constants
    import maria.functions as functions
    import maria.utils.linalg as linalg
    import maria.utils.rotations as rotations
    import maria.utils.signal as signal

    # maria/utils/__init__.py re-exports its rotations module; the bare parent package
    # gets the reference's own function under the name plan/patterns.py imports
    sys.modules["maria.utils"].get_rotation_matrix_2d = rotations.get_rotation_matrix_2d
    import maria.plan.patterns as patterns

    return constants, functions, beam, linalg, rotations, signal, patterns


def main():
    constants, functions, beam, linalg, rotations, signal, patterns = _load_reference_leaves()
    g = {"_generator": "oracle/gen_golden.py", "_reference": "thomaswmorris/maria @ /root/reference"}

    g["constants"] = {"k_B": constants.k_B, "c": constants.c}

    r = np.array([0, 0.5, 2, 17, 250, 1000, 3333, 2e4, 2e6], float)
    g["matern"] = {"r": r.tolist(), "cases": []}
    for nu, r0 in [(5 / 6, 1e3), (1 / 3, 1e3), (5 / 6, 2345.0), (1 / 3, 17.0)]:
        g["matern"]["cases"].append(
            {
                "nu": nu,
                "r0": r0,
                "approximate_normalized_matern": functions.approximate_normalized_matern(r, nu=nu, r0=r0).tolist(),
                "normalized_matern": functions.normalized_matern(r / r0, nu).tolist(),
            }
        )
    rr = np.geomspace(1e-3, 5e4, 64)
    g["matern_dense"] = {
        "r": rr.tolist(),
        "nu": 5 / 6,
        "r0": 1e3,
        "value": functions.approximate_normalized_matern(rr, nu=5 / 6, r0=1e3).tolist(),
    }

    z = np.array([250.0, 750.0, 2500.0, 1e4])
    g["beam"] = {
        "z": z.tolist(),
        "physical_fwhm_100m_90GHz": beam.compute_physical_fwhm(100, z=z, nu=90e9).tolist(),
        "physical_fwhm_6m_150GHz": beam.compute_physical_fwhm(6, z=z, nu=150e9).tolist(),
        "angular_fwhm_50m_inf_150GHz": float(beam.compute_angular_fwhm(50, z=np.inf, nu=150e9)),
        "angular_fwhm_12m_z_230GHz": beam.compute_angular_fwhm(12, z=z, nu=230e9).tolist(),
    }
    g["radiometry"] = {
        "rayleigh_jeans_spectrum_1K_150GHz": float(functions.rayleigh_jeans_spectrum(1, 150e9)),
        "planck_spectrum_2.72548K_150GHz": float(functions.planck_spectrum(2.72548, 150e9)),
    }

    # fast_psd_inverse on a seeded SPD matrix (utils/linalg.py:95-102)
    rng = np.random.default_rng(11)
    a = rng.standard_normal((7, 7))
    spd = a @ a.T + 7 * np.eye(7)
    g["fast_psd_inverse"] = {"M": spd.tolist(), "inv": linalg.fast_psd_inverse(spd).tolist()}

    # rotations (utils/rotations.py:25-77)
    g["orthogonal_transform"] = {
        "signature": [True, True, False],
        "entries": [0.37],
        "R": rotations.get_orthogonal_transform((True, True, False), [0.37]).tolist(),
        "signature3": [True, True, True],
        "entries3": [0.1, -0.2, 0.3],
        "R3": rotations.get_orthogonal_transform((True, True, True), [0.1, -0.2, 0.3]).tolist(),
    }
    # an elongated, rotated cloud: drifting footprint like atmosphere.py:161-186
    pts_rng = np.random.default_rng(5)
    n = 400
    along = np.linspace(0, 3000, n) + 20 * pts_rng.standard_normal(n)
    across = 40 * pts_rng.standard_normal(n)
    ang = np.radians(33.0)
    pts = np.c_[along * np.cos(ang) - across * np.sin(ang), along * np.sin(ang) + across * np.cos(ang), 1500 + 1e-6 * pts_rng.standard_normal(n)]
    np.random.seed(1234)
    R = rotations.compute_aligning_transform(pts, signature=(True, True, False))
    g["aligning_transform"] = {"points": pts.tolist(), "numpy_seed": 1234, "R": R.tolist()}

    # pointing-matrix ingredients (utils/linalg.py:9-58): the sampling rule of map sampling.
    # Sides as the map front end hands them over: (t: one node, eta: descending after the
    # parity flip, xi: ascending); points inside, on nodes, and outside the map.
    pm_rng = np.random.default_rng(23)
    eta = np.linspace(0.02, -0.02, 9)
    xi = np.linspace(-0.03, 0.03, 13)
    yy = np.concatenate([pm_rng.uniform(-0.025, 0.025, 40), eta[[0, 3, 8]], [0.5, -0.5]]).reshape(5, 9)
    xx = np.concatenate([pm_rng.uniform(-0.035, 0.035, 40), xi[[0, 5, 12]], [-0.7, 0.7]]).reshape(5, 9)
    tt = np.zeros((5, 9))
    g["pointing_matrix"] = {"eta": eta.tolist(), "xi": xi.tolist(), "y": yy.tolist(), "x": xx.tolist(), "cases": []}
    for bilinear in (True, False):
        smp, pix, wts, n_pix, n_smp = linalg.compute_pointing_matrix_ingredients(
            x_list=(tt, yy, xx), side_list=(np.array([0.0]), eta, xi), bilinear=bilinear)
        g["pointing_matrix"]["cases"].append({"bilinear": bilinear, "samples": smp.tolist(), "pixels": pix.tolist(),
                                             "weights": wts.tolist(), "n_pixels": int(n_pix), "n_samples": int(n_smp)})

    # spatial basis of the correlated noise modes (utils/linalg.py:105-126)
    sb_rng = np.random.default_rng(31)
    offs = sb_rng.uniform(-0.004, 0.004, (25, 2))
    g["spatial_basis"] = {"offsets": offs.tolist(), "k": 5, "n_side": 16, "scale": 0.01,
                          "B": linalg.generate_spatial_basis(offs, k=5, n_side=16, scale=0.01).tolist()}

    # TOD pre-processing leaves (utils/signal/__init__.py, utils/signal/filters.py)
    sg_rng = np.random.default_rng(41)
    tt = 100.0 + np.arange(400) / 20.0  # 20 s at 20 Hz
    Dm = np.cumsum(sg_rng.normal(size=(6, 400)), axis=1) + 3.0 * sg_rng.normal(size=(6, 1))
    common = np.sin(2 * np.pi * (tt - tt[0]) / 7.0)
    Dm += np.outer(sg_rng.uniform(2, 5, 6), common) * 10
    A, Bm = signal.decompose(Dm, k=3)
    g["signal"] = {
        "t": tt.tolist(), "D": Dm.tolist(),
        "bspline_basis_spacing_3_order_3": signal.bspline_basis(tt, spacing=3.0, order=3).tolist(),
        "remove_slope": signal.remove_slope(Dm).tolist(),
        "lowpass_fc1.5_order1": signal.lowpass(Dm, fc=1.5, sample_rate=20.0, order=1).tolist(),
        "highpass_fc0.2_order1": signal.highpass(Dm, fc=0.2, sample_rate=20.0, order=1).tolist(),
        "lowpass_fc2.5_order2": signal.lowpass(Dm, fc=2.5, sample_rate=20.0, order=2).tolist(),
        "decompose_k3_first2_modes": (A[:, :2] @ Bm[:2]).tolist(),
    }

    # the daisy scan pattern every BASELINE config names (plan/patterns.py:108-155)
    g["daisy"] = []
    for kw in (
        dict(n=600, fs=10.0, x_throw=0.5, y_throw=0.5, speed=0.5),
        dict(n=900, fs=20.0, x_throw=0.25, y_throw=0.4, speed=0.3, petals=2.0, miss_factor=0.35, miss_freq=0.17),
    ):
        tt = 1.7e9 + np.arange(kw["n"]) / kw["fs"]
        args = {k: v for k, v in kw.items() if k not in ("n", "fs")}
        g["daisy"].append({"n": kw["n"], "fs": kw["fs"], "t0": 1.7e9, "kwargs": args, "offsets": patterns.daisy(tt, **args).tolist()})

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    jax_steps()


JAX_OUT = os.path.join(os.path.dirname(OUT), "jax_steps.json")


def jax_step_inputs():
    """The inputs of the jax leg (seeded; tests/test_oracle_golden.py rebuilds the oracle's answers from the
    copies stored in the file, so this function is only called here)."""
    rng = np.random.default_rng(20260612)
    offsets = np.radians(rng.uniform(-1.0, 1.0, (24, 2)))
    offsets[0] = 0.0                                    # the boresight itself
    offsets[1] = np.radians([0.0, 0.75])                # on an axis: atan2(-0, -dy)
    az = np.radians(np.r_[rng.uniform(0, 360, 12), 359.999, 0.0, 45.0, 180.0])
    el = np.radians(np.r_[rng.uniform(20, 85, 12), 60.0, 89.2, 30.0, 45.0])
    # a screen-shaped lookup (atmosphere.py:359-366): float64 axes of a ribbon far from its origin, points inside,
    # on nodes, on the last node and outside
    ext = -3000.0 + 5.0 * np.arange(48)
    cross = np.linspace(1234.5 - 7.0, 1234.5 + 120.0 + 7.0, 19)
    values = rng.standard_normal((48, 19))
    pe = np.r_[rng.uniform(ext[0], ext[-1], 40), ext[[0, 7, 47]], ext[0] - 1.0, ext[-1] + 1e-3, 0.5 * (ext[3] + ext[4])]
    pc = np.r_[rng.uniform(cross[0], cross[-1], 40), cross[[0, 9, 18]], cross[5], cross[5], cross[-1] + 2.0]
    # a table-shaped lookup (band.py:283-286): (scalar T0, pwv, elevation) on a (T, pwv, el) grid whose last elevation
    # node is 90.1 deg (spectrum/atmosphere.py:48-50)
    T = np.array([250.0, 270.0, 290.0])
    pwv = np.linspace(0.0, 10.0, 21)
    elg = np.radians(np.r_[np.linspace(10.0, 87.5, 15), 90.1])
    tab = 30.0 * (T[:, None, None] / 270.0) * (1 - np.exp(-(0.04 + 0.025 * pwv[None, :, None]) / np.sin(np.minimum(elg, np.pi / 2))[None, None, :]))
    qp = np.r_[rng.uniform(0.2, 3.0, 30), 0.0, 10.0, 10.5, 1.0]
    qe = np.radians(np.r_[rng.uniform(25, 85, 30), 10.0, 90.0, 60.0, 9.0])
    return dict(offsets=offsets, az=az, el=el, ext=ext, cross=cross, values=values, pe=pe, pc=pc, T=T, pwv=pwv, elg=elg, tab=tab,
                T0=273.15, qp=qp, qe=qe)


def jax_steps():
    """Optional leg: where jax is installed, pin the three steps SURVEY 8(c) lists as unpinned -- the float32 pointing
    chain (coords/transforms.py:10-29, the reference's own function, loaded unmodified through the bare-parent loader)
    and jax.scipy.interpolate.RegularGridInterpolator as called at atmosphere/atmosphere.py:359-366 and
    band/band.py:283-286 -- into tests/golden/jax_steps.json.  jax is never stubbed: without it this leg does nothing
    and tests/test_oracle_golden.py::test_jax_* skip."""
    try:
        import jax  # noqa: F401
        import jax.scipy as jsp
    except ImportError:
        print("jax is not installed: tests/golden/jax_steps.json not written (the jax steps stay unpinned)")
        return
    m = types.ModuleType("maria.coords")
    m.__path__ = [os.path.join(REF, "maria", "coords")]
    sys.modules["maria.coords"] = m
    import maria.coords.transforms as transforms

    x = jax_step_inputs()
    lst = lambda a: np.asarray(a).tolist()  # noqa: E731
    g = {"_generator": "oracle/gen_golden.py (jax leg)", "_jax_version": jax.__version__,
         "_x64": bool(jax.config.read("jax_enable_x64"))}
    # Coordinates.broadcast (coordinates.py:378-386): offsets[..., None, :] against the boresight arrays
    pt = np.asarray(transforms.unjitted_offsets_to_phi_theta(x["offsets"][:, None, :], x["az"], x["el"]))
    g["offsets_to_phi_theta"] = {"offsets": lst(x["offsets"]), "az": lst(x["az"]), "el": lst(x["el"]), "dtype": str(pt.dtype),
                                 "phi": lst(pt[..., 0].astype(np.float64)), "theta": lst(pt[..., 1].astype(np.float64))}
    y = np.asarray(jsp.interpolate.RegularGridInterpolator((x["ext"], x["cross"]), x["values"], method="linear")(
        np.stack([x["pe"], x["pc"]], axis=-1)))
    g["rgi_screen"] = {"extrusion": lst(x["ext"]), "cross_section": lst(x["cross"]), "values": lst(x["values"]),
                       "points_e": lst(x["pe"]), "points_c": lst(x["pc"]), "dtype": str(y.dtype),
                       "y": [None if v != v else float(v) for v in y.astype(np.float64)]}
    p = np.asarray(jsp.interpolate.RegularGridInterpolator((x["T"], x["pwv"], x["elg"]), x["tab"])((x["T0"], x["qp"], x["qe"])))
    g["rgi_table"] = {"T": lst(x["T"]), "pwv": lst(x["pwv"]), "el": lst(x["elg"]), "values": lst(x["tab"]), "T0": x["T0"],
                      "points_pwv": lst(x["qp"]), "points_el": lst(x["qe"]), "dtype": str(p.dtype),
                      "p": [None if v != v else float(v) for v in p.astype(np.float64)]}
    with open(JAX_OUT, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", JAX_OUT, os.path.getsize(JAX_OUT), "bytes")


if __name__ == "__main__":
    main()
