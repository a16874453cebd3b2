"""CPU oracle for maria's atmosphere -> TOD hot path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy restatement of what the reference computes
between ``Atmosphere.simulate_pwv`` and ``_compute_atmospheric_loading``
(SURVEY.md section 8(a), rows a8-a17 and a19).  It exists to check the HIP
kernels and to provide the timed CPU baseline; nothing in ``maria_amd`` may
import it (only ``tests/``, ``__graft_entry__.smoke`` and ``bench.py``'s
``cpu_baseline`` leg do).

PARITY PINNING.  The reference cannot be imported here (jax, dask, astropy,
h5py are absent) and its own tests pin no number on this path, so
this restatement is pinned only at its leaves: ``oracle/functions.py`` and
``oracle/geometry.py`` are checked against golden vectors produced by the three
reference modules that do import (``maria.functions``, ``maria.beam``,
``maria.utils.{linalg,rotations}``; see ``oracle/gen_golden.py``), and the
scipy calls below are the very calls the reference makes.  The jax arithmetic
(float32 ``RegularGridInterpolator``, float32 pointing) is restated from jax's
documented behaviour and is NOT pinned by a run of jax: **parity unpinned** for
those steps.

Every function cites the reference lines it follows (paths under maria/).
"""

from __future__ import annotations

import itertools

import numpy as np
import scipy as sp
import scipy.interpolate
import scipy.ndimage

f32 = np.float32
HALF_PI_F32 = f32(np.pi / 2)  # jnp.pi / 2 is weakly typed -> float32 next to a float32 array


class Transcendentals:
    """The float32 elementary functions of the pointing chain.  XLA's float32 sin / cos / atan2 / asin / tan / sqrt are
    not numpy's to the last ulp and cannot be restated without jax; the functions below go through this table so that
    tests/test_oracle_golden.py::test_one_ulp_envelope_of_the_float32_transcendentals can replace every one of them
    by a version that is off by +-1 ulp and measure what that does to the loading."""

    sqrt, arctan2, sin, cos, arcsin, tan = np.sqrt, np.arctan2, np.sin, np.cos, np.arcsin, np.tan


TR = Transcendentals


# ---------------------------------------------------------------------------
# pointing
# ---------------------------------------------------------------------------


def offsets_to_phi_theta(dx, dy, cphi, ctheta):
    """coords/transforms.py:10-29 (``unjitted_offsets_to_phi_theta``) in float32.

    jax without ``enable_x64`` demotes every input to float32.  The complex
    product is written out the way XLA expands it (two multiplies and an add per
    component, no fused multiply-add).
    """
    dx, dy = np.asarray(dx, f32), np.asarray(dy, f32)
    cphi, ctheta = np.asarray(cphi, f32), np.asarray(ctheta, f32)
    r = TR.sqrt(dx * dx + dy * dy)
    p = TR.arctan2(-dx, -dy)
    a_re = TR.sin(r) * TR.cos(p)
    a_im = TR.cos(r)
    ang = ctheta - HALF_PI_F32
    b_re, b_im = TR.cos(ang), TR.sin(ang)
    re = a_re * b_re - a_im * b_im
    im = a_re * b_im + a_im * b_re
    phi = TR.arctan2(TR.sin(r) * TR.sin(p), re) + cphi
    theta = TR.arcsin(im)
    return phi.astype(f32), theta.astype(f32)


def phi_theta_to_offsets(phi, theta, cphi, ctheta):
    """coords/transforms.py:36-53, the inverse of ``offsets_to_phi_theta`` (float64 here: it
    only closes the reference's own round-trip test, tests/coordinates/test_coordinates.py:7-19)."""
    phi, theta = np.asarray(phi, np.float64), np.asarray(theta, np.float64)
    dphi = phi - cphi
    proj_from_east = (np.cos(dphi) * np.cos(theta) + 1j * np.sin(theta)) * np.exp(1j * (np.pi / 2 - ctheta))
    dz = np.sin(dphi) * np.cos(theta) + 1j * np.real(proj_from_east)
    r = np.abs(dz)
    dz = dz * np.arcsin(r) / np.where(r > 0, r, 1.0)
    return np.stack([-np.real(dz), -np.imag(dz)], axis=-1)


def broadcast(offsets, az, el):
    """coords/coordinates.py:378-386: boresight [Ta] x offsets [D,2] -> [D,Ta] float32."""
    offsets = np.asarray(offsets)
    return offsets_to_phi_theta(offsets[:, 0][:, None], offsets[:, 1][:, None], np.asarray(az)[None, :], np.asarray(el)[None, :])


def downsample(t, az, el, timestep):
    """coords/coordinates.py:286-304: coarse time grid + linear interpolation."""
    t = np.asarray(t, float)
    ds_t = np.arange(t.min(), t.max(), timestep)
    kw = dict(axis=-1, bounds_error=False, fill_value="extrapolate")
    ds_az = sp.interpolate.interp1d(t, az, **kw)(ds_t)
    ds_el = sp.interpolate.interp1d(t, el, **kw)(ds_t)
    return ds_t, ds_az, ds_el


def project_unit(phi, theta):
    """coords/coordinates.py:333-349 with z=1 and the observer at the origin.

    ``phi``/``theta`` are the float32 arrays ``broadcast`` returned: numpy takes
    tan/cos/sin and the quotient in float32, and the product with the float64
    ``(z - self.z)`` promotes the result to float64.
    """
    phi, theta = np.asarray(phi, f32), np.asarray(theta, f32)
    tan_theta = TR.tan(theta)
    px = (TR.cos(phi) / tan_theta).astype(np.float64)
    py = (TR.sin(phi) / tan_theta).astype(np.float64)
    return np.stack([px, py, np.ones_like(px)], axis=-1)


# ---------------------------------------------------------------------------
# jax.scipy.interpolate.RegularGridInterpolator(method="linear"), float32
# ---------------------------------------------------------------------------


def rgi_linear_f32(axes, values, xi):
    """jax ``RegularGridInterpolator.__call__`` for ``method="linear"``,
    ``bounds_error=False``, ``fill_value=nan`` (call sites
    atmosphere/atmosphere.py:359-366 and band/band.py:283-286).

    Everything is float32.  Per axis ``i = clip(searchsorted(g, x) - 1, 0, n-2)``
    and ``w = (x - g[i]) / (g[i+1] - g[i])``; a point is out of bounds when
    ``x < g[0]`` or ``x > g[-1]``.  The value is the sum over the corner tuples in
    ``itertools.product`` order of ``values[corner] * weight`` with the weight
    built as ``((1 * w_0) * w_1) ...``, accumulated into 0.0.
    """
    axes = [np.asarray(a, f32) for a in axes]
    values = np.asarray(values, f32)
    xi = [np.asarray(x, f32) for x in xi]
    shape = np.broadcast_shapes(*[x.shape for x in xi])
    xi = [np.broadcast_to(x, shape).ravel() for x in xi]
    idx, wts = [], []
    oob = np.zeros(xi[0].shape, bool)
    for x, g in zip(xi, axes):
        i = np.searchsorted(g, x, side="left") - 1
        i = np.where(i < 0, 0, i)
        i = np.where(i > g.size - 2, g.size - 2, i)
        idx.append(i)
        wts.append(((x - g[i]) / (g[i + 1] - g[i])).astype(f32))
        oob |= x < g[0]
        oob |= x > g[-1]
    out = np.zeros(xi[0].shape, f32)
    for corner in itertools.product(*[[0, 1] for _ in axes]):
        weight = np.ones(xi[0].shape, f32)
        for c, w in zip(corner, wts):
            weight = weight * (w if c else (f32(1) - w))
        vals = values[tuple(i + c for i, c in zip(idx, corner))]
        out = out + vals * weight
    out = np.where(oob, f32(np.nan), out).astype(f32)
    return out.reshape(shape)


# ---------------------------------------------------------------------------
# turbulence sampling
# ---------------------------------------------------------------------------


def wind_translation(timestep, vx, vy):
    """atmosphere/atmosphere.py:318-319: cumulative drift [Ta,3]."""
    wind = np.c_[vx, vy, np.zeros(np.shape(vx))]
    return np.cumsum(timestep * wind, axis=0)


def simulate_pwv(pp, layers, pwv0, timestep):
    """atmosphere/atmosphere.py:309-373 given the smoothed screens.

    ``pp``: [D,Ta,3] from :func:`project_unit`.  Each layer is a dict with
    ``values`` [E,C] (already smoothed), ``extrusion`` [E], ``cross_section`` [C],
    ``transform`` [3,3], ``vx``/``vy`` [Ta], ``h``, ``pwv_rms``.  Returns the
    float64 zenith-scaled pwv [D,Ta]; raises like the reference when a sample
    leaves a screen (:368-369).
    """
    pwv = float(pwv0) * np.ones(pp.shape[:-1])
    for k, layer in enumerate(layers):
        translation = wind_translation(timestep, layer["vx"], layer["vy"])
        p = layer["h"] * pp + translation[None]
        tp = p @ np.asarray(layer["transform"], float)
        y = rgi_linear_f32(
            (layer["extrusion"], layer["cross_section"]),
            layer["values"],
            (tp[..., 0], tp[..., 1]),
        )
        if np.isnan(y).any():
            raise RuntimeError(f"Layer {k} introduced nans into PWV simulation.")
        # a numpy float64 scalar times a jax array defers to jax (float32 product);
        # the in-place add into the float64 numpy array then widens it.
        pwv += (f32(layer["pwv_rms"]) * y).astype(np.float64)
    return pwv


def layer_offsets(layer, timestep):
    """Host-side f64 constants the C ABI takes per layer (include/mrx.h, mrx_layer):
    off[t] = (translation[t] + (0,0,h)) @ transform, columns 0 and 1."""
    translation = wind_translation(timestep, layer["vx"], layer["vy"])
    q = (translation + np.array([0.0, 0.0, layer["h"]])) @ np.asarray(layer["transform"], float)
    return q[:, 0].copy(), q[:, 1].copy()


# ---------------------------------------------------------------------------
# emission
# ---------------------------------------------------------------------------


def atmosphere_power(table, base_temperature, zenith_pwv, elevation):
    """band/band.py:264-286 (``method="linear"``): trilinear float32 lookup of the
    band-integrated emission table ``table["values"]`` [nT,npwv,nel] on
    ``table["T"], table["pwv"], table["el"]``."""
    return rgi_linear_f32(
        (table["T"], table["pwv"], table["el"]),
        table["values"],
        (np.asarray(base_temperature), zenith_pwv, elevation),
    )


def atmosphere_power_cubic(table, base_temperature, zenith_pwv, elevation):
    """band/band.py:288-300 (``method != "linear"``): linear ``interp1d`` along the temperature
    axis, then scipy's ``RegularGridInterpolator(method="cubic")`` on (pwv, el); float64."""
    tiv = sp.interpolate.interp1d(np.asarray(table["T"], float), np.asarray(table["values"], float), kind="linear", axis=0)(base_temperature)
    rgi = sp.interpolate.RegularGridInterpolator((np.asarray(table["pwv"], float), np.asarray(table["el"], float)), tiv, method="cubic")
    return rgi((zenith_pwv, elevation))


def mueller00(gamma):
    """array/array.py:204-218, element [0,0]: 1 unpolarised (gamma NaN), else 0.5."""
    gamma = np.asarray(gamma, float)
    return np.where(np.isnan(gamma), 0.5 * np.sqrt(2) ** 2, 0.5)


def coarse_loading(pwv, theta, band_index, tables, base_temperature, m00, method="linear"):
    """sim/atmosphere.py:39-65: per band emission x Mueller weight -> [D,Ta] float32.
    ``method``: ``obs.atmosphere.interpolation_method`` (band/band.py:283-300)."""
    loading = np.zeros(pwv.shape, f32)
    for b, table in enumerate(tables):
        mask = np.asarray(band_index) == b
        if not mask.any():
            continue
        el = np.asarray(theta, f32)[mask].clip(max=np.pi / 2)
        if method == "linear":
            p = atmosphere_power(table, base_temperature, pwv[mask], el)
            loading[mask] = np.asarray(m00, f32)[mask][:, None] * p
        else:  # float64 from scipy, times the float64 Mueller element, stored into the float32 array
            p = atmosphere_power_cubic(table, base_temperature, pwv[mask], el)
            loading[mask] = np.asarray(m00, np.float64)[mask][:, None] * p
    return loading


def upsample_cubic(ta, loading_a, t, dtype=f32):
    """sim/atmosphere.py:72-82."""
    return sp.interpolate.interp1d(
        ta, loading_a, kind="cubic", bounds_error=False, fill_value="extrapolate", axis=-1
    )(t).astype(dtype)


def upsample_linear(ta, pwv, t):
    """sim/atmosphere.py:30-37."""
    return sp.interpolate.interp1d(ta, pwv, bounds_error=False, fill_value="extrapolate")(t)


# ---------------------------------------------------------------------------
# TOD.to("K_RJ")
# ---------------------------------------------------------------------------

K_B = 1.380649e-23  # maria/constants.py


def transmission_integral_grid(passband, side_nu, opacity):
    """band/band.py:248-252 with the default nu range: trapezoid over nu of
    passband(nu) * exp(-opacity) on the spectrum's (T, pwv, el) grid."""
    return np.trapezoid(passband(side_nu) * np.exp(-opacity), x=side_nu, axis=-1)


def power_to_rayleigh_jeans(P_pW, table, base_temperature, zenith_pwv, elevation, polarized=False):
    """tod/tod.py:106-142 -> Calibration("pW -> K_RJ") -> calibration/functions.py:73-90.

    ``P_pW`` [Db, T] float32 is first brought to base units (x 1e-12, float32 array times
    python float), the transmission integral is the float32 trilinear lookup of
    ``table["values"]`` at (scalar T0, scalar pwv, per-sample elevation) (band.py:253-255), and
    the quotient is taken in float32: python-float factors fold to float32 next to a float32
    array (NEP 50)."""
    P = np.asarray(P_pW, f32) * f32(1e-12)
    integral = rgi_linear_f32((table["T"], table["pwv"], table["el"]), table["values"],
                              (np.asarray(base_temperature), np.asarray(zenith_pwv), np.asarray(elevation, f32)))
    den = f32((0.5 if polarized else 1.0) * K_B) * integral
    return (P / den).astype(f32)


def calibrate_to_krj(tod_pW, band_index, cal_tables, base_temperature, zenith_pwv, el_full, polarized=None):
    """Per band as ``TOD.to`` loops (tod.py:124-137); ``el_full`` [D, T] float32 are the
    detector elevations of observation.coords (observation.py:55-58)."""
    out = np.empty_like(np.asarray(tod_pW, f32))
    for b, table in enumerate(cal_tables):
        mask = np.asarray(band_index) == b
        if mask.any():
            pol = bool(polarized[b]) if polarized is not None else False
            out[mask] = power_to_rayleigh_jeans(tod_pW[mask], table, base_temperature, zenith_pwv, el_full[mask], pol)
    return out


# ---------------------------------------------------------------------------
# screens
# ---------------------------------------------------------------------------


def smooth_screen(values, sigma_e_px, sigma_c_px):
    """atmosphere/atmosphere.py:341-344."""
    return sp.ndimage.gaussian_filter(values, sigma=(sigma_e_px, sigma_c_px))


def map_smooth(data, weight, sigma_y_px, sigma_x_px):
    """map/projection.py:485-504 on one [ny,nx] (or stacked [...,ny,nx]) map."""
    data = np.asarray(data)
    weight = np.ones_like(data) if weight is None else np.asarray(weight)
    sig = (sigma_y_px, sigma_x_px)
    numer = sp.ndimage.gaussian_filter(data * weight, sigma=sig, axes=(-2, -1))
    denom = sp.ndimage.gaussian_filter(weight, sigma=sig, axes=(-2, -1))
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(denom > 0, numer / denom, 0)
    return out, denom


# ---------------------------------------------------------------------------
# the whole deterministic path, for parity tests and the CPU baseline
# ---------------------------------------------------------------------------


def run_path(problem, return_intermediates=False):
    """pointing -> gather -> emission -> cubic upsample on a problem dict
    (the layout ``maria_amd.synthetic.make_problem`` and the tests build):

    ``t`` [T] f64, ``ta`` [Ta], ``az_a``/``el_a`` [Ta] f64 coarse boresight,
    ``offsets`` [D,2], ``band_index`` [D], ``m00`` [D], ``layers`` (see
    :func:`simulate_pwv`), ``tables``, ``T0``, ``pwv0``, ``timestep``, optional
    ``gain`` [D].
    """
    phi, theta = broadcast(problem["offsets"], problem["az_a"], problem["el_a"])
    pp = project_unit(phi, theta)
    pwv = simulate_pwv(pp, problem["layers"], problem["pwv0"], problem["timestep"])
    loading_a = coarse_loading(pwv, theta, problem["band_index"], problem["tables"], problem["T0"], problem["m00"],
                               method=problem.get("interpolation_method", "linear"))
    tod = upsample_cubic(problem["ta"], loading_a, problem["t"])
    if problem.get("gain") is not None:
        tod = (tod * np.asarray(problem["gain"], f32)[:, None]).astype(f32)
    if return_intermediates:
        return tod, {"phi": phi, "theta": theta, "pwv": pwv, "loading_a": loading_a}
    return tod
