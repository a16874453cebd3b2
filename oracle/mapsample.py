"""Map sampling restated in numpy.  TEST INFRASTRUCTURE ONLY.

Follows maria/sim/map.py:76-172 (``_sample_maps``), maria/map/projection.py:134-179
(pointing-matrix ingredients with Stokes weights), maria/utils/linalg.py:9-58
(``compute_pointing_matrix_ingredients``), maria/coords/transforms.py:36-80 (float32
``phi_theta_to_offsets``, ``phi_theta_to_xyz``, ``xyz_to_phi_theta``) and
maria/coords/coordinates.py:184-236 (frame transform: float32 points times a float64
3x3 per sample, back to float32 angles).  ``pointing_matrix_ingredients`` is pinned by
tests/golden/leaves.json (generated from the reference's own function); the float32 jax
steps are restated from jax semantics (parity unpinned, like the rest of the chain).
"""

from __future__ import annotations

import numpy as np
import scipy.ndimage

from . import hotpath

f32 = np.float32
K_B = 1.380649e-23


def phi_theta_to_xyz(phi, theta):
    """transforms.py:56-65, float32."""
    phi, theta = np.asarray(phi, f32), np.asarray(theta, f32)
    c = np.cos(theta)
    return np.stack([np.cos(phi) * c, np.sin(phi) * c, np.sin(theta)], axis=-1).astype(f32)


def xyz_to_phi_theta(xyz):
    """transforms.py:68-75, float32 (the argument is demoted on entry to the jitted function)."""
    xyz = np.asarray(xyz, f32)
    two_pi = f32(2 * np.pi)
    phi = np.mod(np.arctan2(xyz[..., 1], xyz[..., 0]), two_pi).astype(f32)
    theta = np.arcsin(xyz[..., 2] / np.sqrt(np.sum(xyz * xyz, axis=-1, dtype=f32))).astype(f32)
    return phi, theta


def phi_theta_to_offsets(phi, theta, cphi, ctheta):
    """transforms.py:36-53, float32; cphi/ctheta are static python floats (weakly typed)."""
    phi, theta = np.asarray(phi, f32), np.asarray(theta, f32)
    dphi = (phi - f32(cphi)).astype(f32)
    rot = np.exp(1j * (np.pi / 2 - ctheta)).astype(np.complex64)
    proj = ((np.cos(dphi) * np.cos(theta)).astype(f32) + 1j * np.sin(theta).astype(f32)).astype(np.complex64) * rot
    dz = ((np.sin(dphi) * np.cos(theta)).astype(f32) + 1j * proj.real.astype(f32)).astype(np.complex64)
    r = np.abs(dz).astype(f32)
    dz = dz * (np.arcsin(r) / np.where(r > 0, r, f32(1.0))).astype(f32)
    return np.stack([-dz.real, -dz.imag], axis=-1).astype(f32)


def frame_angles(az, el, transform_stack=None):
    """coordinates.py:220-230: detector az/el [D, T] float32 -> map-frame (phi, theta) float32.
    ``transform_stack`` [T, 3, 3] float64, or None for a map in the az/el frame."""
    if transform_stack is None:
        return np.asarray(az, f32), np.asarray(el, f32)
    pts = phi_theta_to_xyz(az, el)  # [D, T, 3] float32
    out = (np.expand_dims(pts, -2).astype(np.float64) @ np.asarray(transform_stack, np.float64)).squeeze(-2)
    return xyz_to_phi_theta(out)


def pointing_matrix_ingredients(x_list, side_list, bilinear=True):
    """utils/linalg.py:9-58."""
    if isinstance(bilinear, bool):
        bilinear = len(x_list) * [bilinear]
    shape = np.broadcast_shapes(*[np.shape(x) for x in x_list])
    x_list = [np.reshape(x, shape) for x in x_list]
    samples = np.arange(x_list[0].size, dtype=int).reshape(shape)
    pixels = np.zeros(shape, dtype=int)
    weights = np.ones(shape, dtype=float)
    n_pixels = 1
    for dim, (x, side, bil) in enumerate(zip(x_list, side_list, bilinear)):
        if np.size(side) <= 1:
            continue
        side = np.asarray(side, float)
        pixels = pixels * len(side)
        n_pixels *= len(side)
        padded = np.array([-np.inf, *side, np.inf])
        if bil:
            b = np.digitize(x, bins=side)
            with np.errstate(invalid="ignore"):
                p = (x - padded[b]) / np.diff(padded)[b]
            p = np.where(p > 0, p, 0)
            dim_pixels = np.stack([b - 1, b], axis=0).clip(0, len(side) - 1)
            dim_weights = np.stack([1 - p, p], axis=0)
        else:
            b = np.digitize(x, bins=0.5 * (side[1:] + side[:-1]))
            dim_pixels = b[None]
            dim_weights = np.ones_like(x, dtype=float)[None]
        for add in range(dim):
            dim_pixels = np.expand_dims(dim_pixels, add + 1)
            dim_weights = np.expand_dims(dim_weights, add + 1)
        samples = samples + np.zeros_like(dim_pixels)
        pixels = pixels + dim_pixels
        weights = weights * dim_weights
    return samples.reshape(-1, *shape), pixels.reshape(-1, *shape), weights.reshape(-1, *shape), n_pixels, x_list[0].size


def mueller_row(gamma):
    """array/array.py:204-221: row 0 of the detector Mueller matrices, [D, 4] (I, Q, U, V)."""
    a = np.asarray(gamma, float)
    m = np.stack([np.where(np.isnan(a), np.sqrt(2), 1), np.where(np.isnan(a), 0, np.cos(2 * a)),
                  np.where(np.isnan(a), 0, np.sin(2 * a)), np.zeros_like(a)], axis=1)
    return (0.5 * m[..., None] * m[..., None, :])[:, 0]


def sample_channel(offsets, eta, xi, channel_map, stokes_weights, bilinear=True):
    """projection.py:134-179 + `P @ map` of sim/map.py:153: offsets [D, T, 2] float32 (dx, dy),
    eta/xi the map's axes after parity, channel_map [S, n_eta, n_xi], stokes_weights [D, S]
    -> float64 [D, T]."""
    ox, oy = offsets[..., 0], offsets[..., 1]
    _, pixels, weights, n_pixels, _ = pointing_matrix_ingredients((oy, ox), (eta, xi), bilinear)
    flat = np.asarray(channel_map).reshape(len(channel_map), -1)
    out = np.zeros(ox.shape, np.float64)
    for s in range(flat.shape[0]):
        w = weights * np.asarray(stokes_weights, float)[:, s][None, :, None]
        out += (w * flat[s][pixels]).sum(axis=0)
    return out


def channel_calibration(table, axes, base_temperature, zenith_pwv, elevation):
    """band/band.py:235-255 with a spectrum: float32 trilinear lookup of the channel's
    transmission integral at (scalar T0, pwv [D, T], el [D, T]), then 1e12 k_B (map.py:131-135)."""
    T0 = np.full(np.shape(zenith_pwv), base_temperature)
    integral = hotpath.rgi_linear_f32(axes, table, (T0, zenith_pwv, elevation))
    return (1e12 * K_B) * integral  # float32 array times python scalars: float32


def sample_maps(az, el, t, coarse_t, coarse_pwv, map_eta, map_xi, center, channel_maps, stokes_weights,
                cal_tables=None, cal_axes=None, cal_scalars=None, base_temperature=None, transform_stack=None, bilinear=True):
    """sim/map.py:76-172 for one band.  az/el [D, T] float32 detector pointing; channel_maps
    [C, S, n_eta, n_xi] in K_RJ; calibration per channel either from ``cal_tables`` [C][nT, npwv, nel]
    (atmosphere present) or ``cal_scalars`` [C] (no atmosphere).  Returns float32 [D, T] in pW."""
    phi, theta = frame_angles(az, el, transform_stack)
    offsets = phi_theta_to_offsets(phi, theta, center[0], center[1])
    loading = np.zeros(np.shape(az), f32)
    pwv = None
    if cal_tables is not None:
        pwv = hotpath.upsample_linear(coarse_t, coarse_pwv, t)  # sim/atmosphere.py:30-37
    for c, cmap in enumerate(channel_maps):
        if cal_tables is not None:
            pw_per_k = channel_calibration(cal_tables[c], cal_axes, base_temperature, pwv, el)
        else:
            pw_per_k = 1e12 * K_B * cal_scalars[c]
        pw = pw_per_k * sample_channel(offsets, map_eta, map_xi, cmap, stokes_weights, bilinear)
        loading += pw  # float32 accumulator (map.py:155)
    return scipy.ndimage.convolve1d(loading, weights=np.array([0.25, 0.5, 0.25]), axis=-1)  # map.py:170


def bin_map(az, el, tod, weight, map_eta, map_xi, center, stokes_weights, n_stokes, channel=None, n_channels=1,
            transform_stack=None, bilinear=False):
    """mappers/bin_mapper.py:84-120: ``map_sum += (W * D) @ P`` and ``map_wgt += W @ |P|`` with the
    Stokes-weighted pointing matrix of map/projection.py:134-179.  az/el [D, T] float32 detector
    pointing; tod, weight [D, T].  Returns (sum, wgt) [n_stokes, n_channels, n_eta, n_xi] float64."""
    phi, theta = frame_angles(az, el, transform_stack)
    offsets = phi_theta_to_offsets(phi, theta, center[0], center[1])
    _, pixels, weights, n_pixels, _ = pointing_matrix_ingredients((offsets[..., 1], offsets[..., 0]), (map_eta, map_xi), bilinear)
    if channel is not None:
        pixels = pixels + (np.asarray(channel)[None, :, None] * n_pixels)
    n_plane = n_pixels * n_channels
    W = np.ones(np.shape(tod)) if weight is None else np.asarray(weight, float)
    WD = W * np.asarray(tod, float)
    msum, mwgt = np.zeros(n_stokes * n_plane), np.zeros(n_stokes * n_plane)
    for s in range(n_stokes):
        w = weights * np.asarray(stokes_weights, float)[:, s][None, :, None]
        np.add.at(msum, (pixels + s * n_plane).ravel(), (w * WD[None]).ravel())
        np.add.at(mwgt, (pixels + s * n_plane).ravel(), (np.abs(w) * W[None]).ravel())
    shape = (n_stokes, n_channels, len(map_eta), len(map_xi))
    return msum.reshape(shape), mwgt.reshape(shape)
