"""numpy spectral screen generator for CPU-only tests and the CPU baseline.
TEST INFRASTRUCTURE ONLY (see oracle/hotpath.py).

Same construction as the HIP generator (maria_amd/csrc/mrx_screen.hip): complex
white noise in k space times sqrt(PSD), PSD ~ (k0^2 + |k|^2)^-(nu+1) with
k0 = sqrt(2 nu)/r0, the 2-D transform of the Matern covariance of
functions/__init__.py:30-39; inverse FFT; real part scaled to unit variance.
The random stream is numpy's, not Philox, so screens differ sample by sample
from the GPU's: parity of the generator is statistical.
"""

from __future__ import annotations

import numpy as np


def psd_amplitude(ny, nx, dy, dx, r0, nu):
    ky = 2 * np.pi * np.fft.fftfreq(ny, dy)[:, None]
    kx = 2 * np.pi * np.fft.fftfreq(nx, dx)[None, :]
    k0sq = 2.0 * nu / r0**2
    return (k0sq + kx**2 + ky**2) ** (-(nu + 1.0) / 2.0)


def periodic_covariance(shape, steps, r0, nu, x_cut=30.0):
    """The exact Matern correlation (functions/__init__.py:30-39) summed over the periodic images of the
    grid ``shape`` (2 or 3 axes) that lie within ``x_cut`` outer scales: the covariance of the field
    mrx_screen_amplitudes defines, on the full grid."""
    from itertools import product

    from .functions import normalized_matern

    half = [np.arange(n // 2 + 1) * float(d) for n, d in zip(shape, steps)]
    periods = [n * float(d) for n, d in zip(shape, steps)]
    reach = [int(np.ceil(x_cut * r0 / L + 0.5)) for L in periods]
    acc = np.zeros([len(h) for h in half])
    for image in product(*[range(-m, m + 1) for m in reach]):
        axes = [(h + k * L) ** 2 for h, k, L in zip(half, image, periods)]
        if sum(a.min() for a in axes) >= (x_cut * r0) ** 2:
            continue
        r = np.sqrt(sum(np.meshgrid(*axes, indexing="ij"))) / r0
        near = r < x_cut
        acc[near] += normalized_matern(r[near], nu)
    fold = [np.minimum(np.arange(n), n - np.arange(n)) for n in shape]
    return acc[np.ix_(*fold)]


def covariance_amplitude(shape, steps, r0, nu, x_cut=30.0):
    """sqrt of the eigenvalues of periodic_covariance, scaled so that a field drawn with them has Matern's
    STRUCTURE FUNCTION (variance = the zero-lag image sum): the table mrx_screen_amplitudes builds by
    cosine sums, here by numpy's FFT.  Returns (amp, sum of amp^2 / zero-lag value)."""
    rho = periodic_covariance(shape, steps, r0, nu, x_cut)
    amp = np.sqrt(np.maximum(np.fft.fftn(rho).real, 0.0))
    return amp, rho.reshape(-1)[0]


def numpy_screen(ny, nx, dy, dx, r0, nu, rng):
    amp = psd_amplitude(ny, nx, dy, dx, r0, nu)
    noise = rng.standard_normal((ny, nx)) + 1j * rng.standard_normal((ny, nx))
    field = np.fft.ifft2(amp * noise).real * (ny * nx)
    return (field / np.sqrt((amp**2).sum())).astype(np.float32)


def hermitian_philox_screen(philox4x32, seed, stream, ny, nx, dy, dx, r0, nu, amp=None):
    """The screen mrx_screen_generate defines (include/mrx.h): the Hermitian half spectrum
    rebuilt cell by cell from the library's own Philox routine, then numpy's irfft2.
    ``philox4x32(seed, counter) -> 4 words`` (maria_amd._lib.philox4x32, host-evaluated).
    ``amp``: [ny, nx] amplitudes (covariance_amplitude) instead of the power law."""
    if amp is None:
        amp = psd_amplitude(ny, nx, dy, dx, r0, nu)
    half = ny // 2
    H = np.zeros((ny, nx // 2 + 1), complex)

    def normal_pair(a, b):
        u1 = ((a >> 8) + 0.5) / 16777216.0
        u2 = (b >> 8) / 16777216.0
        rad = np.sqrt(-2 * np.log(u1))
        return rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)

    for ix in range(nx // 2 + 1):
        edge = ix == 0 or ix == nx // 2
        for iy in range(half):
            w = philox4x32(seed, (ix, iy, stream, 0))
            g0, g1 = normal_pair(w[0], w[1]), normal_pair(w[2], w[3])
            if not edge:
                H[iy, ix] = amp[iy, ix] * complex(*g0) / np.sqrt(2)
                H[iy + half, ix] = amp[iy + half, ix] * complex(*g1) / np.sqrt(2)
            elif iy == 0:
                H[0, ix] = amp[0, ix] * g0[0]
                H[half, ix] = amp[half, ix] * g1[0]
            else:
                H[iy, ix] = amp[iy, ix] * complex(*g0) / np.sqrt(2)
                H[ny - iy, ix] = np.conj(H[iy, ix])
    return np.fft.irfft2(H, s=(ny, nx)) * (ny * nx) / np.sqrt((amp**2).sum())


def psd_amplitude_3d(nh, ny, nx, dh, dy, dx, r0, nu):
    kz = 2 * np.pi * np.fft.fftfreq(nh, dh)[:, None, None]
    ky = 2 * np.pi * np.fft.fftfreq(ny, dy)[None, :, None]
    kx = 2 * np.pi * np.fft.fftfreq(nx, dx)[None, None, :]
    return (2.0 * nu / r0**2 + kx**2 + ky**2 + kz**2) ** (-(nu + 1.5) / 2.0)


def hermitian_philox_screens_3d(philox4x32, seed, stream, nh, ny, nx, dh, dy, dx, r0, nu, plane_pos, plane_scale=None, amp=None):
    """The height planes mrx_screen_generate_3d defines (include/mrx.h), rebuilt on the host:
    independent cells on kx <= nx/2, inverse FFT along h, linear interpolation to the planes,
    Hermitian symmetrisation of the two self-mirrored columns, numpy's irfft2 per plane."""
    if amp is None:
        amp = psd_amplitude_3d(nh, ny, nx, dh, dy, dx, r0, nu)
    H = np.zeros((nh, ny, nx // 2 + 1), complex)

    def normal_pair(a, b):
        u1 = ((a >> 8) + 0.5) / 16777216.0
        u2 = (b >> 8) / 16777216.0
        rad = np.sqrt(-2 * np.log(u1))
        return complex(rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2))

    for ix in range(nx // 2 + 1):
        for iy in range(ny):
            for iz in range(nh // 2):
                w = philox4x32(seed, (ix, iy, (stream << 16) | iz, 0x33440000))
                H[iz, iy, ix] = amp[iz, iy, ix] * normal_pair(w[0], w[1]) / np.sqrt(2)
                H[iz + nh // 2, iy, ix] = amp[iz + nh // 2, iy, ix] * normal_pair(w[2], w[3]) / np.sqrt(2)
    S = np.fft.ifft(H, axis=0) * nh  # [h, ky, kx]
    norm = np.sqrt((amp**2).sum())
    half = ny // 2
    out = []
    for p, pos in enumerate(plane_pos):
        h0 = min(int(pos), nh - 2)
        w = pos - h0
        P = ((1 - w) * S[h0] + w * S[h0 + 1]) * (1.0 if plane_scale is None else plane_scale[p])
        for ix in (0, nx // 2):
            col = P[:, ix].copy()
            new = col.copy()
            for iy in range(1, half):
                new[iy] = (col[iy] + np.conj(col[ny - iy])) / np.sqrt(2)
                new[ny - iy] = np.conj(new[iy])
            new[0], new[half] = np.sqrt(2) * col[0].real, np.sqrt(2) * col[half].real
            P[:, ix] = new
        out.append(np.fft.irfft2(P, s=(ny, nx)) * (ny * nx) / norm)
    return out


def radial_covariance(screen, dy, dx, lags_px):
    """Empirical covariance of a periodic screen at integer-pixel lags along both
    axes (FFT autocorrelation); returns (r_metres, cov) for each axis."""
    f = np.fft.fft2(screen.astype(np.float64))
    acf = np.fft.ifft2(np.abs(f) ** 2).real / screen.size
    lags_px = np.asarray(lags_px)
    return (lags_px * dy, acf[lags_px, 0]), (lags_px * dx, acf[0, lags_px])
