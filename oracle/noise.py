"""The reference's detector-noise model.  TEST INFRASTRUCTURE ONLY.

Restates maria/noise/generation.py:11-51 (``generate_noise_with_knee``),
maria/utils/linalg.py:105-126 (``generate_spatial_basis``) and the band loop of
maria/sim/noise.py:18-63.  The reference draws the pink noise's white input from
``jax.random.normal(jax.random.key(12345))`` -- the same key on every call; jax is not
available here, so numpy's generator stands in: the oracle is a statistical target
(spectrum, variances, cross-detector covariance), not a sample-by-sample one.
``generate_spatial_basis`` is pinned by tests/golden/leaves.json (utils/linalg.py imports on
its own); ``generate_noise_with_knee`` is parity unpinned (noise/generation.py needs jax).
"""

from __future__ import annotations

import numpy as np
import scipy as sp
import scipy.interpolate


def matern_five_halves(r):
    """functions/__init__.py:26-27."""
    return (1 + np.sqrt(3) * r + (5.0 / 3.0) * r**2) * np.exp(-np.sqrt(5) * r)


def generate_spatial_basis(offsets, k: int = 5, n_side: int = 8, scale: float = 1):
    """utils/linalg.py:105-126."""
    x = np.linspace(offsets[..., 0].min(), offsets[..., 0].max(), n_side)
    y = np.linspace(offsets[..., 1].min(), offsets[..., 1].max(), n_side)
    X, Y = np.meshgrid(x, y)
    sample_offsets = np.stack([X.ravel(), Y.ravel()], axis=-1)
    D_eff = np.sqrt(np.square(sample_offsets - sample_offsets[:, None]).sum(axis=-1)) / scale
    C = matern_five_halves(D_eff)
    u, s, v = np.linalg.svd(C)
    basis = u[:, :k] * np.sqrt(s[:k])
    B = sp.interpolate.RegularGridInterpolator((x, y), basis.reshape(n_side, n_side, -1), method="cubic")(offsets)
    B *= np.sign(B[:, 0].mean())
    return B


def generate_noise_with_knee(shape, sample_rate=1.0, knee=0.0, beta=1.0, basis=None, corr_prop=0.0, rng=None):
    """noise/generation.py:11-51 with an explicit numpy generator for every draw."""
    rng = rng or np.random.default_rng()
    noise = np.sqrt(sample_rate) * rng.standard_normal(shape)
    if knee > 0:
        f = np.fft.fftfreq(n=shape[-1], d=1 / sample_rate)
        a = knee / 2
        with np.errstate(divide="ignore", invalid="ignore"):
            pink_noise_power_spectrum = np.where(f != 0, a / (np.abs(f) ** beta), 0)
        weights = np.sqrt(2 * sample_rate * pink_noise_power_spectrum)
        pink_noise = np.real(np.fft.ifft(weights * np.fft.fft(rng.standard_normal(shape))))
        if basis is not None:
            noise_modes = generate_noise_with_knee((basis.shape[-1], shape[-1]), sample_rate=sample_rate, knee=knee, rng=rng)
            pink_noise = np.sqrt(corr_prop) * basis @ noise_modes + np.sqrt(1 - corr_prop) * pink_noise
        noise += pink_noise
    return noise


def one_sided_psd_model(f, sample_rate, knee, scale=1.0):
    """One-sided PSD of scale * generate_noise_with_knee(...) without the correlated part:
    white level 2 scale^2 (variance sample_rate over a band sample_rate/2), pink 2 scale^2 knee/f."""
    return 2.0 * scale**2 * (1.0 + knee / np.asarray(f, float))
