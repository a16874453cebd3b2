"""Matern covariances and beam widths.  TEST INFRASTRUCTURE ONLY.

Restates maria/functions/__init__.py:30-74 and maria/beam/__init__.py:9-29.
PINNED: tests/test_oracle_golden.py checks these against tests/golden/leaves.json,
produced by importing the reference's own modules (oracle/gen_golden.py).
"""

from __future__ import annotations

import numpy as np
import scipy.special

C_LIGHT = 299792458.0  # maria/constants.py
K_B = 1.380649e-23


def normalized_matern(r, nu):
    """functions/__init__.py:30-39."""
    z = np.sqrt(2 * nu) * r + 1e-16
    return 2 ** (1 - nu) / scipy.special.gamma(nu) * scipy.special.kv(nu, z) * z**nu


def approximate_normalized_matern(r, nu=1 / 3, r0=1e0, n_test_points=1024):
    """functions/__init__.py:42-74: log-log table of the exact correlation, blended with
    the structure function so that both ends keep their precision."""
    r = np.asarray(r, float)
    r_eff = r / r0
    r_eff_min, r_eff_max = 1e-6, 1e3
    r_eff_safe = np.atleast_1d(np.abs(r_eff)).clip(min=r_eff_min)
    nonzero = r_eff_safe[r_eff_safe < r_eff_max]
    samples = np.geomspace(r_eff_min, r_eff_max, n_test_points)
    cov_samples = normalized_matern(samples, nu=nu)
    with np.errstate(divide="ignore"):
        sf = np.exp(np.interp(np.log(nonzero), np.log(samples), np.log(1 - cov_samples)))
        cov = np.exp(np.interp(np.log(nonzero), np.log(samples), np.log(cov_samples)))
    t = 1 / (1 + nonzero**2)
    res = np.zeros(r.shape)
    res[r_eff_safe < r_eff_max] = t * (1 - sf) + (1 - t) * cov
    return res


def compute_angular_fwhm(fwhm_0, z=np.inf, n=1.0, nu=None, l=None):  # noqa: E741
    """beam/__init__.py:9-25."""
    if nu is None and l is None:
        raise ValueError("You must supply either a frequency 'f' or wavelength 'l'.")
    w_0 = fwhm_0 / 2
    z_r = np.pi * w_0**2 * n / (l or C_LIGHT / nu)
    return 2 * w_0 * np.sqrt(1 / z**2 + 1 / z_r**2)


def compute_physical_fwhm(fwhm_0, z=np.inf, n=1, nu=None, l=None):  # noqa: E741
    """beam/__init__.py:28-29."""
    return z * compute_angular_fwhm(fwhm_0=fwhm_0, z=z, n=n, nu=nu, l=l)

H_PLANCK = 6.62607015e-34  # maria/constants.py


def rayleigh_jeans_spectrum(T_RJ, nu):
    """functions/radiometry.py:6-7."""
    return 2 * K_B * nu**2 * T_RJ / C_LIGHT**2


def planck_spectrum(T_b, nu):
    """functions/radiometry.py:14-15."""
    return 2 * H_PLANCK * nu**3 / (C_LIGHT**2 * np.expm1(H_PLANCK * nu / (K_B * T_b)))
