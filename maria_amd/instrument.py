"""Light-weight detector and band tables: the slice of maria's ``Array`` / ``Band``
objects the atmosphere path reads (SURVEY section 2, rows ``maria/array`` and
``maria/band``).  Host-side numpy; configuration parsing, noise levels and focal
plane generation from YAML stay with maria's own front end.
"""

from __future__ import annotations

import numpy as np
import scipy.spatial

C_LIGHT = 299792458.0  # maria/constants.py
K_B = 1.380649e-23


# The passband shapes of band/band.py:62-86, all of one family: tau = 2^-(|2 (nu - center) / width|^p) has half its
# peak at center +- width / 2 for every p -- a Gaussian of that FWHM (p = 2), a flat-topped curve (p = 8), and in the limit
# p -> infinity the boxcar of that width.  Per shape: (half of the sampled range in units of the width, p).
PASSBAND_SHAPES = {"gaussian": (1.5, 2.0), "top_hat": (1.0, 8.0), "flat": (0.6, np.inf)}


def generate_passband(center, width, shape, samples=256):
    """``samples`` frequencies across the band and its transmission there (band/band.py:62-86)."""
    if shape not in PASSBAND_SHAPES:
        raise ValueError(f"Invalid shape '{shape}'")
    half_range, power = PASSBAND_SHAPES[shape]
    nu = np.linspace(center - half_range * width, center + half_range * width, samples)
    x = np.abs(2.0 * (nu - center) / width)
    tau = np.where(x < 1.0, 1, 0) if np.isinf(power) else np.exp(np.log(0.5) * x**power)
    if np.trapezoid(tau, x=nu) < 1e-2 * 2.0 * half_range * width:  # (the reference's sanity check, :83-84)
        raise ValueError("Error generating band")
    return nu, tau


class Band:
    """band/band.py:89-160 (constructor) and :317-323 (``passband``)."""

    def __init__(self, center=None, width=None, nu=None, tau=None, name=None, shape="gaussian", efficiency=0.5, gain_error=0.0,
                 NEP=1e-17, NEP_per_loading=0.0, knee=1.0):
        auto = center is not None and width is not None
        manual = nu is not None and tau is not None
        if not auto ^ manual:
            raise ValueError("You must pass either both 'center' and 'width' or both 'nu' and 'tau'.")
        if auto:
            self.nu, self.tau = generate_passband(center, width, shape, samples=1024)
        else:
            tau = np.asarray(tau, float)
            tau_max = tau.max()
            efficiency *= tau_max
            self.nu, self.tau = np.asarray(nu, float), tau / tau_max
            if self.nu.ndim != 1 or self.nu.shape != self.tau.shape:
                raise ValueError(f"'nu' and 'tau' have mismatched shapes ({self.nu.shape} and {self.tau.shape}).")
        self.efficiency = efficiency
        self.gain_error = gain_error
        self.NEP, self.NEP_per_loading, self.knee = NEP, NEP_per_loading, knee  # W sqrt(s), -, Hz (band.py:103-108)
        self.shape = shape
        self.center = float(np.trapezoid(self.nu * self.tau, self.nu) / np.trapezoid(self.tau, self.nu)) if center is None else float(center)
        self.name = name or f"f{10 ** (np.log10(self.center) % 3):>03.0f}"

    def passband(self, nu):
        """band/band.py:317-323: efficiency x linear interpolation of tau, 0 outside."""
        return self.efficiency * np.interp(nu, self.nu, self.tau, left=0.0, right=0.0)

    def emission_table(self, spectrum):
        """band/band.py:272-280: 1e12 k_B trapezoid(T_RJ x passband, nu) on the spectrum's
        (T, pwv, el) grid, in pW."""
        return 1e12 * K_B * np.trapezoid(spectrum._emission * self.passband(spectrum.side_nu), spectrum.side_nu, axis=-1)


    def transmission_table(self, spectrum):
        """band/band.py:248-252: trapezoid(passband x exp(-opacity), nu) on the spectrum's
        (T, pwv, el) grid, the integral ``TOD.to("K_RJ")`` divides by."""
        return np.trapezoid(self.passband(spectrum.side_nu) * np.exp(-spectrum._opacity), x=spectrum.side_nu, axis=-1)


def compute_angular_fwhm(fwhm_0, z=np.inf, n=1.0, nu=None):
    """beam/__init__.py:9-25."""
    w_0 = fwhm_0 / 2
    z_r = np.pi * w_0**2 * n / (C_LIGHT / nu)
    with np.errstate(divide="ignore"):
        return 2 * w_0 * np.sqrt(1 / np.square(z) + 1 / z_r**2)


class Detectors:
    """The detector table (``instrument.dets``): offsets, band membership, beams."""

    def __init__(self, offsets, bands, band_index=None, primary_size=10.0, gamma=None):
        self.offsets = np.atleast_2d(np.asarray(offsets, float))
        self.bands = list(bands)
        self.n = len(self.offsets)
        self.band_index = np.zeros(self.n, np.int32) if band_index is None else np.asarray(band_index, np.int32)
        self.primary_size = np.broadcast_to(np.asarray(primary_size, float), (self.n,)).copy()
        self.gamma = np.full(self.n, np.nan) if gamma is None else np.asarray(gamma, float)
        if self.band_index.shape != (self.n,) or (self.n and (self.band_index.min() < 0 or self.band_index.max() >= len(self.bands))):
            raise ValueError("band_index must map every detector to one of the bands")

    @classmethod
    def hexagon(cls, n, field_of_view_deg, bands, primary_size=10.0):
        """``n`` positions x ``len(bands)`` bands, each band its own block of rows
        (array/array.py:496-502)."""
        from .synthetic import hex_pack

        pos = hex_pack(n, np.radians(field_of_view_deg))
        nb = len(bands)
        return cls(np.tile(pos, (nb, 1)), bands, np.repeat(np.arange(nb), n), primary_size)

    @property
    def band_name(self):
        return np.array([b.name for b in self.bands])[self.band_index]

    @property
    def band_center(self):
        return np.array([b.center for b in self.bands], float)[self.band_index]

    @property
    def field_of_view(self):
        """array/array.py:178-179 (diameter of the offsets), radians."""
        if self.n < 2:
            return 0.0
        pts = self.offsets
        try:
            pts = pts[scipy.spatial.ConvexHull(pts).vertices]
        except Exception:
            pass
        return float(scipy.spatial.distance.pdist(pts).max())

    def angular_fwhm(self, z=np.inf):
        """array/array.py:223-227, radians."""
        return compute_angular_fwhm(z=z, fwhm_0=self.primary_size, nu=self.band_center)

    def physical_fwhm(self, z):
        """array/array.py:229-233, metres."""
        return z * self.angular_fwhm(z)

    def mueller00(self):
        """array/array.py:204-218, element [0, 0]."""
        m0 = np.where(np.isnan(self.gamma), np.sqrt(2), 1.0)
        return 0.5 * m0 * m0

    def outer(self):
        """array/array.py:156-162: the convex-hull detectors."""
        try:
            idx = np.sort(scipy.spatial.ConvexHull(self.offsets).vertices)
        except Exception:
            return self
        return self.subset(idx)

    def one_detector_from_each_band(self):
        """array/array.py:149-154."""
        _, first = np.unique(self.band_index, return_index=True)
        return self.subset(np.sort(first))

    def subset(self, idx):
        return Detectors(self.offsets[idx], self.bands, self.band_index[idx], self.primary_size[idx], self.gamma[idx])

    def mask(self, band_name):
        return self.band_name == band_name


class Instrument:
    def __init__(self, dets, name="instrument"):
        self.dets = dets
        self.name = name


class Site:
    def __init__(self, altitude=0.0, region="synthetic", latitude=-23.0, longitude=-67.8):
        self.altitude = float(altitude)
        self.region = region
        self.latitude, self.longitude = float(latitude), float(longitude)  # degrees (site/site.py)
