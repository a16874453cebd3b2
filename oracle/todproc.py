"""TOD pre-processing restated for the tests.  TEST INFRASTRUCTURE ONLY.

Follows maria/tod/processing.py:91-204 (``process_tod``: operation order, dtypes of the
in-place numpy updates) and maria/utils/signal/__init__.py (``remove_slope`` :151-152,
``bspline_knots`` / ``bspline_basis`` :91-123, ``decompose`` :59-89) and
maria/utils/signal/filters.py:46-69 (Bessel low / high pass through ``scipy.signal.sosfilt``).
Pinned by tests/golden/leaves.json ("signal": outputs of the reference's own functions).
"""

from __future__ import annotations

import numpy as np
import scipy.interpolate
import scipy.signal
import scipy.sparse.linalg


def remove_slope(D):
    return D - np.linspace(D[..., 0], D[..., -1], D.shape[-1]).T


def bspline_basis(t, spacing, order=3):
    """The same basis through scipy's own B-spline evaluation on the reference's knots
    (utils/signal/__init__.py:91-104), an independent route to the Cox-de Boor recursion."""
    t = np.asarray(t, float)
    tmin, tmax = t.min(), t.max()
    n_bins = int(np.maximum((tmax - tmin) // spacing, 1))
    k = spacing * np.arange(n_bins, dtype=float)
    k += float(tmax + tmin) / 2 - k.mean()
    k = np.r_[k[0] + spacing * np.arange(-order - 1, 0), k, k[-1] + spacing * np.arange(1, order + 2)]
    n_basis = len(k) - order - 1
    out = np.zeros((n_basis, len(t)))
    for i in range(n_basis):
        c = np.zeros(n_basis)
        c[i] = 1.0
        out[i] = scipy.interpolate.BSpline(k, c, order, extrapolate=False)(t)
    return np.nan_to_num(out)


def bessel(D, fc, sample_rate, order, btype):
    sos = scipy.signal.bessel(2 * (order + 1), 2 * fc / sample_rate, analog=False, btype=btype, output="sos")
    return scipy.signal.sosfilt(sos, D, axis=-1)


def remove_modes(D, m, k=None):
    """decompose + the subtraction of processing.py:178-186."""
    k = k or m + 1
    dnorm = np.sqrt(np.sum(np.square(D), axis=-1))
    dnorm = np.where(dnorm > 0, dnorm, 1)
    u, s, v = scipy.sparse.linalg.svds(D / dnorm[..., None], k=k)
    order = np.argsort(-s)
    u, s, v = u[:, order], s[order], v[order]
    return D - (dnorm[:, None] * u[:, :m] * s[:m]) @ v[:m]


def process_tod(signal, t, el, config):
    """tod/processing.py:91-204 on a float32 [D, T] signal; returns (float32 total, weight)."""
    D = np.array(signal, np.float32)
    W = np.ones(D.shape[-1])
    sample_rate = 1.0 / np.mean(np.diff(t))
    if "remove_slope" in config:
        D -= np.linspace(D[..., 0], D[..., -1], D.shape[-1]).T
    if "remove_spline" in config:
        sub = config["remove_spline"]
        B = bspline_basis(t, spacing=sub["knot_spacing"], order=sub.get("order", 3))
        eo = sub.get("remove_el_gradient_order", 2 if sub.get("remove_el_gradient", False) else None)
        if eo is not None:
            rel = (el - el.min()) / np.ptp(el)
            B = np.concatenate([B * rel**i for i in range(eo + 1)], axis=0)
        A = (np.linalg.inv(B @ B.T) @ B @ D.swapaxes(-2, -1)).swapaxes(-2, -1)
        D -= A @ B
    if "window" in config:
        w = getattr(scipy.signal.windows, config["window"]["name"])(D.shape[-1], **config["window"].get("kwargs", {}))
        W = W * w
        D *= W
    if "filter" in config:
        sub = config["filter"]
        D = remove_slope(D)
        if "f_upper" in sub:
            D = bessel(D, sub["f_upper"], sample_rate, sub.get("order", 1), "low")
        if "f_lower" in sub:
            D = bessel(D, sub["f_lower"], sample_rate, sub.get("order", 1), "high")
    if "remove_modes" in config:
        D = remove_modes(D, config["remove_modes"]["modes_to_remove"])
    return D.astype(np.float32), W
