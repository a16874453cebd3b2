"""``process_tod``: the TOD pre-processing the mappers run before binning
(maria/tod/processing.py:91-204), device-backed.

The streaming operations -- ``remove_slope``, ``window`` and the Bessel ``filter`` (scipy's
``sosfilt`` made time-parallel) -- are HIP kernels in ``libmrx`` (csrc/mrx_tod.hip).  The two
GEMM-shaped ones are plain library products on the device: ``remove_spline`` is a least-squares
fit on a small B-spline basis, ``remove_modes`` a projection onto the leading singular vectors
(``torch.matmul`` / ``torch.linalg.eigh``, i.e. rocBLAS / rocSOLVER).  The signal stays a float32
``[D, T]`` device tensor throughout (the reference ends in float32 too; its float64
intermediates between ``filter`` and ``remove_modes`` are rounded once more here).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.signal
import torch

from ._lib import Context, ptr

# tod/processing.py:16-37: operation -> parameter -> (type, aliases of the keyword form)
OPERATION_KWARGS = {
    "remove_slope": {},
    "window": {"name": (str, ["window"]), "kwargs": (dict, ["window_kwargs"])},
    "filter": {"f_lower": (float, ["f_lower"]), "f_upper": (float, ["f_upper"]), "order": (int, ["filter_order"]),
               "method": (str, ["filter_method"])},
    "remove_modes": {"modes_to_remove": (int, ["modes_to_remove"])},
    "remove_spline": {"knot_spacing": (float, ["remove_spline_knot_spacing"]), "remove_el_gradient": (bool, ["remove_el_gradient"]),
                      "remove_el_gradient_order": (int, ["remove_el_gradient_order"]), "order": (int, ["depline_order"])},
}


def process_operation_kwargs(**kwargs):
    """tod/processing.py:40-60: the keyword form (``f_lower=...``) to the config form."""
    config = {}
    for operation, params in OPERATION_KWARGS.items():
        sub = {}
        for key, (_, aliases) in params.items():
            for kw in list(kwargs):
                if kw in aliases:
                    sub[key] = kwargs.pop(kw)
        if sub:
            config[operation] = sub
    if kwargs:
        raise ValueError(f"Invalid kwargs for TOD processing: {kwargs}.")
    return config


def validate_process_config(config):
    """tod/processing.py:63-88."""
    for operation, params in config.items():
        if operation not in OPERATION_KWARGS:
            raise ValueError(f"Invalid operation '{operation}'. Valid operations are {list(OPERATION_KWARGS)}")
        for key, value in params.items():
            if key not in OPERATION_KWARGS[operation]:
                raise ValueError(f"Invalid param '{key}' for operation '{operation}'. Valid parameters for this operation are "
                                 f"{list(OPERATION_KWARGS[operation])}")
            dtype = OPERATION_KWARGS[operation][key][0]
            if not isinstance(value, dtype):
                try:
                    config[operation][key] = dtype(value)
                except Exception:
                    raise TypeError(f"Could not convert param {{{key!r}: {value!r}}} for operation '{operation}' to requisite type "
                                    f"'{dtype.__name__}'.")
    return config


# ---- host pieces ------------------------------------------------------------------------------

def bspline_knots(t, spacing, order):
    """utils/signal/__init__.py:91-104: uniform knots straddling the data, padded by order + 1."""
    tmin, tmax = float(np.min(t)), float(np.max(t))
    n_bins = int(max((tmax - tmin) // spacing, 1))
    k = spacing * np.arange(n_bins, dtype=float)
    k += (tmax + tmin) / 2 - k.mean()
    return np.r_[k[0] + spacing * np.arange(-order - 1, 0), k, k[-1] + spacing * np.arange(1, order + 2)]


def bspline_basis(t, spacing, order=3):
    """utils/signal/__init__.py:107-123: the B-spline basis [n_basis, T] on those knots by the
    Cox-de Boor recursion, degree 0 being the indicator of np.digitize's bin."""
    t = np.asarray(t, float)
    k = bspline_knots(t, spacing, order)
    n_basis = len(k) - order - 1
    prev = np.zeros((len(k) + 1, len(t)))
    prev[np.digitize(t, k) - 1, np.arange(len(t))] = 1.0
    for p in range(1, order + 1):
        cur = np.zeros_like(prev)
        i = np.arange(len(k) - p - 1)
        left = (t[None, :] - k[i][:, None]) / (k[i + p] - k[i])[:, None]
        right = (k[i + p + 1][:, None] - t[None, :]) / (k[i + p + 1] - k[i + 1])[:, None]
        cur[i] = prev[i] * left + prev[i + 1] * right
        prev = cur
    return prev[:n_basis]


def bessel_sos(fc, sample_rate, order, btype):
    """utils/signal/filters.py:46-69."""
    return scipy.signal.bessel(2 * (order + 1), 2 * fc / sample_rate, analog=False, btype=btype, output="sos")


def chunk_matrix(sos, length):
    """State transition of the cascade over ``length`` samples with zero input: column j is the
    state after running the transposed-direct-form-II recursion from the unit state e_j
    (state order z0, z1 per section) -- what ``mrx_sosfilt`` chains its chunks with."""
    sos = np.asarray(sos, float)
    S = len(sos)
    b = sos[:, :3] / sos[:, 3:4]
    a = sos[:, 4:] / sos[:, 3:4]
    M = np.zeros((2 * S, 2 * S))
    for j in range(2 * S):
        z = np.zeros((S, 2))
        z[j // 2, j % 2] = 1.0
        for _ in range(length):
            x = 0.0
            for s in range(S):
                y = b[s, 0] * x + z[s, 0]
                z[s, 0] = b[s, 1] * x - a[s, 0] * y + z[s, 1]
                z[s, 1] = b[s, 2] * x - a[s, 1] * y
                x = y
        M[:, j] = z.ravel()
    return M


# ---- the pipeline --------------------------------------------------------------------------------

class ProcessedTOD:
    """What ``process_tod`` returns: the reference builds a TOD with the single field "total",
    the window as weight and the config attached (processing.py:191-204)."""

    def __init__(self, tod, total, weight, config):
        self.data = {"total": total}
        self.weight = weight
        self.dets, self.coords, self.units, self.metadata = tod.dets, tod.coords, tod.units, tod.metadata
        self.processing_config = config

    @property
    def fields(self):
        return ["total"]


def _signal(tod, device):
    total = None
    for field in tod.data.values():
        f = field if isinstance(field, torch.Tensor) else torch.as_tensor(np.asarray(field))
        f = f.to(device, torch.float32)
        total = f.clone() if total is None else total.add_(f)
    return total.contiguous()


def process_tod(tod, config=None, ctx=None, device="cuda:0", **kwargs):
    """tod/processing.py:91-204 on the device.  ``tod``: a ``maria_amd.sim.TOD`` (fields on the
    host or the device); returns a :class:`ProcessedTOD` whose ``data["total"]`` is a float32
    device tensor.  Operations run in the reference's order: remove_slope, remove_spline,
    window, filter, remove_modes."""
    config = validate_process_config(dict(config) if config else process_operation_kwargs(**kwargs))
    dev = torch.device(device)
    ctx = ctx or Context(dev.index or 0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    D = _signal(tod, dev)
    n_det, n_samp = D.shape
    t = np.asarray(tod.coords.t, float)
    sample_rate = 1.0 / np.mean(np.diff(t)) if n_samp > 1 else 1.0
    anchors = torch.empty(2 * n_det + 16, dtype=torch.float64, device=dev)
    weight = np.ones(n_samp)

    def check(name):
        if bool(torch.isnan(D).any()):
            raise ValueError(f"tod operation '{name}' introduced NaNs")

    if "remove_slope" in config:
        ctx.call("mrx_tod_detrend_window", ptr(D), D.stride(0), n_det, n_samp, 1, None, ptr(anchors))
        check("remove_slope")

    if "remove_spline" in config:
        sub = config["remove_spline"]
        B = bspline_basis(t, spacing=sub["knot_spacing"], order=sub.get("order", 3))
        if sub.get("remove_el_gradient", False) and "remove_el_gradient_order" not in sub:
            sub["remove_el_gradient_order"] = 2
        if "remove_el_gradient_order" in sub:
            el = np.asarray(tod.coords._bel, float)
            if np.ptp(el) == 0:
                raise ValueError("Cannot remove elevation gradient when elevation is constant")
            rel = (el - el.min()) / np.ptp(el)
            B = np.concatenate([B * rel**i for i in range(sub["remove_el_gradient_order"] + 1)], axis=0)
        proj = torch.as_tensor(np.linalg.inv(B @ B.T) @ B).to(dev)  # [nb, T] float64
        Bd = torch.as_tensor(B).to(dev)
        rows = max(1, int(2e9 // (8 * n_samp)))  # float64 staging of ~2 GB of rows at a time
        for lo in range(0, n_det, rows):
            blk = D[lo : lo + rows].to(torch.float64)
            A = blk @ proj.T                      # (inv(B B^T) B D^T)^T
            D[lo : lo + rows] = (blk - A @ Bd).to(torch.float32)
        check("remove_spline")

    if "window" in config:
        w = getattr(scipy.signal.windows, config["window"]["name"])(n_samp, **config["window"].get("kwargs", {}))
        weight = weight * w
        d_w = torch.as_tensor(np.ascontiguousarray(w, np.float64)).to(dev)
        ctx.call("mrx_tod_detrend_window", ptr(D), D.stride(0), n_det, n_samp, 0, ptr(d_w), ptr(anchors))
        check("window")

    if "filter" in config:
        sub = config["filter"]
        order = sub.get("order", 1)
        sections = []
        if "f_upper" in sub:
            sections.append(bessel_sos(sub["f_upper"], sample_rate, order, "low"))
        if "f_lower" in sub:
            sections.append(bessel_sos(sub["f_lower"], sample_rate, order, "high"))
        if sections:
            sos = np.ascontiguousarray(np.concatenate(sections, axis=0), np.float64)
            chunk = ctx.lib.mrx_sosfilt_chunk()
            M = torch.as_tensor(np.ascontiguousarray(chunk_matrix(sos, chunk))).to(dev)
            need = C.c_size_t()
            ctx.lib.mrx_sosfilt_work_doubles(n_det, n_samp, len(sos), C.byref(need))
            work = torch.empty(need.value, dtype=torch.float64, device=dev)
            ctx.call("mrx_sosfilt", sos.ctypes.data_as(C.POINTER(C.c_double)), len(sos), ptr(M), ptr(D), D.stride(0), n_det, n_samp,
                     1, ptr(D), D.stride(0), ptr(work))
        else:  # the reference removes the slope before looking for filters (processing.py:151)
            ctx.call("mrx_tod_detrend_window", ptr(D), D.stride(0), n_det, n_samp, 1, None, ptr(anchors))
        check("filter")

    if "remove_modes" in config:
        m = config["remove_modes"]["modes_to_remove"]
        if not isinstance(m, int):
            raise TypeError("'modes_to_remove' must be an integer.")
        if m > 0:
            # utils/signal/__init__.py:59-89: svds of D / |D_row|; removing the first m modes is
            # D -= |D_row| U_m U_m^T (D / |D_row|), U_m the leading left singular vectors
            D64 = D.to(torch.float64) if n_det * n_samp <= 2.5e8 else None
            src = D64 if D64 is not None else D
            dnorm = torch.sqrt((src.to(torch.float64) ** 2).sum(dim=1)) if D64 is None else torch.sqrt((D64 * D64).sum(dim=1))
            dnorm = torch.where(dnorm > 0, dnorm, torch.ones_like(dnorm))
            Dn = (src / dnorm[:, None].to(src.dtype))
            G = (Dn @ Dn.T).to(torch.float64)
            _, vecs = torch.linalg.eigh(G)
            U = vecs[:, -m:].to(Dn.dtype)  # leading eigenvectors of Dn Dn^T = left singular vectors
            D.sub_(((dnorm[:, None].to(Dn.dtype) * U) @ (U.T @ Dn)).to(torch.float32))
        check("remove_modes")

    return ProcessedTOD(tod, D, weight, config)
