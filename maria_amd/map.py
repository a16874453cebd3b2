"""Sky maps: beam smoothing (``ProjectionMap.smooth``, map/projection.py:485-504) and sampling
into the TOD (``MapMixin._sample_maps``, sim/map.py:76-172)."""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import Context, ptr


def smooth(data, weight=None, sigma=None, fwhm=None, x_res=1.0, y_res=1.0, device="cuda:0", ctx=None):
    """``numer = G(data*weight); denom = G(weight); data = where(denom > 0, numer/denom, 0)``
    over the last two axes, sigma in the map's angular units, resolutions per pixel.

    ``data`` (and ``weight``): [..., ny, nx] numpy arrays or device tensors.  Returns
    ``(smoothed, denom)`` of the input kind; ``denom`` is the reference's new weight.
    """
    if not (sigma is None) ^ (fwhm is None):
        raise ValueError("You must supply exactly one of 'sigma' or 'fwhm'.")
    sigma = sigma if sigma is not None else fwhm / np.sqrt(8 * np.log(2))
    x_sigma_pixels, y_sigma_pixels = abs(sigma / x_res), abs(sigma / y_res)
    as_numpy = not isinstance(data, torch.Tensor)
    dev = torch.device(device)
    d = torch.as_tensor(np.ascontiguousarray(data, np.float32)).to(dev) if as_numpy else data.to(dev, torch.float32).contiguous()
    w = None
    if weight is not None:
        w = torch.as_tensor(np.array(np.broadcast_to(weight, data.shape), dtype=np.float32)).to(dev) if as_numpy else weight.to(dev, torch.float32).expand_as(d).contiguous()
    torch.cuda.set_device(dev)
    ctx = ctx or Context(dev.index or 0)
    ctx.set_stream(torch.cuda.current_stream(dev))
    ny, nx = d.shape[-2:]
    out, den = torch.empty_like(d), torch.empty_like(d)
    tmp = torch.empty(2 * ny * nx, dtype=torch.float32, device=dev)
    flat_d, flat_o, flat_n = d.reshape(-1, ny, nx), out.reshape(-1, ny, nx), den.reshape(-1, ny, nx)
    flat_w = None if w is None else w.reshape(-1, ny, nx)
    for k in range(flat_d.shape[0]):
        ctx.call(
            "mrx_map_smooth", ptr(flat_d[k]), ptr(None if flat_w is None else flat_w[k]), ptr(flat_o[k]),
            ptr(flat_n[k]), ptr(tmp), ny, nx, float(y_sigma_pixels), float(x_sigma_pixels),
        )
    if as_numpy:
        return out.cpu().numpy(), den.cpu().numpy()
    return out, den


def mueller_row(gamma):
    """array/array.py:204-221: row 0 of each detector's Mueller matrix, [D, 4] (I, Q, U, V)."""
    a = np.asarray(gamma, float)
    m = np.stack([np.where(np.isnan(a), np.sqrt(2), 1.0), np.where(np.isnan(a), 0.0, np.cos(2 * a)),
                  np.where(np.isnan(a), 0.0, np.sin(2 * a)), np.zeros_like(a)], axis=1)
    return 0.5 * m[:, :1] * m


def collapse_temperature(table, axis_T, base_temperature):
    """Host part of the channel calibration (band/band.py:235-255): the (T, pwv, el) grid of a
    channel's transmission integral collapsed at the scalar base temperature with jax's float32
    index/weight rule -> [npwv, nel] float32 (NaN when the temperature is off the grid)."""
    g = np.asarray(axis_T, np.float32)
    x = np.float32(base_temperature)
    i = min(max(int(np.searchsorted(g, x, side="left")) - 1, 0), len(g) - 2)
    w = np.float32((x - g[i]) / (g[i + 1] - g[i]))
    v = np.asarray(table, np.float32)
    out = (np.float32(1) - w) * v[i] + w * v[i + 1]
    if x < g[0] or x > g[-1]:
        out = np.full_like(out, np.nan)
    return out.astype(np.float32)


def steps_per_tile(t, dta, tile=1024):
    """``mrx_map_cal.steps_per_tile``: the most steps of a coarse series of step ``dta`` that ``tile`` consecutive samples
    (and the two either side) of the sample times ``t`` meet -- what sizes the LDS table of the map sampler's per-sample
    calibration (csrc/mrx_map.hip, the interval form)."""
    t = np.asarray(t, float)
    if len(t) < 2:
        return 1
    span = float(np.max(t[min(tile + 1, len(t) - 1):] - t[: len(t) - min(tile + 1, len(t) - 1)]))
    return int(np.ceil(span / float(dta))) + 1


def sample_map(ctx, values, eta, xi, center, az, el, offsets, stokes_weights, out=None, transform=None, bilinear=True,
               cal_tables=None, cal_axis_pwv=None, cal_axis_el=None, coarse_pwv=None, ta0=0.0, dta=1.0, t=None,
               cal_scalars=None, device="cuda:0", sync=True, krj=None, scale=None, steps_per_tile=None):
    """``mrx_map_sample`` for the detectors of one band (sim/map.py:76-172).

    values [C, S, n_eta, n_xi] K_RJ (smoothed, parity applied); eta, xi the map axes in radians;
    center (phi, theta) rad; az, el the full-rate boresight [T]; offsets [D, 2] rad; stokes_weights
    [D, S]; transform [T, 3, 3] float64 or None (az/el-frame map).  Calibration: either
    ``cal_tables`` [C, npwv, nel] (collapsed at the base temperature) with the pwv / elevation axes,
    the coarse zenith-scaled pwv [Ta, D] (device tensor or array), its time grid and the sample
    times ``t``; or ``cal_scalars`` [C].  Arrays or device tensors throughout (tensors of the right type are used as
    they are).  ``sync=False``: return without waiting for the kernel -- for callers whose inputs outlive the launch
    or live on the launch's stream (torch's allocator then orders their reuse).  Returns the [D, T] float32 device
    tensor in pW.

    ``krj``: the field in K_RJ instead (``mrx_map_sample_krj``: ``TOD.to("K_RJ")`` on the sampler's store, the same bits
    as ``mrx_tod_to_krj`` of the pW field) -- a dict of device tensors for THESE rows: ``bore_el`` [T], ``dx``, ``dy``
    [D] float32, ``band`` [D] int32, ``axis`` [n_el], ``values`` [n_bands, n_el] float32 (``DevicePath.krj_row_tables()``,
    its per-row arrays indexed by the band's rows); ``scale`` [D] float32: the gain error, multiplied in before the
    division (with ``krj`` only).

    ``steps_per_tile`` (with ``cal_tables``): ``mrx_map_cal.steps_per_tile`` -- the most coarse steps of the pwv series
    that 1024 consecutive samples meet; worked out from ``t`` and ``dta`` when ``t`` is an array on the host, the
    library's default for a device tensor (:func:`steps_per_tile`)."""
    from ._lib import MrxMapCal, MrxSkyMap

    dev = torch.device(device)
    f32 = lambda a: a.to(dev, torch.float32).contiguous() if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    f64 = lambda a: a.to(dev, torch.float64).contiguous() if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(dev)  # noqa: E731
    values = np.asarray(values) if not isinstance(values, torch.Tensor) else values
    C_, S_, n_eta, n_xi = values.shape
    D, T = len(offsets), len(az)
    eta, xi = np.asarray(eta, float), np.asarray(xi, float)
    deta, dxi = (eta[-1] - eta[0]) / (n_eta - 1), (xi[-1] - xi[0]) / (n_xi - 1)
    if not (np.allclose(np.diff(eta), deta, rtol=1e-6, atol=0) and np.allclose(np.diff(xi), dxi, rtol=1e-6, atol=0)):
        raise ValueError("map axes must be uniform (np.linspace, map/projection.py:122-123)")
    keep = dict(values=f32(values), az=f32(az), el=f32(el), dx=f32(offsets[:, 0]), dy=f32(offsets[:, 1]),
                w=f64(stokes_weights[:, :S_]))
    sky = MrxSkyMap(ptr(keep["values"]), C_, S_, n_eta, n_xi, float(eta[0]), float(deta), float(xi[0]), float(dxi),
                    float(center[0]), float(center[1]), 1 if bilinear else 0, 0)
    cal = MrxMapCal()
    if cal_tables is not None:
        keep.update(tab=f32(cal_tables), ap=f32(cal_axis_pwv), ae=f32(cal_axis_el), pwv=f64(coarse_pwv), t=f64(t))
        assert tuple(keep["tab"].shape) == (C_, len(cal_axis_pwv), len(cal_axis_el)) and keep["pwv"].shape[1] == D
        cal.d_table, cal.d_axis_pwv, cal.d_axis_el = ptr(keep["tab"]), ptr(keep["ap"]), ptr(keep["ae"])
        cal.n_pwv, cal.n_el = len(cal_axis_pwv), len(cal_axis_el)
        cal.d_pwv, cal.Ta, cal.ta0, cal.dta, cal.d_t = ptr(keep["pwv"]), keep["pwv"].shape[0], float(ta0), float(dta), ptr(keep["t"])
        if steps_per_tile is None:
            steps_per_tile = 0 if isinstance(t, torch.Tensor) else globals()["steps_per_tile"](t, dta)
        cal.steps_per_tile = int(steps_per_tile)
    else:
        keep.update(sc=f64(np.atleast_1d(cal_scalars)))
        assert keep["sc"].shape == (C_,)
        cal.d_scalar = ptr(keep["sc"])
    if transform is not None:
        keep["tr"] = f64(np.asarray(transform).reshape(T, 9) if not isinstance(transform, torch.Tensor) else transform.reshape(T, 9))
    if out is None:
        out = torch.empty((D, T), dtype=torch.float32, device=dev)
    if krj is None:
        if scale is not None:
            raise ValueError("scale goes with krj (the pW field takes its gain from the caller)")
        ctx.call("mrx_map_sample", C.byref(sky), C.byref(cal), ptr(keep["az"]), ptr(keep["el"]), T, ptr(keep.get("tr")),
                 ptr(keep["dx"]), ptr(keep["dy"]), ptr(keep["w"]), D, ptr(out), out.stride(0))
    else:
        keep.update(k_el=f32(krj["bore_el"]), k_dx=f32(krj["dx"]), k_dy=f32(krj["dy"]), k_axis=f32(krj["axis"]), k_values=f32(krj["values"]),
                    k_band=krj["band"].to(dev, torch.int32).contiguous(), k_scale=None if scale is None else f32(scale))
        n_el = int(keep["k_axis"].numel())
        n_bands = int(keep["k_values"].numel()) // n_el
        assert keep["k_el"].numel() == T and keep["k_dx"].numel() == D and keep["k_dy"].numel() == D and keep["k_band"].numel() == D
        assert keep["k_scale"] is None or keep["k_scale"].numel() == D
        ctx.call("mrx_map_sample_krj", C.byref(sky), C.byref(cal), ptr(keep["az"]), ptr(keep["el"]), T, ptr(keep.get("tr")),
                 ptr(keep["dx"]), ptr(keep["dy"]), ptr(keep["w"]), D, ptr(keep["k_scale"]), ptr(keep["k_el"]), ptr(keep["k_dx"]),
                 ptr(keep["k_dy"]), ptr(keep["k_band"]), ptr(keep["k_axis"]), ptr(keep["k_values"]), n_el, n_bands,
                 ptr(out), out.stride(0))
    if sync:
        torch.cuda.current_stream(dev).synchronize()  # the temporaries in `keep` may go once the kernel is done
    return out


class ProjectionMap:
    """The slice of ``maria.map.ProjectionMap`` map sampling reads (map/projection.py:37-133):
    ``data`` [stokes, nu, eta, xi] in K_RJ on a uniform tangent-plane grid around ``center``.

    ``data`` may be [n_eta, n_xi], [n_nu, n_eta, n_xi] or [n_stokes, n_nu, n_eta, n_xi]; ``nu``
    in Hz (one entry per channel), ``stokes`` a string out of "IQUV"; ``width`` / ``height`` /
    ``resolution`` and ``center`` in degrees (``degrees=True``) or radians; ``frame`` "ra/dec"
    or "az/el".  Row i of the data sits at eta[i], eta ascending as given -- the reference then
    flips axis and data together (``apply_parity``), which changes nothing for sampling and is
    mirrored here so that the device sees the same descending axis.  Units other than K_RJ, the
    time dimension, FITS/HDF input and plotting stay with maria's front end."""

    def __init__(self, data, nu=None, stokes=None, width=None, height=None, resolution=None, center=(0.0, 0.0),
                 frame="ra/dec", degrees=True, units="K_RJ"):
        if units != "K_RJ":
            raise NotImplementedError("maps are sampled in K_RJ; other map units go through maria's calibration graph")
        if frame not in ("ra/dec", "az/el"):
            raise NotImplementedError(f"frame '{frame}': only 'ra/dec' and 'az/el' are built")
        data = np.asarray(data, np.float32)
        while data.ndim < 4:
            data = data[None]
        if data.ndim != 4:
            raise ValueError("data must be [eta, xi], [nu, eta, xi] or [stokes, nu, eta, xi]")
        self.stokes = stokes or "I"
        self.nu = np.atleast_1d(np.asarray(150e9 if nu is None else nu, float))
        if data.shape[0] != len(self.stokes) or data.shape[1] != len(self.nu):
            raise ValueError(f"data of shape {data.shape} does not match stokes '{self.stokes}' and {len(self.nu)} channel(s)")
        unit = np.pi / 180 if degrees else 1.0
        n_eta, n_xi = data.shape[-2:]
        if all(v is None for v in (width, height, resolution)):
            raise ValueError("You must pass at least one of 'width', 'height', 'resolution'.")
        xi_res = eta_res = None  # projection.py:104-123
        if width is not None:
            xi_res = width / (n_xi - 1)
            eta_res = xi_res if height is None else None
        if height is not None:
            eta_res = height / (n_eta - 1)
            xi_res = abs(eta_res) if width is None else xi_res
        if resolution is not None:
            xi_res = eta_res = resolution
        self.xi = unit * xi_res * (n_xi - 1) * np.linspace(-0.5, 0.5, n_xi)
        self.eta = unit * eta_res * (n_eta - 1) * np.linspace(-0.5, 0.5, n_eta)
        self.x_res, self.y_res = unit * xi_res, unit * eta_res
        # the parity convention of projection.py:128-130: eta descending
        self.eta, self.data = self.eta[::-1].copy(), data[:, :, ::-1].copy()
        self.center = (unit * center[0], unit * center[1])
        self.frame, self.units = frame, units

    @property
    def nu_bin_bounds(self):
        """map/base.py:452-454."""
        edges = [0.0, *((self.nu[:-1] + self.nu[1:]) / 2), np.inf]
        return list(zip(edges[:-1], edges[1:]))

    def smooth(self, fwhm, ctx=None, device="cuda:0"):
        """map/projection.py:485-504 on the device, uniform weights; fwhm in radians."""
        if fwhm <= 0:
            return self.data
        out, _ = smooth(self.data, fwhm=fwhm, x_res=self.x_res, y_res=self.y_res, device=device, ctx=ctx)
        return out
