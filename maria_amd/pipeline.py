"""Device-resident atmosphere -> TOD path: the host side above the C ABI.

``DevicePath`` owns the device copies of one observation's inputs (as torch
tensors: torch is the allocator and the stream owner, nothing else) and runs

    mrx_atm_sample -> mrx_spline_upsample_fused

for a contiguous block of detector rows.  The stages are the reference's
``Atmosphere.simulate_pwv`` (atmosphere/atmosphere.py:293-380) and
``AtmosphereMixin._compute_atmospheric_loading`` (sim/atmosphere.py:39-84).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import Context, MrxBandTable, MrxError, MrxLayer, ptr


class _range:
    """Profiler range (no-op if the marker library is unavailable)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        try:
            torch.cuda.nvtx.range_push(self.name)
            self.ok = True
        except Exception:
            self.ok = False

    def __exit__(self, *exc):
        if self.ok:
            torch.cuda.nvtx.range_pop()
        return False


def _dev(a, dtype, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(device)


def matern_log_tables(nu, n=8192, lo=1e-6, hi=1e3, eps=1e-10):
    """log of the exact Matern correlation (functions/__init__.py:30-39) and of its complement at ``n``
    log-spaced r / r0 -- the construction of the reference's ``approximate_normalized_matern`` (:42-74), eight
    times denser because the device interpolates it to 1e-11 (``mrx_screen_amplitudes``).  Returns
    (log_first, log_step, log_cov, log_sf, x_cut); values the float64 range cannot hold (rho underflows
    beyond r ~ 700 r0) are pinned at log(1e-300); ``x_cut``: the first node where the correlation has fallen
    below ``eps`` (periodic images farther away are not summed)."""
    import scipy.special

    x = np.geomspace(lo, hi, n)
    z = np.sqrt(2 * nu) * x + 1e-16
    cov = 2 ** (1 - nu) / scipy.special.gamma(nu) * scipy.special.kv(nu, z) * z**nu
    with np.errstate(divide="ignore"):
        log_cov = np.maximum(np.log(cov), np.log(1e-300))
        log_sf = np.maximum(np.log(1 - cov), np.log(1e-300))
    lx = np.log(x)
    x_cut = float(x[np.argmax(cov < eps)]) if (cov < eps).any() else float(hi)
    return float(lx[0]), float((lx[-1] - lx[0]) / (n - 1)), np.ascontiguousarray(log_cov), np.ascontiguousarray(log_sf), x_cut


def morton_order(offsets):
    """Permutation that sorts focal-plane offsets along a Z-order curve, so that
    consecutive detectors (the lanes of a wave, the 256 rows of a workgroup) form a
    compact patch on the sky: their lines of sight then hit the same few cache lines
    of every screen.  Pure host-side indexing; results do not depend on it."""
    off = np.asarray(offsets, float)
    if len(off) < 2:
        return np.arange(len(off))
    off = np.where(np.isfinite(off), off, 0.0)  # (a NaN offset sorts anywhere; its samples are NaN whatever its place)
    lo, span = off.min(axis=0), np.maximum(np.ptp(off, axis=0), 1e-300)
    q = np.minimum(((off - lo) / span * 65535.0).astype(np.uint64), 65535)

    def spread(v):  # 16 bits -> every other bit of 32
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555

    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1), kind="stable")


def table_slabs(table, T0):
    """Host part of the emission lookup: the two temperature slabs bracketing
    ``T0`` and T0's float32 normalised distance, computed exactly as jax's
    ``_find_indices`` would (band/band.py:283-286)."""
    Tg = np.asarray(table["T"], np.float32)
    x = np.float32(T0)
    i = int(np.searchsorted(Tg, x, side="left")) - 1
    i = min(max(i, 0), len(Tg) - 2)
    w = np.float32((x - Tg[i]) / (Tg[i + 1] - Tg[i]))
    oob = bool(x < Tg[0] or x > Tg[-1])
    vals = np.asarray(table["values"], np.float32)[i : i + 2]
    return np.ascontiguousarray(vals), w, oob


def table_cubic_cells(table, T0):
    """Host part of ``interpolation_method="cubic"`` (band/band.py:288-300): the band table
    interpolated linearly to ``T0`` (scipy ``interp1d``, which raises ValueError outside the
    temperature axis, as in the reference) and scipy's tensor-product not-a-knot cubic spline on
    (pwv, el) expanded into one bicubic per grid cell (Taylor coefficients at the cell's lower
    corner).  Returns the float64 buffer of ``mrx_band_table.d_cubic``:
    [pwv nodes][el nodes][cells][16]."""
    import scipy.interpolate

    Tg, x, y = (np.asarray(table[k], float) for k in ("T", "pwv", "el"))
    V = scipy.interpolate.interp1d(Tg, np.asarray(table["values"], float), kind="linear", axis=0)(T0)  # [n_pwv, n_el]
    if len(x) < 4 or len(y) < 4:
        raise ValueError("cubic interpolation needs at least 4 nodes per axis")
    fact = (1.0, 1.0, 2.0, 6.0)
    # The reference calls scipy's RegularGridInterpolator(method="cubic").  From scipy 1.13 on that
    # is an NdBSpline whose coefficients come from an ITERATIVE solver (gcrotmk, atol 1e-6): it
    # differs from the exact tensor-product spline by ~6e-6 of the table's scale.  To reproduce the
    # reference and not the textbook, the cells are expanded from scipy's own spline object; older
    # scipy (recursive 1-D splines, exact) and the fallback below give the exact tensor spline.
    spline = getattr(scipy.interpolate.RegularGridInterpolator((x, y), V, method="cubic"), "_spline", None)
    if spline is not None and hasattr(spline, "t"):
        X0, Y0 = np.meshgrid(x[:-1], y[:-1], indexing="ij")
        pts = np.stack([X0.ravel(), Y0.ravel()], axis=-1)
        Cc = np.empty((len(x) - 1, len(y) - 1, 4, 4))
        for k in range(4):
            for m in range(4):
                Cc[:, :, k, m] = spline(pts, nu=(m, k)).reshape(len(x) - 1, len(y) - 1) / (fact[m] * fact[k])
    else:  # separable operator: Taylor coefficients in el of every pwv row's spline, then along pwv
        sy = scipy.interpolate.make_interp_spline(y, V, k=3, axis=1)
        A = np.stack([sy.derivative(k)(y[:-1]) / fact[k] if k else sy(y[:-1]) for k in range(4)], axis=-1)  # [n_pwv, n_el-1, 4]
        sx = scipy.interpolate.make_interp_spline(x, A, k=3, axis=0)
        Cc = np.stack([sx.derivative(m)(x[:-1]) / fact[m] if m else sx(x[:-1]) for m in range(4)], axis=-1)  # [.., 4(k), 4(m)]
    return np.concatenate([x, y, np.ascontiguousarray(Cc).reshape(-1)])


class DevicePath:
    """One observation (or one detector shard of it) on one GPU."""

    def __init__(self, problem, device="cuda:0", det_slice=None, ctx=None, keep_pwv=False, locality_sort=True):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("maria_amd runs on a gfx950 GPU only; there is no CPU path")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        torch.cuda.set_device(index)
        self.ctx = ctx or Context(index)
        self.ctx.set_stream(torch.cuda.current_stream(self.device))
        self.problem = problem
        method = problem.get("interpolation_method", "linear")
        if method not in ("linear", "cubic"):
            raise ValueError(f"interpolation_method must be 'linear' or 'cubic', not {method!r}")
        self.cubic = method == "cubic"
        sl = det_slice or slice(0, len(problem["offsets"]))
        self.det_slice = sl
        dev = self.device

        off = np.asarray(problem["offsets"], float)[sl]
        self.D = int(off.shape[0])
        # internal detector order (Z-order on the focal plane); `order[k]` is the
        # caller's row of internal detector k.  The coarse arrays live in internal
        # order; the TOD is written straight into the caller's rows (d_rows).
        self.order = morton_order(off) if locality_sort else np.arange(self.D)
        self.inverse = np.argsort(self.order)
        off = off[self.order]
        pick = lambda a: np.asarray(a)[sl][self.order]  # noqa: E731
        self.Ta = int(len(problem["ta"]))
        self.T = int(len(problem["t"]))
        # jax demotes the float64 offsets and boresight to float32 on entry
        self.d_dx = _dev(off[:, 0], torch.float32, dev)
        self.d_dy = _dev(off[:, 1], torch.float32, dev)
        self.d_az = _dev(problem["az_a"], torch.float32, dev)
        self.d_el = _dev(problem["el_a"], torch.float32, dev)
        self.d_band = _dev(pick(problem["band_index"]), torch.int32, dev)
        self.d_m00 = _dev(pick(problem["m00"]), torch.float32, dev)
        self.d_rows = _dev(self.order, torch.int32, dev) if locality_sort else None
        self._d_inverse = _dev(self.inverse, torch.int64, dev)
        self.d_t = _dev(problem["t"], torch.float64, dev)
        gain = problem.get("gain")
        self.d_gain = _dev(pick(gain), torch.float32, dev) if gain is not None else None
        self.ta0 = float(problem["ta"][0])
        # the coarse step from the grid's whole span: the difference of two neighbouring nodes carries the
        # float64 rounding of a unix time (1e-7 s: 1e-6 of the step, 0.7 ms of time shift -- 1.3e-6 of the
        # loading -- at the end of a 600 s scan)
        ta = np.asarray(problem["ta"], float)
        self.dta = float((ta[-1] - ta[0]) / (len(ta) - 1)) if len(ta) > 1 else float(problem.get("timestep", 1.0))
        self.pwv0 = float(problem["pwv0"])

        self.d_flags = torch.zeros(1, dtype=torch.int32, device=dev)
        self.d_loading = torch.empty((self.Ta, self.D), dtype=torch.float32, device=dev)
        self.d_ym = None  # (y, m) knots of the two-call spline form: allocated by prepare()
        # keep_pwv: every sample() also writes the float64 zenith-scaled pwv (the map mixin reads it) and
        # run() keeps the stages serial; without it coarse_pwv() still works -- it re-runs the sampler on
        # the bound screens into a buffer made on first use
        self.keep_pwv = bool(keep_pwv)
        self.d_pwv = torch.empty((self.Ta, self.D), dtype=torch.float64, device=dev) if keep_pwv else None
        self._pwv_stale = True

        self._layer_bufs = []
        self._table_bufs = []
        self.plan = None
        self._upload_tables()
        if all(l.get("values") is not None for l in problem["layers"]):
            self.set_screens([l["values"] for l in problem["layers"]])

    # -- set-up ------------------------------------------------------------
    def _upload_tables(self):
        dev = self.device
        self._tables = (MrxBandTable * len(self.problem["tables"]))()
        for b, table in enumerate(self.problem["tables"]):
            vals, w, oob = table_slabs(table, self.problem["T0"])
            bufs = [
                _dev(vals, torch.float32, dev),
                _dev(table["pwv"], torch.float32, dev),
                _dev(table["el"], torch.float32, dev),
            ]
            tb = self._tables[b]
            tb.d_values, tb.d_axis_pwv, tb.d_axis_el = (x.data_ptr() for x in bufs)
            tb.n_pwv, tb.n_el = len(table["pwv"]), len(table["el"])
            tb.w_t, tb.t_oob = float(w), int(oob)
            tb.d_cubic = None
            if self.cubic:
                bufs.append(_dev(table_cubic_cells(table, self.problem["T0"]), torch.float64, dev))
                tb.d_cubic = bufs[-1].data_ptr()
            self._table_bufs.append(tuple(bufs))

    def layer_offsets(self, layer):
        """f64 per-time offsets of mrx_layer (include/mrx.h):
        (cumsum(timestep*(vx,vy,0)) + (0,0,h)) @ transform, columns 0 and 1
        (atmosphere/atmosphere.py:318-319,346-347)."""
        dt = self.problem["timestep"]
        tr = np.cumsum(dt * np.c_[layer["vx"], layer["vy"], np.zeros(len(layer["vx"]))], axis=0)
        q = (tr + np.array([0.0, 0.0, layer["h"]])) @ np.asarray(layer["transform"], float)
        return q[:, 0], q[:, 1]

    def set_screens(self, screens):
        """Bind one smoothed screen per layer (numpy arrays or device tensors [E,C])."""
        if self.plan is not None:
            self.ctx.call("mrx_atm_plan_destroy", self.plan)
            self.plan = None
        self.plan, self._layers, self._layer_bufs = self._make_plan(screens)
        self._la = None  # (a look-ahead set up for generated screens ends here)

    def _make_plan(self, screens):
        """(plan handle, mrx_layer array, device buffers) of one set of screens."""
        dev = self.device
        layers = self.problem["layers"]
        c_layers = (MrxLayer * len(layers))()
        layer_bufs = []
        for l, (layer, scr) in enumerate(zip(layers, screens)):
            vals = scr if isinstance(scr, torch.Tensor) else _dev(scr, torch.float32, dev)
            if vals.device != dev or vals.dtype != torch.float32 or not vals.is_contiguous():
                vals = vals.to(device=dev, dtype=torch.float32).contiguous()
            oe, oc = self.layer_offsets(layer)
            bufs = (
                vals,
                _dev(layer["extrusion"], torch.float32, dev),
                _dev(layer["cross_section"], torch.float32, dev),
                _dev(oe, torch.float64, dev),
                _dev(oc, torch.float64, dev),
            )
            assert bufs[3].numel() == self.Ta, "layer wind arrays must have Ta entries"
            assert tuple(vals.shape) == (len(layer["extrusion"]), len(layer["cross_section"]))
            layer_bufs.append(bufs)
            R = np.asarray(layer["transform"], float)
            ly = c_layers[l]
            ly.d_values, ly.d_axis_e, ly.d_axis_c, ly.d_off_e, ly.d_off_c = (x.data_ptr() for x in bufs)
            ly.n_e, ly.n_c = vals.shape
            ly.h = float(layer["h"])
            ly.r00, ly.r10, ly.r01, ly.r11 = R[0, 0], R[1, 0], R[0, 1], R[1, 1]
            ly.pwv_rms = float(np.float32(layer["pwv_rms"]))
            # uniform-axis hint (include/mrx.h): the f64 grids behind the f32 axes
            ex, cs = np.asarray(layer["extrusion"], float), np.asarray(layer["cross_section"], float)
            # (the step from the whole span: the difference of two neighbouring nodes carries their rounding)
            ly.e0, ly.de = float(ex[0]), float((ex[-1] - ex[0]) / (len(ex) - 1))
            ly.c0, ly.dc = float(cs[0]), float((cs[-1] - cs[0]) / (len(cs) - 1))
        plan = C.c_void_p()
        self.ctx.call("mrx_atm_plan_create", c_layers, len(layers), self._tables, len(self._tables), self.Ta, C.byref(plan))
        return plan, c_layers, layer_bufs

    def plan_info(self):
        """(number of layer axes on the recomputed-node fast path, tables staged in LDS)."""
        ua, tl = C.c_int(), C.c_int()
        self.ctx.call("mrx_atm_plan_info", self.plan, C.byref(ua), C.byref(tl))
        return ua.value, bool(tl.value)

    @staticmethod
    def _grid_steps(layer):
        """(extrusion step, cross-section step) of the grid a layer's screen is GENERATED on."""
        de = float(layer["extrusion"][1] - layer["extrusion"][0])
        cross = layer["gen"]["cross"] if layer.get("gen") is not None else layer["cross_section"]
        return de, float(cross[1] - cross[0])

    def _amplitude_table(self, nh, ny, nx, dh, dy, dx, r0, nu):
        """The device table of ``mrx_screen_amplitudes`` for one periodic domain, built on first use; None when the
        path was asked for the power-law spectrum."""
        if getattr(self, "_amp", None) is None or self.problem.get("turbulence_spectrum", "covariance") != "covariance":
            return None
        key = (nh, ny, nx, dh, dy, dx, r0, nu)
        if key not in self._amp:
            log_first, log_step, log_cov, log_sf, x_cut = matern_log_tables(nu)
            n_t, n_w = C.c_size_t(), C.c_size_t()
            _lib.load().mrx_screen_amp_floats(nh, ny, nx, len(log_cov), C.byref(n_t), C.byref(n_w))
            table = torch.empty(n_t.value, dtype=torch.float32, device=self.device)
            work = torch.empty(n_w.value, dtype=torch.float32, device=self.device)
            as_d = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            self.ctx.call("mrx_screen_amplitudes", nh, ny, nx, dh, dy, dx, r0, as_d(log_cov), as_d(log_sf), len(log_cov),
                          log_first, log_step, x_cut, ptr(table), ptr(work), work.numel())
            self._amp[key] = table
        return self._amp[key]

    def sampled_margins_px(self):
        """Per layer, the smallest distance in pixels between any line of sight of the observation and an edge of the
        layer's grid: (margin along the extrusion axis, margin across).  Of the WHOLE focal plane, not of this path's
        detector shard: the margins decide how a screen is generated (the beam as a stencil or as a factor of the
        spectrum, generate_screens), screens are shared by all shards -- regenerated by every rank or, layer-sharded,
        made by one rank for all --, and a shard's TOD must not depend on how the detectors were cut.  Host-side and
        conservative: the boresight track with a ring of 24 directions around the focal plane's outermost detector,
        pushed out to circumscribe the circle, through the float64 form of the pointing
        (coords/transforms.py:10-29) and the layer's projection (atmosphere/atmosphere.py:346-347)."""
        if getattr(self, "_margins", None) is None:
            p = self.problem
            off = np.asarray(p["offsets"], float)
            rad = float(np.hypot(off[:, 0], off[:, 1]).max()) if len(off) else 0.0
            ang = np.linspace(0.0, 2.0 * np.pi, 24, endpoint=False)
            ring = np.r_[np.zeros((1, 2)), (rad / np.cos(np.pi / 24) * 1.001 + 1e-9) * np.c_[np.cos(ang), np.sin(ang)]]
            az, el = np.asarray(p["az_a"], float), np.asarray(p["el_a"], float)
            dx, dy = ring[:, 0][:, None], ring[:, 1][:, None]
            r, q = np.hypot(dx, dy), np.arctan2(-dx, -dy)
            z = (np.sin(r) * np.cos(q) + 1j * np.cos(r)) * np.exp(1j * (el[None, :] - np.pi / 2))
            phi, theta = np.arctan2(np.sin(r) * np.sin(q), z.real) + az[None, :], np.arcsin(z.imag)
            with np.errstate(divide="ignore", invalid="ignore"):
                px, py = np.cos(phi) / np.tan(theta), np.sin(phi) / np.tan(theta)
            out = []
            for layer in p["layers"]:
                oe, oc = self.layer_offsets(layer)
                R = np.asarray(layer["transform"], float)
                e = layer["h"] * (px * R[0, 0] + py * R[1, 0]) + oe[None, :]
                c = layer["h"] * (px * R[0, 1] + py * R[1, 1]) + oc[None, :]
                ex, cs = np.asarray(layer["extrusion"], float), np.asarray(layer["cross_section"], float)
                de, dc = (ex[-1] - ex[0]) / (len(ex) - 1), (cs[-1] - cs[0]) / (len(cs) - 1)
                if not (np.isfinite(e).all() and np.isfinite(c).all()):
                    out.append((-np.inf, -np.inf))
                    continue
                out.append((float(min(e.min() - ex[0], ex[-1] - e.max()) / de), float(min(c.min() - cs[0], cs[-1] - c.max()) / dc)))
            self._margins = out
        return self._margins

    def generate_screens(self, smooth=True, only=None, exchange=None):
        """Philox + k-space filter + complex-to-real iFFT on the device with the beam
        smoothing (atmosphere/atmosphere.py:328-344) folded into the two FFT passes
        (mrx_screen_generate_batch), into persistent screen buffers.  The first call
        allocates the buffers and binds them; later calls only launch kernels: two per
        group of layers that share an FFT domain.  ``only``: generate just these layer indices
        (sharded generation); ``exchange``: a callable that takes the list of screens and fills in the
        other ranks' layers (maria_amd.dist), run on the stream the screens are generated on -- with
        enable_lookahead() the exchange of the next observation's screens then rides beside this one's
        synthesis.  Returns the device tensors."""
        dev = self.device
        layers = self.problem["layers"]
        la = getattr(self, "_la", None)
        shapes = [(len(l["extrusion"]), len(l["cross_section"])) for l in layers]
        # a layer may ask for a larger periodic FFT domain ("fft_shape") than its grid:
        # the screen is then the top-left block of it (atmosphere.py ribbons are not
        # periodic; the padding decorrelates opposite edges)
        fft_shapes = [tuple(l.get("fft_shape") or sh) for l, sh in zip(layers, shapes)]
        if getattr(self, "_gen_screens", None) is None:
            self._gen_screens = [torch.empty(sh, dtype=torch.float32, device=dev) for sh in shapes]
            # layers that are slices of one 3-D volume (model="3d") are generated together;
            # the others in groups that share a 2-D FFT domain
            self._gen_groups, self._gen_volumes = {}, {}
            for l, (layer, f) in enumerate(zip(layers, fft_shapes)):
                vol = layer.get("volume")
                if vol is not None:
                    self._gen_volumes.setdefault((vol["id"], vol["nh"], f), []).append(l)
                else:
                    self._gen_groups.setdefault(f, []).append(l)
            need = 0
            n = C.c_size_t()
            for (fe, fc), members in self._gen_groups.items():
                _lib.load().mrx_screen_work_floats(fe, fc, len(members), C.byref(n))
                need = max(need, n.value)
            for (_, nh, (fe, fc)), members in self._gen_volumes.items():
                _lib.load().mrx_screen3d_work_floats(nh, fe, fc, len(members), C.byref(n))
                need = max(need, n.value)
            self._gen_work = torch.empty(need, dtype=torch.float32, device=dev)
            # amplitude tables (mrx_screen_amplitudes: the eigenvalues of the Matern covariance on the periodic
            # grid), one per distinct (domain, steps, r0, nu); problem["turbulence_spectrum"] = "power_law" keeps the
            # closed-form spectrum
            self._amp = {}
            if self.problem.get("turbulence_spectrum", "covariance") == "covariance":
                for (fe, fc), members in self._gen_groups.items():
                    for l in members:
                        de, dc = self._grid_steps(layers[l])
                        self._amplitude_table(0, fe, fc, 0.0, de, dc, float(layers[l]["r0"]), float(layers[l]["nu"]))
                for (_, nh, (fe, fc)), members in self._gen_volumes.items():
                    first = layers[members[0]]
                    de, dc = self._grid_steps(first)
                    self._amplitude_table(nh, fe, fc, float(first["volume"]["dh"]), de, dc, float(first["r0"]), float(first["nu"]))
            # model="3d" with per-layer cross-section grids (layer["gen"]): the volume's planes are generated on
            # the process's generation grid, then resampled onto each layer's own grid and smoothed there
            self._gen_fine = {}
            for l, layer in enumerate(layers):
                g = layer.get("gen")
                if g is not None:
                    ne, n_l = shapes[l]
                    self._gen_fine[l] = dict(
                        plane=torch.empty((ne, len(g["cross"])), dtype=torch.float32, device=dev),
                        idx=_dev(g["idx"], torch.int32, dev), w=_dev(g["w"], torch.float32, dev), scale=_dev(g["scale"], torch.float32, dev))
            if self._gen_fine:
                ne = max(shapes[l][0] for l in self._gen_fine)
                nc = max(shapes[l][1] for l in self._gen_fine)
                self._gen_tmp = torch.empty((2, ne * nc), dtype=torch.float32, device=dev)
            self.set_screens(self._gen_screens)
        # The beam as a factor of the spectrum (mrx_screen_desc.periodic_beam: no stencils, a third of the generator's
        # arithmetic less) where every line of sight of the observation keeps at least the stencil's radius + 2 pixels
        # from the edges of the layer's grid: what those pixels hold is then scipy's reflect-mode result to rounding,
        # and the pixels that differ (the rim) are never sampled.  The reference's own ribbons leave a margin of one or
        # two pixels (atmosphere.py:208-245): they keep the stencils, and scipy's edges.
        if getattr(self, "_beam_in_spectrum", None) is None or getattr(self, "_beam_smooth", None) != smooth:
            self._beam_smooth = smooth
            self._beam_in_spectrum = [False] * len(layers)
            if smooth and self.problem.get("beam_in_spectrum", True):
                margins = None
                for l, layer in enumerate(layers):
                    sigma = float(layer.get("beam_sigma", 0) or 0)
                    if sigma <= 0 or layer.get("volume") is not None or l in self._gen_fine:
                        continue
                    if margins is None:
                        margins = self.sampled_margins_px()
                    de = float(layer["extrusion"][1] - layer["extrusion"][0])
                    dc = float(layer["cross_section"][1] - layer["cross_section"][0])
                    ry, rx = int(4.0 * sigma / de + 0.5), int(4.0 * sigma / float(layer.get("res") or dc) + 0.5)
                    fe, fc = fft_shapes[l]
                    self._beam_in_spectrum[l] = bool(margins[l][0] >= ry + 2 and margins[l][1] >= rx + 2 and ry < fe // 2 and rx < fc // 2)

        def pixel_sigmas(layer):
            """(sigma_e, sigma_c) of the beam in pixels as the reference forms them (atmosphere.py:338-339):
            beam sigma / mean extrusion step and beam sigma / layer.res -- the layer's RESOLUTION, which the
            step of its linspace grid exceeds slightly (synthetic layers without a "res": the grid step)."""
            de = float(layer["extrusion"][1] - layer["extrusion"][0])
            dc = float(layer["cross_section"][1] - layer["cross_section"][0])
            sigma = float(layer.get("beam_sigma", 0) or 0) if smooth else 0.0
            return sigma / de, sigma / float(layer.get("res") or dc)

        # look-ahead (enable_lookahead): this call fills the OTHER set of screens on the screens' own stream, behind
        # the samplers that last read that set; run() then samples it behind an event
        gen_ctx, target = self.ctx, self._gen_screens
        if la is not None:
            which = la["count"] % 2
            target, gen_ctx = la["screens"][which], la["ctx"]
            if la["sampled"][which]:
                la["stream"].wait_event(la["samplers_done"][which])

        def describe(members):
            descs = (_lib.MrxScreenDesc * len(members))()
            for d, l in zip(descs, members):
                layer = layers[l]
                de = float(layer["extrusion"][1] - layer["extrusion"][0])
                fine = self._gen_fine.get(l)
                if fine is None:
                    out = target[l]
                    dc = float(layer["cross_section"][1] - layer["cross_section"][0])
                    d.sigma_y, d.sigma_x = pixel_sigmas(layer)
                    d.periodic_beam = int(self._beam_in_spectrum[l])
                else:  # unsmoothed, on the generation grid
                    out = fine["plane"]
                    dc = float(layer["gen"]["cross"][1] - layer["gen"]["cross"][0])
                    d.sigma_y = d.sigma_x = 0.0
                d.d_out, d.stream = out.data_ptr(), l
                d.out_ny, d.out_nx, d.ld_out = out.shape[0], out.shape[1], out.stride(0)
                d.dy, d.dx, d.r0, d.nu = de, dc, float(layer["r0"]), float(layer["nu"])
                if layer.get("volume") is None:
                    amp = self._amplitude_table(0, *fft_shapes[l], 0.0, de, dc, d.r0, d.nu)
                    d.d_amp = amp.data_ptr() if amp is not None else None
            return descs

        with _range("Generating turbulence"):
            for (fe, fc), members in self._gen_groups.items():
                if only is not None:
                    members = [l for l in members if l in only]
                    if not members:
                        continue
                gen_ctx.call(
                    "mrx_screen_generate_batch", self.problem["seed"], fe, fc, describe(members), len(members),
                    ptr(self._gen_work), self._gen_work.numel(),
                )
            for (vid, nh, (fe, fc)), members in self._gen_volumes.items():
                # one volume is one draw: `only` does not split it (a rank that owns any of its layers makes all)
                if only is not None and not any(l in only for l in members):
                    continue
                first = layers[members[0]]
                vol = first["volume"]
                de = float(first["extrusion"][1] - first["extrusion"][0])
                cross = first["gen"]["cross"] if first.get("gen") is not None else first["cross_section"]
                dc = float(cross[1] - cross[0])
                pos = (C.c_double * len(members))(*[layers[l]["volume"]["pos"] for l in members])
                scl = (C.c_double * len(members))(*[layers[l]["volume"]["scale"] for l in members])
                amp = self._amplitude_table(nh, fe, fc, float(vol["dh"]), de, dc, float(first["r0"]), float(first["nu"]))
                self.ctx.call(
                    "mrx_screen_generate_3d", self.problem["seed"], int(vid) & 0xFFFF, nh, fe, fc, float(vol["dh"]), de, dc,
                    float(first["r0"]), float(first["nu"]), pos, scl, describe(members), len(members),
                    ptr(self._gen_work), self._gen_work.numel(), ptr(amp) if amp is not None else None,
                )
                for l in members:  # onto the layer's own grid, then the beam on that grid (atmosphere.py:341-344)
                    fine = self._gen_fine.get(l)
                    if fine is None:
                        continue
                    out, plane = self._gen_screens[l], fine["plane"]
                    ne, n_l = out.shape
                    sy, sx = pixel_sigmas(layers[l])
                    dst = out if (sy <= 1e-15 and sx <= 1e-15) else self._gen_tmp[0]
                    self.ctx.call("mrx_resample_columns", ptr(plane), ne, plane.shape[1], plane.stride(0), ptr(fine["idx"]), ptr(fine["w"]),
                                  ptr(fine["scale"]), n_l, ptr(dst), n_l)
                    if dst is not out:
                        self.ctx.call("mrx_gauss_smooth2d", ptr(dst), ptr(out), ptr(self._gen_tmp[1]), ne, n_l, sy, sx, 4.0)
        if la is not None:
            which = la["count"] % 2
            if exchange is not None:
                with torch.cuda.stream(la["stream"]):
                    exchange(target)
            la["screens_done"][which].record(la["stream"])
            la["count"] += 1
            la["current"] = which
            self.plan, self._layers, self._layer_bufs = la["plans"][which]
            self._gen_screens = target
            return target
        if exchange is not None:
            exchange(self._gen_screens)
        return self._gen_screens

    def enable_lookahead(self):
        """Let successive observations overlap (Simulation.run's loop over its plans, sim/simulation.py:201-211;
        bench.py's steps): the screens of the NEXT call to generate_screens() are made on a stream of their own, into
        a second set of buffers, while the samplers and writers of this run() are still at work, and the samplers (all
        of them on the side stream, block 0 included) start as soon as their screens and their coarse buffers are
        free instead of behind the previous run's last writer.  Nothing changes in what is computed: the same
        launches, ordered by events; the TOD of run() is stream-ordered on the caller's stream as before.  The
        returned screens are NOT ordered against the caller's stream: call wait_screens() before reading them.
        Returns False (and changes nothing) for paths whose screens are planes of a 3-D volume."""
        if getattr(self, "_la", None) is not None:
            return True
        layers = self.problem["layers"]
        if any(l.get("volume") is not None or l.get("gen") is not None for l in layers) or self.keep_pwv:
            return False
        if getattr(self, "_gen_screens", None) is None:
            self.generate_screens()
        torch.cuda.synchronize(self.device)
        main = torch.cuda.current_stream(self.device)
        side = self._pipeline_state(max(self.default_blocks(), 2), main)["side"]
        self.ctx.set_stream(main)
        probe = Context(self.ctx.device)
        probe.set_stream(side)
        stream = None
        for _ in range(8):  # a hardware queue of its own beside the caller's and the samplers' (HIP has four)
            stream = torch.cuda.Stream(device=self.device)
            if self.ctx.streams_concurrent(stream) and probe.streams_concurrent(stream):
                break
        ctx3 = Context(self.ctx.device)
        ctx3.set_stream(stream)
        for opt in (_lib.OPT_SCREEN_STOCKHAM,):
            ctx3.set_option(opt, self.ctx.get_option(opt))
        other = [torch.empty_like(t) for t in self._gen_screens]
        plans = [(self.plan, self._layers, self._layer_bufs), self._make_plan(other)]
        ev = lambda: torch.cuda.Event()  # noqa: E731
        self._la = dict(stream=stream, ctx=ctx3, screens=[self._gen_screens, other], plans=plans, count=0, current=0,
                        screens_done=[ev(), ev()], samplers_done=[ev(), ev()], sampled=[False, False],
                        writer_done={}, probe=probe)
        # the set bound now holds valid screens (generated above, on the caller's stream)
        self._la["screens_done"][0].record(main)
        self._la["count"] = 1
        return True

    def _mark_sampled(self, stream=None):
        """enable_lookahead: everything queued on ``stream`` (default: the current one) so far has read the bound set of
        screens -- the screens' stream may refill that set behind it.  Every reader of the bound screens ends with this
        (sample(), the serial and K_RJ forms of run(), synthesize(); the two-stream pipeline marks its side stream)."""
        la = getattr(self, "_la", None)
        if la is not None:
            la["samplers_done"][la["current"]].record(stream or torch.cuda.current_stream(self.device))
            la["sampled"][la["current"]] = True

    def wait_screens(self, stream=None):
        """Order ``stream`` (default: the current one) behind the generation of the screens now bound."""
        la = getattr(self, "_la", None)
        if la is not None:
            (stream or torch.cuda.current_stream(self.device)).wait_event(la["screens_done"][la["current"]])

    # -- hot path ------------------------------------------------------------
    def sample(self, want_pwv=False):
        self._pipelined = False
        self._synthesized = False
        if self.plan is None:
            raise RuntimeError("no screens bound: call set_screens() or generate_screens() first")
        self.wait_screens()
        self.ctx.call(
            "mrx_atm_sample", self.plan, ptr(self.d_az), ptr(self.d_el), self.Ta,
            ptr(self.d_dx), ptr(self.d_dy), ptr(self.d_band), ptr(self.d_m00), self.D,
            self.pwv0, ptr(self.d_pwv) if (self.keep_pwv or want_pwv) else None, ptr(self.d_loading), ptr(self.d_flags),
        )
        self._pwv_stale = not (self.keep_pwv or want_pwv)
        self._pwv_blocked = False
        self._mark_sampled()

    def prepare(self, krj=False):
        """Second derivatives of the coarse loading (``krj``: of the coarse loading in K_RJ that
        coarse_to_krj() made) for upsample() / upsample_krj(): the two-call form of the spline.
        run() uses upsample_fused(), which needs neither this call nor its buffer."""
        if self.d_ym is None:
            self.d_ym = torch.empty((self.Ta, self.D, 2), dtype=torch.float32, device=self.device)
        self.ctx.call("mrx_spline_prepare", ptr(self.d_loading_krj if krj else self.d_loading), self.D, self.Ta, ptr(self.d_ym))

    def upsample(self, out):
        self.ctx.call(
            "mrx_spline_upsample", ptr(self.d_ym), self.D, self.Ta, self.ta0, self.dta,
            ptr(self.d_t), self.T, ptr(self.d_gain), ptr(self.d_rows), ptr(out), out.stride(0),
        )

    def upsample_fused(self, out, krj=False):
        """mrx_spline_upsample_fused: spline solve + evaluation of the coarse loading in one
        kernel (no (y, m) buffer; ``krj``: of the coarse loading coarse_to_krj() made -- up to the last
        knot; the samples the reference extrapolates beyond it are divided one by one, _krj_tail)."""
        n_main = self._krj_split() if krj else self.T
        self.ctx.call(
            "mrx_spline_upsample_fused", ptr(self.d_loading_krj if krj else self.d_loading), self.D, self.Ta, self.ta0, self.dta,
            ptr(self.d_t), n_main, ptr(self.d_gain), ptr(self.d_rows), ptr(out), out.stride(0),
        )
        if krj and n_main < self.T:
            k = self._krj_tail_knots()
            self._krj_tail(self.d_loading[self.Ta - k :], self.D, slice(0, self.D), out, ptr(self.d_rows), self.ctx)

    def _krj_split(self):
        """Samples up to the last coarse knot.  Past it the reference EXTRAPOLATES its spline of the loading and divides
        by the true denominator; the coarse-grid form would extrapolate loading / denominator instead, and an extrapolated
        cubic misses the denominator's motion by 50x what an interior interval does (9e-5 in the last four samples of a
        tight fast scan, found by the randomised front-end sweep) -- so those samples, at most one coarse step of them,
        take the per-sample form: the two-call spline of the last knots and mrx_spline_upsample_krj on that window."""
        if getattr(self, "_n_main", None) is None:
            self._n_main = int(np.searchsorted(np.asarray(self.problem["t"], float)[: self.T], float(self.problem["ta"][-1]), side="right"))
        return self._n_main

    def _krj_tail_knots(self):
        return min(self.Ta, 48)  # the end of the spline forgets knots farther back as 0.268^k

    def _krj_tail(self, y_tail, n, rows, out, d_rows, ctx):
        """The samples past the last knot in K_RJ, per sample: ``y_tail`` the last _krj_tail_knots() knots of the coarse
        loading in pW ([k, n], contiguous) of the detector rows ``rows`` (internal order)."""
        c = self._cal
        k, s0 = self._krj_tail_knots(), self._krj_split()
        ym = torch.empty((k, n, 2), dtype=torch.float32, device=self.device)
        ctx.call("mrx_spline_prepare", ptr(y_tail), n, k, ptr(ym))
        dst = out[:, s0:] if d_rows.value else out[rows, s0:]
        ctx.call(
            "mrx_spline_upsample_krj", ptr(ym), n, k, float(self.problem["ta"][self.Ta - k]), self.dta, ptr(self.d_t[s0:]), self.T - s0,
            None if self.d_gain is None else ptr(self.d_gain[rows]), d_rows, ptr(c["bore_el"][s0:]), ptr(c["dx"][rows]), ptr(c["dy"][rows]),
            ptr(self.d_band[rows]), ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"], ptr(dst), out.stride(0),
        )

    def set_gain(self, gain):
        """Per-detector scale of the TOD, in the caller's detector order (or None)."""
        self.d_gain = None if gain is None else _dev(np.asarray(gain, np.float32)[self.order], torch.float32, self.device)

    def coarse_loading(self):
        """[D, Ta] float32 coarse loading in the caller's detector order (device tensor)."""
        if getattr(self, "_synthesized", False):  # the last run was one launch: blocks of [Ta][pitch] (mrx_atm_synthesize)
            if getattr(self, "_synthesized_krj", False):
                raise RuntimeError("the last run() wrote its coarse loading in K_RJ: call sample() (pW) before coarse_loading()")
            br, parts = self._synth_block_rows, []
            for lo in range(0, self.D, br):
                n = min(br, self.D - lo)
                pitch = (n + 31) // 32 * 32
                parts.append(self._coarse_blocks[self.Ta * lo : self.Ta * lo + self.Ta * pitch].view(self.Ta, pitch)[:, :n])
            return torch.cat(parts, dim=1).T.index_select(0, self._d_inverse)
        if getattr(self, "_pipelined", False):  # the last run kept it in per-block buffers
            if getattr(self, "_pipelined_krj", False):
                raise RuntimeError("the last run() converted its coarse buffers to K_RJ in place: call sample() (pW) before coarse_loading()")
            return torch.cat(self._pipe["loading"], dim=1).T.index_select(0, self._d_inverse)
        return self.d_loading.T.index_select(0, self._d_inverse)

    def coarse_pwv(self):
        """[D, Ta] float64 zenith-scaled pwv in the caller's detector order (of the screens now bound:
        without ``keep_pwv`` the sampler runs once more to produce it)."""
        if self.d_pwv is None:
            self.d_pwv = torch.empty((self.Ta, self.D), dtype=torch.float64, device=self.device)
        if self._pwv_stale:
            self.sample(want_pwv=True)
        if getattr(self, "_pwv_blocked", False) and self._synth_block_rows < self.D:  # the one-launch form's blocks
            br, flat, parts = self._synth_block_rows, self.d_pwv.view(-1), []
            for lo in range(0, self.D, br):
                n = min(br, self.D - lo)
                parts.append(flat[self.Ta * lo : self.Ta * (lo + n)].view(self.Ta, n))
            return torch.cat(parts, dim=1).T.index_select(0, self._d_inverse)
        return self.d_pwv.T.index_select(0, self._d_inverse)

    def coarse_pwv_time_major(self, rows=None):
        """[Ta, n] float64, time-major: the zenith-scaled pwv of the caller's rows ``rows`` (a device int64 tensor; None:
        all of them, in the caller's order) in the layout ``mrx_map_sample`` reads -- one gather along the detector axis of
        the sampler's own [Ta][D] array (which sits in the path's internal detector order) instead of coarse_pwv()'s
        transpose, the caller's row selection and a transpose back: four passes over 480 MB at 10 000 x 6 000."""
        if self.d_pwv is None or self._pwv_stale or (getattr(self, "_pwv_blocked", False) and self._synth_block_rows < self.D):
            full = self.coarse_pwv()  # (samples if need be; puts the one launch's detector blocks together)
            return (full if rows is None else full.index_select(0, rows)).T.contiguous()
        cols = self._d_inverse if rows is None else self._d_inverse.index_select(0, rows)
        return self.d_pwv.view(self.Ta, self.D).index_select(1, cols)

    def run(self, out=None, blocks=None, writer_events=None, krj=False):
        """The whole path for this shard; returns the [D, T] float32 TOD tensor.  The
        stages carry the reference's progress-bar names as profiler ranges (roctx via
        torch.cuda.nvtx).

        ``blocks``: detector rows are independent from sampling to the TOD, so the shard is
        cut into blocks and the arithmetic-bound sampler of block b+1 (a small resident grid on a
        side stream) runs beside the HBM-bound writer of block b: 2.79 against 3.10 ms on
        atlast_10k with 4 blocks.  Default: ``default_blocks()`` (never with
        ``keep_pwv``: its consumers want whole coarse arrays); ``blocks=1`` runs the stages back
        to back on the caller's stream.  ``krj``: the TOD in K_RJ (set_calibration first): the
        conversion is applied to the coarse loading before the spline when coarse_krj_bound() allows
        (the pW writer then writes K_RJ), per sample by mrx_spline_upsample_krj otherwise.
        Everything is queued, nothing waited for: call check_flags() before trusting the TOD -- a line of sight off its
        screen (the reference's RuntimeError), an emission table left, or a hand-over that gave up inside the one launch
        (MRX_FLAG_HANDOVER: the TOD is then invalid) are reported there.  Simulation.run_obs does."""
        if out is None:
            out = torch.empty((self.D, self.T), dtype=torch.float32, device=self.device)
        coarse_form = krj and krj != "sample" and self.coarse_krj_bound() <= self.COARSE_KRJ_LIMIT
        if blocks is None and (not krj or (coarse_form and not getattr(self, "_synth_krj_unsupported", False))) and self.synthesize_applies():
            try:
                return self.synthesize(out, writer_events=writer_events, krj=bool(krj))
            except MrxError as e:
                if e.code != -4:  # MRX_ERR_UNSUPPORTED
                    raise
                # (a layer off a uniform axis, a literal option, cubic tables -- or, in K_RJ, a cell table too large for the
                #  launch's LDS: the two-call forms)
                setattr(self, "_synth_krj_unsupported" if krj else "_synth_unsupported", True)
        if blocks is None:
            blocks = self.default_blocks()
        if krj and not self.coarse_krj_bound() <= self.COARSE_KRJ_LIMIT:
            # the per-sample conversion fused into the writer (set_calibration first)
            self.sample()
            self.prepare()
            self.upsample_krj(out)
            return out
        if blocks > 1 and not self.keep_pwv:
            return self._run_pipelined(out, blocks, writer_events=writer_events, krj=krj)
        with _range("Sampling turbulence + Computing atmospheric emission"):
            self.sample()
        with _range("Upsampling atmospheric loading"):
            if krj:
                self.coarse_to_krj()
            self.upsample_fused(out, krj=krj)
        return out

    def synthesize_applies(self):
        """Does run() take the one-launch form by default?  Wherever the library's form applies (every layer on a
        uniform axis, the default cell rule and pointing, linear tables: the call itself says so) from 1024 rows (640 rows:
        0.22 ms against 0.21 for the stages back to back; 1 264: 0.32 against 0.34; 2 512: 0.54 against 0.65; 10 000: 2.0
        against 2.3).  ``keep_pwv`` rides along: the sampler's float64 pwv is the launch's optional second output.

        The two-call forms stay selectable (same bits, the stages back to back or pipelined on two streams): set
        ``path.one_launch = False``, or MARIA_AMD_ONE_LAUNCH=0 in the environment for every path of the process
        (``Simulation`` included) -- the fallback for a device on which the launch's hand-over raises MRX_FLAG_HANDOVER."""
        if not getattr(self, "one_launch", True) or os.environ.get("MARIA_AMD_ONE_LAUNCH", "1") == "0":
            return False
        return self.D >= 1024 and not getattr(self, "_synth_unsupported", False)

    def synthesize(self, out=None, block_rows=None, sampler_wgs_per_cu=None, chunk=None, writer_events=None, krj=False, sampler_wgs=0):
        """Atmosphere -> TOD in ONE launch (mrx_atm_synthesize): sampler work items and TOD tiles as two queues of
        one resident grid, the hand-over between them on the device, time chunk by time chunk.  Same bits as the
        two-call forms.  ``block_rows``: detectors per block of the coarse array (default: the library's -- one block
        where it stays below 2 GiB); ``sampler_wgs_per_cu``: workgroups per CU that only sample while items remain
        (default 2; 8 or more: none -- every workgroup writes and samples only where it would wait);
        ``chunk``: coarse steps per time chunk, the unit of the hand-over (default: the library's, 32);
        ``sampler_wgs``: the dedicated samplers as a number of workgroups instead of per CU (0: not given).
        ``krj``: the TOD in K_RJ by the coarse-grid form (mrx_atm_synthesize_krj: the division in the sampler's
        epilogue; set_calibration first, and the caller has checked coarse_krj_bound() as run() does); the samples
        past the last knot take the per-sample form afterwards."""
        if self.plan is None:
            raise RuntimeError("no screens bound: call set_screens() or generate_screens() first")
        if out is None:
            out = torch.empty((self.D, self.T), dtype=torch.float32, device=self.device)
        main = torch.cuda.current_stream(self.device)
        self.ctx.set_stream(main)
        self.wait_screens(main)
        if block_rows is None:
            # the library's choice -- but a pwv that is kept (the map mixin reads it) in ONE block where the launch allows:
            # d_pwv is then the plain [Ta][D] array its reader wants (coarse_pwv_time_major), and taking the blocks apart
            # again costs more (four passes over the array, ~1 ms at 10 000 x 6 000) than blocks gain the launch (3-7 %)
            block_rows = (self.D + 255) // 256 * 256 if self.keep_pwv else 0
        if writer_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(main)
        if getattr(self, "_coarse_blocks", None) is None:
            self._coarse_blocks = torch.empty(self.Ta * ((self.D + 31) // 32 * 32), dtype=torch.float32, device=self.device)
        saved = (self.ctx.get_option(_lib.OPT_SAMPLE_WGS_PER_CU), self.ctx.get_option(_lib.OPT_SAMPLE_CHUNK))
        if sampler_wgs_per_cu is not None:
            self.ctx.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, int(sampler_wgs_per_cu))
        if chunk is not None:
            self.ctx.set_option(_lib.OPT_SAMPLE_CHUNK, int(chunk))
        args = [self.plan, ptr(self.d_az), ptr(self.d_el), self.Ta, ptr(self.d_dx), ptr(self.d_dy),
                ptr(self.d_band), ptr(self.d_m00), self.D, self.pwv0, ptr(self._coarse_blocks), int(block_rows), int(sampler_wgs),
                ptr(self.d_flags), self.ta0, self.dta, ptr(self.d_t), self._krj_split() if krj else self.T,
                None if self.d_gain is None else ptr(self.d_gain), None if self.d_rows is None else ptr(self.d_rows),
                ptr(out), out.stride(0)]
        tail = None
        if krj:
            c = self._cal
            if self._krj_split() < self.T:  # the samples past the last knot: per sample, from the last knots in pW
                if getattr(self, "_synth_tail", None) is None:
                    self._synth_tail = torch.empty((self._krj_tail_knots(), self.D), dtype=torch.float32, device=self.device)
                tail = self._synth_tail
            args += [ptr(c["dx"]), ptr(c["dy"]), ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"],
                     ptr(tail), 0 if tail is None else tail.shape[0], 0 if tail is None else tail.stride(0)]
        # keep_pwv: the sampler's second output, block by block like the loading (coarse_pwv() puts the blocks together)
        args.append(ptr(self.d_pwv) if self.keep_pwv else None)
        try:
            with _range("Sampling turbulence + Computing atmospheric emission + Upsampling atmospheric loading"):
                self.ctx.call("mrx_atm_synthesize_krj" if krj else "mrx_atm_synthesize", *args)
                if tail is not None:
                    self._krj_tail(tail, self.D, slice(0, self.D), out, ptr(self.d_rows), self.ctx)
        finally:
            self.ctx.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, saved[0])
            self.ctx.set_option(_lib.OPT_SAMPLE_CHUNK, saved[1])
        if writer_events is not None:
            ev[1].record(main)
            writer_events.append(ev)
        self._mark_sampled(main)  # (enable_lookahead: the screens' stream may refill this set once this launch is through)
        self._synth_block_rows = self.synth_block_rows(int(block_rows))
        self._synthesized = True
        self._synthesized_krj = bool(krj)  # (the coarse blocks then hold K_RJ, not pW)
        self._pipelined = False
        self._pwv_stale = not self.keep_pwv
        self._pwv_blocked = self.keep_pwv  # (d_pwv holds blocks of [Ta][rows], not one [Ta][D] array)
        return out

    def synth_block_rows(self, block_rows=0):
        """Rows per block of the coarse array as mrx_atm_synthesize lays it out (the library tells:
        mrx_atm_synthesize_block_rows): the caller's number in whole groups of 256 detectors, or -- 0 -- its own choice."""
        rows = C.c_int()
        self.ctx.call("mrx_atm_synthesize_block_rows", self.plan, self.D, self.Ta, int(block_rows), C.byref(rows))
        return rows.value

    def default_blocks(self):
        """Detector blocks of the pipelined run: about 0.65e9 samples each from 4096 rows up -- 4 for atlast_10k
        (2.12 ms with 4 blocks, 2.17 with 8, 2.24-2.34 with 12, 2.69 serial), 12 for the per-GPU share of
        atlast_50k (6 250 x 1 440 000 samples, 16 layers: 9.1-9.4 ms with 12 or 16 blocks, 9.8 with 8, 10.4 with 4,
        12.6 serial; round 4, scripts/exp_50k_sweep.py) --, 2 from 2048 rows (the shard of a 4-GPU run, 2 512 rows:
        0.63 ms against 0.66 serial and 0.69 with 4), 1 below (1 264 rows: 0.35 against 0.37;
        scripts/exp_small_blocks.py)."""
        if self.keep_pwv or self.D < 2048:
            return 1
        if self.D < 4096:
            return 2
        return int(min(max(round(self.D * self.T / 0.65e9), 4), 16, self.D // 256))

    def default_resident_wgs(self):
        """Workgroups per CU of the sampler that runs beside a writer.  The register file is what the two share: 3
        where the writer is the longer of the two (atlast_10k: 8 layers, sampler 0.7 ms against the writer's 1.8),
        4 where the sampler has as much to do (atlast_50k: 16 layers, 5.5 against 6.8 ms: 9.1 ms with 4, 9.9 with 3
        or 5, 10.8 with 6).  The measure is layer-samples per TOD sample."""
        work = len(self.problem["layers"]) * self.Ta / max(self.T, 1)
        return 4 if work >= 0.3 else 3

    def _side_stream(self, main):
        """A stream that really runs beside ``main``: HIP spreads streams round-robin over four hardware queues,
        and one new stream in four lands on the caller's -- the two then take turns and the pipelined step is as
        slow as the serial one (mrx_streams_concurrent)."""
        self.ctx.set_stream(main)
        side = None
        for _ in range(6):
            side = torch.cuda.Stream(device=self.device)
            if self.ctx.streams_concurrent(side):
                break
        return side

    def _pipeline_state(self, blocks, main=None):
        st = getattr(self, "_pipe", None)
        main = main if main is not None else torch.cuda.current_stream(self.device)
        if st is not None and st["blocks"] == blocks:
            if st["main"] != main.cuda_stream:  # another caller's stream: the side stream must be checked against it
                st["side"] = self._side_stream(main)
                st["ctx2"].set_stream(st["side"])
                st["main"] = main.cuda_stream
            return st
        # whole sampler workgroups (and writer tiles) per block, equal blocks (measured on two boxes against a
        # first block half as long as the others, whose writer then waits for the second sampler: 2.02-2.05 ms
        # against 2.07-2.12 at four blocks; scripts/exp_block_shares.py)
        units = -(-self.D // 256)
        share = getattr(self, "block_shares", None) or [1] * blocks
        cuts = np.floor(np.cumsum(share) / float(sum(share)) * units + 0.5).astype(int)
        edges = [0] + [min(int(c) * 256, self.D) for c in cuts]
        edges[-1] = self.D
        bounds = [(lo, hi) for lo, hi in zip(edges[:-1], edges[1:]) if hi > lo]
        if st is not None and st["main"] == main.cuda_stream:  # another cut of the rows: the streams stay
            side, ctx2 = st["side"], st["ctx2"]
        else:
            side = self._side_stream(main)
            ctx2 = Context(self.ctx.device)
            ctx2.set_stream(side)
        st = dict(blocks=blocks, bounds=bounds, side=side, ctx2=ctx2, main=main.cuda_stream,
                  ready=[torch.cuda.Event() for _ in bounds], start=torch.cuda.Event(), tail_done=torch.cuda.Event(),
                  loading=[torch.empty((self.Ta, hi - lo), dtype=torch.float32, device=self.device) for lo, hi in bounds])
        self._pipe = st
        return st

    def _run_pipelined(self, out, blocks, resident_wgs_per_cu=None, writer_events=None, serial_events=None, krj=False,
                       resident_times=1):
        """The sampler of block b on the side stream, the writer of block b (spline solve fused
        in: mrx_spline_upsample_fused) on the caller's stream behind an event; block 0's sampler
        takes the whole chip (nothing to run beside).
        ``writer_events``: a list that receives one (start, end) pair of timing events per writer
        launch, recorded on the stream the writer runs on (bench.py's live kernel timing).
        ``serial_events``: run the same block launches back to back on the caller's stream instead
        (no overlap) and append (t0, t1, t2) timing events per block around sample / writer: the
        per-stage breakdown of exactly the launches the pipelined step makes."""
        if self.plan is None:
            raise RuntimeError("no screens bound: call set_screens() or generate_screens() first")
        if resident_wgs_per_cu is None:
            resident_wgs_per_cu = self.default_resident_wgs()
        resident_wgs_per_cu = int(os.environ.get("MRX_AB_RESIDENT_WGS", resident_wgs_per_cu))  # (A/B runs)
        main = torch.cuda.current_stream(self.device)
        st = self._pipeline_state(blocks, main)
        side, ctx2 = st["side"], st["ctx2"]
        # the writers go through self.ctx: its stream must be the one the events below are
        # recorded on, whatever stream was current when this DevicePath was made
        self.ctx.set_stream(main)
        serial = serial_events is not None
        la = getattr(self, "_la", None) if not (serial or krj) else None
        if la is None:
            self.wait_screens(main)
        else:
            side.wait_event(la["screens_done"][la["current"]])
        if serial:
            side, ctx2 = main, self.ctx
        if not serial:
            # the sampler runs through the side context: what the caller set on this path's context (the cell rule, the
            # pointing chain: MRX_OPT_AXIS_LITERAL, MRX_OPT_POINTING_CHAIN) applies there too
            for opt in (_lib.OPT_POINTING_CHAIN, _lib.OPT_AXIS_LITERAL):
                if ctx2.get_option(opt) != self.ctx.get_option(opt):
                    ctx2.set_option(opt, self.ctx.get_option(opt))
        saved = (ctx2.get_option(_lib.OPT_SAMPLE_WGS_PER_CU), ctx2.get_option(_lib.OPT_SAMPLE_TIMES))
        sl = lambda t, lo, hi: None if t is None else ptr(t[lo:hi])  # noqa: E731
        for i, (lo, hi) in enumerate(st["bounds"]):
            n = hi - lo
            # beside a writer: a resident grid of default_resident_wgs() workgroups per CU, the layer loop
            # software-pipelined (atlast_10k: 2.12 ms at 3 per CU, 2.15 at 4, 2.27 at 5, 2.44 at 2; two steps
            # per thread at 96 registers: 2.18)
            alone = (i == 0 and la is None) or serial
            # block 0 has nothing to run beside: its sampler goes on the caller's stream, straight behind the screens and
            # straight before its writer (a kernel follows a kernel of its own stream after ~6 us, an event of another
            # stream after ~20: the kernel trace of the step showed five such waits on its critical path, two of them here)
            # -- unless successive runs overlap (enable_lookahead): every sampler then runs on the side stream, block
            # 0's beside the previous run's last writer, each behind the writer that last read its coarse buffer
            c = self.ctx if (i == 0 and la is None) else ctx2
            if la is not None and (blocks, i) in la["writer_done"]:
                side.wait_event(la["writer_done"][(blocks, i)])
            if c is ctx2:
                ctx2.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, saved[0] if alone else resident_wgs_per_cu)
                ctx2.set_option(_lib.OPT_SAMPLE_TIMES, saved[1] if alone else resident_times)
            if serial:
                tev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                tev[0].record(main)
            c.call(
                "mrx_atm_sample", self.plan, ptr(self.d_az), ptr(self.d_el), self.Ta,
                sl(self.d_dx, lo, hi), sl(self.d_dy, lo, hi), sl(self.d_band, lo, hi), sl(self.d_m00, lo, hi), n,
                self.pwv0, None, ptr(st["loading"][i]), ptr(self.d_flags),
            )
            if krj and krj != "sample":  # TOD.to("K_RJ") on the coarse grid: the writer below then writes K_RJ at the pW writer's cost
                tail = None
                if self._krj_split() < self.T:  # (the samples past the last knot: per sample, from the loading in pW,
                    if "tail" not in st:        #  in one pass over all rows beside the last block's writer)
                        st["tail"] = torch.empty((self._krj_tail_knots(), self.D), dtype=torch.float32, device=self.device)
                    tail = st["tail"][:, lo:hi]  # filled by the conversion kernel as it reads those knots
                self.coarse_to_krj(st["loading"][i], n, slice(lo, hi), c, tail=tail)
            if serial:
                tev[1].record(main)
            elif i == 0 and la is None:
                st["start"].record(main)  # the side stream starts block 1 behind the screens, the previous run's writers
                side.wait_event(st["start"])  # (they read the coarse buffers it is about to fill) and block 0's sampler
            else:
                st["ready"][i].record(side)
                main.wait_event(st["ready"][i])
            if self.d_rows is not None:
                dst, rows = out, sl(self.d_rows, lo, hi)
            else:
                dst, rows = out[lo:hi], None
            if writer_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(main)
            # a writer the next block's sampler has to find room beside: one workgroup per tile (a resident grid over
            # the tile queue -- the stand-alone default -- holds every slot of the chip until its last tile); the last
            # block's writer has the chip to itself
            self.ctx.set_option(_lib.OPT_WRITER_PER_TILE, 0 if (serial or (i == len(st["bounds"]) - 1 and la is None)) else 1)
            self.ctx.call(
                "mrx_spline_upsample_fused", ptr(st["loading"][i]), n, self.Ta, self.ta0, self.dta,
                ptr(self.d_t), self._krj_split() if krj else self.T, sl(self.d_gain, lo, hi), rows, ptr(dst), out.stride(0),
            )

            if writer_events is not None:
                ev[1].record(main)
                writer_events.append(ev)
            if la is not None:
                la["writer_done"].setdefault((blocks, i), torch.cuda.Event()).record(main)
            if serial:
                tev[2].record(main)
                serial_events.append(tev)
        ctx2.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, saved[0])
        ctx2.set_option(_lib.OPT_SAMPLE_TIMES, saved[1])
        self.ctx.set_option(_lib.OPT_WRITER_PER_TILE, 0)
        if la is not None:  # the screens' stream may refill this set once these samplers are through
            la["samplers_done"][la["current"]].record(side)
            la["sampled"][la["current"]] = True
        else:  # (the serial and K_RJ forms sample on streams the caller's stream is ordered behind by now)
            self._mark_sampled(main)
        if krj and krj != "sample" and self._krj_split() < self.T:
            if serial:
                self._krj_tail(st["tail"], self.D, slice(0, self.D), out, ptr(self.d_rows), self.ctx)
            else:
                # on the side stream, beside the last block's writer (other samples of the same rows): the copies into
                # st["tail"] precede it there, block 0's precedes the start event; its scratch belongs to that stream
                with torch.cuda.stream(side):
                    self._krj_tail(st["tail"], self.D, slice(0, self.D), out, ptr(self.d_rows), ctx2)
                st["tail_done"].record(side)
                main.wait_event(st["tail_done"])
        self._pipelined = True
        self._synthesized = False
        self._pipelined_krj = bool(krj and krj != "sample")  # (the coarse buffers then hold K_RJ, not pW)
        self._pwv_stale = True
        return out

    # -- TOD.to("K_RJ") fused into the upsample -----------------------------------------
    def set_calibration(self, cal_tables, base_temperature, zenith_pwv, bore_el, coords_offsets, polarized=None):
        """Host part of the pW -> K_RJ conversion (tod/tod.py:90-142,
        calibration/functions.py:73-90, band/band.py:235-255).

        ``cal_tables``: per band a dict with axes ``T``, ``pwv``, ``el`` and ``values``
        [nT, npwv, nel] = trapezoid(passband * exp(-opacity), nu); all bands share the
        elevation axis (one spectrum).  The table is collapsed at the scalar
        (``base_temperature``, ``zenith_pwv``) with jax's float32 index/weight rule onto the
        elevation axis and scaled to den = factor * k_B * 1e12 * integral, so that
        K_RJ = pW / den.  ``bore_el``: full-rate boresight elevation [T]; ``coords_offsets``:
        the (rolled) offsets of observation.coords [D, 2] in the caller's detector order."""
        dev = self.device
        k_B = 1.380649e-23
        nb = len(cal_tables)
        el_axis = np.asarray(cal_tables[0]["el"], float)
        polarized = np.zeros(nb, bool) if polarized is None else np.asarray(polarized, bool)
        dens = np.zeros((nb, len(el_axis)), np.float32)
        for b, tab in enumerate(cal_tables):
            assert np.array_equal(np.asarray(tab["el"], float), el_axis), "bands must share the elevation axis"
            vals = np.asarray(tab["values"], np.float32).astype(np.float64)
            wts = []
            for axis, x in ((tab["T"], base_temperature), (tab["pwv"], zenith_pwv)):
                g = np.asarray(axis, np.float32)
                xf = np.float32(x)
                i = int(np.searchsorted(g, xf, side="left")) - 1
                i = min(max(i, 0), len(g) - 2)
                w = np.float32((xf - g[i]) / (g[i + 1] - g[i]))
                oob = bool(xf < g[0] or xf > g[-1])
                wts.append((i, float(w), oob))
            (it, wt, ot), (ip, wp, op) = wts
            sl = vals[it : it + 2, ip : ip + 2]  # [2, 2, nel]
            col = ((1 - wt) * (1 - wp)) * sl[0, 0] + ((1 - wt) * wp) * sl[0, 1] + (wt * (1 - wp)) * sl[1, 0] + (wt * wp) * sl[1, 1]
            if ot or op:
                col = np.full_like(col, np.nan)
            dens[b] = ((0.5 if polarized[b] else 1.0) * k_B * 1e12 * col).astype(np.float32)
        off = np.asarray(coords_offsets, float)[self.det_slice][self.order]
        self._cal = dict(
            axis=_dev(el_axis, torch.float32, dev), values=_dev(dens, torch.float32, dev), n_el=len(el_axis), n_bands=nb,
            bore_el=_dev(bore_el, torch.float32, dev), dx=_dev(off[:, 0], torch.float32, dev), dy=_dev(off[:, 1], torch.float32, dev),
            axis_np=el_axis.astype(np.float32).astype(np.float64), values_np=dens.astype(np.float64),
            # (of the WHOLE focal plane, not of this shard's rows: which form of the conversion a run takes must not
            # depend on how its detectors are sharded -- shards are bit-identical to the unsharded rows)
            radius=float(np.hypot(*np.asarray(coords_offsets, float).T).max()) if len(coords_offsets) else 0.0,
        )

    # the coarse-grid form of the K_RJ conversion is taken when this estimate of its deviation from the
    # per-sample form stays below 0.4 of the parity tolerance (1e-5): the float32 path itself takes 3e-6 of it
    # at full size, which leaves a quarter of the tolerance unspent
    COARSE_KRJ_LIMIT = 4.0e-6
    SPLINE_KINK = 0.1708  # max |spline - f| / (slope jump x knot spacing) for a kink between uniform knots

    def coarse_krj_bound(self):
        """Estimate of max |S[y/g] - S[y]/g| / |S[y]/g| (S: the spline in time, g: the K_RJ
        denominator at the detector's elevation), i.e. of what dividing the COARSE loading by g
        (mrx_coarse_to_krj, then the pW writer) changes against dividing every full-rate sample
        (mrx_spline_upsample_krj, the reference's order, tod/tod.py:106-142).  Both are the same
        linear functional of y but for the spline's interpolation error on g(t) = den(el(t)):
        (a) where a detector's elevation crosses a node of the table's axis between two knots g has
        a kink, and the not-a-knot cubic spline through uniform knots misses a kink by at most 0.1708 x
        (slope jump) x (knot spacing) -- the kink in the middle of a knot interval; 0.085 on a knot; a
        linear interpolant: 0.25 -- (tests/test_host_geometry.py::test_spline_error_at_a_kink computes it);
        measured on the daisy scan: 0.15;
        (b) inside a cell g is linear in el, so the error is the spline's error on el(t),
        (5/384) h^4 d4el/dt4 -- estimated from fourth differences of the coarse boresight.
        inf when the form does not apply: a NaN in the collapsed table, a detector that may leave
        the table's elevation axis, or one that comes within 7 deg of the zenith (its elevation
        is not smooth in time there)."""
        c = self._cal
        ax, den = c["axis_np"], c["values_np"]
        el = np.asarray(self.problem["el_a"], float)
        if not np.isfinite(den).all() or len(el) < 5:
            return float("inf")
        lo, hi = el.min() - 1.05 * c["radius"], el.max() + 1.05 * c["radius"]
        if lo < ax[0] or hi > ax[-1] or hi > np.radians(83.0):
            return float("inf")
        # only the part of the axis the detectors visit counts: the cells that overlap [lo, hi] and the
        # nodes between them
        i0 = max(int(np.searchsorted(ax, lo, side="right")) - 1, 0)
        i1 = min(int(np.searchsorted(ax, hi, side="left")), len(ax) - 1)  # cells i0 .. i1 - 1
        slope = np.diff(den, axis=1) / np.diff(ax)[None, :]
        inner = slice(i0, i1 - 1)  # jumps between cells k and k + 1, k = i0 .. i1 - 2, sit at node k + 1
        rel_jump = (np.abs(np.diff(slope, axis=1))[:, inner] / np.abs(den[:, 1:-1][:, inner])).max() if i1 - i0 > 1 else 0.0
        cells = slice(i0, i1)
        rel_slope = (np.abs(slope[:, cells]) / np.minimum(np.abs(den[:, 1:]), np.abs(den[:, :-1]))[:, cells]).max()
        step = np.abs(np.diff(el)).max()
        d4 = np.abs(np.diff(el, n=4)).max()
        # Samples BEFORE the first knot (none in the reference, whose coarse grid starts at the first sample) would be
        # EXTRAPOLATED by both forms, and the cubic's error on g at a distance
        # delta h beyond the end is delta (delta+1) (delta+2) (delta+3) / 24 times h^4 d4g/dt4 -- 0.95 at delta = 1
        # against the 5/384 of an interior interval -- and a kink there is missed by delta x (slope jump) x h.  (A
        # randomised sweep found the form 9e-5 off in the last four samples of a tight, fast scan: a 0.13 deg daisy
        # at 0.6 deg/s, 12 knots per turn.)
        t, ta = np.asarray(self.problem["t"], float), np.asarray(self.problem["ta"], float)
        h = (ta[-1] - ta[0]) / max(len(ta) - 1, 1)
        delta = max(0.0, (ta[0] - t.min()) / h) if len(t) else 0.0  # (past the last knot the samples are divided one by one)
        smooth = max(5.0 / 384.0, delta * (delta + 1) * (delta + 2) * (delta + 3) / 24.0)
        kink = max(self.SPLINE_KINK, delta)
        # + 4e-7: the two forms round differently in float32 (and the per-sample writer interpolates
        # the reciprocal over 4 samples)
        return float(1.1 * (kink * rel_jump * step + smooth * rel_slope * d4) + 4e-7)

    def coarse_to_krj(self, loading=None, n=None, rows=slice(None), ctx=None, tail=None):
        """mrx_coarse_to_krj on the coarse loading (``loading``: a block's [Ta, n] buffer, in
        place; default: the whole shard's into a buffer of its own, which prepare(krj=True) reads).
        ``tail``: a [k, n] view that receives the last k knots in pW as they are read (_krj_tail's input)."""
        c = self._cal
        if loading is None:
            if getattr(self, "d_loading_krj", None) is None:
                self.d_loading_krj = torch.empty_like(self.d_loading)
            src, dst, n = self.d_loading, self.d_loading_krj, self.D
        else:
            src = dst = loading
        (ctx or self.ctx).call(
            "mrx_coarse_to_krj_keep_tail", ptr(src), n, self.Ta, ptr(self.d_el), ptr(c["dx"][rows]), ptr(c["dy"][rows]), ptr(self.d_band[rows]),
            ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"], ptr(dst),
            ptr(tail), 0 if tail is None else tail.shape[0], 0 if tail is None else tail.stride(0),
        )

    def upsample_krj(self, out):
        """mrx_spline_upsample_krj: the TOD in K_RJ (set_calibration first)."""
        c = self._cal
        self.ctx.call(
            "mrx_spline_upsample_krj", ptr(self.d_ym), self.D, self.Ta, self.ta0, self.dta, ptr(self.d_t), self.T,
            ptr(self.d_gain), ptr(self.d_rows), ptr(c["bore_el"]), ptr(c["dx"]), ptr(c["dy"]), ptr(self.d_band),
            ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"], ptr(out), out.stride(0),
        )

    def krj_row_tables(self):
        """What mrx_noise_generate_krj takes besides the field: the calibration of set_calibration() with the per-detector
        arrays in the CALLER's row order (the rows of the field it writes)."""
        c = self._cal
        if "rows" not in c:
            c["rows"] = dict(dx=c["dx"].index_select(0, self._d_inverse), dy=c["dy"].index_select(0, self._d_inverse),
                             band=self.d_band.index_select(0, self._d_inverse))
        return dict(c["rows"], bore_el=c["bore_el"], axis=c["axis"], values=c["values"], n_el=c["n_el"], n_bands=c["n_bands"])

    def to_krj(self, data):
        """mrx_tod_to_krj: convert a full-rate [D, T] pW field (caller's row order) to K_RJ in
        place with the calibration of ``set_calibration`` (tod/tod.py:106-142)."""
        c = self._cal
        assert tuple(data.shape) == (self.D, self.T) and data.stride(1) == 1
        self.ctx.call(
            "mrx_tod_to_krj", ptr(data), data.stride(0), self.D, self.T, None, ptr(self.d_rows), ptr(c["bore_el"]),
            ptr(c["dx"]), ptr(c["dy"]), ptr(self.d_band), ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"],
        )
        return data

    def from_krj(self, data):
        """mrx_tod_from_krj: the way back, a [D, T] K_RJ field to pW in place."""
        c = self._cal
        assert tuple(data.shape) == (self.D, self.T) and data.stride(1) == 1
        self.ctx.call(
            "mrx_tod_from_krj", ptr(data), data.stride(0), self.D, self.T, None, ptr(self.d_rows), ptr(c["bore_el"]),
            ptr(c["dx"]), ptr(c["dy"]), ptr(self.d_band), ptr(c["axis"]), ptr(c["values"]), c["n_el"], c["n_bands"],
        )
        return data

    def check_flags(self):
        """Raise the reference's errors if a sample left a screen or a table."""
        word = C.c_uint32()
        self.ctx.call("mrx_read_flags", ptr(self.d_flags), C.byref(word))
        if word.value & _lib.FLAG_HANDOVER:
            raise RuntimeError("mrx_atm_synthesize: a writer gave up waiting for the sampler (MRX_FLAG_HANDOVER): the TOD of that run is invalid")
        if word.value & _lib.FLAG_SCREEN_OOB:
            # atmosphere/atmosphere.py:368-369
            raise RuntimeError("A layer introduced nans into PWV simulation (line of sight left its screen).")
        if self.cubic and word.value & _lib.FLAG_TABLE_OOB:
            # scipy's RegularGridInterpolator(bounds_error=True), band/band.py:296-300
            raise ValueError("One of the requested xi is out of bounds of the emission table (pwv, elevation).")
        return word.value

    def clear_flags(self):
        self.ctx.call("mrx_clear_flags", ptr(self.d_flags))

    # -- accounting -------------------------------------------------------------
    def algorithmic_bytes(self):
        """B_alg of BASELINE.md section 4 for this shard."""
        screens = sum(len(l["extrusion"]) * len(l["cross_section"]) for l in self.problem["layers"])
        tables = sum(np.asarray(t["values"]).shape[1] * np.asarray(t["values"]).shape[2] for t in self.problem["tables"])
        return 4 * self.D * self.T + 8 * self.D * self.Ta + 4 * screens + 8 * self.Ta + 8 * self.D + 4 * tables
