"""Detector noise: the host side of ``NoiseMixin._simulate_noise`` (sim/noise.py:18-63).

The time-domain synthesis (white + 1/f with correlated modes) runs in ``libmrx``
(``mrx_noise_generate``); the spatial basis of the correlated modes is a small SVD +
cubic interpolation on the host, as in the reference (utils/linalg.py:105-126).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.interpolate
import scipy.spatial
import torch

from ._lib import ptr

DEFAULT_NOISE_SIM_KWARGS = {"correlated_noise_proportion": 0.5, "correlated_noise_spatial_scale": 1.0}  # sim/noise.py:11
# One keyword beyond the reference's: ``exact_spectrum`` (False).  From 32 768 samples on, where the pink part at a quarter
# (or half) of the Nyquist frequency is below 2 % of the white level, the generator makes the pink parts at a quarter (half)
# of the sample rate and interpolates them (csrc/mrx_noise.hip, the two-rate form: 5 instead of 8 ms at 10 000 x 240 000):
# the pink power above fs / (2 rate) -- under 2 % of the spectrum there -- is left out and the octave below loses up to 4 %
# of its pink part.  ``Simulation(noise_kwargs={"exact_spectrum": True})`` keeps the reference's a / |f| up to fs / 2
# (noise/generation.py:27-38; the one-rate form) at that price.
EXACT_SPECTRUM_BIT = 8  # of MRX_OPT_NOISE_GENERIC


def spatial_basis(offsets, k: int = 5, n_side: int = 16, scale: float = 1.0):
    """utils/linalg.py:105-126: the k leading modes of a Matern-5/2 kernel on an
    n_side x n_side grid over the focal plane, cubic-interpolated to the detectors."""
    offsets = np.asarray(offsets, float)
    x = np.linspace(offsets[:, 0].min(), offsets[:, 0].max(), n_side)
    y = np.linspace(offsets[:, 1].min(), offsets[:, 1].max(), n_side)
    X, Y = np.meshgrid(x, y)
    pts = np.stack([X.ravel(), Y.ravel()], axis=-1)
    r = np.sqrt(np.square(pts - pts[:, None]).sum(axis=-1)) / scale
    cov = (1 + np.sqrt(3) * r + (5.0 / 3.0) * r**2) * np.exp(-np.sqrt(5) * r)
    u, s, _ = np.linalg.svd(cov)
    basis = u[:, :k] * np.sqrt(s[:k])
    B = scipy.interpolate.RegularGridInterpolator((x, y), basis.reshape(n_side, n_side, -1), method="cubic")(offsets)
    return B * np.sign(B[:, 0].mean())


def diameter(offsets):
    """utils/__init__.py:56-: largest pairwise distance (via the hull)."""
    pts = np.asarray(offsets, float)
    if len(pts) < 2:
        return 0.0
    try:
        pts = pts[scipy.spatial.ConvexHull(pts).vertices]
    except Exception:
        pass
    return float(scipy.spatial.distance.pdist(pts).max())


def simulate_noise(ctx, dets, T, sample_rate, seed, noise_kwargs=None, device="cuda:0", batch=1024, out=None, loading=None,
                   det_slice=None, krj=None):
    """sim/noise.py:18-63: one band at a time, [ndet, T] float32 in pW on the device.
    ``dets`` is a ``maria_amd.instrument.Detectors``; bands carry ``NEP`` (W sqrt(s)), ``knee``
    (Hz) and ``NEP_per_loading``; ``loading`` is the [ndet, T] float32 device tensor of the summed
    loadings in pW, needed only by bands whose NEP grows with it (noise.py:35-37).
    ``det_slice``: generate only these rows of every band-major table (a detector shard; it may
    begin or end inside a detector pair): a shard's rows equal the same rows of the unsharded call,
    modes included.  ``krj``: ``DevicePath.krj_row_tables()`` -- the field is then written in K_RJ
    (mrx_noise_generate_krj: TOD.to("K_RJ"), tod/tod.py:106-142, on the generator's own store where its two-rate form applies)."""
    from ._lib import OPT_NOISE_GENERIC

    kw = dict(DEFAULT_NOISE_SIM_KWARGS)
    kw.update(noise_kwargs or {})
    if kw.get("exact_spectrum"):
        # the one-rate form for this call; whatever the option held (its other bits are cross-checks of the tests) comes back
        before = ctx.__dict__.get("_options", {}).get(OPT_NOISE_GENERIC, 0)
        ctx.set_option(OPT_NOISE_GENERIC, before | EXACT_SPECTRUM_BIT)
        try:
            return simulate_noise(ctx, dets, T, sample_rate, seed, dict(kw, exact_spectrum=False), device=device, batch=batch, out=out,
                                  loading=loading, det_slice=det_slice, krj=krj)
        finally:
            ctx.set_option(OPT_NOISE_GENERIC, before)
    dev = torch.device(device)
    lo, hi = (0, dets.n) if det_slice is None else (det_slice.start or 0, dets.n if det_slice.stop is None else det_slice.stop)
    if out is None:
        out = torch.empty((hi - lo, T), dtype=torch.float32, device=dev)
    assert out.shape[0] == hi - lo and (loading is None or loading.shape[0] == hi - lo)
    for b, band in enumerate(dets.bands):
        idx = np.nonzero(dets.band_index == b)[0]
        if len(idx) == 0:
            continue
        # the reference masks rows by band name (sim/noise.py:32): a band's rows need not be neighbours.  The draws
        # are keyed by a detector's index WITHIN its band; this shard holds the band's members k0 .. k1 - 1
        k0, k1 = int(np.searchsorted(idx, lo, side="left")), int(np.searchsorted(idx, hi, side="left"))
        if k1 <= k0:
            continue
        mine = idx[k0:k1]
        contiguous = bool((np.diff(mine) == 1).all())
        first, last = int(mine[0]), int(mine[0]) + (k1 - k0)  # (row range of the contiguous case)
        offset = k0  # index within the band: what the draws are keyed by
        per_loading = float(getattr(band, "NEP_per_loading", 0.0))
        if per_loading and loading is None:
            raise ValueError(f"band {band.name} has NEP_per_loading != 0: pass the summed loading (sim/noise.py:35-37)")
        # the basis depends on the band's focal-plane geometry alone (a 256 x 256 SVD and a cubic
        # interpolation on the host: 40 ms a band): kept on the Detectors object between observations
        cache = dets.__dict__.setdefault("_noise_basis_cache", {})
        key = (b, len(idx), float(kw.get("correlated_noise_spatial_scale", 0)), hash(dets.offsets[idx].tobytes()))
        basis = cache.get(key)
        if basis is None:
            offs = dets.offsets[idx]
            fov = diameter(offs)
            if fov > 0 and len(idx) > 16:  # sim/noise.py:42-50
                basis = spatial_basis(offs, k=5, n_side=16, scale=fov * kw.get("correlated_noise_spatial_scale", 0))
            else:
                basis = np.ones((len(idx), 1))
            cache[key] = basis
        count = last - first
        n_modes = basis.shape[1]
        need = C.c_size_t()
        ctx.lib.mrx_noise_work_floats(int(T), int(n_modes), int(min(batch, count)), C.byref(need))
        work = torch.empty(need.value, dtype=torch.float32, device=dev)
        # What a run needs on the device besides its output -- the band's basis, its NEP scale, this shard's rows of the
        # field -- does not change from run to run: uploaded once and kept beside the basis.  (Uploaded per call, each of
        # the small synchronous copies made the host wait for whatever the stream still held -- 7 ms apiece behind the
        # map sampler's kernel -- and nothing of the noise could be queued meanwhile.)
        dkey = (key, str(dev), lo, hi)
        held = cache.get(dkey)
        if held is None:
            held = dict(basis=torch.as_tensor(np.ascontiguousarray(basis, np.float32)).to(dev), rows=torch.as_tensor(mine - lo, device=dev),
                        scale=torch.full((len(idx),), float(1e12 * band.NEP), dtype=torch.float32, device=dev), nep=float(band.NEP), krj={})
            cache[dkey] = held
        if held["nep"] != float(band.NEP):
            held["scale"].fill_(float(1e12 * band.NEP))
            held["nep"] = float(band.NEP)

        def generate(row0, n_rows, dst, dst_loading, rows_of_out):
            """rows row0 .. row0 + n_rows - 1 of the band (row0 even: the pink series come in pairs) into ``dst``;
            ``rows_of_out``: the rows of ``out`` they are (an index tensor), for the K_RJ tables"""
            d_basis = held["basis"][row0 : row0 + n_rows]  # (whole rows of a contiguous array: a view)
            d_scale = held["scale"][:n_rows]  # noise.py:62
            args = (int(seed) + 7919 * b, n_rows, row0, int(T), float(sample_rate), float(band.knee),
                    float(kw.get("correlated_noise_proportion", 0)), ptr(d_basis), int(n_modes), ptr(d_scale),
                    ptr(dst_loading) if per_loading else None, dst_loading.stride(0) if per_loading else 0,
                    1e12 * per_loading, ptr(dst), dst.stride(0))
            if krj is None:
                ctx.call("mrx_noise_generate", *args, 0, ptr(work), need.value)
            else:
                # the calibration's per-row arrays for these rows (kept while the tables are the same objects)
                kk = (id(krj["dx"]), row0, n_rows)
                if kk not in held["krj"]:
                    if len(held["krj"]) > 8:
                        held["krj"].clear()
                    held["krj"][kk] = (krj["dx"], tuple(krj[k].index_select(0, rows_of_out) for k in ("dx", "dy", "band")))
                dx, dy, bd = held["krj"][kk][1]
                ctx.call("mrx_noise_generate_krj", *args, ptr(work), need.value, ptr(krj["bore_el"]), ptr(dx), ptr(dy), ptr(bd),
                         ptr(krj["axis"]), ptr(krj["values"]), krj["n_el"], krj["n_bands"])

        d_rows = held["rows"]  # the band's rows of ``out``
        if contiguous:
            view = out[first - lo : last - lo]
            lview = loading[first - lo : last - lo] if per_loading else None
        else:  # scattered rows: drawn into a buffer of their own and copied to their rows below
            view = torch.empty((k1 - k0, T), dtype=torch.float32, device=dev)
            lview = loading.index_select(0, d_rows) if per_loading else None
        start = 0
        if offset % 2:
            # the shard begins on the second detector of a pair: that pair is drawn whole into two scratch rows (its first
            # detector belongs to the neighbouring shard, which draws the same pair) and the second row kept
            pair = torch.empty((2, T), dtype=torch.float32, device=dev)
            pl = None
            if per_loading:  # the first row's loading is not ours: any finite values do for a row that is dropped
                pl = torch.stack([lview[0], lview[0]])
            generate(offset - 1, 2, pair, pl, d_rows[[0, 0]])
            view[0].copy_(pair[1])
            start = 1
        # ... and likewise when it ends on the FIRST detector of a pair whose second one exists (in the next shard): drawn
        # alone, that row would get the same draws but another rounding (the pair's two rows share one complex transform)
        stop = count
        if (offset + count) % 2 and offset + count < len(idx) and count > start:
            pair = torch.empty((2, T), dtype=torch.float32, device=dev)
            pl = torch.stack([lview[count - 1], lview[count - 1]]) if per_loading else None
            generate(offset + count - 1, 2, pair, pl, d_rows[[count - 1, count - 1]])
            view[count - 1].copy_(pair[0])
            stop = count - 1
        if stop > start:
            generate(offset + start, stop - start, view[start:stop], None if lview is None else lview[start:stop], d_rows[start:stop])
        if not contiguous:
            out.index_copy_(0, d_rows, view)
        del work
    return out
