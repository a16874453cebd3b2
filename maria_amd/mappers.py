"""``BinMapper`` (maria/mappers/bin_mapper.py): the TODs binned back onto a tangent-plane grid,
``map = ((W * D) @ P) / (W @ |P|)`` with the Stokes-weighted pointing matrix of
map/projection.py:134-179 -- on the device (``mrx_bin_map_bucketed`` / ``mrx_bin_map``), never
materialising P.  ``tod_preprocessing`` runs ``maria_amd.tod_processing.process_tod`` first, as
mappers/base.py:138 does; the map post-processing pipeline stays with maria's front end."""

from __future__ import annotations

import ctypes as C
import logging

import numpy as np
import torch

from ._lib import Context, MrxSkyMap, ptr
from .map import ProjectionMap, mueller_row

logger = logging.getLogger("maria")


# work buffer of the bucketed binning: all samples in one go needs 16 bytes per sample (64 bilinear); beyond this
# the call walks the time axis in chunks
BIN_WORK_LIMIT_BYTES = 24 << 30


def bin_map(ctx, sky, signal, weight, az, el, transform, dx, dy, stokes_w, channel, msum, mwgt, bucketed=None):
    """``map_sum += (W * D) @ P``, ``map_wgt += W @ |P|`` for one TOD on the device
    (mappers/bin_mapper.py:84-120).  Maps of up to 2048 regions of 64 x 32 pixels take
    ``mrx_bin_map_bucketed`` (samples routed to map regions, summed in LDS: no scattered global
    atomics); larger ones ``mrx_bin_map`` (float64 atomics).  ``bucketed``: force (True) or
    forbid (False) the first form."""
    D, T = signal.shape
    args = (C.byref(sky), ptr(signal), signal.stride(0), ptr(weight), 0 if weight is None else weight.stride(0),
            ptr(az), ptr(el), T, ptr(transform), ptr(dx), ptr(dy), ptr(stokes_w), ptr(channel), D, ptr(msum), ptr(mwgt))
    lo, full = C.c_size_t(), C.c_size_t()
    fits = ctx.lib.mrx_bin_map_work_bytes(C.byref(sky), D, T, C.byref(lo), C.byref(full)) == 0
    if bucketed is True and not fits:
        raise ValueError("the bucketed binning takes maps of at most 2048 regions of 64 x 32 pixels")
    if fits and bucketed is not False:
        free = torch.cuda.mem_get_info(signal.device)[0]
        size = max(lo.value, min(full.value, BIN_WORK_LIMIT_BYTES, max(free // 2, lo.value)))
        work = torch.empty(size, dtype=torch.uint8, device=signal.device)
        ctx.call("mrx_bin_map_bucketed", *args, ptr(work), work.numel())
        torch.cuda.current_stream(signal.device).synchronize()  # the buffer goes back to the allocator
        del work
    else:
        ctx.call("mrx_bin_map", *args)


class BinMapper:
    def __init__(self, tods, center, width=None, height=None, resolution=None, stokes="I", nu=None, frame="ra/dec",
                 units="K_RJ", degrees=True, bilinear=False, tod_preprocessing=None, map_postprocessing=None, device="cuda:0"):
        if map_postprocessing:
            # BinMapper.run overrides BaseMapper.run (mappers/bin_mapper.py:84 vs base.py:162-198):
            # the reference accepts the argument and never applies it
            logger.warning("BinMapper does not apply 'map_postprocessing' (neither does the reference's)")
        self.tod_preprocessing = dict(tod_preprocessing or {})  # mappers/base.py:138: tod.process(config=...)
        if frame not in ("ra/dec", "az/el"):
            raise NotImplementedError(f"frame '{frame}': only 'ra/dec' and 'az/el' are built")
        self.tods = list(tods)
        for tod in self.tods:
            if tod.units != units:
                raise ValueError(f"the TOD is in {tod.units}; ask Simulation.run(units='{units}')")
        unit = np.pi / 180 if degrees else 1.0
        if resolution is None or (width is None and height is None):
            raise ValueError("pass 'resolution' and at least one of 'width', 'height'")
        width = height if width is None else width
        height = width if height is None else height
        self.n_xi, self.n_eta = int(max(1, width / resolution)), int(max(1, height / resolution))  # mappers/base.py:295-301
        self.xi = unit * resolution * (self.n_xi - 1) * np.linspace(-0.5, 0.5, self.n_xi)
        self.eta = (unit * resolution * (self.n_eta - 1) * np.linspace(-0.5, 0.5, self.n_eta))[::-1].copy()
        self.center = (unit * center[0], unit * center[1])
        self.resolution, self.degrees = resolution, degrees
        self.stokes, self.frame, self.units, self.bilinear = stokes, frame, units, bool(bilinear)
        self.nu = np.atleast_1d(np.asarray(150e9 if nu is None else nu, float))
        self.device = torch.device(device)
        self.products = None

    def run(self):
        from .sim import sky_transform_stack

        dev = self.device
        ctx = Context(dev.index or 0)
        ctx.set_stream(torch.cuda.current_stream(dev))
        S, Cn = len(self.stokes), len(self.nu)
        msum = torch.zeros((S, Cn, self.n_eta, self.n_xi), dtype=torch.float64, device=dev)
        mwgt = torch.zeros_like(msum)
        deta, dxi = (self.eta[-1] - self.eta[0]) / (self.n_eta - 1), (self.xi[-1] - self.xi[0]) / (self.n_xi - 1)
        sky = MrxSkyMap(None, Cn, S, self.n_eta, self.n_xi, float(self.eta[0]), float(deta), float(self.xi[0]), float(dxi),
                        float(self.center[0]), float(self.center[1]), 1 if self.bilinear else 0, 0)
        f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
        for tod in self.tods:
            dets, coords = tod.dets, tod.coords
            if dets.n == 0:
                continue
            weight = None
            if self.tod_preprocessing:
                from .tod_processing import process_tod

                done = process_tod(tod, config={k: dict(v) for k, v in self.tod_preprocessing.items()}, ctx=ctx, device=dev)
                signal = done.data["total"]
                if not np.all(done.weight == 1.0):  # the window is the processed TOD's weight (processing.py:193)
                    weight = torch.as_tensor(np.ascontiguousarray(done.weight, np.float32)).to(dev).expand(signal.shape[0], -1).contiguous()
            else:
                signal = None
                for field in tod.data.values():  # tod.signal: the sum of the fields
                    f = field if isinstance(field, torch.Tensor) else torch.as_tensor(field)
                    f = f.to(dev, torch.float32)
                    signal = f.clone() if signal is None else signal.add_(f)
                signal = signal.contiguous()
            transform = None
            if self.frame == "ra/dec":
                transform = torch.as_tensor(sky_transform_stack(coords.t, tod.metadata["latitude"], tod.metadata["longitude"]).reshape(-1, 9)).to(dev)
            stokes_w = torch.as_tensor(np.ascontiguousarray(mueller_row(dets.gamma)[:, ["IQUV".index(s) for s in self.stokes]], np.float64)).to(dev)
            # the nu plane whose frequency is the detector's band centre, else plane 0 (projection.py:152-155)
            chan = np.zeros(dets.n, np.int32)
            for k, nu in enumerate(self.nu):
                chan[dets.band_center == nu] = k
            d_chan = torch.as_tensor(chan).to(dev)
            az, el = f32(coords._baz), f32(coords._bel)
            dx, dy = f32(coords.offsets[:, 0]), f32(coords.offsets[:, 1])
            bin_map(ctx, sky, signal, weight, az, el, transform, dx, dy, stokes_w, d_chan, msum, mwgt)
            torch.cuda.current_stream(dev).synchronize()
        data = (msum / mwgt).cpu().numpy()  # 0/0 = nan where nothing was observed, as numpy gives the reference
        self.products = {"data": data, "weight": mwgt.cpu().numpy(), "sum": msum.cpu().numpy()}
        out = ProjectionMap.__new__(ProjectionMap)
        out.data, out.weight = data.astype(np.float32), self.products["weight"]
        out.eta, out.xi, out.center = self.eta, self.xi, self.center
        unit = np.pi / 180 if self.degrees else 1.0
        out.x_res = out.y_res = unit * self.resolution
        out.stokes, out.nu, out.frame, out.units = self.stokes, self.nu, self.frame, self.units
        return out

    @property
    def map(self):
        if self.products is None:
            raise RuntimeError("Mapper has not been run yet!")
        return self.products["data"]
