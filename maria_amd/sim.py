"""``Simulation``: the reference's driver for the atmosphere path, device-backed.

Mirrors ``maria.sim.Simulation`` (sim/simulation.py:66-272) and the atmosphere
mixin (sim/atmosphere.py:24-84) for what BASELINE.json's north star covers:
``Simulation(instrument, plans, site, atmosphere="2d", atmosphere_kwargs=...)``
and ``.run(units)`` returning one ``TOD`` per plan.  The CMB, map and noise mixins
and unit conversions other than pW are follow-on rows and say so.
"""

from __future__ import annotations

import logging
import time as ttime

import numpy as np

from .atmosphere import Atmosphere
from .instrument import Instrument, Site

logger = logging.getLogger("maria")

MIN_ELEVATION_WARN = 10  # sim/observation.py
MIN_ELEVATION_ERROR = 5
HALF_PI_F32 = np.float32(np.pi / 2)


class PointingError(Exception):
    """maria/errors/__init__.py:8."""


class Coordinates:
    """az/el pointing of a boresight or of detectors around it (host-side numpy).

    The slice of ``maria.coords.Coordinates`` the atmosphere set-up needs:
    ``downsample`` (coordinates.py:286-304), ``broadcast`` to detector offsets in
    float32 like jax does (coordinates.py:378-386, transforms.py:10-29) and the
    unit-height ground projection (coordinates.py:333-349)."""

    def __init__(self, t, az, el, offsets=None):
        self.t = np.asarray(t, float)
        self._baz, self._bel = np.asarray(az, float), np.asarray(el, float)
        self.offsets = None if offsets is None else np.atleast_2d(np.asarray(offsets, float))
        self._azel = None

    def _pointing(self):
        if self.offsets is None:
            return self._baz, self._bel
        if self._azel is None:
            f32 = np.float32
            dx, dy = self.offsets[:, 0, None].astype(f32), self.offsets[:, 1, None].astype(f32)
            r = np.sqrt(dx * dx + dy * dy)
            p = np.arctan2(-dx, -dy)
            a = self._bel[None].astype(f32) - HALF_PI_F32
            a_re, a_im = np.sin(r) * np.cos(p), np.cos(r)
            re = a_re * np.cos(a) - a_im * np.sin(a)
            im = a_re * np.sin(a) + a_im * np.cos(a)
            self._azel = (np.arctan2(np.sin(r) * np.sin(p), re) + self._baz[None].astype(f32), np.arcsin(im))
        return self._azel

    @property
    def az(self):
        return self._pointing()[0]

    @property
    def el(self):
        return self._pointing()[1]

    @property
    def shape(self):
        return self.az.shape

    def downsample(self, timestep):
        ds_t = np.arange(self.t.min(), self.t.max(), timestep)
        return Coordinates(ds_t, np.interp(ds_t, self.t, self._baz), np.interp(ds_t, self.t, self._bel))

    def broadcast(self, offsets):
        return Coordinates(self.t, self._baz, self._bel, offsets=offsets)

    def project_unit(self):
        """(cos az / tan el, sin az / tan el) [.., Ta, 2]: ``project(z=1)`` from the origin."""
        az, el = self._pointing()
        tan_el = np.tan(el)
        return np.stack([np.cos(az) / tan_el, np.sin(az) / tan_el], axis=-1).astype(np.float64)


def sky_transform_stack(t, latitude_deg, longitude_deg):
    """[T, 3, 3] float64 with ``xyz(az, el) @ M[t] = xyz(ra, dec)``: the per-sample transform
    coords/coordinates.py:184-236 fits to astropy's AltAz -> ICRS at fiducial times and
    interpolates.  astropy is not available to this package, so the rotation is the
    textbook one -- horizon to hour angle/declination at the site's latitude, then the local
    sidereal angle (GMST of IAU 1982 + longitude) -- without precession, nutation, aberration
    or refraction: coordinates "of date", self-consistent within a simulation.  A caller with
    astropy hands its own stack to ``mrx_map_sample``."""
    t = np.asarray(t, float)
    lat = np.radians(latitude_deg)
    days = t / 86400.0 + 2440587.5 - 2451545.0
    lst = np.radians(15.0 * ((18.697374558 + 24.06570982441908 * days) % 24.0) + longitude_deg)
    A = np.array([[-np.sin(lat), 0.0, np.cos(lat)], [0.0, -1.0, 0.0], [np.cos(lat), 0.0, np.sin(lat)]])  # (N, E, U) -> hour-angle frame
    B = np.zeros((len(t), 3, 3))
    B[:, 0, 0], B[:, 0, 1] = np.cos(lst), np.sin(lst)
    B[:, 1, 0], B[:, 1, 1] = np.sin(lst), -np.cos(lst)
    B[:, 2, 2] = 1.0
    return A[None] @ B


class Plan:
    """Time-ordered boresight in the az/el frame (``maria.plan.Plan``'s ``time``,
    ``phi``, ``theta``; other frames need astropy and stay with maria's front end)."""

    def __init__(self, time, az, el, roll=0.0):
        self.time = np.asarray(time, float)
        self.phi, self.theta = np.asarray(az, float), np.asarray(el, float)
        self.frame, self.roll = "az/el", roll
        if self.time.ndim != 1 or self.phi.shape != self.time.shape or self.theta.shape != self.time.shape:
            raise ValueError("time, az and el must be one-dimensional and of equal length")

    @classmethod
    def daisy(cls, start_time=0.0, duration=60.0, sample_rate=50.0, scan_center=(45.0, 60.0), radius=0.5, speed=0.5):
        """plan/plan.py:55-140 with ``scan_pattern="daisy"`` in az/el, degrees."""
        from .synthetic import daisy_scan

        t = np.arange(start_time, start_time + duration, 1.0 / sample_rate)  # plan.py:83
        az, el = daisy_scan(t, radius, speed, scan_center[0], scan_center[1])
        return cls(t, az, el)

    @property
    def duration(self):
        return float(self.time[-1] - self.time[0])


class Observation:
    """sim/observation.py:27-100."""

    def __init__(self, instrument, plan, site, atmosphere=None, atmosphere_kwargs={}):
        self.instrument, self.plan, self.site = instrument, plan, site
        self.boresight = Coordinates(plan.time, plan.phi, plan.theta)
        c, s = np.cos(np.radians(plan.roll)), np.sin(np.radians(plan.roll))
        self.coords = self.boresight.broadcast(instrument.dets.offsets @ np.array([[c, -s], [s, c]]).T)
        # the elevation checks of observation.py:61-71, on the hull detectors (the extremes)
        el_min = float(self.boresight.downsample(1.0).broadcast(instrument.dets.outer().offsets).el.min()) if len(plan.time) > 1 else float(plan.theta.min())
        el_min = min(el_min, float(plan.theta.min()))
        if el_min < np.radians(MIN_ELEVATION_WARN):
            logger.warning(f"Some detectors come within {MIN_ELEVATION_WARN} degrees of the horizon (el_min = {np.degrees(el_min):.01f} deg)")
        if el_min <= np.radians(MIN_ELEVATION_ERROR):
            raise PointingError(f"Some detectors come within {MIN_ELEVATION_ERROR} degrees of the horizon (el_min = {np.degrees(el_min):.01f} deg)")
        if atmosphere:
            self.atmosphere_kwargs = atmosphere_kwargs
            if isinstance(atmosphere, Atmosphere):
                self.atmosphere = atmosphere
            else:
                self.atmosphere = Atmosphere(model=atmosphere, timestamp=float(plan.time.mean()), region=site.region, altitude=site.altitude, **atmosphere_kwargs)
        self.loading = {}


class TOD:
    """The slice of ``maria.tod.TOD`` this path fills: ``data`` (dict of [ndet, nt]
    float32 arrays, numpy or device tensors), ``dets``, ``coords``, ``units``, ``metadata``."""

    def __init__(self, data, dets, coords, units="pW", metadata=None):
        self.data, self.dets, self.coords, self.units = data, dets, coords, units
        self.metadata = metadata or {}

    @property
    def fields(self):
        return list(self.data)

    @property
    def signal(self):
        return sum(self.data.values())

    @property
    def time(self):
        return self.coords.t

    def to(self, units):
        """tod/tod.py:106-142 between "pW" and "K_RJ" (``mrx_tod_to_krj`` / ``mrx_tod_from_krj`` on
        every field), for a TOD that came out of ``Simulation.run``; other units stay with maria's
        calibration graph."""
        if units == self.units:
            return self
        if {units, self.units} != {"pW", "K_RJ"} or getattr(self, "_calibrator", None) is None:
            raise NotImplementedError(
                f"conversion {self.units} -> {units}: only pW <-> K_RJ of a TOD made by Simulation.run() is built"
            )
        out = TOD(data=self._calibrator(self.data, to_krj=(units == "K_RJ")), dets=self.dets, coords=self.coords, units=units,
                  metadata=self.metadata)
        out._calibrator = self._calibrator
        return out


class Simulation:
    def __init__(
        self,
        instrument,
        plans,
        site,
        atmosphere=None,
        atmosphere_kwargs: dict = {},
        cmb=None,
        cmb_kwargs: dict = {},
        map=None,
        map_kwargs: dict = {},
        noise: bool = True,
        noise_kwargs: dict = {},
        progress_bars: bool = True,
        keep_mean_signal: bool = False,
        dtype: type = np.float32,
        *,
        gain_seed: int = None,
        noise_seed: int = None,
        device_output: bool = False,
        shard=None,
        device: str = "cuda:0",
    ):
        """sim/simulation.py:76-198.  ``device_output=True`` leaves the TOD on the GPU as
        a torch tensor (a 10 k x 240 k TOD is 9.6 GB; the PCIe copy dwarfs the synthesis).
        ``shard``: ``(rank, world_size)`` -- this process simulates its contiguous block of
        detector rows (maria_amd.dist.shard_slice) of every observation, ``"auto"`` takes both
        from an initialised ``torch.distributed`` group, ``None`` simulates every row.  Every
        detector row is independent given the shared geometry (atmosphere.py:346-373 has no
        cross-detector term), so a shard's TOD equals the same rows of the unsharded run."""
        if cmb is not None:
            raise NotImplementedError("the CMB mixin is a follow-on row (SURVEY 8(f))")
        if isinstance(map, str):
            raise NotImplementedError("reading a map from a file stays with maria's io; pass a maria_amd.map.ProjectionMap")
        if map is not None:
            from .map import ProjectionMap

            if not isinstance(map, ProjectionMap):
                raise TypeError("'map' must be a maria_amd.map.ProjectionMap")
        self.map, self.map_kwargs = map, {"bilinear_sampling": True, **dict(map_kwargs)}  # sim/map.py:20
        if np.dtype(dtype) != np.float32:
            raise NotImplementedError("the device path writes float32 TODs (the reference default)")
        if not isinstance(instrument, Instrument):
            raise ValueError("'instrument' must be an Instrument (named configs stay with maria's front end)")
        if not isinstance(site, Site):
            raise ValueError("'site' must be either a Site object or a string.")
        if isinstance(plans, Plan):
            plans = [plans]
        elif not isinstance(plans, list):
            raise TypeError("plans must be a plan or a list of plans")
        self.instrument, self.site, self.plans = instrument, site, plans
        self.atmosphere, self.atmosphere_kwargs = atmosphere, dict(atmosphere_kwargs)
        self.noise, self.dtype = noise, dtype
        self.disable_progress_bars = not progress_bars
        self.device_output = device_output
        self.device = device  # every observation of this Simulation runs on this GPU
        if shard == "auto":
            import torch.distributed as dist

            shard = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else None
        if shard is not None:
            rank, world = (int(v) for v in shard)
            if not 0 <= rank < world:
                raise ValueError(f"shard rank {rank} outside [0, {world})")
            shard = (rank, world)
        self.shard = shard
        if shard is not None and shard[1] > 1:
            bands = instrument.dets.bands if isinstance(instrument, Instrument) else []
            if noise and noise_seed is None:
                raise ValueError("a sharded run needs an explicit noise_seed: the correlated noise modes are common to all shards")
            if gain_seed is None and any(getattr(b, "gain_error", 0.0) for b in bands):
                raise ValueError("a sharded run needs an explicit gain_seed: every rank draws the whole gain vector and keeps its rows")
        self._gain_rng = np.random.default_rng(gain_seed)
        self.noise_kwargs = dict(noise_kwargs)
        self._noise_seed = int(noise_seed if noise_seed is not None else np.random.SeedSequence().entropy % (1 << 62))
        self._noise_runs = 0
        self._noise_ctx = None
        self.obs_list = []
        for plan in self.plans:
            obs = Observation(instrument, plan, site, atmosphere, {"device": device, **self.atmosphere_kwargs})
            if hasattr(obs, "atmosphere"):
                # only the map mixin reads the coarse pwv (sim/map.py:117-135): without it the sampler and
                # the TOD writer run pipelined in one call and the pwv is made if somebody asks for it
                obs.atmosphere.keep_pwv = self.map is not None
                obs.atmosphere.initialize(obs)
            self.obs_list.append(obs)

    def _rows(self, n_det):
        """[lo, hi) of this process's detector rows (all of them without a shard)."""
        if self.shard is None:
            return 0, n_det
        from .dist import shard_bounds

        return shard_bounds(n_det, self.shard[1], self.shard[0])

    def run(self, units: str = "K_RJ", gather: bool = False):
        """sim/simulation.py:201-211: one TOD per plan, in ``units`` ("K_RJ", the reference
        default, or "pW").  With a shard each TOD holds this rank's rows; ``gather=True``
        all-gathers every field over the initialised ``torch.distributed`` group (RCCL over
        xGMI for device tensors) into the whole [ndet, nt] arrays on every rank."""
        if units not in ("K_RJ", "pW"):
            raise NotImplementedError(f"units '{units}': only 'K_RJ' and 'pW' are built (tod/tod.py:106-142)")
        tods = []
        for k, obs in enumerate(self.obs_list):
            t0 = ttime.monotonic()
            tod = self.run_obs(obs, units=units)
            if gather and self.shard is not None and self.shard[1] > 1:
                tod = self._gather(obs, tod)
            tods.append(tod)
            logger.info(f"Simulated observation {k + 1} of {len(self.obs_list)} in {ttime.monotonic() - t0:.2f} s")
        return tods

    # -- sim/atmosphere.py:24-84 -------------------------------------------------------
    def _simulate_atmosphere(self, obs):
        lo, hi = self._rows(obs.instrument.dets.n)
        det_slice = None if self.shard is None else slice(lo, hi)
        # screens only: DevicePath.run() samples and writes in one call -- with a map too: the coarse pwv its calibration
        # reads (sim/map.py:117-135) is the sampler's second output there (keep_pwv)
        obs.atmosphere.new_realisation(instrument=obs.instrument, det_slice=det_slice)

    def _gather(self, obs, tod):
        """All-gather a shard's TOD along the detector axis (equal row blocks, so the gathered
        array is the concatenation: dist.all_gather_tod)."""
        import torch

        from .dist import all_gather_tod

        dets = obs.instrument.dets
        data = {}
        for name, field in tod.data.items():
            on_device = isinstance(field, torch.Tensor)
            full = all_gather_tod(field if on_device else torch.as_tensor(field), dets.n)
            data[name] = full if on_device else full.numpy()
        out = TOD(data=data, dets=dets, coords=obs.coords, units=tod.units, metadata=dict(tod.metadata, shard=None))
        return out

    def _set_calibration(self, obs, metadata):
        """Host part of ``TOD.to("K_RJ")`` (tod/tod.py:90-97): collapse the bands' transmission
        tables at the scalars the TOD metadata carries, rounded as run_obs stores them."""
        atm, dets = obs.atmosphere, obs.instrument.dets
        path = atm._device_path()
        key = (metadata["base_temperature"], metadata["pwv"])
        # (the key lives on the path itself: a path rebuilt for another shard has none, whatever address it got)
        if getattr(path, "_cal_key", None) == key and hasattr(path, "_cal"):  # same path, same scalars: the tables are on the device already
            return
        path._cal_key = key
        sp = atm.spectrum
        tables = [{"T": sp.side_base_temperature, "pwv": sp.side_zenith_pwv, "el": sp.side_elevation,
                   "values": band.transmission_table(sp)} for band in dets.bands]
        polarized = [bool((~np.isnan(dets.gamma[dets.band_index == b])).all()) for b in range(len(dets.bands))]
        atm._device_path().set_calibration(tables, metadata["base_temperature"], metadata["pwv"], obs.boresight.el,
                                           obs.coords.offsets, polarized)

    def _compute_atmospheric_loading(self, obs, gain=None, units="pW", metadata=None):
        """Spline solve + cubic upsample of the coarse loading the sampling kernel already
        wrote (emission and Mueller weight are fused into it), scaled by ``gain``; with
        ``units="K_RJ"`` the division of ``TOD.to`` (tod/tod.py:90-142) is fused in: on the coarse grid
        before the spline where ``DevicePath.coarse_krj_bound`` allows, per sample in the writer otherwise."""
        import torch

        path = obs.atmosphere._device_path()
        path.set_gain(gain)
        out = torch.empty((path.D, path.T), dtype=torch.float32, device=path.device)
        if units == "K_RJ":
            self._set_calibration(obs, metadata)
        path.run(out, krj=units == "K_RJ")  # (nothing is sampled yet: see _simulate_atmosphere)
        path.check_flags()  # RuntimeError "introduced nans" like atmosphere.py:368-369
        return out

    def _sample_maps(self, obs, rows=None, krj=None, gain=None):
        """sim/map.py:76-172 on the device: one ``mrx_map_sample`` per band, [D, T] pW
        (``rows = (lo, hi)``: only this shard's detector rows).  ``krj`` (``DevicePath.krj_row_tables()``): the field in
        K_RJ instead, ``gain`` [D] multiplied in, both on the sampler's own store (``mrx_map_sample_krj``: the bits of
        the pW field times the gain, converted by ``DevicePath.to_krj``, without the two passes over the field)."""
        import torch

        from . import map as mmap
        from ._lib import Context
        from .instrument import compute_angular_fwhm

        lo, hi = rows or (0, obs.instrument.dets.n)
        dets = obs.instrument.dets.subset(np.arange(lo, hi))
        offsets = obs.coords.offsets[lo:hi]
        atm = getattr(obs, "atmosphere", None)
        if atm is not None:
            path = atm._device_path()
            ctx, device = path.ctx, path.device
            # (sync=False below frees this call's temporaries while the kernel is in flight: safe only on torch's current
            # stream, whatever stream path.run() left on the context)
            ctx.set_stream(torch.cuda.current_stream(device))
        else:
            device = torch.device(self.device)
            self._noise_ctx = self._noise_ctx or Context(device.index or 0)
            ctx = self._noise_ctx
            ctx.set_stream(torch.cuda.current_stream(device))
        T = len(obs.coords.t)
        # what does not change from run to run lives on the device, per observation and per band: the boresight, the
        # sample times, the horizon -> equatorial transform (14 ms of host arithmetic and 17 MB a run when it was made
        # per call), the smoothed map channels, the collapsed calibration tables, the detectors' offsets and weights
        # (the map OBJECT is kept and compared with `is`, as are the things the cached tensors were made from -- its data
        # array and the weather's temperature: an id() of a freed map can come back, and an edit in place keeps the id)
        key = (str(device), lo, hi)
        made_from = (self.map, self.map.data, None if atm is None else float(atm.weather.temperature[0]))
        cache = getattr(obs, "_map_cache", None)
        if (cache is None or cache.get("key") != key or cache["made_from"][0] is not made_from[0]
                or cache["made_from"][1] is not made_from[1] or cache["made_from"][2] != made_from[2]):
            f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(device)  # noqa: E731
            f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float64)).to(device)  # noqa: E731
            cache = {"key": key, "made_from": made_from, "az": f32(obs.boresight._baz), "el": f32(obs.boresight._bel),
                     "t": f64(obs.coords.t), "bands": {}}
            if atm is not None:
                cache["steps_per_tile"] = mmap.steps_per_tile(obs.coords.t, atm._device_path().dta)
            cache["transform"] = (f64(sky_transform_stack(obs.coords.t, obs.site.latitude, obs.site.longitude).reshape(T, 9))
                                  if self.map.frame == "ra/dec" else None)
            stokes_rows = mmap.mueller_row(dets.gamma)[:, ["IQUV".index(s) for s in self.map.stokes]]
            covered = np.zeros(dets.n, bool)
            for b, band in enumerate(dets.bands):
                idx = np.nonzero(dets.band_index == b)[0]
                if len(idx) == 0:
                    continue
                # ideally one beam per channel; the reference smooths once per band (map.py:101-104)
                fwhm = float(compute_angular_fwhm(fwhm_0=dets.primary_size.mean(), z=np.inf, nu=band.center))
                smoothed = self.map.smooth(fwhm, ctx=ctx, device=device)  # [S, C, eta, xi]
                channels, tables, scalars = [], [], []
                for c, (nu_min, nu_max) in enumerate(self.map.nu_bin_bounds):
                    if band.nu.max() < nu_min or nu_max < band.nu.min():  # map.py:112-113
                        continue
                    channels.append(c)
                    if atm is not None:  # band/band.py:250-255
                        sp = atm.spectrum
                        mask = (sp.side_nu >= nu_min) & (sp.side_nu < nu_max)
                        nu = sp.side_nu[mask]
                        grid = np.trapezoid(band.passband(nu) * np.exp(-sp._opacity[..., mask]), x=nu, axis=-1)
                        tables.append(mmap.collapse_temperature(grid, sp.side_base_temperature, atm.weather.temperature[0]))
                    else:  # band/band.py:246-248
                        nu = band.nu[(band.nu >= nu_min) & (band.nu < nu_max)]
                        scalars.append(float(np.trapezoid(band.passband(nu), x=nu)))
                if not channels:
                    logger.warning(f"No load from map for band {band.name}")
                    continue
                covered[idx] = True
                sm = smoothed if isinstance(smoothed, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(smoothed, np.float32))
                entry = dict(idx=idx, contiguous=bool((np.diff(idx) == 1).all()),
                             values=sm.to(device, torch.float32)[:, channels].transpose(0, 1).contiguous(),  # [C, S, eta, xi]
                             offsets=f32(offsets[idx]), stokes=f64(stokes_rows[idx]), idx_dev=torch.as_tensor(idx, device=device))
                if atm is not None:
                    sp = atm.spectrum
                    entry.update(tab=f32(np.stack(tables)), ap=f32(sp.side_zenith_pwv), ae=f32(sp.side_elevation))
                else:
                    entry.update(scalars=scalars)
                cache["bands"][b] = entry
            cache["all_rows"] = bool(covered.all())
            obs._map_cache = cache
        # rows of bands that see no channel of the map stay zero (sim/map.py:112-113); where every row is written
        # nothing is cleared first (a [D, T] fill is 9.6 GB at the headline size)
        out = (torch.empty if cache["all_rows"] else torch.zeros)((dets.n, T), dtype=torch.float32, device=device)
        for b, e in cache["bands"].items():
            # the reference masks rows by band name (sim/map.py:90-95): a band's rows need not be neighbours.  Contiguous
            # rows are written in place; scattered ones into a buffer of their own, copied to their rows below
            idx = e["idx"]
            if atm is not None:
                pwv = path.coarse_pwv_time_major(e["idx_dev"])  # [Ta, D_band]
                kw = dict(cal_tables=e["tab"], cal_axis_pwv=e["ap"], cal_axis_el=e["ae"], coarse_pwv=pwv, ta0=path.ta0,
                          dta=path.dta, t=cache["t"], steps_per_tile=cache["steps_per_tile"])
            else:
                kw = dict(cal_scalars=e["scalars"])
            if krj is not None:  # the calibration's per-row arrays for this band's rows (kept while the tables are the same objects)
                if e.get("krj_of") is not krj["dx"]:
                    e["krj_rows"] = {k: krj[k].index_select(0, e["idx_dev"]) for k in ("dx", "dy", "band")}
                    e["krj_of"] = krj["dx"]
                kw.update(krj=dict(e["krj_rows"], bore_el=krj["bore_el"], axis=krj["axis"], values=krj["values"]),
                          scale=None if gain is None else gain.index_select(0, e["idx_dev"]))
            dst = out[int(idx[0]) : int(idx[-1]) + 1] if e["contiguous"] else torch.empty((len(idx), T), dtype=torch.float32, device=device)
            mmap.sample_map(ctx, e["values"], self.map.eta, self.map.xi, self.map.center, cache["az"], cache["el"],
                            e["offsets"], e["stokes"], out=dst, transform=cache["transform"],
                            bilinear=bool(self.map_kwargs["bilinear_sampling"]), device=device, sync=False, **kw)
            if not e["contiguous"]:
                out.index_copy_(0, e["idx_dev"], dst)
        return out

    def _simulate_noise(self, obs, loading=None, rows=None, krj=None):
        """sim/noise.py:18-63 on the device, in pW; ``loading`` (pW, [D, T] on the device) only
        for bands whose NEP grows with the loading.  The gain error does not apply to the
        noise field (simulation.py:243-245).  ``krj``: DevicePath.krj_row_tables() -- the field in K_RJ."""
        import torch

        from . import noise as mnoise
        from ._lib import Context

        dets = obs.instrument.dets
        if hasattr(obs, "atmosphere"):
            ctx, device = obs.atmosphere._device_path().ctx, obs.atmosphere._device_path().device
        else:
            device = torch.device(self.device)
            if self._noise_ctx is None:
                self._noise_ctx = Context(device.index or 0)
            ctx = self._noise_ctx
            ctx.set_stream(torch.cuda.current_stream(device))
        t = obs.coords.t
        fs = 1.0 / np.mean(np.diff(t)) if len(t) > 1 else 1.0
        self._noise_runs += 1
        return mnoise.simulate_noise(ctx, dets, len(t), fs, self._noise_seed + 104729 * self._noise_runs,
                                     self.noise_kwargs, device=device, loading=loading,
                                     det_slice=None if rows is None else slice(*rows), krj=krj)

    def run_obs(self, obs, units: str = "pW") -> TOD:
        """sim/simulation.py:213-272 (followed by ``.to(units)`` of :206), for this process's
        detector rows."""
        import torch

        from . import _lib

        obs.loading = {}
        all_dets = obs.instrument.dets
        lo, hi = self._rows(all_dets.n)
        rows = None if self.shard is None else (lo, hi)
        dets = all_dets if rows is None else all_dets.subset(np.arange(lo, hi))
        # per-detector gain error (simulation.py:239-247): every rank draws the whole vector
        # and keeps its rows
        gain_error = np.array([b.gain_error for b in all_dets.bands], float)[all_dets.band_index]  # (not a loop over the detectors: 0.6 ms for 10 000, with the GPU idle)
        if any(b.gain_error for b in all_dets.bands):
            gain = np.exp(gain_error * self._gain_rng.standard_normal(all_dets.n))[lo:hi]
        else:  # no band has a gain error: nothing to draw (every rank decides alike, so the streams stay in step)
            gain = np.ones(hi - lo)
        has_gain = bool(np.any(gain_error[lo:hi]))
        metadata = {"atmosphere": False, "altitude": float(obs.site.altitude), "region": obs.site.region,
                    "latitude": obs.site.latitude, "longitude": obs.site.longitude,
                    "shard": None if rows is None else {"rank": self.shard[0], "world": self.shard[1], "rows": [lo, hi]}}
        has_atm = hasattr(obs, "atmosphere")
        device = torch.device(obs.atmosphere.device) if has_atm else torch.device(self.device)
        if hi <= lo:
            # a rank past the last block of rows (dist.shard_bounds cuts blocks of 16: 144 detectors on 4 ranks are
            # 48 + 48 + 48 + 0): fields of no rows, and the random streams kept in step with the other ranks
            if has_atm:
                metadata.update(atmosphere=True, pwv=float(np.round(obs.atmosphere.weather.pwv, 3)),
                                base_temperature=float(np.round(obs.atmosphere.weather.temperature[0], 3)))
                obs.atmosphere._realisation += 1
            names = [n for n, on in (("atmosphere", has_atm), ("map", self.map is not None), ("noise", self.noise)) if on]
            empty = torch.empty((0, len(obs.coords.t)), dtype=torch.float32, device=device)
            obs.loading = {n: (empty.clone() if self.device_output else empty.cpu().numpy()) for n in names}
            tod = TOD(data=obs.loading, dets=dets, coords=obs.boresight.broadcast(obs.coords.offsets[lo:hi]), units=units, metadata=metadata)
            tod._calibrator = lambda data, to_krj: {k: (v.clone() if isinstance(v, torch.Tensor) else v.copy()) for k, v in data.items()}
            return tod
        # Bands whose NEP grows with the loading need the loadings in pW, WITHOUT the gain
        # error, before the noise is drawn (sim/noise.py:35-37 runs before simulation.py:239-247
        # multiplies the gain into the non-noise fields); gain and the K_RJ conversion are then
        # applied after the fact.  Otherwise both ride on the upsample's store.
        loading_nep = self.noise and any(getattr(b, "NEP_per_loading", 0.0) for b in dets.bands)
        loading = None
        if has_atm:
            metadata.update(atmosphere=True, pwv=float(np.round(obs.atmosphere.weather.pwv, 3)),
                            base_temperature=float(np.round(obs.atmosphere.weather.temperature[0], 3)))
            self._simulate_atmosphere(obs)
            loading = self._compute_atmospheric_loading(
                obs, gain=gain if has_gain and not loading_nep else None,
                units="pW" if loading_nep else units, metadata=metadata)
        # The map field: in pW, gain and TOD.to("K_RJ") applied below -- or, in the default units with an atmosphere to
        # calibrate by and no noise that wants the loadings in pW first, written in K_RJ with the gain by the sampler
        # itself (mrx_map_sample_krj; not with the literal pointing chain, which that entry refuses)
        map_loading, map_done = None, False
        if self.map is not None:
            if (units == "K_RJ" and has_atm and not loading_nep
                    and not obs.atmosphere._device_path().ctx.get_option(_lib.OPT_POINTING_CHAIN)):
                self._set_calibration(obs, metadata)
                d_gain_rows = torch.as_tensor(gain.astype(np.float32), device=device) if has_gain else None
                map_loading = self._sample_maps(obs, rows, krj=obs.atmosphere._device_path().krj_row_tables(), gain=d_gain_rows)
                map_done = True
            else:
                map_loading = self._sample_maps(obs, rows)  # pW
        noise = None
        if self.noise:
            total = None
            if loading_nep:  # noise.py:35-37 sums every loading field, in pW
                fields = [f for f in (loading, map_loading) if f is not None]
                if fields:
                    total = fields[0] if len(fields) == 1 else fields[0] + fields[1]
                else:
                    total = torch.zeros((dets.n, len(obs.coords.t)), dtype=torch.float32, device=device)
            # in the default units the noise leaves its generator in K_RJ (mrx_noise_generate_krj)
            noise_krj = None
            if units == "K_RJ" and has_atm:
                self._set_calibration(obs, metadata)
                noise_krj = obs.atmosphere._device_path().krj_row_tables()
            noise = self._simulate_noise(obs, loading=total, rows=rows, krj=noise_krj)
        if has_gain:  # the gain error applies to every field but the noise (simulation.py:243-245)
            d_gain = torch.as_tensor(gain.astype(np.float32), device=device)[:, None]
            if map_loading is not None and not map_done:
                map_loading *= d_gain
            if loading is not None and loading_nep:
                loading *= d_gain
        if units == "K_RJ" and has_atm:
            # TOD.to("K_RJ") of the fields that were not written in K_RJ directly
            path = obs.atmosphere._device_path()
            if loading_nep:
                self._set_calibration(obs, metadata)
                path.to_krj(loading)
            if map_loading is not None and not map_done:  # (the noise was written in K_RJ)
                path.to_krj(map_loading)
        elif units == "K_RJ":
            den = torch.as_tensor(self._band_denominators(all_dets)[lo:hi].astype(np.float32), device=device)[:, None]
            for field in (map_loading, noise):
                if field is not None:
                    field /= den
        for name, field in (("atmosphere", loading), ("map", map_loading), ("noise", noise)):
            if field is not None:
                obs.loading[name] = field if self.device_output else field.cpu().numpy()
        coords = obs.coords if rows is None else obs.boresight.broadcast(obs.coords.offsets[lo:hi])
        tod = TOD(data=obs.loading, dets=dets, coords=coords, units=units, metadata=metadata)
        tod._calibrator = self._make_calibrator(obs, metadata, (lo, hi))
        return tod

    @staticmethod
    def _band_denominators(dets):
        """Without an atmosphere the transmission integral of ``TOD.to("K_RJ")`` is the band's own
        Int passband dnu, one number per band (band/band.py:246-248, calibration/functions.py:73-90):
        pW per K_RJ for every detector row."""
        den = np.empty(dets.n)
        for b, band in enumerate(dets.bands):
            rows = dets.band_index == b
            polarized = bool((~np.isnan(dets.gamma[rows])).all()) if rows.any() else False
            den[rows] = (0.5 if polarized else 1.0) * 1e12 * 1.380649e-23 * float(np.trapezoid(band.passband(band.nu), x=band.nu))
        return den

    def _make_calibrator(self, obs, metadata, rows):
        """What ``TOD.to`` needs to move a finished TOD between pW and K_RJ: the observation's
        calibration on the device (atmosphere) or one number per band (none)."""
        lo, hi = rows
        has_atm = hasattr(obs, "atmosphere")

        def convert(data, to_krj):
            import torch

            out = {}
            if has_atm:
                path = obs.atmosphere._device_path()
                self._set_calibration(obs, metadata)
                device = path.device
            else:
                device = torch.device(self.device)
                den = self._band_denominators(obs.instrument.dets)[lo:hi]
            for name, field in data.items():
                on_device = isinstance(field, torch.Tensor)
                f = (field.clone() if on_device else torch.as_tensor(np.ascontiguousarray(field, np.float32)).to(device)).contiguous()
                if has_atm:
                    path.to_krj(f) if to_krj else path.from_krj(f)
                else:
                    d = torch.as_tensor(den.astype(np.float32), device=f.device)[:, None]
                    f = f / d if to_krj else f * d
                out[name] = f if on_device else f.cpu().numpy()
            return out

        return convert
