"""GPU parity of map sampling: mrx_map_sample against the numpy
restatement of sim/map.py:76-172 (oracle/mapsample.py, whose pointing-matrix rule is pinned
by the reference's own function in tests/golden/leaves.json)."""

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _scan(D=37, T=3001, fov_deg=0.4, seed=0):
    from maria_amd import synthetic

    t = 1.7e9 + np.arange(T) / 50.0
    az, el = synthetic.daisy_scan(t)
    off = synthetic.hex_pack(D, np.radians(fov_deg))
    roll = np.radians(11.0)
    R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
    return t, az.astype(np.float32), el.astype(np.float32), off @ R.T


def _sky_rotation(t):
    """A smooth stack of orthonormal 3x3 (row vector times matrix), like the fitted
    az/el -> ra/dec transforms of coords/coordinates.py:205-218: a fixed tilt times a slow
    rotation about the z axis."""
    tilt = np.radians(35.0)
    Rx = np.array([[1, 0, 0], [0, np.cos(tilt), -np.sin(tilt)], [0, np.sin(tilt), np.cos(tilt)]])
    w = 7.292e-5 * (t - t[0]) + 0.3
    Rz = np.zeros((len(t), 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = np.cos(w), -np.sin(w), np.sin(w), np.cos(w), 1
    return Rx[None] @ Rz


def _centre(az, el, transform):
    """Centre of the scanned patch in the map's frame."""
    from oracle import mapsample

    phi, theta = mapsample.frame_angles(az[None, :], el[None, :], transform)
    xyz = mapsample.phi_theta_to_xyz(phi[0], theta[0]).astype(float).mean(axis=0)
    xyz /= np.linalg.norm(xyz)
    return float(np.arctan2(xyz[1], xyz[0]) % (2 * np.pi)), float(np.arcsin(xyz[2]))


def _rounding_bound(values, eta, xi, stokes_weight_sum, pw_per_k_sum):
    """Float32 angles carry a few 1e-7 rad of rounding in either implementation (6e-7 is the
    bound test_offsets_through_ramp_maps holds them to); times the steepest gradient of the
    sampled field that is the agreement one can ask of two float32 evaluations."""
    grad = max(np.abs(np.diff(values, axis=-1)).max() / abs(xi[1] - xi[0]), np.abs(np.diff(values, axis=-2)).max() / abs(eta[1] - eta[0]))
    return 2 * 6e-7 * grad * stokes_weight_sum * pw_per_k_sum


def _blob_map(C, S, n_eta, n_xi, eta, xi, rng):
    X, Y = np.meshgrid(xi, eta)
    w = 0.25 * (abs(xi[-1] - xi[0]) + abs(eta[-1] - eta[0])) / 2
    m = np.zeros((C, S, n_eta, n_xi), np.float32)
    for c in range(C):
        for s in range(S):
            x0, y0 = rng.uniform(-0.5, 0.5, 2) * w
            m[c, s] = (1 + 0.3 * c - 0.2 * s) * np.exp(-((X - x0) ** 2 + (Y - y0) ** 2) / (2 * w * w)) + 0.05 * rng.normal(size=X.shape)
    return m


@pytest.fixture(params=[0, 1], ids=["composed", "chain"])
def pointing_mode(request, gpu_ctx):
    """Both forms of steps 1-3: the composed float64 rotation (default) and the reference's
    float32 chain taken literally (MRX_OPT_POINTING_CHAIN)."""
    gpu_ctx.set_option(0, request.param)
    yield request.param
    gpu_ctx.set_option(0, 0)


@pytest.mark.parametrize("frame", ["az/el", "sky"])
def test_offsets_through_ramp_maps(gpu_ctx, frame, pointing_mode):
    """A map whose value is its own xi (or eta) coordinate returns the sample's offset from
    the map centre under bilinear sampling: the float32 geometry chain (pointing, frame
    rotation, phi_theta_to_offsets) agrees with the restatement to a few float32 ulps of angle."""
    from maria_amd import map as mmap
    from oracle import hotpath, mapsample

    t, az, el, off = _scan()
    transform = _sky_rotation(t) if frame == "sky" else None
    centre = _centre(az, el, transform)
    n = 64
    eta = np.linspace(0.02, -0.02, n)   # descending: the parity flip of sim/map.py:72-73
    xi = np.linspace(-0.02, 0.02, n)
    X, Y = np.meshgrid(xi, eta)
    values = np.stack([X, Y])[:, None].astype(np.float32)  # two "channels": xi ramp, eta ramp
    w = np.ones((len(off), 1))
    az_d, el_d = hotpath.broadcast(off, az, el)
    for c in range(2):
        got = mmap.sample_map(gpu_ctx, values[c : c + 1], eta, xi, centre, az, el, off, w, transform=transform,
                              cal_scalars=[1.0 / (1e12 * mapsample.K_B)]).cpu().numpy()
        ref = mapsample.sample_maps(az_d, el_d, t, None, None, eta, xi, centre, values[c : c + 1], w,
                                    cal_scalars=[1.0 / (1e12 * mapsample.K_B)], transform_stack=transform)
        assert np.abs(ref).max() > 3e-3  # the scan does cover the map
        assert np.abs(got - ref).max() <= 6e-7, (c, np.abs(got - ref).max())


@pytest.mark.parametrize("bilinear", [True, False])
def test_map_sampling_matches_oracle(gpu_ctx, bilinear, pointing_mode):
    """Three Stokes planes, two channels, polarised and unpolarised detectors, coarse pixels (so
    that float32 angle rounding moves no weight by more than 1e-5), samples off the map's edge,
    a TOD length that is no multiple of the tile, rows that are not 16-byte aligned."""
    import torch

    from maria_amd import map as mmap
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(3)
    t, az, el, off = _scan(D=45, T=2503, fov_deg=0.8)
    transform = _sky_rotation(t)
    centre = _centre(az, el, transform)
    az_d, el_d = hotpath.broadcast(off, az, el)
    ox = mapsample.phi_theta_to_offsets(*mapsample.frame_angles(az_d, el_d, transform), *centre)
    # a map smaller than the scanned patch: the samples beyond its edge take the edge pixels
    n_eta, n_xi = 5, 7
    half_eta, half_xi = 0.6 * float(np.abs(ox[..., 1]).max()), 0.6 * float(np.abs(ox[..., 0]).max())
    eta = np.linspace(half_eta, -half_eta, n_eta)
    xi = np.linspace(-half_xi, half_xi, n_xi)
    values = _blob_map(2, 3, n_eta, n_xi, eta, xi, rng)
    gamma = np.where(np.arange(len(off)) % 3 == 0, np.nan, rng.uniform(0, np.pi, len(off)))
    w = mapsample.mueller_row(gamma)[:, :3]
    np.testing.assert_allclose(mmap.mueller_row(gamma), mapsample.mueller_row(gamma), atol=1e-15)
    scal = [2.1e10, 0.7e10]
    ref = mapsample.sample_maps(az_d, el_d, t, None, None, eta, xi, centre, values, w, cal_scalars=scal,
                                transform_stack=transform, bilinear=bilinear)
    buf = torch.full((len(off), len(t) + 3), -5.0, dtype=torch.float32, device="cuda:0")
    out = buf[:, 1 : len(t) + 1]
    mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, out=out, transform=transform, bilinear=bilinear, cal_scalars=scal)
    got = out.cpu().numpy()
    assert bool((buf[:, 0] == -5).all()) and bool((buf[:, len(t) + 1 :] == -5).all())
    if bilinear:
        # within 3e-4 of the peak here (measured: 3e-5)
        bound = _rounding_bound(values, eta, xi, np.abs(w).sum(axis=1).max(), 1e12 * mapsample.K_B * sum(scal))
        assert np.abs(got - ref).max() <= bound and bound <= 3e-4 * np.abs(ref).max()
    else:
        # nearest pixel: a sample within float32 rounding of a pixel boundary may land on
        # either side; everything else is exact
        bad = np.abs(got - ref) > 1e-6 * np.abs(ref).max()
        assert bad.mean() < 2e-3


def test_map_sampling_with_atmospheric_transmission(gpu_ctx, pointing_mode):
    """With an atmosphere the K_RJ -> pW factor is looked up per sample at (zenith pwv,
    elevation) (sim/map.py:117-135): coarse pwv series interpolated linearly, float32
    trilinear table."""
    from maria_amd import map as mmap
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(4)
    t, az, el, off = _scan(D=20, T=1500)
    centre = _centre(az, el, None)
    eta = np.linspace(0.02, -0.02, 9)
    xi = np.linspace(-0.02, 0.02, 9)
    values = _blob_map(2, 1, 9, 9, eta, xi, rng)
    w = np.ones((len(off), 1)) * 0.5
    axis_T = np.array([250.0, 270.0, 290.0])
    axis_pwv = np.linspace(0.0, 6.0, 13)
    axis_el = np.radians(np.linspace(20.0, 90.0, 15))
    tabs = [(1.5e10 + 4e9 * c) * np.exp(-(0.05 + 0.03 * c + 0.04 * axis_pwv[None, :, None]) / np.sin(axis_el)[None, None, :])
            * (axis_T[:, None, None] / 270.0) ** 0.2 for c in range(2)]
    ta = np.arange(t[0], t[-1], 0.5)
    coarse = 1.2 + 0.3 * np.cumsum(rng.normal(0, 0.05, (len(off), len(ta))), axis=1)
    T0 = 273.0
    az_d, el_d = hotpath.broadcast(off, az, el)
    ref = mapsample.sample_maps(az_d, el_d, t, ta, coarse, eta, xi, centre, values, w, cal_tables=tabs,
                                cal_axes=(axis_T, axis_pwv, axis_el), base_temperature=T0)
    collapsed = np.stack([mmap.collapse_temperature(tab, axis_T, T0) for tab in tabs])
    got = mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, cal_tables=collapsed, cal_axis_pwv=axis_pwv,
                          cal_axis_el=axis_el, coarse_pwv=coarse.T, ta0=ta[0], dta=0.5, t=t).cpu().numpy()
    bound = _rounding_bound(values, eta, xi, 0.5, 1e12 * mapsample.K_B * sum(tab.max() for tab in tabs))
    assert np.isfinite(ref).all() and np.abs(got - ref).max() <= bound + 2e-6 * np.abs(ref).max() and bound <= 3e-4 * np.abs(ref).max()
    # a pwv beyond the table gives NaN, as jax's interpolator does
    got2 = mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, cal_tables=collapsed, cal_axis_pwv=axis_pwv,
                           cal_axis_el=axis_el, coarse_pwv=coarse.T + 10.0, ta0=ta[0], dta=0.5, t=t).cpu().numpy()
    assert np.isnan(got2).all()


@pytest.mark.parametrize("atm_cal", [True, False], ids=["cal_tables", "cal_scalars"])
@pytest.mark.parametrize("rate,D,T,el_lo", [(400.0, 37, 3001, 20.0), (20.0, 37, 3001, 20.0), (400.0, 16, 4096, 20.0), (50.0, 5, 1027, 60.0)],
                         ids=["400Hz", "20Hz_per_sample_branch", "whole_tiles", "off_the_axis"])
def test_map_field_written_in_krj_equals_sampling_then_tod_to_krj(gpu_ctx, atm_cal, rate, D, T, el_lo):
    """mrx_map_sample_krj -- the gain and TOD.to("K_RJ") on the sampler's own store -- against mrx_map_sample, a float32
    multiply by the gain and mrx_tod_to_krj in turn: bit for bit (its header's promise), with and without the sampler's
    own atmospheric calibration, two bands, ragged rows and samples, a scan slow enough that a thread's four samples are
    not on a line in elevation (the per-sample branch of the division) and an elevation axis the detectors leave (NaN as
    jax fills, in the same samples)."""
    import torch

    from maria_amd import map as mmap
    from maria_amd import synthetic
    from maria_amd._lib import ptr

    rng = np.random.default_rng(int(rate) + D + T)
    t = 1.7e9 + np.arange(T) / rate
    az, el = synthetic.daisy_scan(t)
    az, el = az.astype(np.float32), el.astype(np.float32)
    off = synthetic.hex_pack(max(D, 2), np.radians(0.4))[:D]
    centre = _centre(az, el, None)
    eta, xi = np.linspace(0.02, -0.02, 9), np.linspace(-0.02, 0.02, 9)
    values = _blob_map(2, 1, 9, 9, eta, xi, rng)
    w = np.ones((D, 1)) * 0.5
    kw = {}
    if atm_cal:
        axis_pwv, axis_el_s = np.linspace(0.0, 6.0, 13), np.radians(np.linspace(20.0, 90.0, 15))
        tabs = np.stack([(1.5e10 + 4e9 * c) * np.exp(-(0.05 + 0.03 * c + 0.04 * axis_pwv[:, None]) / np.sin(axis_el_s)[None, :]) for c in range(2)])
        ta = np.arange(t[0], t[-1] + 1.0, 0.5)
        coarse = 1.2 + 0.3 * np.cumsum(rng.normal(0, 0.05, (D, len(ta))), axis=1)
        kw = dict(cal_tables=tabs.astype(np.float32), cal_axis_pwv=axis_pwv, cal_axis_el=axis_el_s, coarse_pwv=coarse.T, ta0=ta[0], dta=0.5, t=t)
    else:
        kw = dict(cal_scalars=[1.0e10, 1.3e10])
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    # TOD.to("K_RJ")'s tables: an elevation axis with a node at 60 degrees, the middle of the scan (rows straddle it), starting
    # at el_lo degrees (60: the detectors below the boresight's lowest point are off the axis)
    n_el, n_bands = 29, 2
    axis = np.radians(np.linspace(el_lo, 90.0, n_el))
    den = np.stack([(2.0e-2 + 5e-3 * b) * np.exp(-(0.04 + 0.02 * b) / np.sin(axis)) for b in range(n_bands)])
    krj = dict(bore_el=f32(el), dx=f32(off[:, 0]), dy=f32(off[:, 1]), band=torch.as_tensor(rng.integers(0, n_bands, D).astype(np.int32)).to(dev),
               axis=f32(axis), values=f32(den))
    scale = f32(rng.uniform(0.9, 1.1, D))
    for sc in (scale, None):
        ref = mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, **kw)
        if sc is not None:
            ref *= sc[:, None]
        gpu_ctx.call("mrx_tod_to_krj", ptr(ref), ref.stride(0), D, T, None, None, ptr(krj["bore_el"]), ptr(krj["dx"]), ptr(krj["dy"]),
                     ptr(krj["band"]), ptr(krj["axis"]), ptr(krj["values"]), n_el, n_bands)
        got = mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, krj=krj, scale=sc, **kw)
        nan_ref, nan_got = torch.isnan(ref), torch.isnan(got)
        assert torch.equal(nan_ref, nan_got)
        assert torch.equal(torch.where(nan_ref, torch.zeros_like(ref), ref), torch.where(nan_got, torch.zeros_like(got), got)), \
            float((got - ref).abs().nan_to_num().max())
        assert bool(torch.isfinite(ref).any()) and float(ref[torch.isfinite(ref)].abs().max()) > 0
        if el_lo > 20.0:
            assert bool(nan_ref.any()) and not bool(nan_ref.all())  # the case is what it says: some samples off the axis
        else:
            assert not bool(nan_ref.any())


@pytest.mark.parametrize("frame", ["az/el", "ra/dec"])
def test_simulation_with_map(gpu_ctx, frame):
    """Simulation(map=...): tod.data["map"] against the oracle chain fed with the same
    smoothed map, calibration tables and coarse pwv; two bands, one map channel each."""
    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site, compute_angular_fwhm
    from maria_amd.sim import Plan, Simulation, sky_transform_stack
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(8)
    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093"), Band(center=150e9, width=41e9, shape="top_hat", name="f150")]
    inst = Instrument(Detectors.hexagon(19, 0.25, bands, primary_size=6.0))
    plan = Plan.daisy(start_time=1.7e9, duration=40.0, sample_rate=50.0, scan_center=(120.0, 50.0), radius=0.3, speed=0.3)
    site = Site(altitude=1000.0, latitude=-23.0, longitude=-67.8)
    transform = sky_transform_stack(plan.time, site.latitude, site.longitude) if frame == "ra/dec" else None
    az32, el32 = plan.phi.astype(np.float32), plan.theta.astype(np.float32)
    centre = _centre(az32, el32, transform)
    n = 48
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    data = np.stack([np.exp(-((X - 0.2) ** 2 + (Y + 0.1) ** 2) / 0.1), 0.5 * np.exp(-((X + 0.3) ** 2 + Y**2) / 0.2)]).astype(np.float32)
    data += 0.02 * rng.normal(size=data.shape).astype(np.float32)
    sky = mmap.ProjectionMap(data, nu=[93e9, 150e9], width=1.2, center=np.degrees(centre), frame=frame, degrees=True)
    sim = Simulation(inst, plan, site, atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 5}, map=sky, noise=False)
    (tod,) = sim.run(units="pW")
    assert set(tod.fields) == {"atmosphere", "map"}
    got = tod.data["map"]
    obs = sim.obs_list[0]
    atm, dets = obs.atmosphere, inst.dets
    path = atm._device_path()
    coarse = path.coarse_pwv().cpu().numpy()  # [D, Ta]
    ta = path.ta0 + path.dta * np.arange(coarse.shape[1])
    az_d, el_d = hotpath.broadcast(obs.coords.offsets, az32, el32)
    ref = np.zeros_like(got)
    sp = atm.spectrum
    for b, band in enumerate(bands):
        rows = np.nonzero(dets.band_index == b)[0]
        fwhm = float(compute_angular_fwhm(fwhm_0=dets.primary_size.mean(), z=np.inf, nu=band.center))
        sigma_pix = fwhm / np.sqrt(8 * np.log(2)) / abs(sky.x_res)
        smoothed = hotpath.map_smooth(sky.data, None, sigma_pix, sigma_pix)
        smoothed = np.asarray(smoothed[0] if isinstance(smoothed, tuple) else smoothed, np.float32)
        chans, tabs = [], []
        for c, (lo, hi) in enumerate(sky.nu_bin_bounds):
            if band.nu.max() < lo or hi < band.nu.min():
                continue
            mask = (sp.side_nu >= lo) & (sp.side_nu < hi)
            chans.append(c)
            tabs.append(np.trapezoid(band.passband(sp.side_nu[mask]) * np.exp(-sp._opacity[..., mask]), x=sp.side_nu[mask], axis=-1))
        values = np.swapaxes(smoothed[:, chans], 0, 1)
        ref[rows] = mapsample.sample_maps(az_d[rows], el_d[rows], plan.time, ta, coarse[rows], sky.eta, sky.xi, sky.center, values,
                                          mapsample.mueller_row(dets.gamma[rows])[:, :1], cal_tables=tabs,
                                          cal_axes=(sp.side_base_temperature, sp.side_zenith_pwv, sp.side_elevation),
                                          base_temperature=atm.weather.temperature[0], transform_stack=transform)
    assert np.isfinite(got).all() and np.abs(ref).max() > 0
    bound = _rounding_bound(sky.data, sky.eta, sky.xi, 1.0, 1e12 * mapsample.K_B * 5e10)
    assert np.abs(got - ref).max() <= bound + 1e-5 * np.abs(ref).max(), (np.abs(got - ref).max(), bound, np.abs(ref).max())
    # default units: the same per-sample factor as the atmosphere field (tod/tod.py:130-136)
    sim2 = Simulation(inst, plan, site, atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 5}, map=sky, noise=False)
    (tod2,) = sim2.run()
    factor = tod2.data["atmosphere"].astype(np.float64) / tod.data["atmosphere"]
    np.testing.assert_allclose(tod2.data["map"], got * factor, rtol=3e-6, atol=1e-7 * np.abs(got * factor).max())
    # ... and written in K_RJ by the sampler itself (mrx_map_sample_krj): the very bits of the pW field converted afterwards
    np.testing.assert_array_equal(tod2.data["map"], tod.to("K_RJ").data["map"])


@pytest.mark.parametrize("shape,chunked,bilinear", [((12, 16), False, 0), ((70, 150), False, 0), ((70, 150), True, 0), ((33, 64), True, 0),
                                                    ((12, 16), False, 1), ((70, 150), True, 1), ((33, 64), False, 1)])
def test_bucketed_bin_map_matches_the_atomic_form_and_the_oracle(gpu_ctx, shape, chunked, bilinear):
    """mrx_bin_map_bucketed: samples routed to map regions of 64 x 32 pixels and summed in LDS.
    Against mrx_bin_map on the same inputs (float64 rounding: the order of the sums differs) --
    weights, two Stokes planes, two channels, samples beyond the grid, a detector count that is
    no multiple of 16, a length that is no multiple of 1024, maps of one and of several regions,
    and the time axis walked in chunks of one column of tiles (the minimum work buffer); nearest
    pixel (tiles of 16 detectors x 1024 samples) and bilinear (8 x 256, four corners a sample)."""
    import ctypes as C

    import torch

    from maria_amd._lib import MrxSkyMap, ptr
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(16)
    t, az, el, off = _scan(D=37, T=3301, fov_deg=0.5)
    transform = _sky_rotation(t)
    centre = _centre(az, el, transform)
    az_d, el_d = hotpath.broadcast(off, az, el)
    ox = mapsample.phi_theta_to_offsets(*mapsample.frame_angles(az_d, el_d, transform), *centre)
    n_eta, n_xi = shape
    half_eta, half_xi = 0.8 * float(np.abs(ox[..., 1]).max()), 0.8 * float(np.abs(ox[..., 0]).max())
    eta = np.linspace(half_eta, -half_eta, n_eta)
    xi = np.linspace(-half_xi, half_xi, n_xi)
    tod = rng.normal(1.0, 0.5, (len(off), len(t))).astype(np.float32)
    wts = rng.uniform(0.5, 2.0, tod.shape).astype(np.float32)
    gamma = np.where(np.arange(len(off)) % 2 == 0, np.nan, rng.uniform(0, np.pi, len(off)))
    sw = mapsample.mueller_row(gamma)[:, :2]
    chan = (np.arange(len(off)) % 2).astype(np.int32)
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    d_tod, d_w, d_az, d_el, d_dx, d_dy = f32(tod), f32(wts), f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    d_sw = torch.as_tensor(np.ascontiguousarray(sw, np.float64)).to(dev)
    d_tr = torch.as_tensor(transform.reshape(-1, 9)).to(dev)
    d_chan = torch.as_tensor(chan).to(dev)
    sky = MrxSkyMap(None, 2, 2, n_eta, n_xi, float(eta[0]), float(eta[1] - eta[0]), float(xi[0]), float(xi[1] - xi[0]),
                    centre[0], centre[1], bilinear, 0)
    args = (C.byref(sky), ptr(d_tod), d_tod.stride(0), ptr(d_w), d_w.stride(0), ptr(d_az), ptr(d_el), len(t),
            ptr(d_tr), ptr(d_dx), ptr(d_dy), ptr(d_sw), ptr(d_chan), len(off))
    ref = [torch.zeros((2, 2, n_eta, n_xi), dtype=torch.float64, device=dev) for _ in range(2)]
    gpu_ctx.call("mrx_bin_map", *args, ptr(ref[0]), ptr(ref[1]))
    lo, full = C.c_size_t(), C.c_size_t()
    assert gpu_ctx.lib.mrx_bin_map_work_bytes(C.byref(sky), len(off), len(t), C.byref(lo), C.byref(full)) == 0
    assert full.value == (13 if bilinear else 4) * lo.value  # ceil(3301 / 256) or ceil(3301 / 1024) columns of tiles
    work = torch.empty(lo.value if chunked else full.value, dtype=torch.uint8, device=dev)
    got = [torch.full((2, 2, n_eta, n_xi), 1.0, dtype=torch.float64, device=dev) for _ in range(2)]  # adds to what is there
    gpu_ctx.call("mrx_bin_map_bucketed", *args, ptr(got[0]), ptr(got[1]), ptr(work), work.numel())
    for g, r in zip(got, ref):
        g, r = g.cpu().numpy() - 1.0, r.cpu().numpy()
        assert np.abs(r).max() > 0 and np.abs(g - r).max() <= 1e-12 * np.abs(r).max()
    ref_sum, ref_wgt = mapsample.bin_map(az_d, el_d, tod, wts, eta, xi, centre, sw, 2, channel=chan, n_channels=2,
                                         transform_stack=transform, bilinear=bool(bilinear))
    assert abs(float(got[1].sum() - got[1].numel()) / ref_wgt.sum() - 1) < 1e-9  # every sample lands somewhere
    # what the bucketed form does not take: more than 2048 regions
    big = MrxSkyMap(None, 2, 2, 4096, 4096, 1.0, -1e-3, -1.0, 1e-3, centre[0], centre[1], 0, 0)
    assert gpu_ctx.lib.mrx_bin_map_work_bytes(C.byref(big), len(off), len(t), C.byref(lo), C.byref(full)) != 0


@pytest.mark.parametrize("bilinear", [False, True])
def test_bin_map_matches_oracle(gpu_ctx, bilinear, pointing_mode):
    """mrx_bin_map (BinMapper.run, mappers/bin_mapper.py:84-120): the transpose of the pointing
    matrix as float64 atomics.  Random TOD and weights, two Stokes planes, two channels, samples
    beyond the grid (clamped to the edge pixels, as the reference's clip does)."""
    import ctypes as C

    import torch

    from maria_amd._lib import MrxSkyMap, ptr
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(6)
    t, az, el, off = _scan(D=40, T=2051, fov_deg=0.5)
    transform = _sky_rotation(t)
    centre = _centre(az, el, transform)
    az_d, el_d = hotpath.broadcast(off, az, el)
    ox = mapsample.phi_theta_to_offsets(*mapsample.frame_angles(az_d, el_d, transform), *centre)
    n_eta, n_xi = 12, 16
    half_eta, half_xi = 0.8 * float(np.abs(ox[..., 1]).max()), 0.8 * float(np.abs(ox[..., 0]).max())
    eta = np.linspace(half_eta, -half_eta, n_eta)
    xi = np.linspace(-half_xi, half_xi, n_xi)
    tod = rng.normal(1.0, 0.5, (len(off), len(t))).astype(np.float32)
    wts = rng.uniform(0.5, 2.0, tod.shape).astype(np.float32)
    gamma = np.where(np.arange(len(off)) % 2 == 0, np.nan, rng.uniform(0, np.pi, len(off)))
    sw = mapsample.mueller_row(gamma)[:, :2]
    chan = (np.arange(len(off)) % 2).astype(np.int32)
    ref_sum, ref_wgt = mapsample.bin_map(az_d, el_d, tod, wts, eta, xi, centre, sw, 2, channel=chan, n_channels=2,
                                         transform_stack=transform, bilinear=bilinear)
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    d_tod, d_w, d_az, d_el, d_dx, d_dy = f32(tod), f32(wts), f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    d_sw = torch.as_tensor(np.ascontiguousarray(sw, np.float64)).to(dev)
    d_tr = torch.as_tensor(transform.reshape(-1, 9)).to(dev)
    d_chan = torch.as_tensor(chan).to(dev)
    msum = torch.zeros((2, 2, n_eta, n_xi), dtype=torch.float64, device=dev)
    mwgt = torch.zeros_like(msum)
    sky = MrxSkyMap(None, 2, 2, n_eta, n_xi, float(eta[0]), float(eta[1] - eta[0]), float(xi[0]), float(xi[1] - xi[0]),
                    centre[0], centre[1], 1 if bilinear else 0, 0)
    gpu_ctx.call("mrx_bin_map", C.byref(sky), ptr(d_tod), d_tod.stride(0), ptr(d_w), d_w.stride(0), ptr(d_az), ptr(d_el), len(t),
                 ptr(d_tr), ptr(d_dx), ptr(d_dy), ptr(d_sw), ptr(d_chan), len(off), ptr(msum), ptr(mwgt))
    got_sum, got_wgt = msum.cpu().numpy(), mwgt.cpu().numpy()
    assert ref_wgt.sum() > 0 and abs(got_wgt.sum() / ref_wgt.sum() - 1) < 1e-9  # every sample lands somewhere
    if bilinear:
        # weights move by the float32 rounding of the offsets (<= 6e-7 rad of a ~1e-3 rad pixel)
        tol = 6e-7 / abs(xi[1] - xi[0]) * 4
        assert np.abs(got_sum - ref_sum).max() <= tol * np.abs(ref_sum).max()
        assert np.abs(got_wgt - ref_wgt).max() <= tol * np.abs(ref_wgt).max()
    else:
        # nearest pixel: a handful of samples within float32 rounding of a pixel edge may swap bins
        moved = np.abs(got_wgt - ref_wgt).sum() / ref_wgt.sum()
        assert moved < 1e-3, moved
        # pixels no sample moved into or out of, judged on the I plane (a polarised detector's Q
        # weight can be arbitrarily small): there the sums agree to float64 rounding
        ok = np.broadcast_to((np.abs(got_wgt - ref_wgt) <= 1e-9 * ref_wgt.max())[:1], ref_wgt.shape)
        assert ok.mean() > 0.9 and np.abs(got_sum - ref_sum)[ok].max() <= 1e-9 * np.abs(ref_sum).max()


def test_bin_mapper_recovers_the_sampled_map(gpu_ctx):
    """Simulation(map=...) then BinMapper: binning the noiseless map TOD gives back the
    beam-smoothed input where the scan covered it (the round trip the reference's
    tests/mappers exercise), and sum / weight bookkeeping is consistent."""
    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.mappers import BinMapper
    from maria_amd.sim import Plan, Simulation, sky_transform_stack

    bands = [Band(center=150e9, width=30e9, shape="top_hat", name="f150")]
    inst = Instrument(Detectors.hexagon(61, 0.3, bands, primary_size=30.0))
    plan = Plan.daisy(start_time=1.7e9, duration=120.0, sample_rate=50.0, scan_center=(100.0, 60.0), radius=0.25, speed=0.5)
    site = Site(altitude=1000.0)
    transform = sky_transform_stack(plan.time, site.latitude, site.longitude)
    centre = _centre(plan.phi.astype(np.float32), plan.theta.astype(np.float32), transform)
    n = 96
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    data = (np.exp(-((X - 0.1) ** 2 + (Y + 0.15) ** 2) / 0.05) + 0.5 * np.exp(-((X + 0.3) ** 2 + (Y - 0.2) ** 2) / 0.02)).astype(np.float32)
    sky = mmap.ProjectionMap(data, nu=150e9, width=1.0, center=np.degrees(centre), frame="ra/dec")
    sim = Simulation(inst, plan, site, map=sky, noise=False)
    (tod,) = sim.run(units="pW")
    assert set(tod.fields) == {"map"} and np.isfinite(tod.data["map"]).all()
    mapper = BinMapper([tod], center=np.degrees(centre), width=0.6, resolution=1.0 / 60, stokes="I", nu=150e9, frame="ra/dec", units="pW")
    with pytest.raises(RuntimeError):
        _ = mapper.map
    out = mapper.run()
    hit = mapper.products["weight"][0, 0] > 20
    assert hit.mean() > 0.3
    np.testing.assert_allclose(mapper.products["sum"][0, 0][hit] / mapper.products["weight"][0, 0][hit], out.data[0, 0][hit], rtol=1e-6)
    # compare with the input map (pW per K_RJ is a constant without an atmosphere) on the hit pixels
    scale = 1e12 * 1.380649e-23 * float(np.trapezoid(bands[0].passband(bands[0].nu), x=bands[0].nu)) * 0.5 * 2  # mueller I weight = 1
    from scipy.interpolate import RegularGridInterpolator

    interp = RegularGridInterpolator((sky.eta[::-1], sky.xi), sky.data[0, 0][::-1], bounds_error=False, fill_value=np.nan)
    E, Xg = np.meshgrid(out.eta, out.xi, indexing="ij")
    expect = interp(np.stack([E, Xg], axis=-1)) * scale
    rec = out.data[0, 0]
    good = hit & np.isfinite(expect)
    # beam smoothing (a 30 m dish at 150 GHz: ~17 arcsec) and 1-arcmin nearest-pixel binning blur the blobs a little
    assert np.corrcoef(rec[good], expect[good])[0, 1] > 0.99
    assert abs(np.sum(rec[good]) / np.sum(expect[good]) - 1) < 0.05


def test_recover_map_as_the_reference_asserts_it(gpu_ctx):
    """The one numeric pin the reference's tests hold on the map chain (maria/tests/map/test_recover_map.py:15-69): 300
    positions x 3 bands (90 / 150 / 220 GHz) behind a beam-free dish (primary_size 1000 m), field of view = half the
    map, a 60 s daisy of radius width / 3 at 50 Hz, no noise, no atmosphere; the TOD (default units K_RJ) binned by
    BinMapper at the input map's centre, width and resolution; and, per band,
        sqrt(nansum(w (m1 - m0)^2) / nansum(w)) < 1e-3   [K_RJ]
    between the binned map m1 and the input m0.  The reference's input is a downloaded cluster map (mean subtracted);
    here a synthetic one of the same kind -- a beta-model decrement of 5 mK with arcminute-scale structure, mean
    subtracted -- and, since a bound of 1e-3 K says little about a 5-mK map, the same residual is also held below 1 %
    of the map's peak."""
    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.mappers import BinMapper
    from maria_amd.sim import Plan, Simulation, sky_transform_stack

    bands = [Band(center=90e9, width=30e9, name="f090"), Band(center=150e9, width=40e9, name="f150"), Band(center=220e9, width=50e9, name="f220")]
    n, width = 128, 1.0  # degrees
    res = width / (n - 1)
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    rng = np.random.default_rng(8)
    field = np.fft.irfft2(np.fft.rfft2(rng.standard_normal((n, n))) * np.exp(-0.5 * (np.hypot(*np.meshgrid(np.fft.rfftfreq(n), np.fft.fftfreq(n))) * 12.0) ** 2), s=(n, n))
    data = -5e-3 * (1 + ((X - 0.1) ** 2 + (Y + 0.05) ** 2) / 0.04) ** -1.0 + 4e-4 * field / field.std()
    data = (data - data.mean()).astype(np.float32)
    inst = Instrument(Detectors.hexagon(300, width / 2, bands, primary_size=1000.0))
    plan = Plan.daisy(start_time=1.7e9, duration=60.0, sample_rate=50.0, scan_center=(120.0, 55.0), radius=width / 3, speed=0.5)
    site = Site(altitude=5190.0)
    centre = _centre(plan.phi.astype(np.float32), plan.theta.astype(np.float32), sky_transform_stack(plan.time, site.latitude, site.longitude))
    sky = mmap.ProjectionMap(data, nu=150e9, width=width, center=np.degrees(centre), frame="ra/dec")
    sim = Simulation(inst, plan, site, map=sky, noise=False)
    (tod,) = sim.run()
    assert tod.units == "K_RJ" and set(tod.fields) == {"map"}
    # the input map's own grid: n pixels of its resolution around its centre
    mapper = BinMapper([tod], center=np.degrees(centre), width=(n + 0.5) * res, resolution=res, stokes="I",
                       nu=[b.center for b in bands], frame="ra/dec", units="K_RJ")
    out = mapper.run()
    assert out.data.shape[-2:] == (n, n) and np.allclose(out.xi, sky.xi, atol=1e-12) and np.allclose(out.eta, sky.eta, atol=1e-12)
    m0 = sky.data[0, 0]
    m1 = out.data[0, :]                      # [band, eta, xi]
    w = mapper.products["weight"][0, -1]     # (the reference weighs every band with the last channel's hits)
    assert (w > 0).mean() > 0.5
    relsqres = np.sqrt(np.nansum(w * (m1 - m0) ** 2, axis=(-1, -2)) / np.nansum(w))
    print("weighted rms residual per band [K_RJ]:", relsqres, "map peak", np.abs(m0).max())
    assert relsqres.shape == (3,) and np.all(relsqres < 1e-3)       # the reference's assertion
    assert np.all(relsqres < 0.01 * np.abs(m0).max())              # ... and one that a 5-mK map can fail


def test_end_to_end_polarised_multifrequency_pipeline(gpu_ctx):
    """The shape of the reference's end-to-end tests (tests/sim/test_pipeline.py:21-54,
    test_polarization.py, test_multifrequency.py): atmosphere + map + noise in the default
    units, polarised detectors on an IQU map with one plane per band, then a BinMapper; their
    assertions: no NaN anywhere, and the mapper's weight sums to something positive."""
    from maria_amd import map as mmap
    from maria_amd import synthetic
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.mappers import BinMapper
    from maria_amd.sim import Plan, Simulation, sky_transform_stack

    rng = np.random.default_rng(11)
    bands = [Band(center=90e9, width=30e9, name="f090", NEP=3e-17, knee=1.0, gain_error=0.02),
             Band(center=150e9, width=40e9, name="f150", NEP=4e-17, knee=1.0, gain_error=0.02)]
    pos = synthetic.hex_pack(30, np.radians(0.3))
    gamma = np.tile(np.where(np.arange(30) % 3 == 0, np.nan, rng.uniform(0, np.pi, 30)), 2)  # some unpolarised
    dets = Detectors(np.tile(pos, (2, 1)), bands, np.repeat([0, 1], 30), primary_size=12.0, gamma=gamma)
    inst = Instrument(dets)
    plan = Plan.daisy(start_time=1.71e9, duration=60.0, sample_rate=50.0, scan_center=(200.0, 55.0), radius=0.25, speed=0.4)
    site = Site(altitude=5000.0, latitude=-23.0, longitude=-67.8)
    centre = _centre(plan.phi.astype(np.float32), plan.theta.astype(np.float32), sky_transform_stack(plan.time, site.latitude, site.longitude))
    n = 64
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    blob = np.exp(-(X**2 + Y**2) / 0.1).astype(np.float32)
    data = np.stack([np.stack([(1 + c) * s * blob for c in range(2)]) for s in (1.0, 0.1, -0.05)])  # [IQU, nu, eta, xi]
    sky = mmap.ProjectionMap(data * 1e-3, nu=[90e9, 150e9], stokes="IQU", width=1.0, center=np.degrees(centre), frame="ra/dec")
    sim = Simulation(inst, plan, site, atmosphere="2d", atmosphere_kwargs={"n_layers": 3, "seed": 1}, map=sky, noise=True, noise_seed=2)
    (tod,) = sim.run()
    assert tod.units == "K_RJ" and set(tod.fields) == {"atmosphere", "map", "noise"}
    for field in tod.fields:
        assert np.isfinite(tod.data[field]).all(), field
    assert tod.data["map"].std() > 0 and tod.data["noise"].std() > 0
    # polarised detectors see Q and U: their map signal differs from the unpolarised neighbour's
    assert not np.allclose(tod.data["map"][1], tod.data["map"][0] * tod.data["map"][1].mean() / tod.data["map"][0].mean(), rtol=1e-3)
    mapper = BinMapper([tod], center=np.degrees(centre), width=0.8, resolution=1.0 / 30, stokes="IQU", nu=[90e9, 150e9], frame="ra/dec", units="K_RJ")
    out = mapper.run()
    w = mapper.products["weight"]
    assert w.shape == (3, 2, 24, 24) and w.sum() > 0 and (w[0] > 0).mean() > 0.3
    assert np.isfinite(out.data[w > 0]).all()


def test_default_units_without_an_atmosphere(gpu_ctx):
    """TOD.to("K_RJ") with spectrum = None (tod/tod.py:98-100): one transmission integral per
    band, Int passband dnu; the map-only run in K_RJ is the pW run divided by it."""
    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    band = Band(center=150e9, width=30e9, shape="top_hat", name="f150", NEP=2e-17, knee=0.0)
    inst = Instrument(Detectors.hexagon(19, 0.2, [band], primary_size=20.0))
    plan = Plan.daisy(start_time=1.7e9, duration=20.0, sample_rate=50.0, scan_center=(100.0, 60.0), radius=0.2, speed=0.4)
    n = 32
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    sky = mmap.ProjectionMap(np.exp(-(X**2 + Y**2) / 0.2).astype(np.float32), nu=150e9, width=1.0, center=(100.0, 60.0), frame="az/el")
    runs = {}
    for units in ("pW", "K_RJ"):
        sim = Simulation(inst, plan, Site(altitude=1000.0), map=sky, noise=True, noise_seed=5)
        (runs[units],) = sim.run(units=units)
    den = 1e12 * 1.380649e-23 * float(np.trapezoid(band.passband(band.nu), x=band.nu))  # unpolarised
    for field in ("map", "noise"):
        np.testing.assert_allclose(runs["K_RJ"].data[field], runs["pW"].data[field] / np.float32(den), rtol=2e-6)
    # a 1 K_RJ source comes back as ~1 K_RJ in the TOD (peak of the smoothed blob below 1)
    assert 0.5 < runs["K_RJ"].data["map"].max() <= 1.0


def test_map_entry_points_reject_bad_arguments(gpu_ctx):
    """Negative status and a message instead of a launch: the error behaviour of the map rows'
    entry points."""
    import ctypes as C

    import torch

    from maria_amd._lib import MrxError, MrxMapCal, MrxSkyMap, ptr

    dev = "cuda:0"
    vals = torch.zeros((1, 1, 4, 4), dtype=torch.float32, device=dev)
    one = torch.zeros(8, dtype=torch.float32, device=dev)
    one64 = torch.ones(8, dtype=torch.float64, device=dev)
    out = torch.zeros((2, 8), dtype=torch.float32, device=dev)
    cal = MrxMapCal()
    cal.d_scalar = ptr(one64)

    def call(sky, ld=8):
        gpu_ctx.call("mrx_map_sample", C.byref(sky), C.byref(cal), ptr(one), ptr(one), 8, None, ptr(one), ptr(one), ptr(one64), 2,
                     ptr(out), ld)

    good = MrxSkyMap(ptr(vals), 1, 1, 4, 4, 0.1, -0.05, -0.1, 0.05, 0.0, 1.0, 1, 0)
    call(good)
    for bad, msg in [(MrxSkyMap(ptr(vals), 1, 5, 4, 4, 0.1, -0.05, -0.1, 0.05, 0.0, 1.0, 1, 0), "n_stokes"),
                     (MrxSkyMap(ptr(vals), 1, 1, 4, 4, 0.1, 0.0, -0.1, 0.05, 0.0, 1.0, 1, 0), "non-zero step"),
                     (MrxSkyMap(None, 1, 1, 4, 4, 0.1, -0.05, -0.1, 0.05, 0.0, 1.0, 1, 0), "null")]:
        with pytest.raises(MrxError, match=msg):
            call(bad)
    with pytest.raises(MrxError, match="ld_out"):
        call(good, ld=4)
    with pytest.raises(MrxError, match="leading dimension"):
        gpu_ctx.call("mrx_bin_map", C.byref(good), ptr(out), 4, None, 0, ptr(one), ptr(one), 8, None, ptr(one), ptr(one), ptr(one64),
                     None, 2, ptr(one64), ptr(one64))


@pytest.mark.parametrize("D,T", [(1, 1), (1, 3), (3, 5), (17, 1025)])
def test_tiny_and_ragged_sizes(gpu_ctx, D, T):
    """One detector, one sample, lengths below the 4-sample group and one past the tile:
    map sampling and binning agree with the restatement (the 3-tap kernel reflects at both
    ends, as scipy does for a length-1 series too)."""
    import ctypes as C

    import torch

    from maria_amd import map as mmap
    from maria_amd import synthetic
    from maria_amd._lib import MrxSkyMap, ptr
    from oracle import hotpath, mapsample

    t = 1.7e9 + np.arange(T) / 50.0
    az, el = synthetic.daisy_scan(t)
    az, el = az.astype(np.float32), el.astype(np.float32)
    off = synthetic.hex_pack(max(D, 2), np.radians(0.2))[:D]
    centre = (float(az.mean()), float(el.mean()))
    eta, xi = np.linspace(0.01, -0.01, 8), np.linspace(-0.01, 0.01, 8)
    X, Y = np.meshgrid(xi, eta)
    values = (1.0 + 30 * X - 20 * Y).astype(np.float32)[None, None]
    w = np.ones((D, 1))
    az_d, el_d = hotpath.broadcast(off, az, el)
    ref = mapsample.sample_maps(az_d, el_d, t, None, None, eta, xi, centre, values, w, cal_scalars=[1e10])
    got = mmap.sample_map(gpu_ctx, values, eta, xi, centre, az, el, off, w, cal_scalars=[1e10]).cpu().numpy()
    assert got.shape == (D, T) and np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max()
    tod = np.random.default_rng(D + T).normal(size=(D, T)).astype(np.float32)
    rs, rw = mapsample.bin_map(az_d, el_d, tod, None, eta, xi, centre, w, 1)
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    d_tod, d_az, d_el, d_dx, d_dy = f32(tod), f32(az), f32(el), f32(off[:, 0]), f32(off[:, 1])
    d_sw = torch.ones((D, 1), dtype=torch.float64, device=dev)
    ms, mw = torch.zeros((1, 1, 8, 8), dtype=torch.float64, device=dev), torch.zeros((1, 1, 8, 8), dtype=torch.float64, device=dev)
    sky = MrxSkyMap(None, 1, 1, 8, 8, float(eta[0]), float(eta[1] - eta[0]), float(xi[0]), float(xi[1] - xi[0]), centre[0], centre[1], 0, 0)
    gpu_ctx.call("mrx_bin_map", C.byref(sky), ptr(d_tod), d_tod.stride(0), None, 0, ptr(d_az), ptr(d_el), T, None, ptr(d_dx), ptr(d_dy),
                 ptr(d_sw), None, D, ptr(ms), ptr(mw))
    assert abs(mw.sum().item() - D * T) < 1e-9 and abs(mw.cpu().numpy() - rw).sum() <= 2  # at most a boundary sample moved
