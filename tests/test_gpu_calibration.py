"""GPU tests of the follow-on rows built behind the path:
full-rate detector pointing and the pW -> K_RJ conversion fused into the TOD writer."""

import numpy as np
import pytest

from helpers import rel_err, small_problem

pytestmark = pytest.mark.gpu


def test_full_rate_pointing_matches_oracle(gpu_ctx):
    """mrx_pointing_broadcast vs coords/transforms.py:10-29 restated in numpy float32."""
    import torch

    from maria_amd._lib import ptr
    from oracle import hotpath

    rng = np.random.default_rng(0)
    D, T = 77, 4099  # neither a multiple of the tile
    az = np.cumsum(rng.normal(0, 1e-4, T)) + 0.8
    el = np.cumsum(rng.normal(0, 1e-4, T)) + 1.0
    off = rng.normal(0, 0.01, (D, 2))
    ref_az, ref_el = hotpath.broadcast(off, az, el)
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    ld = T + 1
    out_az = torch.full((D, ld), -9.0, dtype=torch.float32, device=dev)
    out_el = torch.full((D, ld), -9.0, dtype=torch.float32, device=dev)
    d_az, d_el, d_dx, d_dy = t(az), t(el), t(off[:, 0]), t(off[:, 1])  # keep the inputs alive
    gpu_ctx.call("mrx_pointing_broadcast", ptr(d_az), ptr(d_el), T, ptr(d_dx), ptr(d_dy), D, ptr(out_az), ptr(out_el), ld)
    ga, ge = out_az.cpu().numpy(), out_el.cpu().numpy()
    assert (ga[:, T:] == -9).all() and (ge[:, T:] == -9).all()
    # float32 trigonometry of two libraries: a few ulp at most
    assert np.abs(ga[:, :T] - ref_az).max() <= 6e-7 and np.abs(ge[:, :T] - ref_el).max() <= 6e-7


def test_pointing_round_trip_like_the_reference_test(gpu_ctx):
    """The reference's own numeric test of this transform (tests/coordinates/test_coordinates.py:
    7-19), on the HIP path: offsets -> (phi, theta) by mrx_pointing_broadcast about random centres
    over the whole sphere, back through phi_theta_to_offsets; mean squared error < 1e-5 (the
    reference's bound; float32 gives ~1e-13 away from the poles)."""
    import torch

    from maria_amd._lib import ptr
    from oracle import hotpath

    rng = np.random.default_rng(123)
    n = 256
    cphi = rng.uniform(0, 2 * np.pi, 5)
    ctheta = rng.uniform(-np.pi / 2, np.pi / 2, 5)
    az, el = np.repeat(cphi, 5), np.tile(ctheta, 5)  # the 25 centres as 25 "samples"
    offsets = np.radians(rng.uniform(-0.5, 0.5, (n, 2)))
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    d_az, d_el, d_dx, d_dy = t(az), t(el), t(offsets[:, 0]), t(offsets[:, 1])
    out_az = torch.empty((n, 25), dtype=torch.float32, device=dev)
    out_el = torch.empty((n, 25), dtype=torch.float32, device=dev)
    gpu_ctx.call("mrx_pointing_broadcast", ptr(d_az), ptr(d_el), 25, ptr(d_dx), ptr(d_dy), n, ptr(out_az), ptr(out_el), 25)
    phi, theta = out_az.cpu().numpy(), out_el.cpu().numpy()
    worst = 0.0
    for k in range(25):
        back = hotpath.phi_theta_to_offsets(phi[:, k], theta[:, k], np.float32(az[k]), np.float32(el[k]))
        mse = float(np.mean(np.square(offsets.astype(np.float32) - back)))
        worst = max(worst, mse)
        assert mse < 1e-5, (k, az[k], el[k], mse)
    assert worst < 1e-9  # what float32 actually delivers for centres at least ~1 deg from a pole


def _cal_tables(n_bands):
    """Synthetic transmission-integral tables on a (T, pwv, el) grid (band.py:248-252 shape)."""
    T = np.array([250.0, 270.0, 290.0])
    pwv = np.linspace(0.0, 10.0, 21)
    el = np.radians(np.linspace(10.0, 90.0, 33))
    el[-1] = np.radians(90.1)
    tables = []
    for b in range(n_bands):
        tau = (0.03 + 0.02 * b + (0.01 + 0.02 * b) * pwv[None, :, None]) / np.sin(np.minimum(el, np.pi / 2))[None, None, :]
        tables.append({"T": T, "pwv": pwv, "el": el, "values": (20e9 + 5e9 * b) * (T[:, None, None] / 270.0) ** 0.1 * np.exp(-tau)})
    return tables


def test_krj_upsample_matches_oracle(gpu_ctx):
    """mrx_spline_upsample_krj vs TOD.to("K_RJ") restated (tod/tod.py:106-142,
    calibration/functions.py:73-90): the oracle divides the oracle's pW TOD by the
    float32 trilinear transmission integral at the detectors' full-rate elevations."""
    import torch

    from maria_amd.pipeline import DevicePath
    from maria_amd import synthetic
    from oracle import hotpath

    p = small_problem(n_det=45, n_bands=2, n_layers=2, gain=True)
    az_full, el_full = synthetic.daisy_scan(p["t"])  # the boresight the coarse grid came from
    roll = np.radians(17.0)
    R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
    coords_offsets = p["offsets"] @ R.T  # observation.py:55-58
    tables = _cal_tables(2)
    T0r, pwvr = 273.15, 1.0
    polarized = [False, True]

    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.sample()
    path.prepare()
    path.set_calibration(tables, T0r, pwvr, el_full, coords_offsets, polarized)
    out = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
    path.upsample_krj(out)
    got = out.cpu().numpy()

    tod_pw = hotpath.run_path(p)
    _, el_det = hotpath.broadcast(coords_offsets, az_full, el_full)
    ref = hotpath.calibrate_to_krj(tod_pw, p["band_index"], tables, T0r, pwvr, el_det, polarized)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert rel_err(got, ref) <= 1e-5
    # the polarized band is divided by half the integral: exactly twice the unpolarized value
    path.set_calibration(tables, T0r, pwvr, el_full, coords_offsets, [False, False])
    out2 = torch.empty_like(out)
    path.upsample_krj(out2)
    got2 = out2.cpu().numpy()
    b1 = p["band_index"] == 1
    assert np.array_equal(got[b1], 2.0 * got2[b1]) and np.array_equal(got[~b1], got2[~b1])


def test_coarse_krj_form_stays_within_its_bound(gpu_ctx):
    """TOD.to("K_RJ") applied to the coarse loading before the spline (mrx_coarse_to_krj, then the
    pW writer) against the per-sample conversion in the writer (the reference's order): the two
    differ by the spline's interpolation error on the denominator alone, which
    DevicePath.coarse_krj_bound estimates on the host -- below a quarter of the parity tolerance
    for the daisy scan; the pipelined and the serial run agree bit for bit; and the form is refused
    near the zenith, off the table's axis and for a fast elevation slew."""
    import torch

    from maria_amd import synthetic
    from maria_amd.pipeline import DevicePath

    p = small_problem(n_det=300, n_bands=2, n_layers=2, gain=True)
    az_full, el_full = synthetic.daisy_scan(p["t"])
    roll = np.radians(17.0)
    R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
    tables = _cal_tables(2)
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.set_calibration(tables, 273.15, 1.0, el_full, p["offsets"] @ R.T, [False, True])
    bound = path.coarse_krj_bound()
    assert 0 < bound <= path.COARSE_KRJ_LIMIT, bound
    ref = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
    path.sample()
    path.prepare()
    path.upsample_krj(ref)
    got = path.run(torch.empty_like(ref), blocks=1, krj=True)
    piped = path.run(torch.empty_like(ref), blocks=3, krj=True)
    assert torch.equal(got, piped)
    dev = float(((got - ref).abs() / ref.abs()).max())
    assert dev <= bound, (dev, bound)
    assert dev > 0  # it IS another order of operations: not bit-identical
    # past the last knot the reference extrapolates its spline and divides by the true denominator: the coarse form does
    # NOT extrapolate loading / denominator there (an extrapolated cubic misses the denominator 50x worse than an
    # interior interval: 9e-5 in the last samples of a tight fast scan, found by scripts/fuzz_frontend.py) but divides
    # those samples one by one -- the per-sample writer on the last knots' window: the same values to rounding
    s0 = path._krj_split()
    assert 0 < s0 < path.T
    tail = float(((got[:, s0:] - ref[:, s0:]).abs() / ref[:, s0:].abs()).max())
    assert tail <= 5e-7, tail  # (measured 2e-7: a float32 ulp or two)
    # refused: a table whose axis the focal plane may leave, the zenith, an elevation slew of 1 deg per knot
    low = [dict(t, el=np.radians(np.linspace(59.5, 90.1, 33))) for t in tables]
    path.set_calibration(low, 273.15, 1.0, el_full, p["offsets"] @ R.T, [False, True])
    assert path.coarse_krj_bound() == float("inf")
    for el_a in (np.radians(np.linspace(80.0, 84.0, path.Ta)), np.radians(np.linspace(20.0, 20.0 + 1.0 * path.Ta, path.Ta) % 60 + 15)):
        q = DevicePath(dict(p, el_a=el_a), device="cuda:0", ctx=gpu_ctx)
        q.set_calibration(tables, 273.15, 1.0, el_full, p["offsets"] @ R.T, [False, True])
        assert not q.coarse_krj_bound() <= q.COARSE_KRJ_LIMIT


def test_coarse_to_krj_keeps_the_last_knots_aside(gpu_ctx):
    """mrx_coarse_to_krj_keep_tail: the same output as mrx_coarse_to_krj (also in place), and the last k steps of the
    INPUT -- the loading in pW, which the per-sample conversion past the last knot starts from -- in the caller's
    [k][ld] buffer, columns of a wider array included; bad arguments refused."""
    import torch

    from maria_amd._lib import ptr

    rng = np.random.default_rng(12)
    D, Ta, k, n_el, nb = 333, 77, 9, 21, 2
    dev = "cuda:0"
    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)  # noqa: E731
    loading = f32(rng.uniform(1.0, 2.0, (Ta, D)))
    el = f32(np.radians(55.0) + 0.02 * np.sin(np.arange(Ta) / 9.0))
    dx, dy = f32(rng.normal(0, 3e-3, D)), f32(rng.normal(0, 3e-3, D))
    band = torch.as_tensor(rng.integers(0, nb, D).astype(np.int32)).to(dev)
    axis = f32(np.radians(np.linspace(20.0, 90.1, n_el)))
    values = f32(rng.uniform(0.5, 1.5, (nb, n_el)))
    args = (ptr(el), ptr(dx), ptr(dy), ptr(band), ptr(axis), ptr(values), n_el, nb)
    ref = torch.empty_like(loading)
    gpu_ctx.call("mrx_coarse_to_krj", ptr(loading), D, Ta, *args, ptr(ref))
    assert torch.isfinite(ref).all() and not torch.equal(ref, loading)
    wide = torch.full((k, D + 40), -1.0, dtype=torch.float32, device=dev)  # the block's columns of a wider array
    out = torch.empty_like(loading)
    gpu_ctx.call("mrx_coarse_to_krj_keep_tail", ptr(loading), D, Ta, *args, ptr(out), ptr(wide[:, 25:]), k, wide.stride(0))
    assert torch.equal(out, ref)
    assert torch.equal(wide[:, 25 : 25 + D], loading[Ta - k :])
    assert (wide[:, :25] == -1).all() and (wide[:, 25 + D :] == -1).all()
    inplace = loading.clone()
    tail = torch.empty((Ta, D), dtype=torch.float32, device=dev)  # every step kept: k = Ta
    gpu_ctx.call("mrx_coarse_to_krj_keep_tail", ptr(inplace), D, Ta, *args, ptr(inplace), ptr(tail), Ta, D)
    assert torch.equal(inplace, ref) and torch.equal(tail, loading)
    gpu_ctx.call("mrx_coarse_to_krj_keep_tail", ptr(loading), D, Ta, *args, ptr(out), None, 0, 0)  # no tail asked for
    assert torch.equal(out, ref)
    for bad in ((ptr(tail), Ta + 1, D), (ptr(tail), -1, D), (ptr(tail), k, D - 1)):
        with pytest.raises(RuntimeError):
            gpu_ctx.call("mrx_coarse_to_krj_keep_tail", ptr(loading), D, Ta, *args, ptr(out), *bad)


@pytest.mark.parametrize("el_range_deg", [(25.0, 80.0), (50.0, 52.2), (50.0, 51.5), (84.0, 89.5)])
def test_krj_conversion_under_an_elevation_slew(gpu_ctx, el_range_deg):
    """The K_RJ kernels model a detector's elevation per 1024-sample tile as linear in the
    boresight elevation (fitted over +-0.02 rad); a tile whose boresight sweeps farther -- a fast
    elevation slew at a low sample rate -- must take the full formula per sample, and so must
    one that ends near the zenith.  One tile of 1000 samples over 55 deg, over 2.2 deg (just past
    the model's range), over 1.5 deg (inside it) and up to 89.5 deg, wide focal plane."""
    import torch

    from maria_amd.pipeline import DevicePath
    from oracle import hotpath

    p = small_problem(n_det=45, n_bands=2, n_layers=1)
    T = len(p["t"])
    rng = np.random.default_rng(3)
    coords_offsets = np.radians(rng.uniform(-1.0, 1.0, (45, 2)))  # a 2 deg focal plane
    el_slew = np.radians(np.linspace(el_range_deg[0], el_range_deg[1], T))
    tables = _cal_tables(2)
    T0r, pwvr = 273.15, 1.0
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.set_calibration(tables, T0r, pwvr, el_slew, coords_offsets, [False, True])
    data = torch.ones((path.D, path.T), dtype=torch.float32, device="cuda:0")
    path.to_krj(data)
    got = data.cpu().numpy()
    _, el_det = hotpath.broadcast(coords_offsets, np.zeros(T), el_slew)
    ref = hotpath.calibrate_to_krj(np.ones((45, T), np.float32), p["band_index"], tables, T0r, pwvr, el_det, [False, True])
    ok = np.isfinite(ref)  # beyond the table's last node (90.1 deg) both give NaN
    assert np.array_equal(np.isfinite(got), ok)
    assert np.abs(got[ok] / ref[ok] - 1).max() <= 1e-5
    back = path.from_krj(data).cpu().numpy()
    assert np.abs(back[ok] - 1).max() <= 2e-6


def test_run_default_units_are_krj(gpu_ctx):
    """Simulation.run() with the reference's default units against the oracle chain."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation
    from oracle import hotpath

    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093"), Band(center=150e9, width=41e9, shape="top_hat", name="f150")]
    inst = Instrument(Detectors.hexagon(37, 0.3, bands, primary_size=6.0))
    plan = Plan.daisy(start_time=1.7e9, duration=30.0, sample_rate=50.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
    sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 3, "seed": 2}, noise=False)
    (tod,) = sim.run()
    assert tod.units == "K_RJ"
    k = tod.data["atmosphere"]
    obs = sim.obs_list[0]
    atm, dets = obs.atmosphere, inst.dets
    # same realisation in pW through the plain writer
    path = atm._device_path()
    pw = path.run().cpu().numpy()
    sp = atm.spectrum
    tables = [{"T": sp.side_base_temperature, "pwv": sp.side_zenith_pwv, "el": sp.side_elevation,
               "values": hotpath.transmission_integral_grid(b.passband, sp.side_nu, sp._opacity)} for b in dets.bands]
    _, el_det = hotpath.broadcast(obs.coords.offsets, obs.boresight.az, obs.boresight.el)
    ref = hotpath.calibrate_to_krj(pw, dets.band_index, tables, tod.metadata["base_temperature"], tod.metadata["pwv"], el_det)
    assert rel_err(k, ref) <= 1e-5
    assert 1.0 < np.median(k) < 300.0  # Rayleigh-Jeans kelvin of a ~1 mm pwv sky


def test_standalone_krj_matches_fused_and_oracle(gpu_ctx):
    """mrx_tod_to_krj (TOD.to("K_RJ") of a full-rate field in place) on the pW TOD gives
    what the fused writer gives and what the oracle gives, for ragged sizes and a padded
    leading dimension."""
    import torch

    from maria_amd.pipeline import DevicePath
    from maria_amd import synthetic
    from oracle import hotpath

    p = small_problem(n_det=45, n_bands=2, n_layers=2)
    az_full, el_full = synthetic.daisy_scan(p["t"])
    tables = _cal_tables(2)
    polarized = [True, False]
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.sample()
    path.prepare()
    path.set_calibration(tables, 280.0, 2.5, el_full, p["offsets"], polarized)
    fused = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
    path.upsample_krj(fused)
    for pad in (0, 3):  # pad = 3: rows not 16-byte aligned -> scalar path
        buf = torch.full((path.D, path.T + pad), 7.0, dtype=torch.float32, device="cuda:0")
        pw = buf[:, : path.T]
        path.upsample(pw)
        pw_host = pw.cpu().numpy()
        path.to_krj(pw)
        got = pw.cpu().numpy()
        assert bool((buf[:, path.T :] == 7.0).all())
        # same lookup, one more float32 rounding (the pW value is stored before the division)
        assert rel_err(got, fused.cpu().numpy()) <= 3e-7
        _, el_det = hotpath.broadcast(p["offsets"], az_full, el_full)
        ref = hotpath.calibrate_to_krj(pw_host, p["band_index"], tables, 280.0, 2.5, el_det, polarized)
        assert rel_err(got, ref) <= 1e-5


def test_noise_field_in_krj_and_loading_dependent_nep(gpu_ctx):
    """Simulation(noise=True).run() in the default units: the noise field is converted with
    the same per-sample factor as the atmosphere (tod/tod.py:130-136), and a band with
    NEP_per_loading draws its noise from the pW loading (sim/noise.py:35-37)."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    def run(units, npl):
        bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093", NEP=2e-17, knee=0.0, NEP_per_loading=npl)]
        inst = Instrument(Detectors.hexagon(40, 0.3, bands, primary_size=6.0))
        plan = Plan.daisy(start_time=1.7e9, duration=40.0, sample_rate=100.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
        sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 3},
                         noise=True, noise_seed=11)
        (tod,) = sim.run(units=units)
        return tod

    pw, krj = run("pW", 0.0), run("K_RJ", 0.0)
    # the atmosphere is converted on the coarse grid here (DevicePath.coarse_krj_bound <= 3e-6), the
    # noise sample by sample: the two factors agree to that bound
    factor = krj.data["atmosphere"].astype(np.float64) / pw.data["atmosphere"]
    np.testing.assert_allclose(krj.data["noise"], pw.data["noise"] * factor, rtol=4e-6)
    # loading-dependent NEP: white noise of standard deviation sqrt(fs) 1e12 (NEP + npl L)
    npl = 1e-18
    pw2, krj2 = run("pW", npl), run("K_RJ", npl)
    amp = 1e12 * (2e-17 + npl * pw2.data["atmosphere"].astype(np.float64))
    z = pw2.data["noise"] / (np.sqrt(100.0) * amp)
    assert abs(z.std() - 1) < 0.01 and abs(z.mean()) < 0.01
    assert amp.mean() > 1.2 * 1e12 * 2e-17  # the loading term matters in this configuration
    # same seeds: the unit-variance draw is the one of the run without the loading term
    np.testing.assert_allclose(z, pw.data["noise"] / (np.sqrt(100.0) * 1e12 * 2e-17), atol=2e-5)
    # deferred conversion (pW first, then both fields in place, sample by sample) against the
    # fused one (on the coarse grid): within the bound of the coarse form
    np.testing.assert_allclose(krj2.data["atmosphere"], krj.data["atmosphere"], rtol=3e-6)
    np.testing.assert_allclose(krj2.data["noise"], pw2.data["noise"] * factor, rtol=4e-6)


def test_tod_to_round_trip_like_the_reference_noise_test(gpu_ctx):
    """tests/noise/test_noise.py:14: ``sim.run()[0].to("pW")`` -- the default-unit TOD converted
    back to pW (mrx_tod_from_krj) equals the pW run of the same realisation; and forth again."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093", knee=0.0), Band(center=150e9, width=41e9, shape="top_hat", name="f150", knee=0.0)]
    inst = Instrument(Detectors.hexagon(19, 0.3, bands, primary_size=6.0))
    plan = Plan.daisy(start_time=1.7e9, duration=20.0, sample_rate=50.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)

    def run(units):
        sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 9}, noise=True, noise_seed=3)
        return sim.run(units=units)[0]

    k, p = run("K_RJ"), run("pW")
    back = k.to("pW")
    assert back.units == "pW" and set(back.fields) == {"atmosphere", "noise"} and k.units == "K_RJ"
    for f in back.fields:
        np.testing.assert_allclose(back.data[f], p.data[f], rtol=3e-6, atol=1e-6 * np.abs(p.data[f]).max())
    again = back.to("K_RJ")
    for f in again.fields:
        np.testing.assert_allclose(again.data[f], k.data[f], rtol=3e-6, atol=1e-6 * np.abs(k.data[f]).max())
    assert k.to("K_RJ") is k


@pytest.mark.parametrize("T,fs,knee", [(70001, 400.0, 1.0), (40000, 400.0, 2.0), (20000, 100.0, 1.0), (9000, 100.0, 0.0)])
def test_noise_written_in_krj(gpu_ctx, T, fs, knee):
    """mrx_noise_generate_krj: the field of mrx_noise_generate divided by den_band(d)(el(d, t)) as mrx_tod_to_krj divides
    it -- on the two-rate writer's own store (rates 4 and 2: the same arithmetic, bit for bit) or, where the one-rate
    form or the white-only path applies, by that pass itself."""
    import ctypes as C

    import torch

    from maria_amd import _lib
    from maria_amd._lib import ptr
    from maria_amd.pipeline import DevicePath

    p = small_problem(n_det=46, n_bands=2, n_layers=1)
    rng = np.random.default_rng(5)
    coords_offsets = np.radians(rng.uniform(-0.4, 0.4, (46, 2)))
    tt = np.arange(T) / fs
    bore_el = np.radians(55.0) + np.radians(0.4) * np.sin(2 * np.pi * tt / 7.3)
    path = DevicePath(dict(p, t=tt), device="cuda:0", ctx=gpu_ctx)
    path.set_calibration(_cal_tables(2), 273.15, 1.0, bore_el, coords_offsets, [False, True])
    k = path.krj_row_tables()
    D, n_modes = path.D, 3
    basis = torch.as_tensor(rng.normal(size=(D, n_modes)).astype(np.float32)).to("cuda:0")
    scale = torch.as_tensor(rng.uniform(1.0, 2.0, D).astype(np.float32)).to("cuda:0")
    need = C.c_size_t()
    _lib.load().mrx_noise_work_floats(T, n_modes, D, C.byref(need))
    work = torch.empty(need.value, dtype=torch.float32, device="cuda:0")
    head = (77, D, 0, T, fs, knee, 0.5, ptr(basis), n_modes, ptr(scale), None, 0, 0.0)
    plain = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
    gpu_ctx.call("mrx_noise_generate", *head, ptr(plain), plain.stride(0), 0, ptr(work), need.value)
    fused = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
    gpu_ctx.call("mrx_noise_generate_krj", *head, ptr(fused), fused.stride(0), ptr(work), need.value, ptr(k["bore_el"]), ptr(k["dx"]),
                 ptr(k["dy"]), ptr(k["band"]), ptr(k["axis"]), ptr(k["values"]), k["n_el"], k["n_bands"])
    ref = path.to_krj(plain.clone())
    torch.cuda.synchronize()
    assert torch.isfinite(fused).all() and not torch.equal(fused, plain)
    assert torch.equal(fused, ref)
