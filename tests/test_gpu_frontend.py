"""GPU tests of the reference-shaped front end: Simulation(...).run() and Atmosphere."""

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _setup(n=61, fov=0.3, duration=40.0, fs=50.0, **atm):
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan

    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093", gain_error=0.05),
             Band(center=150e9, width=41e9, shape="top_hat", name="f150", gain_error=0.05)]
    inst = Instrument(Detectors.hexagon(n, fov, bands, primary_size=6.0))
    plan = Plan.daisy(start_time=1.7e9, duration=duration, sample_rate=fs, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
    return inst, plan, Site(altitude=1000.0, region="synthetic")


def _oracle_problem(sim, obs):
    """The oracle's input dict from the objects the front end built (screens downloaded)."""
    atm = obs.atmosphere
    path = atm._device_path()
    layers = []
    for l, bufs in zip(sorted(atm.processes), path._layer_bufs):
        layers.append(dict(atm.processes[l], values=bufs[0].cpu().numpy()))
    dets = obs.instrument.dets
    return dict(
        t=obs.coords.t, ta=atm.boresight.t, az_a=atm.boresight.az, el_a=atm.boresight.el,
        offsets=dets.offsets, band_index=dets.band_index, m00=dets.mueller00(), layers=layers,
        tables=atm._tables(dets), T0=float(atm.weather.temperature[0]), pwv0=float(atm.weather.pwv),
        timestep=float(atm.timestep), gain=None,
    )


def test_simulation_run_matches_oracle(gpu_ctx):
    from maria_amd.sim import Simulation
    from oracle import hotpath

    inst, plan, site = _setup()
    sim = Simulation(inst, plan, site, atmosphere="2d", atmosphere_kwargs={"weather": {"pwv": 1.5}, "seed": 5, "n_layers": 4},
                     noise=False, gain_seed=1)
    (tod,) = sim.run(units="pW")
    data = tod.data["atmosphere"]
    D, T = inst.dets.n, len(plan.time)
    assert data.shape == (D, T) and data.dtype == np.float32 and tod.units == "pW"
    assert not np.isnan(data).any()
    assert tod.metadata["atmosphere"] and tod.metadata["pwv"] == 1.5
    obs = sim.obs_list[0]
    atm = obs.atmosphere
    assert atm.zenith_scaled_pwv.shape == (D, len(atm.boresight.t))
    assert len(atm.processes) == 4 and atm.timestep >= 0.1
    # the reference's chain on the same screens, then the same gains
    ref = hotpath.run_path(_oracle_problem(sim, obs))
    gain = data[:, T // 2] / ref[:, T // 2]  # per-detector gain error exp(0.05 N(0,1))
    assert 0.7 < gain.min() and gain.max() < 1.4 and gain.std() > 0.01
    assert rel_err(data, ref * gain[:, None]) <= 2e-5
    # pwv: mean near the weather value, fluctuations of the order pwv_rms_frac
    pwv = atm.zenith_scaled_pwv
    assert abs(pwv.mean() - 1.5) < 0.1 and 1e-4 < pwv.std() < 0.1
    # a second run is a new realisation (reference: fresh noise every run)
    (tod2,) = sim.run(units="pW")
    assert not np.array_equal(tod2.data["atmosphere"], data)


def test_reference_error_behaviour(gpu_ctx):
    from maria_amd.atmosphere import Atmosphere
    from maria_amd.sim import Plan, PointingError, Simulation

    inst, plan, site = _setup(n=19, duration=10.0)
    with pytest.raises(ValueError, match="Invalid model"):
        Atmosphere(model="4d")
    with pytest.raises(RuntimeError, match="must be initialized"):
        Atmosphere().simulate_pwv()
    low = Plan(plan.time, plan.phi, np.full_like(plan.theta, np.radians(4.0)))
    with pytest.raises(PointingError):
        Simulation(inst, low, site, atmosphere="2d", noise=False)
    with pytest.raises(TypeError):
        Simulation(inst, "daisy", site)
    with pytest.raises(NotImplementedError):
        Simulation(inst, plan, site, atmosphere="2d", map="some_file.fits")  # map io stays with maria
    with pytest.raises(NotImplementedError):
        Simulation(inst, plan, site, atmosphere="2d", cmb="generate")
    sim = Simulation(inst, plan, site, atmosphere="2d", noise=False)
    with pytest.raises(NotImplementedError, match="K_CMB"):
        sim.run(units="K_CMB")  # only the reference's default K_RJ and pW are built
    (tod,) = sim.run(units="pW")
    with pytest.raises(NotImplementedError):
        tod.to("K_CMB")  # only pW <-> K_RJ of an existing TOD is built


def test_screen_statistics_through_the_front_end(gpu_ctx):
    """Ribbon screens cropped from the padded periodic domain keep unit variance and
    the beam smoothing lowers it."""
    from maria_amd.sim import Simulation

    inst, plan, site = _setup(n=37, duration=120.0, fs=20.0)
    sim = Simulation(inst, plan, site, atmosphere="2d", atmosphere_kwargs={"n_layers": 2}, noise=False)
    atm = sim.obs_list[0].atmosphere
    # a ribbon spans about one outer scale, so a single realisation is nearly one draw of the
    # large-scale modes: the second moment about ZERO (not the ribbon's own mean) is what
    # averages to the field's unit variance, over realisations
    from oracle import functions

    msq = []
    lags = np.array([1, 2, 4, 8])
    sf = np.zeros((2, len(lags)))
    n_real = 96  # (a ribbon of 750 x 4 pixels holds few independent increments: 12 realisations scatter by 10 %)
    for _ in range(n_real):
        atm.simulate_pwv(instrument=None)  # no smoothing
        raw = [b[0].cpu().numpy() for b in atm._device_path()._layer_bufs]
        assert all(np.isfinite(s).all() for s in raw)
        msq.append([float((s.astype(np.float64) ** 2).mean()) for s in raw])
        for i, s in enumerate(raw):  # structure function along the extrusion axis (differences: the large modes cancel)
            s = s.astype(np.float64)
            sf[i] += [np.mean((s[k:] - s[:-k]) ** 2) / n_real for k in lags]
    assert 0.4 < np.mean(msq) < 1.9, msq
    # ... and the small-scale structure is Matern's (functions/__init__.py:30-39) from ONE pixel of the ribbon up: the
    # generator's amplitudes are the covariance's own eigenvalues on the padded periodic domain (mrx_screen_amplitudes)
    for i, l in enumerate(sorted(atm.processes)):
        pr = atm.processes[l]
        de = float(pr["extrusion"][1] - pr["extrusion"][0])
        want = 2 * (1 - functions.normalized_matern(lags * de / pr["r0"], pr["nu"]))
        assert np.abs(sf[i] / want - 1).max() < 0.03, (sf[i], want)  # measured 0.2 ... 1.0 %
    atm._realisation -= 1  # same realisation, smoothed
    atm.simulate_pwv(instrument=inst)
    smooth = [b[0].cpu().numpy() for b in atm._device_path()._layer_bufs]
    assert all(a.var() < b.var() for a, b in zip(smooth, raw))


def test_map_smooth_front_end(gpu_ctx):
    from maria_amd import map as mmap
    from oracle import hotpath

    rng = np.random.default_rng(0)
    data = rng.standard_normal((2, 3, 64, 96)).astype(np.float32)  # [stokes, nu, ny, nx]
    weight = rng.random((2, 3, 64, 96)).astype(np.float32)
    got, den = mmap.smooth(data, weight, fwhm=3.0, x_res=-1.0, y_res=0.8)
    sig = 3.0 / np.sqrt(8 * np.log(2))
    ref, ref_den = hotpath.map_smooth(data, weight, sig / 0.8, sig / 1.0)
    assert got.shape == data.shape
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.abs(den - ref_den).max() <= 1e-6
    with pytest.raises(ValueError):
        mmap.smooth(data, weight)


def test_reference_atmosphere_test_case_zenith_stare(gpu_ctx):
    """maria/tests/atmosphere/test_atmosphere.py:21-28 runs
    Simulation("MUSTANG-2", "ten_second_zenith_stare", "green_bank", atmosphere="2d").run()
    (no assertion there).  The same shape here: 217 detectors over 0.07 deg behind a
    100 m primary at 93 GHz (instrument/configs/m2.yml), a 10 s stare at (az 0, el 90)
    sampled at 50 Hz (plan/plans/test.yml:1-9), default units; checked against the oracle."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation
    from oracle import hotpath

    m2 = Band(nu=np.linspace(74e9, 105e9, 31), tau=np.r_[0.0, np.linspace(1.0, 0.3, 29), 0.0], name="m2/f093", efficiency=0.1, gain_error=0.05)
    inst = Instrument(Detectors.hexagon(217, 0.07, [m2], primary_size=100.0), name="MUSTANG-2")
    t = 1.7e9 + np.arange(0, 10, 1 / 50.0)
    plan = Plan(t, np.zeros_like(t), np.full_like(t, np.pi / 2))
    sim = Simulation(inst, plan, Site(altitude=825.0, region="green_bank"), atmosphere="2d", noise=False, gain_seed=3)
    (tod,) = sim.run()
    k = tod.data["atmosphere"]
    assert k.shape == (217, 500) and tod.units == "K_RJ" and np.isfinite(k).all()
    obs = sim.obs_list[0]
    atm = obs.atmosphere
    # atmosphere.py:96-99: max(0.1 s, min_fwhm / max_wind); the 100 m dish's near-field beam gives ~0.4 s
    assert len(atm.processes) == 8 and 0.1 <= atm.timestep < 1.0 and len(atm.boresight.t) >= 4
    # oracle: pW chain on the same screens, gains recovered from the device pW run, then K_RJ
    path = atm._device_path()
    pw = path.run().cpu().numpy()
    ref_pw = hotpath.run_path(_oracle_problem(sim, obs))
    gain = pw[:, 250] / ref_pw[:, 250]
    assert rel_err(pw, ref_pw * gain[:, None]) <= 2e-5
    sp = atm.spectrum
    tables = [{"T": sp.side_base_temperature, "pwv": sp.side_zenith_pwv, "el": sp.side_elevation,
               "values": hotpath.transmission_integral_grid(m2.passband, sp.side_nu, sp._opacity)}]
    _, el_det = hotpath.broadcast(obs.coords.offsets, obs.boresight.az, obs.boresight.el)
    ref = hotpath.calibrate_to_krj(pw, inst.dets.band_index, tables, tod.metadata["base_temperature"], tod.metadata["pwv"], el_det)
    assert rel_err(k, ref) <= 1e-5


@pytest.mark.parametrize("units,n", [("pW", 64), ("K_RJ", 64), ("K_RJ", 61), ("K_RJ", 72)])
def test_sharded_simulation_rows_equal_the_unsharded_run(gpu_ctx, units, n):
    """Simulation(shard=(rank, world)) simulates its block of detector rows only: atmosphere, map
    and noise fields of every shard are the same rows of the unsharded run, bit for bit (no
    cross-detector term: atmosphere/atmosphere.py:346-373; draws keyed by the global row).
    n = 61: bands of odd size, so that shards begin and end inside the detector pairs that share a noise
    transform (the straddling pair is drawn whole on both sides: a randomised sweep found the lone row
    1e-7 off), and the choice between the two K_RJ forms must not depend on the shard's own detectors.
    n = 72 on four ranks: blocks of 16 rows make that 48 + 48 + 48 + 0 -- the last rank gets fields of no rows."""
    from maria_amd import map as mmap
    from maria_amd.dist import shard_bounds
    from maria_amd.sim import Simulation

    inst, plan, site = _setup(n=n, duration=20.0)  # 2 bands x n: shards cut through band 1 and 2
    X, Y = np.meshgrid(np.linspace(-1, 1, 48), np.linspace(-1, 1, 48))
    sky = mmap.ProjectionMap(0.02 * np.exp(-(X**2 + Y**2) / 0.05).astype(np.float32), nu=120e9, width=1.5, center=(45.0, 55.0), frame="az/el")
    kw = dict(atmosphere="2d", atmosphere_kwargs={"seed": 3, "n_layers": 3}, map=sky, noise=True, gain_seed=11, noise_seed=77)
    (full,) = Simulation(inst, plan, site, **kw).run(units=units)
    world = 4 if n == 72 else 3
    seen = 0
    assert n != 72 or shard_bounds(inst.dets.n, world, world - 1) == (144, 144)
    for rank in range(world):
        sim = Simulation(inst, plan, site, shard=(rank, world), **kw)
        (tod,) = sim.run(units=units)
        lo, hi = shard_bounds(inst.dets.n, world, rank)
        assert tod.metadata["shard"]["rows"] == [lo, hi] and tod.dets.n == hi - lo
        assert tod.coords.az.shape == (hi - lo, len(plan.time))
        for name in ("atmosphere", "map", "noise"):
            assert tod.data[name].shape == (hi - lo, len(plan.time))
            assert np.array_equal(tod.data[name], full.data[name][lo:hi]), (rank, name)
        back = tod.to("pW" if units == "K_RJ" else "K_RJ").to(units)
        assert back.data["atmosphere"].shape == tod.data["atmosphere"].shape
        assert hi == lo or rel_err(back.data["atmosphere"], tod.data["atmosphere"]) < 1e-6
        seen += hi - lo
    assert seen == inst.dets.n
    with pytest.raises(ValueError, match="noise_seed"):
        Simulation(inst, plan, site, shard=(0, 2), atmosphere="2d", noise=True)
    with pytest.raises(ValueError, match="gain_seed"):
        Simulation(inst, plan, site, shard=(0, 2), atmosphere="2d", noise=False)
    with pytest.raises(ValueError, match="outside"):
        Simulation(inst, plan, site, shard=(2, 2), atmosphere="2d", noise=False, gain_seed=1)


def test_noise_is_drawn_from_the_loading_before_the_gain_error(gpu_ctx):
    """sim/noise.py:35-37 reads the loadings before simulation.py:239-247 multiplies the gain
    error into the non-noise fields: with NEP_per_loading > 0 the noise field must not depend
    on the gain draw, while the atmosphere field carries it."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    def run(gain_error):
        band = Band(center=93e9, width=27e9, shape="top_hat", name="f093", NEP=2e-17, knee=0.5, NEP_per_loading=0.3, gain_error=gain_error)
        inst = Instrument(Detectors.hexagon(32, 0.3, [band], primary_size=6.0))
        plan = Plan.daisy(start_time=1.7e9, duration=20.0, sample_rate=50.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
        sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"seed": 9, "n_layers": 2},
                         noise=True, gain_seed=5, noise_seed=6)
        return sim.run(units="pW")[0]

    flat, gained = run(0.0), run(0.3)
    assert np.array_equal(flat.data["noise"], gained.data["noise"])
    ratio = gained.data["atmosphere"][:, 500] / flat.data["atmosphere"][:, 500]
    assert ratio.std() > 0.1 and np.allclose(gained.data["atmosphere"], flat.data["atmosphere"] * ratio[:, None], rtol=2e-6)


def test_simulation_with_cubic_interpolation(gpu_ctx):
    """Atmosphere(interpolation_method="cubic") through the front end, against the oracle chain."""
    from maria_amd.atmosphere import Atmosphere
    from maria_amd.sim import Simulation
    from oracle import hotpath

    inst, plan, site = _setup(n=37, duration=20.0)
    sim = Simulation(inst, plan, site, atmosphere="2d", noise=False, gain_seed=1,
                     atmosphere_kwargs={"seed": 5, "n_layers": 3, "interpolation_method": "cubic"})
    (tod,) = sim.run(units="pW")
    obs = sim.obs_list[0]
    assert obs.atmosphere.interpolation_method == "cubic"
    prob = _oracle_problem(sim, obs)
    prob["interpolation_method"] = "cubic"
    ref = hotpath.run_path(prob)
    data = tod.data["atmosphere"]
    gain = data[:, 100] / ref[:, 100]
    assert rel_err(data, ref * gain[:, None]) <= 2e-5
    with pytest.raises(ValueError, match="interpolation_method"):
        Atmosphere(interpolation_method="quintic")


def test_simulation_3d_model(gpu_ctx):
    """Simulation(atmosphere="3d") (atmosphere/atmosphere.py:28,141-279, extrusion.py:69-77): one
    process of many thin layers sampled like any other layer stack (oracle chain on the downloaded
    screens), whose screens are vertically correlated slices of one volume."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation
    from oracle import functions, hotpath

    bands = [Band(center=150e9, width=30e9, name="f150")]
    inst = Instrument(Detectors.hexagon(37, 0.3, bands, primary_size=25.0))
    plan = Plan.daisy(start_time=1.7e9, duration=30.0, sample_rate=20.0, scan_center=(130.0, 52.0), radius=0.3, speed=0.3)
    sim = Simulation(inst, plan, Site(altitude=1800.0), atmosphere="3d", noise=False,
                     atmosphere_kwargs={"weather": {"pwv": 1.2}, "max_height": 400.0, "seed": 21})
    (tod,) = sim.run(units="pW")
    obs = sim.obs_list[0]
    atm = obs.atmosphere
    L = len(atm.layers["h"])
    assert L > 10 and len(atm.processes) == 1 and atm.processes[0]["nu"] == pytest.approx(1 / 3)
    data = tod.data["atmosphere"]
    assert data.shape == (inst.dets.n, len(plan.time)) and np.isfinite(data).all()
    path = atm._device_path()
    screens = [b[0].cpu().numpy() for b in path._layer_bufs]
    assert len(screens) == L
    prob = dict(
        t=obs.coords.t, ta=atm.boresight.t, az_a=atm.boresight.az, el_a=atm.boresight.el,
        offsets=inst.dets.offsets, band_index=inst.dets.band_index, m00=inst.dets.mueller00(),
        layers=[dict(l, values=s) for l, s in zip(atm._layer_list(), screens)],
        tables=atm._tables(inst.dets), T0=float(atm.weather.temperature[0]), pwv0=float(atm.weather.pwv),
        timestep=float(atm.timestep), gain=None,
    )
    ref = hotpath.run_path(prob)
    assert rel_err(data, ref) <= 1e-5
    # neighbouring layers are strongly correlated, distant ones less: the Matern(1/3) of their separation
    atm._realisation -= 1
    atm.simulate_pwv(instrument=None)  # the same volume, unsmoothed
    raw = [b[0].cpu().numpy().astype(np.float64) for b in path._layer_bufs]
    h = atm.layers["h"]
    sf = lambda a, b: np.mean((a - b) ** 2)  # noqa: E731
    near, far = sf(raw[0], raw[1]), sf(raw[0], raw[L - 1])
    want_near = 2 * (1 - functions.approximate_normalized_matern(np.array([h[1] - h[0]]), nu=1 / 3, r0=atm.processes[0]["r0"])[0])
    want_far = 2 * (1 - functions.approximate_normalized_matern(np.array([h[L - 1] - h[0]]), nu=1 / 3, r0=atm.processes[0]["r0"])[0])
    assert near < 0.5 * far
    assert 0.3 * want_near < near < 1.5 * want_near and 0.5 * want_far < far < 1.6 * want_far, (near, want_near, far, want_far)
    with pytest.raises(ValueError, match="Invalid model"):
        Simulation(inst, plan, Site(altitude=1800.0), atmosphere="5d", noise=False)


def test_3d_layers_on_their_own_grids(gpu_ctx):
    """model="3d" with a resolution that grows with height (extrusion.py:56-60): every layer's screen lives on its own
    cross-section grid (atmosphere.py:208-219).  The chain behind it -- the volume's planes on one generation grid,
    mrx_resample_columns onto the layer's nodes, the beam smoothing ON THAT GRID with sigma / layer.res pixels
    (atmosphere.py:338-344) -- is checked layer by layer against numpy / scipy on the downloaded generation planes,
    and the TOD against the oracle chain on the downloaded screens."""
    import scipy.ndimage

    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation
    from oracle import hotpath

    bands = [Band(center=150e9, width=30e9, name="f150")]
    inst = Instrument(Detectors.hexagon(37, 3.0, bands, primary_size=25.0))
    plan = Plan.daisy(start_time=1.7e9, duration=20.0, sample_rate=20.0, scan_center=(130.0, 52.0), radius=0.3, speed=0.3)
    sim = Simulation(inst, plan, Site(altitude=1800.0), atmosphere="3d", noise=False,
                     atmosphere_kwargs={"weather": {"pwv": 1.2}, "max_height": 3000.0, "seed": 5})
    (tod,) = sim.run(units="pW")
    obs = sim.obs_list[0]
    atm = obs.atmosphere
    layers = atm._layer_list()
    counts = [len(l["cross_section"]) for l in layers]
    assert len(set(counts)) > 1 and counts[0] > counts[-1]
    path = atm._device_path()
    screens = [b[0].cpu().numpy() for b in path._layer_bufs]
    for l in (0, len(layers) // 2, len(layers) - 1):
        layer, g = layers[l], layers[l]["gen"]
        plane = path._gen_fine[l]["plane"].cpu().numpy().astype(np.float64)
        assert np.isfinite(plane).all() and 0.05 < (plane**2).mean() < 20.0  # (a patch smaller than the 1 km outer scale: no tighter claim)
        coarse = (g["scale"] * ((1 - g["w"]) * plane[:, g["idx"]] + g["w"] * plane[:, g["idx"] + 1])).astype(np.float32)
        de = layer["extrusion"][1] - layer["extrusion"][0]
        ref = scipy.ndimage.gaussian_filter(coarse, sigma=(layer["beam_sigma"] / de, layer["beam_sigma"] / layer["res"]))
        assert screens[l].shape == ref.shape == (len(layer["extrusion"]), counts[l])
        assert np.abs(screens[l] - ref).max() <= 1e-5 * np.abs(ref).max(), l
    prob = dict(
        t=obs.coords.t, ta=atm.boresight.t, az_a=atm.boresight.az, el_a=atm.boresight.el,
        offsets=inst.dets.offsets, band_index=inst.dets.band_index, m00=inst.dets.mueller00(),
        layers=[dict(l, values=s) for l, s in zip(layers, screens)],
        tables=atm._tables(inst.dets), T0=float(atm.weather.temperature[0]), pwv0=float(atm.weather.pwv),
        timestep=float(atm.timestep), gain=None,
    )
    assert rel_err(tod.data["atmosphere"], hotpath.run_path(prob)) <= 1e-5


def test_bands_whose_rows_are_not_neighbours(gpu_ctx):
    """The reference selects a band's detectors by name (sim/noise.py:32, sim/atmosphere.py:54, sim/map.py:90-95), so an
    array may interleave its bands.  Every field of such an instrument -- atmosphere, noise (drawn per index within
    the band, correlated modes included), in K_RJ -- equals the same detectors' rows of the band-major instrument."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093", NEP=2e-17, knee=0.8),
             Band(center=150e9, width=41e9, shape="gaussian", name="f150", NEP=3e-17, knee=1.5)]
    dets = Detectors.hexagon(19, 0.3, bands, primary_size=6.0)
    perm = np.arange(dets.n).reshape(2, -1).T.ravel()  # f093, f150, f093, f150, ...
    plan = Plan.daisy(start_time=1.7e9, duration=30.0, sample_rate=50.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
    runs = []
    for d in (dets, dets.subset(perm)):
        sim = Simulation(Instrument(d), plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 3},
                         noise=True, noise_seed=11)
        runs.append(sim.run()[0])
    major, mixed = runs
    assert not (np.diff(np.nonzero(mixed.dets.band_index == 0)[0]) == 1).all()
    for name in ("atmosphere", "noise"):
        a, b = major.data[name][perm], mixed.data[name]
        assert np.isfinite(b).all() and np.abs(a - b).max() <= 2e-6 * np.abs(a).max(), name
