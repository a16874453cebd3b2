"""GPU tests of detector noise: statistical parity with the
reference model (noise/generation.py:11-51, sim/noise.py:18-63) -- spectrum, white
level, knee, correlated modes -- since the generator and its period differ by design."""

import ctypes as C

import numpy as np
import pytest
import scipy.signal

pytestmark = pytest.mark.gpu


def _generate(ctx, D, T, fs, knee, corr=0.0, basis=None, scale=None, seed=5, accumulate=0, out=None, batch=64,
              loading=None, per_loading=0.0, det_offset=0):
    import torch

    from maria_amd._lib import ptr

    dev = "cuda:0"
    n_modes = 0 if basis is None else basis.shape[1]
    need = C.c_size_t()
    assert ctx.lib.mrx_noise_work_floats(T, n_modes, min(batch, D), C.byref(need)) == 0
    work = torch.empty(need.value, dtype=torch.float32, device=dev)
    d_basis = None if basis is None else torch.as_tensor(np.ascontiguousarray(basis, np.float32)).to(dev)
    d_scale = None if scale is None else torch.as_tensor(np.asarray(scale, np.float32)).to(dev)
    if out is None:
        out = torch.zeros((D, T), dtype=torch.float32, device=dev)
    ctx.call("mrx_noise_generate", seed, D, det_offset, T, float(fs), float(knee), float(corr), ptr(d_basis), n_modes,
             ptr(d_scale), ptr(loading), 0 if loading is None else loading.stride(0), float(per_loading),
             ptr(out), out.stride(0), accumulate, ptr(work), need.value)
    return out


def test_period_helper(gpu_ctx):
    n1, n2 = C.c_int(), C.c_int()
    for T, want in [(1, 4096), (3000, 4096), (4097, 8192), (240000, 262144), (400000, 1 << 19), (1440000, 2097152), (1 << 23, 1 << 23)]:
        assert gpu_ctx.lib.mrx_noise_period(T, C.byref(n1), C.byref(n2)) == 0
        # first transform: N / 64 up to 2^18, then 4096 (the register transform's length), 8192 for 2^23 alone
        first = want // 64 if want <= 1 << 18 else 4096 if want < 1 << 23 else 8192
        assert n1.value * n2.value == want and 64 <= n1.value <= 1024 and n2.value == first
    assert gpu_ctx.lib.mrx_noise_period((1 << 23) + 1, C.byref(n1), C.byref(n2)) != 0  # unsupported length


def test_white_noise_level_and_independence(gpu_ctx):
    """knee = 0: sqrt(fs) N(0,1) per sample, scaled per detector (generation.py:25)."""
    D, T, fs = 64, 50000, 200.0
    scale = np.linspace(0.5, 2.0, D)
    x = _generate(gpu_ctx, D, T, fs, knee=0.0, scale=scale).cpu().numpy().astype(np.float64)
    assert np.abs(x.mean(axis=1)).max() < 5 * np.sqrt(fs / T) * 2.0
    np.testing.assert_allclose(x.var(axis=1), fs * scale**2, rtol=0.05)
    c = np.corrcoef(x)
    assert np.abs(c - np.eye(D)).max() < 0.03
    # white: lag-1 autocorrelation vanishes
    assert abs(np.mean(x[:, 1:] * x[:, :-1]) / np.mean(x * x)) < 0.01


@pytest.mark.parametrize("T,fs,knee,generic", [(240000, 400.0, 1.0, 0), (30000, 50.0, 5.0, 0), (5000, 50.0, 0.3, 0),
                                                 (30000, 50.0, 5.0, 1), (600000, 400.0, 2.0, 0)])
def test_spectrum_matches_reference_model(gpu_ctx, T, fs, knee, generic):
    """One-sided PSD = 2 (1 + knee/f): flat white level, 1/f below the knee
    (generation.py:27-37), for several lengths (periods 2^18, 2^15, 2^13, 2^20) and both
    forms of the second transform (registers for periods up to 2^19, LDS beyond or on request)."""
    from oracle import noise as onoise

    D = 96 if T < 500000 else 24
    gpu_ctx.set_option(5, generic)  # MRX_OPT_NOISE_GENERIC
    try:
        x = _generate(gpu_ctx, D, T, fs, knee).cpu().numpy().astype(np.float64)
    finally:
        gpu_ctx.set_option(5, 0)
    nper = min(T, 1 << 14)
    f, p = scipy.signal.welch(x, fs=fs, nperseg=nper, noverlap=nper // 2, detrend=False, axis=-1)
    p = p.mean(axis=0)
    edges = np.geomspace(4 * fs / nper, 0.45 * fs, 10)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (f >= lo) & (f < hi)
        if m.sum() < 3:
            continue
        got, want = p[m].mean(), onoise.one_sided_psd_model(f[m], fs, knee).mean()
        # 5 % per band; the lowest band (4-9 bins of the segment) reads 6 % high through the Hann
        # window's leakage on the 1/f slope
        assert abs(got / want - 1) < (0.10 if lo == edges[0] else 0.05), (lo, hi, got, want)


@pytest.mark.parametrize("modes,T", [(0, 150000), (2, 150000), (5, 150000), (5, 100000), (2, 50000), (0, 33000), (5, 300000)])
def test_register_first_pass_matches_the_stockham_one(gpu_ctx, modes, T):
    """First transforms of 1024, 2048 and 4096 points (periods of 2^16, 2^17 and 2^18 ... 2^22
    samples) build the spectrum in registers and transform it with fft_regs (16 x RB x 16, 16 / RB
    series side by side in a workgroup); option bit 1 keeps the LDS Stockham kernel.  Same draws,
    same cells: the two series agree to float32 rounding (1/f noise: compared against the largest
    sample).  11 detectors: an odd pair count, teams past the end."""
    D, fs, knee = 11, 400.0, 3.0
    rng = np.random.default_rng(0)
    basis = None if modes == 0 else rng.normal(size=(D, modes)) / np.sqrt(modes)
    a = _generate(gpu_ctx, D, T, fs, knee, corr=0.4, basis=basis, seed=21).cpu().numpy()
    gpu_ctx.set_option(5, 2)
    try:
        b = _generate(gpu_ctx, D, T, fs, knee, corr=0.4, basis=basis, seed=21).cpu().numpy()
    finally:
        gpu_ctx.set_option(5, 0)
    assert np.abs(a).max() > 10 and np.abs(a - b).max() <= 2e-5 * np.abs(b).max()


@pytest.mark.parametrize("T,modes", [(300000, 2), (600000, 5), (1500000, 0), (2200000, 3)])
def test_register_second_pass_for_long_periods_matches_the_lds_one(gpu_ctx, T, modes):
    """Periods of 2^19 ... 2^22 samples (n1 = 128 ... 1024 = 64 m): 64-point register transforms in
    place over the scratch, then the m-point ones with the epilogue; option bit 0 keeps the LDS
    tiles.  Same spectrum, same samples to float32 rounding; with a scale, a loading-dependent
    level and accumulation into an existing TOD."""
    import torch

    D, fs, knee = 6, 400.0, 3.0
    rng = np.random.default_rng(1)
    basis = None if modes == 0 else rng.normal(size=(D, modes)) / np.sqrt(modes)
    scale = np.linspace(0.5, 2.0, D)
    a = _generate(gpu_ctx, D, T, fs, knee, corr=0.4, basis=basis, scale=scale, seed=33)
    gpu_ctx.set_option(5, 1)
    try:
        b = _generate(gpu_ctx, D, T, fs, knee, corr=0.4, basis=basis, scale=scale, seed=33)
    finally:
        gpu_ctx.set_option(5, 0)
    peak = float(b.abs().max())
    assert peak > 10 and float((a - b).abs().max()) <= 2e-5 * peak
    if modes == 2:  # the epilogue's other branch: loading-dependent level, accumulation
        loading = torch.rand((D, T), dtype=torch.float32, device="cuda:0")
        base = torch.full((D, T), 3.0, dtype=torch.float32, device="cuda:0")
        c = _generate(gpu_ctx, D, T, fs, knee, corr=0.4, basis=basis, scale=scale, seed=33, loading=loading, per_loading=0.7,
                      accumulate=1, out=base.clone())
        want = 3.0 + a * (torch.as_tensor(scale, dtype=torch.float32, device="cuda:0")[:, None] + 0.7 * loading) / \
            torch.as_tensor(scale, dtype=torch.float32, device="cuda:0")[:, None]
        assert float((c - want).abs().max()) <= 3e-5 * peak


def test_matches_oracle_generator_statistics(gpu_ctx):
    """Against the numpy restatement of the reference on the same parameters: equal band
    powers (three octaves below, at and above the knee) within sampling error."""
    from oracle import noise as onoise

    D, T, fs, knee = 64, 20000, 100.0, 2.0
    gpu = _generate(gpu_ctx, D, T, fs, knee).cpu().numpy().astype(np.float64)
    ref = onoise.generate_noise_with_knee((D, T), sample_rate=fs, knee=knee, rng=np.random.default_rng(1))
    f = np.fft.rfftfreq(T, 1 / fs)
    pg = (np.abs(np.fft.rfft(gpu, axis=1)) ** 2).mean(axis=0)
    pr = (np.abs(np.fft.rfft(ref, axis=1)) ** 2).mean(axis=0)
    for lo, hi in [(0.05, 0.4), (0.4, 2.0), (2.0, 10.0), (10.0, 45.0)]:
        m = (f >= lo) & (f < hi)
        assert abs(pg[m].sum() / pr[m].sum() - 1) < 0.1, (lo, hi)


def _band_covariance(x, fs, lo, hi, spectrum):
    """Cross-detector covariance of the Fourier coefficients in [lo, hi) Hz, each bin
    divided by ``spectrum(f)`` (whitening, so that every bin carries the same weight)."""
    T = x.shape[1]
    X = np.fft.rfft(x * np.hanning(T), axis=1)
    f = np.fft.rfftfreq(T, 1 / fs)
    m = (f >= lo) & (f < hi)
    Xw = X[:, m] / np.sqrt(spectrum(f[m]))
    return (Xw @ Xw.conj().T).real / m.sum(), f[m]


def test_correlated_modes_follow_the_basis(gpu_ctx):
    """sqrt(c) B @ modes + sqrt(1-c) pink (generation.py:39-47), B from utils/linalg.py:105-126
    and modes = white + pink: the cross-detector covariance is
    c B B^T (1 + knee/f) + ((1-c) knee/f + 1) I, checked below and above the knee."""
    from maria_amd import noise as mnoise
    from maria_amd import synthetic
    from oracle import noise as onoise

    D, T, fs, c = 120, 60000, 100.0, 0.5
    off = synthetic.hex_pack(D, np.radians(0.5))
    B = mnoise.spatial_basis(off, k=5, n_side=16, scale=mnoise.diameter(off))
    np.testing.assert_allclose(B, onoise.generate_spatial_basis(off, k=5, n_side=16, scale=mnoise.diameter(off)), atol=1e-12)
    off_diag = ~np.eye(D, dtype=bool)
    # r_min: the numpy restatement itself reaches 0.94-0.95 / 0.987 on these bands (sampling error)
    for knee, lo, hi, r_min in [(20.0, 0.2, 4.0, 0.9), (0.5, 10.0, 45.0, 0.97)]:
        x = _generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B).cpu().numpy().astype(np.float64)
        cov, f = _band_covariance(x, fs, lo, hi, lambda f: 1 + knee / f)
        ind = np.mean(((1 - c) * knee / f + 1) / (1 + knee / f))
        model = c * (B @ B.T) + ind * np.eye(D)
        s = np.trace(cov) / np.trace(model)
        r = np.corrcoef(cov[off_diag], model[off_diag])[0, 1]
        assert r > r_min, (knee, r)
        slope = np.sum(cov[off_diag] * model[off_diag]) / np.sum(model[off_diag] ** 2) / s
        assert abs(slope - 1) < 0.08, (knee, slope)
        # the oracle generator on the same parameters gives the same structure
        if knee == 20.0:
            ref = onoise.generate_noise_with_knee((D, T), fs, knee, basis=B, corr_prop=c, rng=np.random.default_rng(3))
            cov_ref, _ = _band_covariance(ref, fs, lo, hi, lambda f: 1 + knee / f)
            assert abs(np.trace(cov) / np.trace(cov_ref) - 1) < 0.05
            assert abs(cov[off_diag].mean() / cov_ref[off_diag].mean() - 1) < 0.15


def test_deterministic_batched_and_accumulating(gpu_ctx):
    import torch

    D, T, fs, knee = 70, 6000, 50.0, 1.0
    a = _generate(gpu_ctx, D, T, fs, knee, seed=9, batch=64)
    b = _generate(gpu_ctx, D, T, fs, knee, seed=9, batch=7)  # other batching: same series ids
    assert torch.equal(a, b)
    c = _generate(gpu_ctx, D, T, fs, knee, seed=10)
    assert not torch.equal(a, c)
    base = torch.full((D, T + 5), 3.0, dtype=torch.float32, device="cuda:0")
    acc = _generate(gpu_ctx, D, T, fs, knee, seed=9, accumulate=1, out=base)
    assert torch.allclose(acc[:, :T], a + 3.0, atol=1e-5) and bool((acc[:, T:] == 3.0).all())


def test_shards_share_modes_and_nothing_else(gpu_ctx):
    """A detector shard (even det_offset) reproduces exactly its rows of the whole band:
    the pink pair series and white draws are keyed by the global index, the modes by the seed."""
    import torch

    from maria_amd import noise as mnoise
    from maria_amd import synthetic

    D, T, fs, knee = 50, 5000, 80.0, 2.0
    off = synthetic.hex_pack(D, np.radians(0.5))
    B = mnoise.spatial_basis(off, k=5, n_side=16, scale=mnoise.diameter(off))
    scale = np.linspace(1.0, 2.0, D)
    whole = _generate(gpu_ctx, D, T, fs, knee, corr=0.5, basis=B, scale=scale, seed=21)
    for lo, hi in [(0, 16), (16, 50), (34, 35)]:
        part = _generate(gpu_ctx, hi - lo, T, fs, knee, corr=0.5, basis=B[lo:hi], scale=scale[lo:hi], seed=21, det_offset=lo)
        if (hi - lo) % 2 == 0 or hi == D:
            assert torch.equal(part, whole[lo:hi])
        else:
            # a shard that ends inside a pair transforms its last row without the partner's
            # mode coefficients in the imaginary part: same value up to float32 rounding
            assert torch.allclose(part, whole[lo:hi], rtol=0, atol=2e-4 * float(whole.abs().max()))
    with pytest.raises(RuntimeError):
        _generate(gpu_ctx, 4, T, fs, knee, det_offset=3)  # odd offset would split a pair


def test_unaligned_rows_and_short_series(gpu_ctx):
    """Rows that are not 16-byte aligned take the scalar store path and give the same values;
    T far below the minimum period (4096) and not a multiple of 4."""
    import torch

    D, T, fs, knee = 9, 1003, 30.0, 1.5
    a = _generate(gpu_ctx, D, T, fs, knee, seed=2)
    buf = torch.full((D, T + 2), -1.0, dtype=torch.float32, device="cuda:0")
    b = _generate(gpu_ctx, D, T, fs, knee, seed=2, out=buf[:, 1 : T + 1])
    assert torch.equal(a, b) and bool((buf[:, 0] == -1).all()) and bool((buf[:, T + 1] == -1).all())
    x = a.cpu().numpy().astype(np.float64)
    assert np.isfinite(x).all() and abs(x.var() / (fs * (1 + 2 * knee / fs * np.sum(1.0 / np.arange(1, 2049)))) - 1) < 0.5


def test_simulation_with_noise(gpu_ctx):
    """Simulation(noise=True), the reference default: a "noise" field in pW with the
    band's NEP (sim/noise.py:62: 1e12 * NEP * unscaled noise)."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    bands = [Band(center=93e9, width=27e9, shape="top_hat", name="f093", NEP=2e-17, knee=1.0),
             Band(center=150e9, width=41e9, shape="top_hat", name="f150", NEP=4e-17, knee=0.0)]
    inst = Instrument(Detectors.hexagon(40, 0.3, bands, primary_size=6.0))
    plan = Plan.daisy(start_time=1.7e9, duration=60.0, sample_rate=100.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
    sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2}, noise=True, noise_seed=4)
    (tod,) = sim.run(units="pW")
    assert set(tod.fields) == {"atmosphere", "noise"}
    n = tod.data["noise"].astype(np.float64)
    assert n.shape == (80, 6000) and np.isfinite(n).all()
    # band 1 has no knee: pure white of variance fs (1e12 NEP)^2
    np.testing.assert_allclose(n[40:].var(axis=1).mean(), 100.0 * (1e12 * 4e-17) ** 2, rtol=0.05)
    # band 0: white + pink: more variance than its white level, and positively correlated rows
    assert n[:40].var(axis=1).mean() > 1.5 * 100.0 * (1e12 * 2e-17) ** 2
    assert np.corrcoef(n[:40])[~np.eye(40, dtype=bool)].mean() > 0.02


def test_reference_white_noise_levels_test_case(gpu_ctx):
    """maria/tests/noise/test_noise.py:7-31: MUSTANG-2 (217 detectors, NEP 1.5e-17, knee 5 Hz,
    band/configs/m2.yml) on the ten-second zenith stare at 50 Hz: the per-detector mean of the
    noise over the TOD, in units of NEP / sqrt(duration), must have a standard deviation in
    [0.7, 1.3] -- i.e. the 1/f part averages to zero over the TOD, as it does in the reference
    whose pink series has the TOD's own period."""
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation

    m2 = Band(nu=np.linspace(74e9, 105e9, 31), tau=np.r_[0.0, np.linspace(1.0, 0.3, 29), 0.0], name="m2/f093", efficiency=0.1,
              NEP=1.5e-17, knee=5.0)
    inst = Instrument(Detectors.hexagon(217, 0.07, [m2], primary_size=100.0), name="MUSTANG-2")
    t = 1.7e9 + np.arange(0, 10, 1 / 50.0)
    plan = Plan(t, np.zeros_like(t), np.full_like(t, np.pi / 2))
    for seed in (1, 2, 3):
        sim = Simulation(inst, plan, Site(altitude=825.0, region="green_bank"), noise=True, noise_seed=seed)
        tod = sim.run()[0].to("pW")  # as the reference's test: default units, then TOD.to
        target_error = 1e12 * 1.5e-17 / np.sqrt(plan.duration + 1 / 50.0)
        scaled = tod.data["noise"].astype(np.float64).mean(axis=1) / target_error
        assert 0.7 < scaled.std() < 1.3, scaled.std()


@pytest.mark.parametrize("T,fs,knee", [(500, 50.0, 5.0), (5000, 100.0, 20.0), (100000, 400.0, 2.0)])
def test_pink_part_has_zero_mean_and_no_power_below_the_tod(gpu_ctx, T, fs, knee):
    """The mean of a detector's noise over the TOD is the mean of its white part: variance
    fs / T (generation.py:27-37 zeroes the f = 0 cell and the period is the TOD).  A pink
    part that kept periods longer than the TOD would add several times that."""
    D = 4096  # the variance of 4096 means: 2.2 % sampling noise (256 detectors left 9 %)
    x = _generate(gpu_ctx, D, T, fs, knee, seed=12, batch=1024)
    m = x.double().mean(dim=1).cpu().numpy()
    assert abs(m.var() / (fs / T) - 1) < 0.08, m.var() / (fs / T)
    x = x[:256].cpu().numpy().astype(np.float64)
    # and the spectrum still follows 2 (1 + knee / f) from the second harmonic of the TOD on
    X = np.abs(np.fft.rfft(x, axis=1)) ** 2
    psd = 2 * X.mean(axis=0) / (fs * T)  # one-sided, per Hz: the white level is 2
    f = np.fft.rfftfreq(T, 1 / fs)
    for j in (2, 3, 5, 8):
        assert abs(psd[j] / (2 * (1 + knee / f[j])) - 1) < 0.35, (j, psd[j], 2 * (1 + knee / f[j]))


@pytest.mark.parametrize("D,T", [(1, 1), (1, 2), (2, 3), (3, 7), (5, 4097)])
def test_tiny_and_odd_sizes(gpu_ctx, D, T):
    """One detector (a pair with no partner), a single sample, lengths below the 4-sample
    group, one past the minimum period: finite output of the right size, nothing written
    beyond it, with and without modes."""
    import torch

    for modes in (0, 2):
        basis = None if modes == 0 else np.ones((D, modes)) / np.sqrt(modes)
        buf = torch.full((D, T + 4), 9.0, dtype=torch.float32, device="cuda:0")
        out = _generate(gpu_ctx, D, T, 50.0, 2.0, corr=0.5, basis=basis, seed=3, out=buf[:, :T])
        x = out.cpu().numpy()
        assert x.shape == (D, T) and np.isfinite(x).all() and bool((buf[:, T:] == 9.0).all())
        if T > 1000:
            assert 0.5 < x.std() / np.sqrt(50.0 * (1 + 2 * 2.0 / 50.0 * np.log(T / 2))) < 2.0


@pytest.mark.parametrize("fs,knee,rate", [(400.0, 1.0, 4), (400.0, 2.0, 2)])
def test_two_rate_form(gpu_ctx, fs, knee, rate):
    """Where the pink part at fs / (2 rate) is below 2 % of the white level (and T >= 32768) the generator makes the
    pink and correlated-pink parts at fs / rate, interpolates them, and draws the white parts per sample
    (noise_two_rate_kernel).  Against the one-rate form (MRX_OPT_NOISE_GENERIC bit 8) on the same parameters: the
    same spectrum band by band (the missing pink power above fs / (2 rate) is < 2 %), the same covariance between
    detectors below and above the knee, unit white level; and as before: shards bit-identical to their rows,
    batch-size independent, accumulating, loading-dependent amplitude."""
    import torch

    from maria_amd import _lib
    from maria_amd import noise as mnoise
    from maria_amd import synthetic

    D, T, c = 96, 100003, 0.5
    off = synthetic.hex_pack(D, np.radians(0.5))
    B = mnoise.spatial_basis(off, k=5, n_side=16, scale=mnoise.diameter(off))
    scale = np.linspace(1.0, 2.0, D)
    two = _generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B, scale=scale, seed=31, batch=64)
    gpu_ctx.set_option(_lib.OPT_NOISE_GENERIC, 8)
    try:
        one = _generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B, scale=scale, seed=31, batch=64)
    finally:
        gpu_ctx.set_option(_lib.OPT_NOISE_GENERIC, 0)
    assert not torch.equal(one, two)  # (another realisation: the white parts are drawn per sample)
    x2, x1 = (t.cpu().numpy().astype(np.float64) / scale[:, None] for t in (two, one))
    f = np.fft.rfftfreq(T, 1 / fs)
    p2, p1 = ((np.abs(np.fft.rfft(x, axis=1)) ** 2).mean(axis=0) for x in (x2, x1))
    cut = fs / (2 * rate)
    for lo, hi in [(0.02, 0.2), (0.2, 2.0), (2.0, 0.25 * cut), (0.25 * cut, 0.5 * cut), (0.5 * cut, cut), (cut, 2 * cut), (2 * cut, fs / 2)]:
        m = (f >= lo) & (f < hi)
        if not m.any():
            continue
        assert abs(p2[m].sum() / p1[m].sum() - 1) < 0.04, (lo, hi, p2[m].sum() / p1[m].sum())
    # the white level alone -- the top of the band, where the pink part is below a per cent -- is that of the model:
    # a detector's own white part plus the modes' through its row of the basis
    m = f > 0.4 * fs
    assert abs(p2[m].mean() / (T * fs * (1 + c * (B**2).sum(axis=1).mean())) - 1) < 0.03
    # covariance between detectors: c B B^T (1 + knee / f) + ((1 - c) knee / f + 1) I, below and above the knee
    off_diag = ~np.eye(D, dtype=bool)
    for lo, hi in [(0.05, knee), (20.0, 0.45 * fs)]:
        got = []
        for x in (x2, x1):  # (the low band holds a few hundred bins of a 250 s series: the one-rate form sets the scale)
            cov, fb = _band_covariance(x, fs, lo, hi, lambda f: 1 + knee / f)
            ind = np.mean(((1 - c) * knee / fb + 1) / (1 + knee / fb))
            model = c * (B @ B.T) + ind * np.eye(D)
            s = np.trace(cov) / np.trace(model)
            got.append((np.corrcoef(cov[off_diag], model[off_diag])[0, 1], np.sum(cov[off_diag] * model[off_diag]) / np.sum(model[off_diag] ** 2) / s))
        (r2, slope2), (r1, slope1) = got
        print(f"band {lo}-{hi} Hz: correlation with the model {r2:.3f} (one-rate {r1:.3f}), slope {slope2:.3f} ({slope1:.3f})")
        assert r2 > r1 - 0.08 and abs(slope2 - slope1) < 0.15, (lo, hi, got)
        if lo > knee:
            assert r2 > 0.9 and abs(slope2 - 1) < 0.1, (lo, hi, got)
    # zero mean of the pink part over the TOD: the mean of a row is that of its white part, sqrt(fs / T) sigma
    means = x2.mean(axis=1)
    assert abs(means.std() / np.sqrt(fs / T * (1 + c * (B**2).sum(axis=1).mean())) - 1) < 0.3
    # shards, batches, accumulation, loading
    for lo, hi in [(0, 16), (16, 96), (34, 35)]:
        part = _generate(gpu_ctx, hi - lo, T, fs, knee, corr=c, basis=B[lo:hi], scale=scale[lo:hi], seed=31, det_offset=lo)
        if (hi - lo) % 2 == 0 or hi == D:
            assert torch.equal(part, two[lo:hi])
        else:
            assert torch.allclose(part, two[lo:hi], rtol=0, atol=2e-4 * float(two.abs().max()))
    assert torch.equal(_generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B, scale=scale, seed=31, batch=10), two)
    base = torch.full((D, T + 5), 3.0, dtype=torch.float32, device="cuda:0")
    acc = _generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B, scale=scale, seed=31, accumulate=1, out=base)
    assert torch.allclose(acc[:, :T], two + 3.0, atol=1e-4) and bool((acc[:, T:] == 3.0).all())
    loading = torch.rand((D, T), device="cuda:0") + 0.5
    lnep = _generate(gpu_ctx, D, T, fs, knee, corr=c, basis=B, scale=scale, seed=31, loading=loading, per_loading=0.7)
    want = two * ((torch.as_tensor(scale, dtype=torch.float32, device="cuda:0")[:, None] + 0.7 * loading) / torch.as_tensor(scale, dtype=torch.float32, device="cuda:0")[:, None])
    assert torch.allclose(lnep, want, rtol=2e-6, atol=1e-5 * float(two.abs().max()))


@pytest.mark.parametrize("T", [32768, 65536 - 3, 131072])
@pytest.mark.parametrize("fs,knee", [(400.0, 1.5), (400.0, 1.0)])  # rate 2, rate 4
def test_work_buffer_size_covers_the_two_rate_form_for_small_batches(gpu_ctx, T, fs, knee):
    """mrx_noise_work_floats sees neither the sample rate nor the knee, so its size holds every form the generator can
    pick: for T at or just below a power of two the slow series' period equals the full one and the two-rate form
    needs MORE than the one-rate form for a batch of a few rows -- a band or a shard with two detectors used to fail
    with 'work buffer too small' although the caller followed the API."""
    from maria_amd import noise as mnoise
    from maria_amd import synthetic

    D = 2
    off = synthetic.hex_pack(D, np.radians(0.1))
    B = mnoise.spatial_basis(off, k=5, n_side=16, scale=max(mnoise.diameter(off), 1e-3))
    x = _generate(gpu_ctx, D, T, fs, knee, corr=0.3, basis=B, batch=D).cpu().numpy().astype(np.float64)
    assert np.isfinite(x).all()
    # the top of the band is white at the model's level: own part + the modes through this row of the basis
    p = (np.abs(np.fft.rfft(x, axis=1)) ** 2)[:, np.fft.rfftfreq(T, 1 / fs) > 0.4 * fs].mean(axis=1)
    np.testing.assert_allclose(p / (T * fs * (1 + 0.3 * (B**2).sum(axis=1))), 1.0, rtol=0.1)


@pytest.mark.parametrize("exact", [False, True], ids=["two_rate_default", "exact_spectrum"])
def test_front_end_spectrum_against_the_reference_law(gpu_ctx, exact):
    """BOTH forms of the generator, as the front end selects them -- the default (two-rate from 32 768 samples on) and
    ``noise_kwargs={"exact_spectrum": True}`` (the one-rate form) -- against the reference's own law, not against each other:
    noise/generation.py:27-38 gives a two-sided pink spectrum a / |f|, a = knee / 2, beside white noise of variance fs, i.e.
    a one-sided density 2 NEP^2 (1 + knee / f).  Octave bands from 16 bins of the TOD up to fs / 2: every band within its own
    statistical error (3 sigma) + 0.7 % -- except that the two-rate form may miss what it says it leaves out:
    the pink power above fs / 8 (rate 4), at most 2 % of the density there, and up to 4 % of the pink part in the octave
    below -- nothing anywhere else."""
    from maria_amd import noise as mnoise
    from maria_amd.instrument import Band, Detectors

    fs, knee, T, D = 400.0, 1.0, 120000, 128
    band = Band(center=150e9, width=30e9, shape="top_hat", name="f150", NEP=1e-12, knee=knee)  # 1e12 NEP = 1: the density in units of NEP^2
    dets = Detectors.hexagon(D, 0.5, [band], primary_size=10.0)
    kwargs = {"correlated_noise_proportion": 0.0, "exact_spectrum": exact}
    x = mnoise.simulate_noise(gpu_ctx, dets, T, fs, seed=11, noise_kwargs=kwargs).cpu().numpy().astype(np.float64)
    assert gpu_ctx.__dict__.get("_options", {}).get(5, 0) == 0  # the option is back where it was
    f = np.fft.rfftfreq(T, 1 / fs)
    p = (np.abs(np.fft.rfft(x, axis=1)) ** 2).mean(axis=0) * 2.0 / (fs * T)  # one-sided density
    want = 2.0 * (1.0 + knee / np.maximum(f, 1e-30))
    hi = fs / 2
    while hi / 2 >= 16 * fs / T:  # octaves downwards from the Nyquist frequency: the slow rate's own edges are among them
        lo = hi / 2
        m = (f >= lo) & (f < hi)
        ratio = p[m].sum() / want[m].sum()
        sigma = 1.0 / np.sqrt(m.sum() * D)
        allowed_low = 0.0
        if not exact:
            pink_share = (knee / f[m]).sum() / (1.0 + knee / f[m]).sum()
            if lo >= fs / 8 * (1 - 1e-9):
                allowed_low = pink_share  # the pink part above the slow Nyquist frequency (fs / 8 at rate 4) is not made at all
            elif hi >= fs / 8 * (1 - 1e-9):
                allowed_low = 0.04 * pink_share  # the cubic's roll-off in the octave below
        assert -(3 * sigma + 0.007 + allowed_low) < ratio - 1 < 3 * sigma + 0.007, (lo, hi, ratio, sigma, allowed_low)
        hi = lo
    if not exact:
        # ... and the default DID take the two-rate form here: another realisation than the exact one
        y = mnoise.simulate_noise(gpu_ctx, dets, T, fs, seed=11, noise_kwargs=dict(kwargs, exact_spectrum=True)).cpu().numpy()
        assert not np.array_equal(x.astype(np.float32), y)
