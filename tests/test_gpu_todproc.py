"""GPU parity of the TOD pre-processing row (tod/processing.py:91-204) against the
numpy / scipy restatement in oracle/todproc.py (pinned by the reference's own utils/signal
functions through tests/golden/leaves.json)."""

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _tod(D=37, T=3001, fs=50.0, seed=0):
    from maria_amd import synthetic
    from maria_amd.instrument import Band, Detectors, Instrument
    from maria_amd.sim import TOD, Coordinates

    rng = np.random.default_rng(seed)
    t = 1.7e9 + np.arange(T) / fs
    az, el = synthetic.daisy_scan(t)
    band = Band(center=150e9, width=30e9, name="f150")
    dets = Detectors(synthetic.hex_pack(D, np.radians(0.3)), [band])
    common = np.cumsum(rng.normal(size=T)) * 0.5
    sig = (30.0 + np.outer(rng.uniform(0.8, 1.2, D), common) + rng.normal(size=(D, T)) + np.linspace(0, 5, T)[None]).astype(np.float32)
    extra = (0.1 * rng.normal(size=(D, T))).astype(np.float32)
    coords = Coordinates(t, az, el, offsets=dets.offsets)
    return TOD(data={"atmosphere": sig, "noise": extra}, dets=dets, coords=coords, units="pW"), sig + extra, t, el


CONFIGS = [
    {"remove_slope": {}},
    {"window": {"name": "tukey", "kwargs": {"alpha": 0.2}}},
    {"filter": {"f_lower": 0.2}},
    {"filter": {"f_upper": 5.0, "f_lower": 0.1, "order": 1}},
    {"filter": {"f_upper": 8.0, "order": 3}},
    {"filter": {}},
    {"remove_spline": {"knot_spacing": 10.0}},
    {"remove_spline": {"knot_spacing": 15.0, "order": 2, "remove_el_gradient": True}},
    {"remove_modes": {"modes_to_remove": 2}},
    {"remove_slope": {}, "remove_spline": {"knot_spacing": 20.0}, "window": {"name": "hann"}, "filter": {"f_lower": 0.3, "f_upper": 10.0},
     "remove_modes": {"modes_to_remove": 1}},
]


@pytest.mark.parametrize("config", CONFIGS, ids=lambda c: "+".join(c))
def test_process_tod_matches_oracle(gpu_ctx, config):
    from maria_amd.tod_processing import process_tod
    from oracle import todproc

    tod, signal, t, el = _tod()
    ref, ref_w = todproc.process_tod(signal, t, el, {k: dict(v) for k, v in config.items()})
    out = process_tod(tod, config={k: dict(v) for k, v in config.items()}, ctx=gpu_ctx)
    got = out.data["total"].cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32 and out.fields == ["total"]
    np.testing.assert_allclose(out.weight, ref_w, rtol=0, atol=1e-15)
    # float32 storage between the operations (the reference keeps float64 after `filter`)
    tol = 2e-6 * np.abs(ref).max()
    if "remove_modes" in config and "filter" not in config:
        # the reference hands svds a float32 matrix here: its own modes carry float32 rounding
        # of the (much larger) unprocessed signal
        tol += 3e-6 * np.abs(signal).max()
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)


@pytest.mark.parametrize("T", [1, 2, 100, 256, 257, 5000])
def test_sosfilt_lengths_and_in_place(gpu_ctx, T):
    """The time-parallel recursion against scipy.signal.sosfilt for lengths below, at and past
    the chunk, 4 sections (low + high pass), in place and out of place."""
    import ctypes as C

    import scipy.signal
    import torch

    from maria_amd import tod_processing as tp
    from maria_amd._lib import ptr

    rng = np.random.default_rng(T)
    D = 9
    x = np.cumsum(rng.normal(size=(D, T)), axis=1).astype(np.float32)
    sos = np.ascontiguousarray(np.concatenate([tp.bessel_sos(4.0, 50.0, 1, "low"), tp.bessel_sos(0.5, 50.0, 1, "high")]))
    ref = scipy.signal.sosfilt(sos, x.astype(np.float64), axis=-1)
    M = torch.as_tensor(tp.chunk_matrix(sos, gpu_ctx.lib.mrx_sosfilt_chunk())).to("cuda:0")
    need = C.c_size_t()
    gpu_ctx.lib.mrx_sosfilt_work_doubles(D, T, len(sos), C.byref(need))
    work = torch.empty(need.value, dtype=torch.float64, device="cuda:0")
    d_in = torch.as_tensor(x).to("cuda:0")
    buf = torch.full((D, T + 3), 5.0, dtype=torch.float32, device="cuda:0")
    gpu_ctx.call("mrx_sosfilt", sos.ctypes.data_as(C.POINTER(C.c_double)), len(sos), ptr(M), ptr(d_in), d_in.stride(0), D, T, 0,
                 ptr(buf), buf.stride(0), ptr(work))
    got = buf[:, :T].cpu().numpy()
    assert bool((buf[:, T:] == 5.0).all())
    scale = np.abs(ref).max() + 1e-30
    assert np.abs(got - ref).max() <= 2e-7 * scale + 1e-7 * np.abs(ref).max()
    gpu_ctx.call("mrx_sosfilt", sos.ctypes.data_as(C.POINTER(C.c_double)), len(sos), ptr(M), ptr(d_in), d_in.stride(0), D, T, 0,
                 ptr(d_in), d_in.stride(0), ptr(work))
    assert torch.equal(d_in, buf[:, :T])


def test_mapper_with_preprocessing_recovers_a_map_under_atmosphere(gpu_ctx):
    """The reference's mapping recipe: atmosphere + map TOD, common modes and slow drifts removed
    (tod_preprocessing), then binned.  Without the pre-processing the atmosphere buries the
    source; with it the source stands out of the map."""
    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.mappers import BinMapper
    from maria_amd.sim import Plan, Simulation

    band = Band(center=150e9, width=30e9, shape="top_hat", name="f150")
    inst = Instrument(Detectors.hexagon(61, 0.3, [band], primary_size=30.0))
    plan = Plan.daisy(start_time=1.7e9, duration=120.0, sample_rate=50.0, scan_center=(100.0, 60.0), radius=0.25, speed=0.5)
    n = 64
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    sky = mmap.ProjectionMap(0.05 * np.exp(-(X**2 + Y**2) / 0.01).astype(np.float32), nu=150e9, width=1.0, center=(100.0, 60.0), frame="az/el")
    sim = Simulation(inst, plan, Site(altitude=5000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2, "seed": 4, "pwv_rms_frac": 0.1}, map=sky, noise=False)
    (tod,) = sim.run(units="K_RJ")
    ratio = float(tod.data["atmosphere"].std() / tod.data["map"].std())
    assert ratio > 10, ratio  # the sky signal is buried
    kw = dict(center=(100.0, 60.0), width=0.5, resolution=1.0 / 60, frame="az/el", units="K_RJ")
    raw = BinMapper([tod], **kw).run().data[0, 0]
    pre = {"remove_modes": {"modes_to_remove": 1}, "remove_spline": {"knot_spacing": 10.0}}
    cleaned = BinMapper([tod], tod_preprocessing=pre, **kw).run().data[0, 0]
    centre = (slice(13, 17), slice(13, 17))
    edge = np.r_[cleaned[:5].ravel(), cleaned[-5:].ravel()]
    edge = edge[np.isfinite(edge)]
    snr_clean = (np.nanmean(cleaned[centre]) - np.nanmean(edge)) / np.nanstd(edge)
    raw_edge = np.r_[raw[:5].ravel(), raw[-5:].ravel()]
    raw_edge = raw_edge[np.isfinite(raw_edge)]
    snr_raw = (np.nanmean(raw[centre]) - np.nanmean(raw_edge)) / np.nanstd(raw_edge)
    assert snr_clean > 5 and snr_clean > 3 * abs(snr_raw), (snr_clean, snr_raw)
