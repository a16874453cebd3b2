#!/usr/bin/env python3
"""Development aid: randomised Simulation(map=...) runs and BinMapper runs against the oracle chain (oracle/mapsample.py):
frames, one or two bands on one or two map planes, polarised detectors on I / IQU maps, rectangular maps, with and without
an atmosphere (the map's calibration then follows the pwv), nearest-pixel and bilinear binning, detector shards.
Usage: python scripts/fuzz_map_frontend.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from maria_amd import map as mmap
from maria_amd.instrument import Band, Detectors, Instrument, Site, compute_angular_fwhm
from maria_amd.mappers import BinMapper
from maria_amd.sim import Plan, Simulation, sky_transform_stack
from oracle import hotpath, mapsample

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
bad = 0


def centre_of(az, el, transform):
    phi, theta = mapsample.frame_angles(az[None, :], el[None, :], transform)
    xyz = mapsample.phi_theta_to_xyz(phi[0], theta[0]).astype(float).mean(axis=0)
    xyz /= np.linalg.norm(xyz)
    return float(np.arctan2(xyz[1], xyz[0]) % (2 * np.pi)), float(np.arcsin(xyz[2]))


for trial in range(trials):
    n_bands = int(rng.integers(1, 3))
    centers = [93e9, 150e9]
    bands = [Band(center=centers[b], width=0.27 * centers[b], shape="top_hat", name=f"b{b}") for b in range(n_bands)]
    n = int(rng.integers(7, 120))
    polarised = bool(rng.random() < 0.5)
    inst = Instrument(Detectors.hexagon(n, float(rng.uniform(0.1, 0.6)), bands, primary_size=float(rng.uniform(5.0, 30.0))))
    dets = inst.dets
    if polarised:
        dets.gamma = np.where(rng.random(dets.n) < 0.3, np.nan, rng.uniform(0, np.pi, dets.n))
    plan = Plan.daisy(start_time=1.7e9, duration=float(rng.uniform(10.0, 60.0)), sample_rate=float(rng.choice([20.0, 50.0, 100.0])),
                      scan_center=(float(rng.uniform(0, 360)), float(rng.uniform(35.0, 70.0))), radius=float(rng.uniform(0.1, 0.5)), speed=float(rng.uniform(0.2, 0.6)))
    site = Site(altitude=float(rng.uniform(0.0, 5000.0)), latitude=float(rng.uniform(-60, 60)), longitude=float(rng.uniform(-180, 180)), region="synthetic")
    frame = str(rng.choice(["az/el", "ra/dec"]))
    with_atm = bool(rng.random() < 0.6)
    transform = sky_transform_stack(plan.time, site.latitude, site.longitude) if frame == "ra/dec" else None
    az32, el32 = plan.phi.astype(np.float32), plan.theta.astype(np.float32)
    centre = centre_of(az32, el32, transform)
    n_eta, n_xi = int(rng.integers(16, 130)), int(rng.integers(16, 130))
    stokes = "IQU" if polarised and rng.random() < 0.7 else "I"
    nus = [centers[b] for b in range(n_bands)] if rng.random() < 0.7 else [centers[0]]
    width = float(rng.uniform(0.6, 2.0))
    res = width / n_xi
    X, Y = np.meshgrid(np.linspace(-1, 1, n_xi), np.linspace(-1, 1, n_eta))
    planes = []
    for s in range(len(stokes)):
        planes.append(np.stack([(1 + c) * (0.1 if s else 1.0) * np.exp(-((X - rng.uniform(-0.3, 0.3)) ** 2 + (Y - rng.uniform(-0.3, 0.3)) ** 2) / rng.uniform(0.02, 0.3))
                                + 0.02 * rng.normal(size=X.shape) for c in range(len(nus))]))
    data = np.stack(planes).astype(np.float32)  # [S, C, eta, xi]
    label = f"trial {trial}: n={dets.n} bands={n_bands} {frame} map {len(stokes)}x{len(nus)}x{n_eta}x{n_xi} atm={with_atm} pol={polarised}"
    try:
        sky = mmap.ProjectionMap(data, nu=nus if len(nus) > 1 else nus[0], stokes=stokes, width=width, height=res * n_eta, center=np.degrees(centre), frame=frame, degrees=True)
        kw = {"n_layers": int(rng.integers(1, 4)), "seed": int(rng.integers(1, 1 << 30))}
        common = dict(map=sky, noise=False, **(dict(atmosphere="2d", atmosphere_kwargs=kw) if with_atm else {}))
        sim = Simulation(inst, plan, site, **common)
        (tod,) = sim.run(units="pW")
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"{label}: {type(exc).__name__}: {exc} BAD", flush=True)
        continue
    got = tod.data["map"]
    obs = sim.obs_list[0]
    az_d, el_d = hotpath.broadcast(obs.coords.offsets, az32, el32)
    ref = np.zeros_like(got)
    grad = 0.0
    if with_atm:
        atm = obs.atmosphere
        path = atm._device_path()
        coarse = path.coarse_pwv().cpu().numpy()
        ta = path.ta0 + path.dta * np.arange(coarse.shape[1])
        sp = atm.spectrum
    sidx = ["IQUV".index(s) for s in stokes]
    for b, band in enumerate(bands):
        rows = np.nonzero(dets.band_index == b)[0]
        fwhm = float(compute_angular_fwhm(fwhm_0=dets.primary_size.mean(), z=np.inf, nu=band.center))
        sig_x, sig_y = fwhm / np.sqrt(8 * np.log(2)) / abs(sky.x_res), fwhm / np.sqrt(8 * np.log(2)) / abs(sky.y_res)
        smoothed = hotpath.map_smooth(sky.data, None, sig_y, sig_x)
        smoothed = np.asarray(smoothed[0] if isinstance(smoothed, tuple) else smoothed, np.float32)
        chans, tabs, scal = [], [], []
        for c, (lo, hi) in enumerate(sky.nu_bin_bounds):
            if band.nu.max() < lo or hi < band.nu.min():
                continue
            chans.append(c)
            if with_atm:
                mask = (sp.side_nu >= lo) & (sp.side_nu < hi)
                tabs.append(np.trapezoid(band.passband(sp.side_nu[mask]) * np.exp(-sp._opacity[..., mask]), x=sp.side_nu[mask], axis=-1))
            else:
                mask = (band.nu >= lo) & (band.nu < hi)
                scal.append(float(np.trapezoid(band.passband(band.nu[mask]), x=band.nu[mask])))
        if not chans:
            continue
        values = np.swapaxes(smoothed[:, chans], 0, 1)
        grad = max(grad, np.abs(np.diff(values, axis=-1)).max() / abs(sky.xi[1] - sky.xi[0]), np.abs(np.diff(values, axis=-2)).max() / abs(sky.eta[1] - sky.eta[0]))
        sw = mapsample.mueller_row(dets.gamma[rows])[:, sidx]
        if with_atm:
            ref[rows] = mapsample.sample_maps(az_d[rows], el_d[rows], plan.time, ta, coarse[rows], sky.eta, sky.xi, sky.center, values, sw, cal_tables=tabs,
                                              cal_axes=(sp.side_base_temperature, sp.side_zenith_pwv, sp.side_elevation),
                                              base_temperature=atm.weather.temperature[0], transform_stack=transform)
        else:
            ref[rows] = mapsample.sample_maps(az_d[rows], el_d[rows], plan.time, None, None, sky.eta, sky.xi, sky.center, values, sw, cal_scalars=scal,
                                              transform_stack=transform)
    bound = 2 * 6e-7 * grad * 1.5 * 1e12 * mapsample.K_B * 5e10 + 1e-5 * np.abs(ref).max()
    err = float(np.abs(got - ref).max())
    ok = np.isfinite(got).all() and np.abs(ref).max() > 0 and err <= bound
    bad += not ok
    print(f"{label} T={got.shape[1]}: |diff| {err:.2e} bound {bound:.2e} max {np.abs(ref).max():.2e} {'ok' if ok else 'BAD'}", flush=True)

    # the default units: the same per-sample factor on the map field as on the atmosphere field (tod/tod.py:130-136)
    if with_atm:
        try:
            (tod_k,) = Simulation(inst, plan, site, **common).run()
            f_atm = tod_k.data["atmosphere"].astype(np.float64) / tod.data["atmosphere"]
            e3 = float(np.abs(tod_k.data["map"] - got * f_atm).max() / max(np.abs(got * f_atm).max(), 1e-300))
            ok3 = tod_k.units == "K_RJ" and e3 <= 5e-6 and np.isfinite(tod_k.data["map"]).all()
            msg3 = f"{e3:.1e}"
        except Exception as exc:  # noqa: BLE001
            ok3, msg3 = False, f"{type(exc).__name__}: {exc}"
        bad += not ok3
        print(f"   K_RJ: map field against pW x the atmosphere's factor {msg3} {'ok' if ok3 else 'BAD'}", flush=True)

    # the mapper on that TOD against the oracle's binning of the same samples
    bilinear = bool(rng.random() < 0.4)
    mres = float(rng.uniform(0.5, 3.0)) * res
    mwidth, mheight = float(rng.uniform(0.4, 1.2)) * width, float(rng.uniform(0.4, 1.2)) * res * n_eta
    try:
        only = type(tod)({"map": got}, tod.dets, tod.coords, units="pW", metadata=tod.metadata)
        mapper = BinMapper([only], center=np.degrees(centre), width=mwidth, height=mheight, resolution=mres, stokes=stokes, nu=nus, frame=frame, units="pW", bilinear=bilinear)
        out = mapper.run()
        chan = np.zeros(dets.n, np.int32)
        for k, nu in enumerate(nus):
            chan[dets.band_center == nu] = k
        ref_sum, ref_wgt = mapsample.bin_map(az_d, el_d, got, None, mapper.eta, mapper.xi, mapper.center, mapsample.mueller_row(dets.gamma)[:, sidx], len(stokes),
                                             channel=chan, n_channels=len(nus), transform_stack=transform, bilinear=bilinear)
        got_sum, got_wgt = mapper.products["sum"], mapper.products["weight"]
        total = abs(got_wgt[0].sum() / ref_wgt[0].sum() - 1)
        if bilinear:
            tol = 6e-7 / abs(mapper.xi[1] - mapper.xi[0]) * 4 if mapper.n_xi > 1 else 1e-9
            e1 = np.abs(got_sum - ref_sum).max() / max(np.abs(ref_sum).max(), 1e-300)
            e2 = np.abs(got_wgt - ref_wgt).max() / ref_wgt.max()
            ok2 = total < 1e-9 and e1 <= tol and e2 <= tol
            msg = f"sum {e1:.1e} wgt {e2:.1e} tol {tol:.1e}"
        else:
            # nearest pixel: samples within the float32 rounding of the offsets (6e-7 rad) of a pixel edge may change bins; with
            # the mapper's unit weights a pixel that loses one and gains one keeps its weight, so the sums are held to the
            # number of samples that can have moved and to their total (tests/test_gpu_map.py holds them pixel by pixel
            # with random weights)
            pixel = min(abs(mapper.xi[1] - mapper.xi[0]) if mapper.n_xi > 1 else 1.0, abs(mapper.eta[1] - mapper.eta[0]) if mapper.n_eta > 1 else 1.0)
            may_move = 4 * 6e-7 / pixel + 1e-4
            moved = np.abs(got_wgt[0] - ref_wgt[0]).sum() / ref_wgt[0].sum()
            e1 = np.abs(got_sum - ref_sum).sum() / (np.abs(got).max() * ref_wgt[0].sum())
            e2 = abs(got_sum[0].sum() - ref_sum[0].sum()) / np.abs(ref_sum[0]).sum()
            ok2 = total < 1e-9 and moved < may_move and e1 <= 2 * may_move and e2 < 1e-9
            msg = f"moved {moved:.1e} (may {may_move:.1e}) sum of |d sum| {e1:.1e} total {e2:.1e}"
    except Exception as exc:  # noqa: BLE001
        ok2, msg = False, f"{type(exc).__name__}: {exc}"
    bad += not ok2
    print(f"   BinMapper {mapper.n_eta}x{mapper.n_xi} {'bilinear' if bilinear else 'nearest'}: {msg} {'ok' if ok2 else 'BAD'}", flush=True)
    del sim, tod
    torch.cuda.empty_cache()
print("BAD" if bad else "all ok", bad)
