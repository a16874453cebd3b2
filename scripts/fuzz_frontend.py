#!/usr/bin/env python3
"""Development aid: randomised instruments, scans and atmospheres through Simulation.run() against the oracle chain on the
downloaded screens (pW and K_RJ, gains, detector shards with the noise on, both turbulence spectra, 1-6 layers or the 3-D model, both
interpolation methods).
Usage: python scripts/fuzz_frontend.py [seed] [trials] [big]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from maria_amd.instrument import Band, Detectors, Instrument, Site
from maria_amd.sim import Plan, Simulation
from oracle import hotpath

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
big = len(sys.argv) > 3 and sys.argv[3] == "big"  # thousands of rows: the pipelined step (2 and 4 detector blocks on two streams)
rng = np.random.default_rng(seed)
bad = 0


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-300))


for trial in range(trials):
    n_bands = int(rng.integers(1, 3))
    centers = [93e9, 150e9, 220e9]
    bands = [Band(center=centers[b], width=0.25 * centers[b], shape="top_hat", name=f"b{b}", gain_error=float(rng.choice([0.0, 0.05])),
                  NEP_per_loading=float(rng.choice([0.0, 0.0, 0.1])), knee=float(rng.choice([0.3, 1.0]))) for b in range(n_bands)]  # (the NEP that grows with the loading: the sharded leg below)
    n = int(rng.integers(7, 400)) if not big else int(rng.integers(2100 // n_bands, 5200 // n_bands))
    inst = Instrument(Detectors.hexagon(n, float(rng.uniform(0.05, 1.0)), bands, primary_size=float(rng.uniform(3.0, 30.0))))
    duration, fs = float(rng.uniform(8.0, 60.0)), float(rng.choice([20.0, 50.0, 100.0]) if not big else rng.choice([50.0, 100.0, 400.0]))
    az0 = float(rng.choice([0.0, 359.95, 180.0])) if rng.random() < 0.25 else float(rng.uniform(0, 360))  # (the wrap of the azimuth)
    el0 = float(rng.uniform(25.0, 84.0))
    pattern = str(rng.choice(["daisy", "daisy", "triangle", "lissajous", "stare"]))
    if pattern == "daisy":
        plan = Plan.daisy(start_time=1.7e9, duration=duration, sample_rate=fs, scan_center=(az0, min(el0, 70.0)), radius=float(rng.uniform(0.1, 0.8)), speed=float(rng.uniform(0.2, 0.8)))
    else:  # boresight tracks given sample by sample (radians): a constant-elevation sweep with sharp turns, a Lissajous box, a stare
        t = np.arange(1.7e9, 1.7e9 + duration, 1.0 / fs)
        s_, A, B = t - t[0], np.radians(rng.uniform(0.1, 1.0)), np.radians(rng.uniform(0.05, 0.5))
        if pattern == "triangle":
            period = float(rng.uniform(4.0, 20.0))
            az = np.radians(az0) + A * (2 * np.abs(2 * ((s_ / period) % 1.0) - 1) - 1) / np.cos(np.radians(el0))
            el = np.full_like(t, np.radians(el0))
        elif pattern == "lissajous":
            az = np.radians(az0) + A * np.sin(2 * np.pi * s_ / rng.uniform(3.0, 15.0)) / np.cos(np.radians(el0))
            el = np.radians(el0) + B * np.sin(2 * np.pi * s_ / rng.uniform(3.0, 15.0) + rng.uniform(0, 6.28))
        else:
            az, el = np.full_like(t, np.radians(az0)), np.full_like(t, np.radians(el0))
        plan = Plan(t, az, el)
    units = str(rng.choice(["pW", "K_RJ"]))
    spectrum = str(rng.choice(["covariance", "power_law"]))
    model = "3d" if rng.random() < 0.3 else "2d"
    method = "cubic" if rng.random() < 0.25 else "linear"
    kw = {"seed": int(rng.integers(1, 1 << 30)), "weather": {"pwv": float(rng.uniform(0.3, 3.0))}, "turbulence_spectrum": spectrum, "interpolation_method": method}
    if model == "3d":  # (a few hundred thin layers of one volume: kept low and short)
        kw["max_height"] = float(rng.uniform(300.0, 2500.0))
    else:
        kw["n_layers"] = int(rng.integers(1, 7))
    altitude = float(rng.uniform(0.0, 5000.0))
    if rng.random() < 0.5:  # another wind (drift direction, ribbon length, time step) and ground temperature
        from maria_amd.atmosphere import SyntheticWeather

        kw["weather_profile"] = SyntheticWeather(base_altitude=altitude, pwv=kw["weather"]["pwv"], base_temperature=float(rng.uniform(270.0, 290.0)),
                                                 wind_speed=float(rng.uniform(2.0, 30.0)), wind_direction_deg=float(rng.uniform(0, 360)))
    try:
        sim = Simulation(inst, plan, Site(altitude=altitude, region="synthetic"), atmosphere=model, atmosphere_kwargs=kw, noise=False,
                         gain_seed=int(rng.integers(1, 1000)))
        (tod,) = sim.run(units=units)
    except Exception as exc:  # noqa: BLE001
        bad += 1
        print(f"trial {trial}: n={inst.dets.n} bands={n_bands} {units} {spectrum} {model} {method} {pattern} az {az0:.1f} el {el0:.1f}: {type(exc).__name__}: {exc} BAD", flush=True)
        continue
    data = tod.data["atmosphere"]
    obs = sim.obs_list[0]
    atm, dets = obs.atmosphere, inst.dets
    path = atm._device_path()
    layers = [dict(l, values=bufs[0].cpu().numpy()) for l, bufs in zip(atm._layer_list(), path._layer_bufs)]
    p = dict(t=obs.coords.t, ta=atm.boresight.t, az_a=atm.boresight.az, el_a=atm.boresight.el, offsets=dets.offsets, band_index=dets.band_index,
             m00=dets.mueller00(), layers=layers, tables=atm._tables(dets), T0=float(atm.weather.temperature[0]), pwv0=float(atm.weather.pwv),
             timestep=float(atm.timestep), gain=None, interpolation_method=method)
    ref = hotpath.run_path(p)
    if units == "K_RJ":
        sp = atm.spectrum
        tables = [{"T": sp.side_base_temperature, "pwv": sp.side_zenith_pwv, "el": sp.side_elevation,
                   "values": hotpath.transmission_integral_grid(b.passband, sp.side_nu, sp._opacity)} for b in dets.bands]
        _, el_det = hotpath.broadcast(obs.coords.offsets, obs.boresight.az, obs.boresight.el)
        ref = hotpath.calibrate_to_krj(ref, dets.band_index, tables, tod.metadata["base_temperature"], tod.metadata["pwv"], el_det)
    T = data.shape[1]
    gain = np.median(data / ref, axis=1)  # the per-detector gain error, exp(0.05 N(0,1)) or 1
    err = rel(data, ref * gain[:, None])
    ok = err <= 2e-5 and np.isfinite(data).all() and 0.7 < gain.min() and gain.max() < 1.4
    bad += not ok
    if not ok and units == "K_RJ":  # which form of the conversion it was
        bound = path.coarse_krj_bound()
        print(f"   coarse-form bound {bound:.2e} (limit {path.COARSE_KRJ_LIMIT:.0e}: {'coarse grid' if bound <= path.COARSE_KRJ_LIMIT else 'per sample'}); "
              f"elevation step per knot {np.abs(np.diff(atm.boresight.el)).max():.2e} rad, scan el range {np.ptp(atm.boresight.el):.3f} rad", flush=True)
    print(f"trial {trial}: n={dets.n} bands={n_bands} T={T} Ta={len(atm.boresight.t)} {model} layers={len(layers)} {method} {units} {spectrum} {pattern} az {az0:.1f} el {el0:.1f}: {err:.2e} "
          f"gain {gain.min():.3f}..{gain.max():.3f} {'ok' if ok else 'BAD'}", flush=True)
    # a detector shard of the same simulation, with the noise on: every field the same rows bit for bit, and the
    # round trip through the other unit within float32
    if rng.random() < 0.5 and dets.n >= 48:
        from maria_amd.dist import shard_bounds

        world = int(rng.integers(2, 5))
        rank = int(rng.integers(0, world))
        common = dict(atmosphere=model, atmosphere_kwargs=kw, noise=True, gain_seed=7, noise_seed=int(rng.integers(1, 1000)))
        try:
            (full,) = Simulation(inst, plan, sim.site, **common).run(units=units)
            (part,) = Simulation(inst, plan, sim.site, shard=(rank, world), **common).run(units=units)
            lo, hi = shard_bounds(dets.n, world, rank)
            same = [f for f in ("atmosphere", "noise") if not np.array_equal(part.data[f], full.data[f][lo:hi])] or True
            other = "pW" if units == "K_RJ" else "K_RJ"
            back = part.to(other).to(units)
            rt = max(rel(back.data[f], part.data[f]) for f in ("atmosphere", "noise")) if hi > lo else 0.0  # (a rank of no rows)
            ok2 = same is True and rt < 2e-6 and np.isfinite(full.data["noise"]).all()
        except Exception as exc:  # noqa: BLE001
            import traceback

            traceback.print_exc()
            ok2, same, rt = False, f"{type(exc).__name__}: {exc}", float("nan")
        bad += not ok2
        print(f"   shard {rank}/{world} with noise: rows identical {same}, unit round trip {rt:.1e} {'ok' if ok2 else 'BAD'}", flush=True)
        if not ok2 and same is not True and not isinstance(same, str):
            starts = [int(np.nonzero(dets.band_index == b)[0][0]) for b in range(n_bands)]
            d = np.abs(part.data["noise"] - full.data["noise"][lo:hi])
            rows_bad = np.nonzero(d.max(axis=1) > 0)[0]
            print(f"      {units}; rows [{lo}, {hi}); bands start at {starts}, n = {dets.n}; differing shard rows {rows_bad[:8]} ... ({len(rows_bad)}), "
                  f"max |diff| / max |noise| {d.max() / np.abs(full.data['noise']).max():.2e}", flush=True)
    del sim, tod
print("BAD" if bad else "all ok", bad)
