#!/usr/bin/env python3
"""Development aid: Simulation(atlast_10k-shaped).run() under rocprofv3 --kernel-trace: which kernels, how long."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd.instrument import Band, Detectors, Instrument, Site
from maria_amd.sim import Plan, Simulation
mode = sys.argv[1] if len(sys.argv) > 1 else "atm"   # atm | noise | map | map_noise
noise = mode.endswith("noise")
band = Band(center=150e9, width=30e9, shape="top_hat", name="f150")
inst = Instrument(Detectors.hexagon(10000, 2.0, [band], primary_size=50.0))
plan = Plan.daisy(start_time=1.7e9, duration=600.0, sample_rate=400.0, scan_center=(45.0, 60.0), radius=0.5, speed=0.5)
site = Site(altitude=5000.0)
sky = None
if mode.startswith("map"):  # bench.py's map rows: a 1024^2 map in the ra/dec frame around the scanned patch
    import numpy as np
    from maria_amd import map as mmap
    from maria_amd.sim import sky_transform_stack
    M = sky_transform_stack(plan.time[::4000], site.latitude, site.longitude)
    az, el = plan.phi[::4000], plan.theta[::4000]
    xyz = np.einsum("ti,tij->tj", np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1), M).mean(axis=0)
    centre = (float(np.degrees(np.arctan2(xyz[1], xyz[0]) % (2 * np.pi))), float(np.degrees(np.arcsin(xyz[2] / np.linalg.norm(xyz)))))
    X, Y = np.meshgrid(np.linspace(-1, 1, 1024), np.linspace(-1, 1, 1024))
    sky = mmap.ProjectionMap((1e-3 * np.exp(-((X - 0.2) ** 2 + (Y + 0.1) ** 2) / 0.05)).astype(np.float32), nu=150e9, width=4.0, center=centre, frame="ra/dec")
sim = Simulation(inst, plan, site, atmosphere="2d", map=sky, noise=noise, device_output=True, noise_seed=1, progress_bars=False)
for k in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    (tod,) = sim.run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"run {k}: {1e3 * (time.perf_counter() - t0):.2f} ms (enqueue {1e3 * (t1 - t0):.2f}); device allocs {st['num_device_alloc']} frees {st['num_device_free']} "
          f"reserved {st['reserved_bytes.all.current'] / 1e9:.1f} GB", flush=True)
    del tod
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
(tod,) = sim.run(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
