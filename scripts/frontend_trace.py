#!/usr/bin/env python3
"""Development aid: Simulation(atlast_10k-shaped).run() under rocprofv3 --kernel-trace: which kernels, how long."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maria_amd.instrument import Band, Detectors, Instrument, Site
from maria_amd.sim import Plan, Simulation
noise = len(sys.argv) > 1 and sys.argv[1] == "noise"
band = Band(center=150e9, width=30e9, shape="top_hat", name="f150")
inst = Instrument(Detectors.hexagon(10000, 2.0, [band], primary_size=50.0))
plan = Plan.daisy(start_time=1.7e9, duration=600.0, sample_rate=400.0, scan_center=(45.0, 60.0), radius=0.5, speed=0.5)
sim = Simulation(inst, plan, Site(altitude=5000.0), atmosphere="2d", noise=noise, device_output=True, noise_seed=1, progress_bars=False)
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
