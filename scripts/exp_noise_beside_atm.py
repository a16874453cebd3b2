#!/usr/bin/env python3
"""Experiment: the detector noise (first pass arithmetic-bound) generated on a stream of its own beside the atmosphere's
TOD synthesis (HBM-bound writer), instead of after it -- Simulation(noise=True) at 10 000 x 240 000, K_RJ."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maria_amd import noise as mnoise
from maria_amd._lib import Context
from maria_amd.instrument import Band, Detectors, Instrument, Site
from maria_amd.sim import Plan, Simulation

band = Band(center=150e9, width=30e9, shape="top_hat", name="f150")
inst = Instrument(Detectors.hexagon(10000, 2.0, [band], primary_size=50.0))
plan = Plan.daisy(start_time=1.7e9, duration=600.0, sample_rate=400.0, scan_center=(45.0, 60.0), radius=0.5, speed=0.5)
sim = Simulation(inst, plan, Site(altitude=5000.0), atmosphere="2d", noise=False, device_output=True, noise_seed=1, progress_bars=False)
(tod,) = sim.run()
obs = sim.obs_list[0]
path = obs.atmosphere._device_path()
dev = path.device
main = torch.cuda.current_stream(dev)
T = len(obs.coords.t)
fs = 400.0
nctx = Context(0)
side = path._side_stream(main)
# a stream for the noise that shares a queue with neither the main stream nor the sampler's
for _ in range(8):
    ns = torch.cuda.Stream(device=dev)
    path.ctx.set_stream(main)
    ok1 = path.ctx.streams_concurrent(ns)
    if ok1:
        break
nctx.set_stream(ns)
noise_out = torch.empty((10000, T), dtype=torch.float32, device=dev)
ev = torch.cuda.Event()


def atmosphere():
    (t,) = sim.run()
    return t


def noise(ctx):
    return mnoise.simulate_noise(ctx, inst.dets, T, fs, 1234, {}, device=dev, out=noise_out)


def sequential():
    t = atmosphere()
    path.ctx.set_stream(main)
    n = noise(path.ctx)
    path.to_krj(n)
    return t


def concurrent():
    ns.wait_stream(main)
    with torch.cuda.stream(ns):
        n = noise(nctx)
        ev.record(ns)
    t = atmosphere()
    main.wait_event(ev)
    path.ctx.set_stream(main)
    path.to_krj(n)
    return t


for name, fn in (("sequential", sequential), ("noise beside the atmosphere", concurrent), ("sequential", sequential), ("noise beside the atmosphere", concurrent)):
    ts = []
    for k in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    print(f"{name}: median {np.median(ts[1:]):.2f} ms min {np.min(ts[1:]):.2f} ms", flush=True)
