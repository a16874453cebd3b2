#!/usr/bin/env python3
"""Per-kernel timing of the hot path on one GPU (development aid; bench.py is the
contract benchmark).  Usage: python scripts/kbench.py [config] [reps]"""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from maria_amd import synthetic  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in evs])
    return float(np.median(ms)), float(ms.min())


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    p = synthetic.config_problem(config)
    path = DevicePath(p, device="cuda:0")
    D, T, Ta = path.D, path.T, path.Ta
    tod = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
    t0 = time.perf_counter()
    path.generate_screens()
    torch.cuda.synchronize()
    print(f"config {config}: D={D} T={T} Ta={Ta} L={len(p['layers'])}; first screens {1e3*(time.perf_counter()-t0):.1f} ms")
    L = len(p["layers"])
    rows = [
        ("screens(gen+smooth)", path.generate_screens, 12.0 * sum(len(l["extrusion"]) * len(l["cross_section"]) for l in p["layers"])),
        ("sample", path.sample, 4.0 * D * Ta),
        ("spline_prepare", path.prepare, 12.0 * D * Ta),
        ("upsample", lambda: path.upsample(tod), 4.0 * D * T + 8.0 * D * Ta + 8.0 * T),
    ]
    total = 0.0
    for name, fn, nbytes in rows:
        med, mn = timeit(fn, reps)
        total += med
        print(f"{name:22s} median {med:8.3f} ms  min {mn:8.3f} ms  alg {nbytes/1e9:7.3f} GB -> {nbytes/med/1e6:8.1f} GB/s")
    print(f"sum of medians {total:.3f} ms -> {D*T/total/1e6:.1f} G det-samples/s; point-layers/s in sample: {D*Ta*L/1e6:.1f} M")
    # the default-units writer (K_RJ fused) and the in-place conversion of a pW field
    Tg = np.array([250.0, 270.0, 290.0])
    pw = np.linspace(0.0, 10.0, 21)
    elg = np.radians(np.linspace(float(os.environ.get("MRX_EL0", "10.0")), 90.0, 33))
    elg[-1] = np.radians(90.1)
    tabs = [{"T": Tg, "pwv": pw, "el": elg, "values": 2e10 * np.exp(-(0.03 + 0.01 * pw[None, :, None]) / np.sin(np.minimum(elg, np.pi / 2))[None, None, :])
             * np.ones((3, 1, 1))} for _ in p["tables"]]
    az_full, el_full = synthetic.daisy_scan(p["t"])
    path.set_calibration(tabs, 273.0, 1.0, el_full, p["offsets"])
    med, mn = timeit(lambda: path.upsample_krj(tod), reps)
    print(f"upsample_krj (fused)   median {med:8.3f} ms  min {mn:8.3f} ms")
    med, mn = timeit(lambda: path.to_krj(tod), reps)
    print(f"tod_to_krj (in place)  median {med:8.3f} ms  min {mn:8.3f} ms")
    for g in (1, 2, 4, 8, 16):
        path.ctx.set_option(4, g)
        med, mn = timeit(lambda: path.upsample(tod), reps)
        print(f"upsample groups={g}: median {med:.3f} ms min {mn:.3f} ms")
    path.ctx.set_option(4, 0)
    # does the writer slow down in sequence?  (event-timed inside the sequence)
    def timed_in_sequence(pre):
        evs = []
        for _ in range(reps):
            pre()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); path.upsample(tod); b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))
    print(f"upsample after upsample : {timed_in_sequence(lambda: path.upsample(tod)):.3f} ms")
    print(f"upsample after prepare  : {timed_in_sequence(path.prepare):.3f} ms")
    print(f"upsample after sample   : {timed_in_sequence(path.sample):.3f} ms")
    print(f"upsample after screens  : {timed_in_sequence(path.generate_screens):.3f} ms")
    print(f"upsample after full step: {timed_in_sequence(lambda: (path.generate_screens(), path.sample(), path.prepare())):.3f} ms")
    # follow-on rows: K_RJ-fused writer and full-rate pointing
    from maria_amd._lib import ptr
    az_full, el_full = synthetic.daisy_scan(p["t"])
    T_, pw_, el_ = np.array([250.0, 270.0, 290.0]), np.linspace(0, 10, 21), np.radians(np.linspace(10, 90, 33))
    el_[-1] = np.radians(90.1)
    tabs = [{"T": T_, "pwv": pw_, "el": el_, "values": 2e10 * np.exp(-(0.03 + 0.01 * pw_[None, :, None]) / np.sin(np.minimum(el_, np.pi / 2))[None, None, :]) * np.ones((3, 1, 1))}
            for _ in p["tables"]]
    path.set_calibration(tabs, 273.15, 1.0, el_full, p["offsets"])
    med, mn = timeit(lambda: path.upsample_krj(tod), reps)
    print(f"upsample + K_RJ        median {med:8.3f} ms  min {mn:8.3f} ms  -> {(4.0*D*T)/med/1e6:8.1f} GB/s written")
    d_az = torch.as_tensor(az_full.astype(np.float32)).cuda()
    d_el = torch.as_tensor(el_full.astype(np.float32)).cuda()
    out_el = torch.empty_like(tod)
    med, mn = timeit(lambda: path.ctx.call("mrx_pointing_broadcast", ptr(d_az), ptr(d_el), T, ptr(path.d_dx), ptr(path.d_dy), D, ptr(tod), ptr(out_el), T), reps)
    print(f"pointing_broadcast     median {med:8.3f} ms  min {mn:8.3f} ms  -> {(8.0*D*T)/med/1e6:8.1f} GB/s written")
    del out_el
    print("plan_info (uniform axes, tables in LDS):", path.plan_info())
    for chain in (0, 1):
        for arrays in (0, 1):
            path.ctx.set_option(0, chain)
            path.ctx.set_option(1, arrays)
            med, mn = timeit(path.sample, reps)
            print(f"sample chain={chain} axis_literal={arrays}: median {med:.3f} ms min {mn:.3f} ms")
    path.ctx.set_option(0, 0)
    path.ctx.set_option(1, 0)
    for kt in (1, 2, 4):
        path.ctx.set_option(2, kt)
        med, mn = timeit(path.sample, reps)
        print(f"sample times_per_thread={kt}: median {med:.3f} ms min {mn:.3f} ms")
    path.ctx.set_option(2, 0)
    for kt in (2, 4):
        for chunk in (4, 8, 16, 32, 64):
            path.ctx.set_option(2, kt)
            path.ctx.set_option(3, chunk)
            med, mn = timeit(path.sample, reps)
            print(f"sample times_per_thread={kt} chunk={chunk}: median {med:.3f} ms min {mn:.3f} ms")
    path.ctx.set_option(2, 0)
    path.ctx.set_option(3, 0)
    assert path.check_flags() == 0


if __name__ == "__main__":
    main()
