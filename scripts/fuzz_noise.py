#!/usr/bin/env python3
"""Development aid: invariants of mrx_noise_generate over random sizes (the spectrum itself is held by tests/test_gpu_noise.py):
rows do not depend on the batch size, the row stride, the call they were drawn in (pair-aligned shards), or the second
transform's form; accumulate adds; values are finite, of zero mean and of the white level's variance at knee = 0.
Usage: python scripts/fuzz_noise.py [seed] [trials]"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from maria_amd._lib import Context, ptr

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
ctx = Context(0)
ctx.set_stream(torch.cuda.current_stream())
dev = "cuda:0"
bad = 0


def generate(D, T, fs, knee, corr, basis, scale, key, batch, det_offset=0, pitch=None, accumulate=0, out=None):
    n_modes = 0 if basis is None else basis.shape[1]
    need = C.c_size_t()
    assert ctx.lib.mrx_noise_work_floats(T, n_modes, min(batch, D), C.byref(need)) == 0
    work = torch.empty(need.value, dtype=torch.float32, device=dev)
    d_basis = None if basis is None else torch.as_tensor(np.ascontiguousarray(basis, np.float32)).to(dev)
    d_scale = None if scale is None else torch.as_tensor(np.asarray(scale, np.float32)).to(dev)
    pitch = pitch or T
    if out is None:
        out = torch.full((D, pitch), 7.0, dtype=torch.float32, device=dev)
    ctx.call("mrx_noise_generate", key, D, det_offset, T, float(fs), float(knee), float(corr), ptr(d_basis), n_modes,
             ptr(d_scale), None, 0, 0.0, ptr(out), out.stride(0), accumulate, ptr(work), need.value)
    torch.cuda.synchronize()
    return out


for trial in range(trials):
    kind = rng.integers(0, 4)
    T = int([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 70000), rng.integers(70000, 600000)][kind])
    D = int(rng.integers(1, 40) if T > 70000 else rng.integers(1, 300))
    fs = float(rng.choice([20.0, 50.0, 200.0, 400.0]))
    knee = float(rng.choice([0.0, 0.1, 1.0, 10.0]))
    n_modes = int(rng.choice([0, 0, 1, 3, 5, 8]))
    corr = float(rng.uniform(0.05, 0.9)) if n_modes else 0.0
    rows_total = D + int(rng.integers(0, 20))
    basis = rng.normal(size=(rows_total, n_modes)) if n_modes else None
    scale = rng.uniform(0.5, 2.0, rows_total) if rng.random() < 0.7 else None
    key = int(rng.integers(1, 1 << 40))
    label = f"trial {trial}: D={D} T={T} fs={fs} knee={knee} modes={n_modes}"
    try:
        ref = generate(D, T, fs, knee, corr, None if basis is None else basis[:D], None if scale is None else scale[:D], key, batch=1024)
        x = ref.cpu().numpy()
        msgs = []
        if not np.isfinite(x).all():
            msgs.append("not finite")
        # another batch size, a padded row pitch: the same bits, and nothing written past T
        b2 = int(rng.choice([1, 2, 3, 7, 16, 64]))
        pitch = T + int(rng.integers(1, 9))
        y = generate(D, T, fs, knee, corr, None if basis is None else basis[:D], None if scale is None else scale[:D], key, batch=b2, pitch=pitch).cpu().numpy()
        if not np.array_equal(y[:, :T], x):
            msgs.append(f"batch {b2} / pitch differs {np.abs(y[:, :T] - x).max():.2e}")
        if not (y[:, T:] == 7.0).all():
            msgs.append("wrote past T")
        # a shard drawn on its own (even first row; it may end inside a pair only at the end of the table)
        if D >= 4:
            a = 2 * int(rng.integers(0, D // 2))
            b = D if rng.random() < 0.5 else min(D, a + 2 * int(rng.integers(1, D // 2 + 1)))
            z = generate(b - a, T, fs, knee, corr, None if basis is None else basis[a:b], None if scale is None else scale[a:b], key, batch=64, det_offset=a).cpu().numpy()
            if not np.array_equal(z, x[a:b]):
                msgs.append(f"rows [{a}, {b}) drawn alone differ {np.abs(z - x[a:b]).max():.2e}")
        # the other form of the second transform
        ctx.set_option(5, 1)
        try:
            g = generate(D, T, fs, knee, corr, None if basis is None else basis[:D], None if scale is None else scale[:D], key, batch=1024).cpu().numpy()
        finally:
            ctx.set_option(5, 0)
        tol = 2e-5 * max(np.abs(x).max(), 1e-30)
        if np.abs(g - x).max() > tol:
            msgs.append(f"generic transform differs {np.abs(g - x).max() / np.abs(x).max():.2e}")
        # accumulate
        acc = generate(D, T, fs, knee, corr, None if basis is None else basis[:D], None if scale is None else scale[:D], key, batch=1024, accumulate=1,
                       out=torch.full((D, T), 3.0, dtype=torch.float32, device=dev)).cpu().numpy()
        if np.abs(acc - (x + 3.0)).max() > 1e-6 * max(1.0, np.abs(x).max()):
            msgs.append(f"accumulate differs {np.abs(acc - (x + 3.0)).max():.2e}")
        # the white level
        if knee == 0.0 and n_modes == 0 and D * T > 20000:
            s = np.ones(D) if scale is None else scale[:D]
            v = (x.astype(np.float64) ** 2).mean(axis=1) / (fs * s**2)
            if abs(v.mean() - 1) > 6 * np.sqrt(2.0 / (D * T)) + 1e-3:
                msgs.append(f"white level {v.mean():.4f}")
        ok = not msgs
    except Exception as exc:  # noqa: BLE001
        ok, msgs = False, [f"{type(exc).__name__}: {exc}"]
    bad += not ok
    print(f"{label}: {'ok' if ok else 'BAD ' + '; '.join(msgs)}", flush=True)
print("BAD" if bad else "all ok", bad)
