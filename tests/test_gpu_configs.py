"""GPU parity on the other BASELINE.json configurations (they are parity cases, not
bench lines): MUSTANG-2 60 s / 600 s against the oracle in full, ACT-like 3 bands and the
AtLAST 50k per-GPU shard (time-chunked TOD, 16 layers of 4096^2) through spot checks."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(name, gpu_ctx, **kw):
    import torch

    from maria_amd import synthetic
    from maria_amd.pipeline import DevicePath

    p = synthetic.config_problem(name, **kw)
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.generate_screens()
    tod = path.run()
    torch.cuda.synchronize()
    assert path.check_flags() == 0
    return p, path, tod


def _oracle_rows(p, path, rows):
    from oracle import hotpath

    sub = dict(p)
    for key in ("offsets", "band_index", "m00"):
        sub[key] = p[key][rows]
    sub["layers"] = [dict(l, values=b[0].cpu().numpy()) for l, b in zip(p["layers"], path._layer_bufs)]
    return hotpath.run_path(sub)


@pytest.mark.parametrize("name", ["mustang2_60s", "mustang2_600s"])
def test_mustang2_configs_match_oracle_in_full(gpu_ctx, name):
    p, path, tod = _run(name, gpu_ctx)
    ref = _oracle_rows(p, path, np.arange(path.D))
    got = tod.cpu().numpy()
    assert got.shape == ref.shape == (217, len(p["t"]))
    assert np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() <= 1e-5


def test_act_like_three_bands(gpu_ctx):
    """9000 rows = 3000 positions x 3 bands, each band its own table."""
    p, path, tod = _run("act_3k", gpu_ctx)
    assert tuple(tod.shape) == (9000, 240000) and len(p["tables"]) == 3
    rng = np.random.default_rng(0)
    rows = np.sort(np.r_[rng.choice(9000, 18, replace=False), [0, 2999, 3000, 5999, 6000, 8999]])
    ref = _oracle_rows(p, path, rows)
    got = tod[rows].cpu().numpy()
    assert np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() <= 1e-5
    # the three bands see different emission: their mean loadings are distinct
    means = [float(tod[b * 3000 : (b + 1) * 3000, ::1000].mean()) for b in range(3)]
    assert len({round(m, 3) for m in means}) == 3


def test_atlast_50k_shard_time_chunked(gpu_ctx):
    """Config 5 on one GPU: a 1/8 detector shard (6250 rows), 16 layers of 4096^2
    screens, 3600 s at 400 Hz.  The 36 GB TOD is written in 600 s chunks through the
    ld_out / d_t window of mrx_spline_upsample_fused, as a consumer that cannot hold it would;
    chunks must tile the un-chunked result exactly."""
    import torch

    from maria_amd import dist, synthetic
    from maria_amd.pipeline import DevicePath
    from maria_amd._lib import ptr

    p = synthetic.config_problem("atlast_50k")
    sl = dist.shard_slice(50000, world_size=8, rank=3)
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx, det_slice=sl)
    assert path.D == 6256 and path.T == 1440000 and path.Ta == 36000
    path.generate_screens()
    assert path._layer_bufs[0][0].shape == (4096, 4096) and len(path._layer_bufs) == 16
    path.sample()
    assert path.check_flags() == 0
    chunk = 240000  # 600 s
    buf = torch.empty((path.D, chunk), dtype=torch.float32, device="cuda:0")
    rng = np.random.default_rng(2)
    rows = np.sort(rng.choice(path.D, 6, replace=False))
    pieces = []
    for s in range(0, path.T, chunk):
        d_t = path.d_t[s : s + chunk]
        path.ctx.call(
            "mrx_spline_upsample_fused", ptr(path.d_loading), path.D, path.Ta, path.ta0, path.dta,
            ptr(d_t), chunk, None, ptr(path.d_rows), ptr(buf), chunk,
        )
        assert bool(torch.isfinite(buf).all())
        pieces.append(buf[rows].cpu().numpy())
    got = np.concatenate(pieces, axis=1)
    ref = _oracle_rows(p, path, np.arange(sl.start, sl.stop)[rows])  # global rows of the shard rows
    assert got.shape == ref.shape
    assert np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max() <= 1e-5
