"""GPU tests at BASELINE.json's full size (atlast_10k: 10 000 det x 240 000 samples,
8 layers of 2048^2 screens) through size-independent properties, plus an oracle
spot check on a random detector subset."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(gpu_ctx):
    import torch

    from maria_amd import synthetic
    from maria_amd.pipeline import DevicePath

    p = synthetic.config_problem("atlast_10k")
    path = DevicePath(p, device="cuda:0", ctx=gpu_ctx)
    path.generate_screens()
    tod = path.run(blocks=1)  # the stages back to back: the tests below call them one by one
    torch.cuda.synchronize()
    assert path.check_flags() == 0
    yield p, path, tod
    del tod
    torch.cuda.empty_cache()


def test_full_size_shapes_and_finiteness(full):
    import torch

    p, path, tod = full
    assert tuple(tod.shape) == (10000, 240000) and tod.dtype == torch.float32
    assert bool(torch.isfinite(tod).all())
    assert len(p["layers"]) == 8 and path._layer_bufs[0][0].shape == (2048, 2048)
    assert path.plan_info() == (16, True)


def test_spline_interpolates_its_knots(full):
    """Every 40th sample sits on a coarse knot: the TOD there is the coarse loading."""
    import torch

    p, path, tod = full
    ratio = round(p["fs"] * p["timestep"])
    at_knots = tod[:, ::ratio][:, : path.Ta]
    coarse = path.coarse_loading()
    err = (at_knots - coarse).abs().max() / coarse.abs().max()
    assert float(err) <= 1e-6
    # and between knots the cubic stays within the local range of the knots (no ringing)
    lo = torch.minimum(coarse[:, :-1], coarse[:, 1:]).min(dim=1).values
    hi = torch.maximum(coarse[:, :-1], coarse[:, 1:]).max(dim=1).values
    span = (hi - lo).clamp_min(1e-12)
    body = tod[:, : (path.Ta - 1) * ratio]
    assert bool(((body.min(dim=1).values - lo) / span > -0.05).all()) and bool(((body.max(dim=1).values - hi) / span < 0.05).all())


def test_runs_are_deterministic_and_shards_bit_identical(full):
    import torch

    from maria_amd.pipeline import DevicePath

    p, path, tod = full
    again = path.run()
    assert torch.equal(again, tod)
    del again
    sl = slice(4992, 5200)  # crosses a 256-detector workgroup boundary
    shard = DevicePath(p, device="cuda:0", ctx=path.ctx, det_slice=sl)
    shard.set_screens([b[0] for b in path._layer_bufs])
    part = shard.run()
    assert torch.equal(part, tod[sl])


def test_pipelined_run_is_bit_identical_to_the_serial_one(full):
    """DevicePath.run() cuts the shard into detector blocks and runs the sampler of block b+1
    beside the writer of block b on two streams (the default from 4096 rows up; the first block is half as long as the others so that the writer starts early): same kernels,
    same rows, same bits -- for several block counts, and twice in a row (buffer reuse)."""
    import torch

    p, path, tod = full
    for blocks in (4, 3, 7, 8, 4):
        out = torch.full_like(tod, float("nan"))
        path.run(out, blocks=blocks)
        torch.cuda.synchronize()
        assert torch.equal(out, tod), blocks
        assert torch.equal(path.coarse_loading(), path.coarse_loading())  # assembled from the block buffers
    assert path.check_flags() == 0
    del out
    path.sample()  # back to the serial state for the tests below


def test_gain_is_linear(full):
    import torch

    p, path, tod = full
    path.set_gain(np.full(path.D, 2.0, np.float32))
    out = torch.empty_like(tod)
    path.upsample_fused(out)
    path.set_gain(None)
    assert torch.equal(out, 2.0 * tod)


def test_oracle_spot_check_at_full_size(full):
    """The numpy/scipy oracle on 24 random detector rows of the full problem."""
    from oracle import hotpath

    p, path, tod = full
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(path.D, 24, replace=False))
    sub = dict(p)
    for key in ("offsets", "band_index", "m00"):
        sub[key] = p[key][rows]
    sub["layers"] = [dict(l, values=b[0].cpu().numpy()) for l, b in zip(p["layers"], path._layer_bufs)]
    ref = hotpath.run_path(sub)
    got = tod[rows].cpu().numpy()
    err = np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err <= 1e-5, err
