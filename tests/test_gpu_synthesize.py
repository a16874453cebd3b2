"""GPU tests of mrx_atm_synthesize -- atmosphere -> TOD in ONE launch, the sampler and the writer as two roles of
one grid with the hand-over on the device (the reference's _simulate_atmosphere from the layer loop to the
interpolation at the sample rate: atmosphere/atmosphere.py:317-373, sim/atmosphere.py:43-82).

The bar is bit-identity with the two-call form (mrx_atm_sample + mrx_spline_upsample_fused, which the oracle tests
pin): the same bodies run in the same order, only who waits for whom differs -- so every word is compared, on
shapes that exercise the block / pitch / tail arithmetic, under uneven load (a head start, one or several resident
sampler workgroups per CU, blocks that do not divide the rows) and repeatedly (the control block is reused)."""

import numpy as np
import pytest

from helpers import small_problem
from maria_amd.synthetic import config_problem

pytestmark = pytest.mark.gpu


def _path(problem, ctx, **kw):
    from maria_amd.pipeline import DevicePath

    return DevicePath(problem, device="cuda:0", ctx=ctx, **kw)


def _two_call(path):
    """The reference form for these tests: the stages back to back on one stream."""
    import torch

    tod = path.run(blocks=1)
    coarse = path.coarse_loading().clone()
    torch.cuda.synchronize()
    return tod, coarse


@pytest.mark.parametrize(
    "n_det,n_layers,n_bands,block_rows,head_rows,wgs",
    [
        (67, 3, 2, 0, 0, 3),        # one short block, pitch 96 > rows
        (300, 1, 1, 256, 0, 2),     # 256 + 44 rows
        (300, 1, 1, 256, 256, 3),   # the first block as head start
        (1000, 8, 3, 512, 300, 1),  # head rounded up to a block; one resident workgroup per CU
        (1000, 8, 3, 256, 5000, 2), # head longer than the shard: everything sampled before the writers enter
        (33, 2, 1, 1000, 0, 7),     # block_rows > rows
    ],
)
def test_one_launch_equals_the_two_calls(gpu_ctx, n_det, n_layers, n_bands, block_rows, head_rows, wgs):
    import torch

    p = small_problem(n_det=n_det, n_layers=n_layers, n_bands=n_bands, gain=True)
    path = _path(p, gpu_ctx)
    path.clear_flags()
    want, coarse = _two_call(path)
    for _ in range(3):  # the launch leaves its control block as it found it
        got = torch.full_like(want, float("nan"))
        path.synthesize(got, block_rows=block_rows, head_rows=head_rows, resident_wgs_per_cu=wgs)
        torch.cuda.synchronize()
        assert path.check_flags() == 0
        assert torch.equal(got, want)
        assert torch.equal(path.coarse_loading(), coarse)


def test_sixteen_layers_and_a_long_coarse_axis(gpu_ctx):
    """BASELINE config 5's shape in small: 16 layers (the sampler's anchors are cut to fit under the writer's LDS
    images), an upsampling ratio of 40, rows that are no multiple of 32."""
    import torch

    p = config_problem("atlast_50k", n_det=777, duration=120.0, side=512)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    want, coarse = _two_call(path)
    assert path.check_flags() == 0
    for block_rows, head_rows, wgs in ((256, 256, 3), (512, 0, 4), (1024, 0, 2)):
        got = torch.full_like(want, float("nan"))
        path.synthesize(got, block_rows=block_rows, head_rows=head_rows, resident_wgs_per_cu=wgs)
        torch.cuda.synchronize()
        assert path.check_flags() == 0
        assert torch.equal(got, want), (block_rows, head_rows, wgs)
        assert torch.equal(path.coarse_loading(), coarse)


def test_caller_row_order_and_strided_output(gpu_ctx):
    """The TOD lands in the caller's row order (d_rows) and in a buffer with a pitch of its own."""
    import torch

    p = small_problem(n_det=300, n_layers=3, n_bands=2, gain=True)
    path = _path(p, gpu_ctx)
    want, _ = _two_call(path)
    wide = torch.full((path.D, path.T + 12), 7.0, dtype=torch.float32, device="cuda:0")
    path.synthesize(wide[:, : path.T], block_rows=256, head_rows=0)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, : path.T], want)
    assert bool((wide[:, path.T :] == 7.0).all())


def test_flags_travel_and_unsupported_plans_are_refused(gpu_ctx):
    """A line of sight off its screen is flagged as by mrx_atm_sample (atmosphere.py:368-369); the literal cell
    rule, the float32 pointing chain and non-uniform axes are the two-call form's: MRX_ERR_UNSUPPORTED, nothing
    launched -- and DevicePath.run() then takes the two calls by itself."""
    import torch

    from maria_amd import _lib

    p = small_problem(n_det=100, n_layers=2)
    p["layers"][1]["extrusion"] = p["layers"][1]["extrusion"] + 400.0
    path = _path(p, gpu_ctx)
    path.clear_flags()
    path.synthesize()
    with pytest.raises(RuntimeError, match="introduced nans"):
        path.check_flags()
    path.clear_flags()

    q = small_problem(n_det=100, n_layers=2)
    good = _path(q, gpu_ctx)
    want, _ = _two_call(good)
    gpu_ctx.set_option(_lib.OPT_AXIS_LITERAL, 1)
    try:
        with pytest.raises(_lib.MrxError) as e:
            good.synthesize()
        assert e.value.code == -4
    finally:
        gpu_ctx.set_option(_lib.OPT_AXIS_LITERAL, 0)
    assert torch.equal(good.synthesize(), want)


def test_screens_beyond_the_pixel_kernels_addressing_take_the_general_kernel(gpu_ctx):
    """The pixel-coordinate sampler forms a 32-bit byte offset with a 24-bit row multiply: a screen side of 2^22 nodes or
    more (or a screen of 4 GiB) must not reach it.  Such a plan is not `all_pixel`: the one-launch form -- which has no
    other sampler -- refuses it, mrx_atm_sample takes the general kernel and flags what leaves the (here tiny) screen
    instead of reading wrong pixels."""
    import torch

    from maria_amd import _lib

    def wide(n_c):
        p = small_problem(n_det=40, n_layers=1)
        lay = p["layers"][0]
        dc = float(lay["cross_section"][1] - lay["cross_section"][0])
        lay["extrusion"] = np.asarray(lay["extrusion"])[:8]
        lay["cross_section"] = float(lay["cross_section"][0]) + dc * np.arange(n_c)
        lay["values"] = np.zeros((8, n_c), np.float32)
        return p

    below = _path(wide((1 << 22) - 8), gpu_ctx)
    below.clear_flags()
    below.synthesize()  # accepted: the pixel kernel's range
    torch.cuda.synchronize()
    below.clear_flags()
    del below
    torch.cuda.empty_cache()
    beyond = _path(wide((1 << 22) + 8), gpu_ctx)
    with pytest.raises(_lib.MrxError) as e:
        beyond.synthesize()
    assert e.value.code == -4
    beyond.clear_flags()
    beyond.sample()  # the general kernel: no fault, and the lines of sight that leave the 8 rows are flagged
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="introduced nans"):
        beyond.check_flags()
    beyond.clear_flags()
    del beyond
    torch.cuda.empty_cache()


def test_krj_on_the_coarse_grid_in_the_same_launch(gpu_ctx):
    """mrx_atm_synthesize_krj: TOD.to("K_RJ") (tod/tod.py:106-142) applied to the coarse loading in the sampler role's
    epilogue -- the same bits as mrx_coarse_to_krj between the two calls, tail past the last knot included; two bands,
    one polarised, rolled calibration offsets, a gain."""
    import torch

    from maria_amd import synthetic
    from test_gpu_calibration import _cal_tables

    p = small_problem(n_det=700, n_bands=2, n_layers=3, gain=True)
    _, el_full = synthetic.daisy_scan(p["t"])
    roll = np.radians(17.0)
    R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
    path = _path(p, gpu_ctx)
    path.set_calibration(_cal_tables(2), 273.15, 1.0, el_full, p["offsets"] @ R.T, [False, True])
    assert 0 < path.coarse_krj_bound() <= path.COARSE_KRJ_LIMIT
    want = path.run(blocks=1, krj=True)
    torch.cuda.synchronize()
    assert 0 < path._krj_split() < path.T
    for block_rows, head_rows, wgs in ((256, 0, 3), (512, 256, 2), (0, 0, 4)):
        got = torch.full_like(want, float("nan"))
        path.synthesize(got, block_rows=block_rows, head_rows=head_rows, resident_wgs_per_cu=wgs, krj=True)
        torch.cuda.synchronize()
        assert path.check_flags() == 0
        assert torch.equal(got, want), (block_rows, head_rows, wgs)
    with pytest.raises(RuntimeError, match="K_RJ"):
        path.coarse_loading()
    # and pW again afterwards
    pw = path.run(blocks=1)
    assert torch.equal(path.synthesize(), pw)


def test_full_size_default_run_is_the_one_launch_form_and_bit_identical(gpu_ctx):
    """atlast_10k at full size: run() takes the one-launch form by default; its TOD equals the stages back to back
    word for word, twenty times over with the launch shapes the sweeps covered (hand-overs under load, every
    consumer's caches warm from the run before)."""
    import torch

    p = config_problem("atlast_10k")
    path = _path(p, gpu_ctx)
    path.generate_screens()
    want = path.run(blocks=1)
    torch.cuda.synchronize()
    assert path.check_flags() == 0
    assert path.synthesize_applies()
    got = torch.empty_like(want)
    shapes = [(None, None, None), (512, 0, 3), (1024, 3000, 2), (256, 1024, 4), (2048, 4096, 1)]
    for rep in range(20):
        br, head, wgs = shapes[rep % len(shapes)]
        got.fill_(float("nan"))
        if br is None:
            path.run(got)
            assert path._synthesized
        else:
            path.synthesize(got, block_rows=br, head_rows=head, resident_wgs_per_cu=wgs)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (rep, br, head, wgs)
    assert path.check_flags() == 0
    # the coarse loading the launch leaves behind is the sampler's
    path.run(got)
    c1 = path.coarse_loading().clone()
    path.sample()
    assert torch.equal(path.coarse_loading(), c1)
    del got, want
    torch.cuda.empty_cache()


def test_hand_over_beside_other_work_on_the_chip(gpu_ctx):
    """The hand-over under uneven load: another stream streams through 8 GB (and a third one spins arithmetic) while the
    launch runs, so that the roles' workgroups are dispatched late and unevenly and every cache is busy with other lines;
    every word of the TOD still equals the two calls', ten times."""
    import torch

    p = config_problem("atlast_10k", n_det=6000)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    want = path.run(blocks=1)
    torch.cuda.synchronize()
    got = torch.empty_like(want)
    big = torch.ones(1 << 30, dtype=torch.float32, device="cuda:0")  # 4 GB read + 4 GB written per pass
    small = torch.rand(1 << 22, dtype=torch.float32, device="cuda:0")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(10):
        got.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for _ in range(1 + rep % 3):
                big.mul_(1.0000001)
        with torch.cuda.stream(s2):
            for _ in range(20):
                small = torch.sin(small) * 1.0001 + 0.1
        path.synthesize(got, block_rows=512, head_rows=(rep % 4) * 1024, resident_wgs_per_cu=2 + rep % 2)
        torch.cuda.synchronize()
        assert torch.equal(got, want), rep
    assert path.check_flags() == 0
    del big, got, want
    torch.cuda.empty_cache()


def test_the_writer_on_its_own_takes_tiles_from_a_queue(gpu_ctx):
    """mrx_spline_upsample_fused is a resident grid over a tile queue: many launches in a row (the queue is left
    zero by the launch itself) on a shape with more tiles than workgroups, against the two-call form's spline."""
    import torch

    from maria_amd._lib import ptr

    rng = np.random.default_rng(5)
    D, Ta, ratio = 3000, 300, 40
    T = (Ta - 1) * ratio + 17
    y = torch.as_tensor(rng.standard_normal((Ta, D)).astype(np.float32)).cuda()
    t = torch.as_tensor(np.arange(T) / 400.0).cuda()
    ref = torch.empty((D, T), dtype=torch.float32, device="cuda:0")
    ym = torch.empty((Ta, D, 2), dtype=torch.float32, device="cuda:0")
    gpu_ctx.call("mrx_spline_prepare", ptr(y), D, Ta, ptr(ym))
    gpu_ctx.call("mrx_spline_upsample", ptr(ym), D, Ta, 0.0, 0.1, ptr(t), T, None, None, ptr(ref), T)
    out = torch.empty_like(ref)
    first = None
    for _ in range(12):
        out.fill_(float("nan"))
        gpu_ctx.call("mrx_spline_upsample_fused", ptr(y), D, Ta, 0.0, 0.1, ptr(t), T, None, None, ptr(out), T)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        if first is None:
            first = out.clone()
            assert float((out - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
        else:
            assert torch.equal(out, first)
