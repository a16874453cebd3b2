"""GPU tests of mrx_atm_synthesize -- atmosphere -> TOD in ONE launch: sampler work items and TOD tiles as two queues of
one resident grid, the hand-over between them on the device, time chunk by time chunk (the reference's
_simulate_atmosphere from the layer loop to the interpolation at the sample rate: atmosphere/atmosphere.py:317-373,
sim/atmosphere.py:43-82).

The bar is bit-identity with the two-call form (mrx_atm_sample + mrx_spline_upsample_fused, which the oracle tests
pin): the same bodies run in the same order, only who waits for whom differs -- so every word is compared, on
shapes that exercise the block / chunk / pitch / tail arithmetic, under uneven load (one to five resident workgroups per
CU, none to five of them sampling only, chunks of 8 to 64 steps, other streams' work on the chip) and repeatedly.

Every repeat loop CHANGES THE DATA between launches (`_Alternating`: two values of pwv0 -- every coarse sample and
every TOD word differ between the two -- and, where the path generates its screens, two seeds), and compares each
launch with the two-call reference of ITS data: the launches of one path share their coarse buffer and their control
block, so a byte a writer read from a cache line of the launch before -- a stale hand-over -- is a wrong byte here.  (With
the same problem launched again and again, as rounds 3-4 did, only the first launch after allocation could fail.)"""

import numpy as np
import pytest

from helpers import attach_numpy_screens, small_problem
from maria_amd.synthetic import config_problem

pytestmark = pytest.mark.gpu


def _path(problem, ctx, **kw):
    from maria_amd.pipeline import DevicePath

    return DevicePath(problem, device="cuda:0", ctx=ctx, **kw)


class _Alternating:
    """Two data sets on ONE path (same buffers, same control block): variant v has its own pwv0 and, for generated
    screens, its own seed; ``want[v]`` / ``coarse[v]`` are its two-call references (the stages back to back on one
    stream: run(blocks=1))."""

    def __init__(self, path, krj=False, generated=False, keep_coarse=True):
        import torch

        self.path, self.krj, self.generated = path, krj, generated
        self.pwv0 = (path.pwv0, path.pwv0 * 1.25)
        self.seed = (int(path.problem["seed"]), int(path.problem["seed"]) + 1)
        self.want, self.coarse = [], []
        for v in (0, 1):
            self.select(v)
            self.want.append(path.run(blocks=1, krj=krj).clone())
            self.coarse.append(path.coarse_loading().clone() if keep_coarse and not krj else None)
            torch.cuda.synchronize()
        assert not torch.equal(self.want[0], self.want[1])

    def select(self, v):
        self.path.pwv0 = self.pwv0[v]
        if self.generated:
            self.path.problem["seed"] = self.seed[v]
            self.path.generate_screens()

    def check(self, v, got, tag=None):
        import torch

        torch.cuda.synchronize()
        assert torch.equal(got, self.want[v]), tag
        if self.coarse[v] is not None:
            assert torch.equal(self.path.coarse_loading(), self.coarse[v]), tag


@pytest.mark.parametrize(
    "n_det,n_layers,n_bands,block_rows,samplers,chunk",
    [
        (67, 3, 2, 0, 2, 32),       # one short block, pitch 96 > rows
        (300, 1, 1, 256, 0, 8),     # 256 + 44 rows in two blocks; nobody only samples
        (300, 1, 1, 0, 3, 16),      # one block of 300 rows
        (1000, 8, 3, 512, 1, 64),   # two blocks, the longest chunks
        (1000, 8, 3, 256, 5, 32),   # four blocks; every workgroup but one samples first
        (33, 2, 1, 1000, 8, 4),     # block_rows > rows; option value 8: no dedicated samplers at all
    ],
)
def test_one_launch_equals_the_two_calls(gpu_ctx, n_det, n_layers, n_bands, block_rows, samplers, chunk):
    import torch

    p = small_problem(n_det=n_det, n_layers=n_layers, n_bands=n_bands, gain=True)
    path = _path(p, gpu_ctx)
    path.clear_flags()
    alt = _Alternating(path)
    got = torch.empty_like(alt.want[0])
    for rep in range(6):  # the launch leaves its control block as it found it; the data change every time
        v = rep % 2
        alt.select(v)
        got.fill_(float("nan"))
        path.synthesize(got, block_rows=block_rows, sampler_wgs_per_cu=samplers, chunk=chunk)
        alt.check(v, got, rep)
        assert path.check_flags() == 0


def test_sixteen_layers_and_a_long_coarse_axis(gpu_ctx):
    """BASELINE config 5's shape in small: 16 layers (the sampler's anchors are cut to fit under the writer's LDS
    images), an upsampling ratio of 40, rows that are no multiple of 32; generated screens, another seed every launch."""
    import torch

    p = config_problem("atlast_50k", n_det=777, duration=120.0, side=512)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    alt = _Alternating(path, generated=True)
    assert path.check_flags() == 0
    got = torch.empty_like(alt.want[0])
    for rep, (block_rows, samplers, chunk) in enumerate(((256, 3, 32), (512, 0, 16), (0, 2, 64), (0, 1, 8), (768, 4, 32), (0, 8, 32))):
        v = rep % 2
        alt.select(v)
        got.fill_(float("nan"))
        path.synthesize(got, block_rows=block_rows, sampler_wgs_per_cu=samplers, chunk=chunk)
        alt.check(v, got, (block_rows, samplers, chunk))
        assert path.check_flags() == 0


def test_caller_row_order_and_strided_output(gpu_ctx):
    """The TOD lands in the caller's row order (d_rows) and in a buffer with a pitch of its own."""
    import torch

    p = small_problem(n_det=300, n_layers=3, n_bands=2, gain=True)
    path = _path(p, gpu_ctx)
    want = path.run(blocks=1)
    torch.cuda.synchronize()
    wide = torch.full((path.D, path.T + 12), 7.0, dtype=torch.float32, device="cuda:0")
    path.synthesize(wide[:, : path.T], block_rows=256)
    torch.cuda.synchronize()
    assert torch.equal(wide[:, : path.T], want)
    assert bool((wide[:, path.T :] == 7.0).all())


def test_coarse_pwv_is_the_launchs_second_output(gpu_ctx):
    """keep_pwv (the map mixin's calibration reads the coarse zenith-scaled pwv, sim/map.py:117-135): the one launch
    writes it beside the loading -- the float64 values mrx_atm_sample writes, in one block of rows and in several --
    and run() keeps the one-launch form for such a path."""
    import torch

    p = small_problem(n_det=1100, n_layers=3, n_bands=2, gain=True)
    path = _path(p, gpu_ctx, keep_pwv=True)
    assert path.synthesize_applies()
    want = path.run(blocks=1)
    pwv = path.coarse_pwv().clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(pwv).all()) and float(pwv.std()) > 0
    # ... and as the map sampler reads it: time-major, the caller's rows of one band (coarse_pwv_time_major: one gather
    # from the launch's own array where that is one block, the blocks put together otherwise)
    rows = torch.as_tensor(np.random.default_rng(3).permutation(path.D)[:700], device=pwv.device)

    def time_major_ok():
        return (torch.equal(path.coarse_pwv_time_major(rows), pwv.index_select(0, rows).T)
                and torch.equal(path.coarse_pwv_time_major(), pwv.T))

    assert time_major_ok()
    for block_rows in (None, 0, 256, 512):
        path.d_pwv.fill_(float("nan"))
        got = path.synthesize(block_rows=block_rows)
        torch.cuda.synchronize()
        assert torch.equal(got, want)
        assert torch.equal(path.coarse_pwv(), pwv), block_rows
        assert time_major_ok(), block_rows
        assert (path._synth_block_rows >= path.D) == (block_rows in (None, 0))  # (kept pwv, library's choice: one block)
    path.d_pwv.fill_(float("nan"))
    assert torch.equal(path.run(), want) and path._synthesized
    assert torch.equal(path.coarse_pwv(), pwv) and time_major_ok()
    path.sample()  # and the two-call form's own layout again afterwards
    assert torch.equal(path.coarse_pwv(), pwv) and time_major_ok()


def test_the_two_call_forms_stay_selectable(gpu_ctx, monkeypatch):
    """``path.one_launch = False`` and MARIA_AMD_ONE_LAUNCH=0 send run() through the two-call forms (the fallback for a
    device whose hand-over raises MRX_FLAG_HANDOVER): the same TOD, and run() says which form it took."""
    import torch

    p = small_problem(n_det=1100, n_layers=3, n_bands=2, gain=True)
    path = _path(p, gpu_ctx)
    assert path.synthesize_applies()
    want = path.run().clone()
    assert path._synthesized
    path.one_launch = False
    assert not path.synthesize_applies()
    got = path.run()
    assert not path._synthesized and torch.equal(got, want)
    path.one_launch = True
    monkeypatch.setenv("MARIA_AMD_ONE_LAUNCH", "0")
    assert not path.synthesize_applies()
    got = path.run()
    assert not path._synthesized and torch.equal(got, want)
    monkeypatch.delenv("MARIA_AMD_ONE_LAUNCH")
    assert path.synthesize_applies() and torch.equal(path.run(), want) and path._synthesized
    assert path.check_flags() == 0


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
    want = good.run(blocks=1)
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
    """mrx_atm_synthesize_krj: TOD.to("K_RJ") (tod/tod.py:106-142) applied to the coarse loading in the sampler's
    epilogue -- the same bits as mrx_coarse_to_krj between the two calls, tail past the last knot included; two bands,
    one polarised, rolled calibration offsets, a gain; the data change between launches."""
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
    alt = _Alternating(path, krj=True)
    assert 0 < path._krj_split() < path.T
    got = torch.empty_like(alt.want[0])
    for rep, (block_rows, samplers, chunk) in enumerate(((256, 3, 32), (512, 2, 16), (0, 4, 64), (0, 0, 8))):
        v = rep % 2
        alt.select(v)
        got.fill_(float("nan"))
        path.synthesize(got, block_rows=block_rows, sampler_wgs_per_cu=samplers, chunk=chunk, krj=True)
        alt.check(v, got, (block_rows, samplers, chunk))
        assert path.check_flags() == 0
    with pytest.raises(RuntimeError, match="K_RJ"):
        path.coarse_loading()
    # and pW again afterwards
    pw = path.run(blocks=1)
    assert torch.equal(path.synthesize(), pw)


def test_full_size_default_run_is_the_one_launch_form_and_bit_identical(gpu_ctx):
    """atlast_10k at full size: run() takes the one-launch form by default; its TOD equals the stages back to back
    word for word, twenty times over with launch shapes around the default, the screens' seed and pwv0 changing between
    launches (hand-overs under load, every consumer's caches warm from the run before -- with the OTHER data)."""
    import torch

    p = config_problem("atlast_10k")
    path = _path(p, gpu_ctx)
    path.generate_screens()
    assert path.synthesize_applies()
    alt = _Alternating(path, generated=True, keep_coarse=False)
    assert path.check_flags() == 0
    got = torch.empty_like(alt.want[0])
    shapes = [(None, None, None), (0, 3, 16), (2048, 2, 64), (512, 1, 32), (0, 0, 8)]
    for rep in range(20):
        v = (rep // 2 + rep) % 2  # 0 1 1 0 0 1 1 0 ...: a launch also follows one with its own data
        alt.select(v)
        br, samplers, chunk = shapes[rep % len(shapes)]
        got.fill_(float("nan"))
        if br is None:
            path.run(got)
            assert path._synthesized
        else:
            path.synthesize(got, block_rows=br, sampler_wgs_per_cu=samplers, chunk=chunk)
        alt.check(v, got, (rep, br, samplers, chunk))
    assert path.check_flags() == 0
    # the coarse loading the launch leaves behind is the sampler's
    path.run(got)
    c1 = path.coarse_loading().clone()
    path.sample()
    assert torch.equal(path.coarse_loading(), c1)
    del got, alt
    torch.cuda.empty_cache()


def test_per_gpu_share_of_atlast_50k_is_bit_identical(gpu_ctx):
    """BASELINE config 5's per-GPU share at full size (6 250 detectors x 1 440 000 samples, 16 layers of 4096^2, a 36 GB
    TOD): where the sampling is as much work as the writing the writers sample where they would wait -- the one-launch
    form is run()'s default here too -- and the TOD equals the stages back to back word for word, in one block of
    rows and in several, the data changing between launches."""
    import torch

    p = config_problem("atlast_50k", n_det=6250)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    assert path.synthesize_applies()
    alt = _Alternating(path, generated=True, keep_coarse=False)
    got = torch.empty_like(alt.want[0])
    for rep, (br, samplers, chunk) in enumerate([(None, None, None), (2048, 3, 32), (0, 1, 16), (None, None, None)]):
        v = rep % 2
        alt.select(v)
        got.fill_(float("nan"))
        if br is None:
            path.run(got)
            assert path._synthesized
        else:
            path.synthesize(got, block_rows=br, sampler_wgs_per_cu=samplers, chunk=chunk)
        alt.check(v, got, (rep, br, samplers, chunk))
    assert path.check_flags() == 0
    del got, alt
    torch.cuda.empty_cache()


def test_hand_over_beside_other_work_on_the_chip(gpu_ctx):
    """The hand-over under uneven load: another stream streams through 8 GB (and a third one spins arithmetic) while the
    launch runs, so that its workgroups are dispatched late and unevenly and every cache is busy with other lines --
    swept over one to five resident workgroups per CU, none to two of them sampling only, chunks of 16 and 64 steps, 30
    launches with the data changing every time: every word of the TOD still equals the two calls' of the same data."""
    import torch

    from maria_amd import _lib

    p = config_problem("atlast_10k", n_det=6000)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    alt = _Alternating(path, generated=True, keep_coarse=False)
    got = torch.empty_like(alt.want[0])
    big = torch.ones(1 << 30, dtype=torch.float32, device="cuda:0")  # 4 GB read + 4 GB written per pass
    small = torch.rand(1 << 22, dtype=torch.float32, device="cuda:0")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    try:
        for rep in range(30):
            per_cu, samplers, chunk = 1 + rep % 5, (rep // 5) % 3, (16, 64)[(rep // 15) % 2]
            v = (rep // 2 + rep) % 2
            alt.select(v)
            got.fill_(float("nan"))
            torch.cuda.synchronize()
            with torch.cuda.stream(s1):
                for _ in range(1 + rep % 3):
                    big.mul_(1.0000001)
            with torch.cuda.stream(s2):
                for _ in range(20):
                    small = torch.sin(small) * 1.0001 + 0.1
            gpu_ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, per_cu)
            path.synthesize(got, sampler_wgs_per_cu=8 if samplers == 0 else min(samplers, per_cu), chunk=chunk)
            alt.check(v, got, (rep, per_cu, samplers, chunk))
    finally:
        gpu_ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, 0)
    assert path.check_flags() == 0
    del big, got, alt
    torch.cuda.empty_cache()


def test_the_acquire_hand_over_is_a_runtime_switch_and_gives_the_same_bits(gpu_ctx):
    """MRX_OPT_SYNTH_ACQUIRE = 1 (round 6): the consumer's agent-scope acquire before a tile
    loads its chunks -- MI355X_MICROARCH.md's always-valid hand-over, one call away at run time instead of a rebuild.  Both
    forms on the same alternating data, five resident workgroups per CU, beside a stream that keeps the caches busy: each
    launch equals the two calls of ITS data in every word, so the two forms equal each other -- on every round's GPU run."""
    import torch

    from maria_amd import _lib

    p = config_problem("atlast_10k", n_det=4000)
    path = _path(p, gpu_ctx)
    path.generate_screens()
    alt = _Alternating(path, generated=True, keep_coarse=False)
    got = torch.empty_like(alt.want[0])
    big = torch.ones(1 << 28, dtype=torch.float32, device="cuda:0")
    side = torch.cuda.Stream()
    try:
        for rep in range(12):
            v = (rep // 2 + rep) % 2
            alt.select(v)
            got.fill_(float("nan"))
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                big.mul_(1.0000001)
            gpu_ctx.set_option(_lib.OPT_SYNTH_ACQUIRE, rep % 2)
            path.synthesize(got, chunk=(16, 32, 64)[rep % 3])
            alt.check(v, got, (rep, "acquire" if rep % 2 else "sc1 loads"))
    finally:
        gpu_ctx.set_option(_lib.OPT_SYNTH_ACQUIRE, 0)
    assert path.check_flags() == 0
    del big, got, alt
    torch.cuda.empty_cache()


def test_random_shapes(gpu_ctx):
    """64 random shapes through the one launch against the two calls, every word (the loop of scripts/fuzz_synth.py with
    a fixed seed): detectors, layers, bands, sample rate, duration and time step (i.e. Ta, T and their ratio), gain, block
    size, chunk length, dedicated samplers and resident workgroups per CU are drawn at random; four in ten run in K_RJ
    where the coarse form's bound allows; every shape is launched twice on alternating data."""
    import torch

    from maria_amd import _lib, synthetic
    from test_gpu_calibration import _cal_tables

    rng = np.random.default_rng(20261005)
    try:
        for trial in range(64):
            n_det = int(np.exp(rng.uniform(0, np.log(3000))))
            n_layers = int(rng.integers(1, 17))
            n_bands = int(rng.integers(1, 4))
            fs = float(rng.choice([20.0, 50.0, 100.0, 400.0]))
            timestep = float(rng.choice([0.1, 0.2, 0.5]))
            duration = float(rng.uniform(4 * timestep + 0.3, 60.0 if fs < 200 else 25.0))
            p = attach_numpy_screens(synthetic.make_problem(
                n_det=n_det, n_bands=min(n_bands, n_det), fov_deg=float(rng.uniform(0.05, 1.0)), fs=fs, duration=duration,
                n_layers=n_layers, side=int(rng.choice([64, 128, 256])), timestep=timestep, seed=int(rng.integers(1 << 30)),
                gain=bool(rng.random() < 0.5)))
            path = _path(p, gpu_ctx)
            path.clear_flags()
            krj = False
            if rng.random() < 0.4:
                _, el_full = synthetic.daisy_scan(p["t"])
                roll = rng.uniform(0, 2 * np.pi)
                R = np.array([[np.cos(roll), -np.sin(roll)], [np.sin(roll), np.cos(roll)]])
                nb = len(p["tables"])
                path.set_calibration(_cal_tables(nb), 273.15, 1.0, el_full, p["offsets"] @ R.T, [bool(rng.random() < 0.5) for _ in range(nb)])
                krj = bool(path.coarse_krj_bound() <= path.COARSE_KRJ_LIMIT)
            alt = _Alternating(path, krj=krj)
            flags0 = int(path.d_flags.item())
            block_rows = int(rng.choice([0, 256, 512, 768, 1024, 4096]))
            chunk = int(rng.choice([1, 4, 8, 16, 32, 64]))
            samplers = int(rng.integers(0, 9))
            gpu_ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, int(rng.integers(0, 6)))
            got = torch.empty_like(alt.want[0])
            for rep in range(2):
                alt.select(rep)
                got.fill_(float("nan"))
                path.synthesize(got, block_rows=block_rows, sampler_wgs_per_cu=samplers, chunk=chunk, krj=krj)
                alt.check(rep, got, (trial, rep, path.D, path.Ta, path.T, n_layers, block_rows, chunk, samplers, krj))
                assert int(path.d_flags.item()) == flags0
            path.clear_flags()
            del path, alt, got
    finally:
        gpu_ctx.set_option(_lib.OPT_SYNTH_WGS_PER_CU, 0)


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
