"""CPU tier: the host side of the noise generator (maria_amd/noise.py) without a device -- which rows of which band
each mrx_noise_generate call is asked for, for whole instruments and for detector shards that begin or end inside
the pairs of detectors that share one complex transform (sim/noise.py:18-63 draws band by band; the pairing is this
library's, csrc/mrx_noise.hip)."""

import ctypes as C

import numpy as np
import pytest
import torch


class _RecordingLib:
    def mrx_noise_work_floats(self, T, n_modes, batch, out):
        out._obj.value = 64
        return 0


class _RecordingContext:
    """Stands in for maria_amd._lib.Context: every generate call writes ``1000 * band + band_row`` into the rows it was
    given (+ 0.5 on rows that were drawn as the lone first detector of a pair whose partner exists: the rounding a
    shard must not see), through the raw pointers, as the library would."""

    def __init__(self, band_sizes):
        self.lib, self.calls, self.band_sizes = _RecordingLib(), [], band_sizes

    def call(self, name, seed, n_rows, row0, T, *rest):
        assert name == "mrx_noise_generate"
        d_out, ld_out = rest[-5], rest[-4]
        band = (seed - 100) // 7919
        self.calls.append((band, row0, n_rows))
        lone = (row0 + n_rows) % 2 == 1 and row0 + n_rows < self.band_sizes[band]
        for r in range(n_rows):
            row = (C.c_float * T).from_address(d_out.value + 4 * ld_out * r)
            tag = 0.5 if (row0 % 2 or (lone and r == n_rows - 1)) else 0.0
            row[:] = [1000.0 * band + row0 + r + tag] * T


def _dets(sizes):
    from maria_amd.instrument import Band, Detectors

    bands = [Band(center=90e9 + 30e9 * b, width=20e9, name=f"b{b}", NEP=1e-17, knee=1.0) for b in range(len(sizes))]
    rng = np.random.default_rng(0)
    return Detectors(rng.normal(0, 1e-3, (sum(sizes), 2)), bands, np.repeat(np.arange(len(sizes)), sizes), primary_size=10.0)


@pytest.mark.parametrize("sizes", [(8,), (7,), (5, 6), (61, 61), (1, 2, 3), (33, 1, 18)])
def test_every_shard_draws_whole_pairs_and_covers_its_rows(sizes):
    from maria_amd.noise import simulate_noise

    dets = _dets(sizes)
    n, T = dets.n, 5
    want = np.concatenate([1000.0 * b + np.arange(s) for b, s in enumerate(sizes)])
    starts = np.concatenate([[0], np.cumsum(sizes)])
    slices = [(lo, hi) for lo in range(n) for hi in range(lo + 1, n + 1)]
    if len(slices) > 400:  # (all of them for the small instruments, a sample of the larger)
        pick = np.random.default_rng(1).choice(len(slices), 400, replace=False)
        slices = [slices[i] for i in pick]
    for lo, hi in slices:
        if True:
            ctx = _RecordingContext(sizes)
            out = simulate_noise(ctx, dets, T, 50.0, 100, device="cpu", det_slice=slice(lo, hi))
            # every row of the shard carries its own (band, row) tag and was never drawn out of its pair
            np.testing.assert_array_equal(out.numpy(), np.repeat(want[lo:hi, None], T, axis=1), err_msg=f"{sizes} [{lo}, {hi})")
            for band, row0, n_rows in ctx.calls:
                assert row0 % 2 == 0 and row0 + n_rows <= sizes[band]
                assert (row0 + n_rows) % 2 == 0 or row0 + n_rows == sizes[band]
                # nothing is drawn beyond the pairs the shard touches
                b_lo, b_hi = max(lo, starts[band]) - starts[band], min(hi, starts[band + 1]) - starts[band]
                assert row0 >= b_lo - b_lo % 2 and row0 + n_rows <= min(b_hi + b_hi % 2, sizes[band])


@pytest.mark.parametrize("sizes", [(4, 4), (5, 3, 6), (9, 2)])
def test_bands_whose_rows_are_interleaved(sizes):
    """The reference masks rows by band name (sim/noise.py:32): a band's rows need not be neighbours.  Rows shuffled
    across bands, every shard: each row still carries (its band, its index within the band), drawn in whole pairs."""
    from maria_amd.instrument import Band, Detectors
    from maria_amd.noise import simulate_noise

    rng = np.random.default_rng(3)
    band_index = rng.permutation(np.repeat(np.arange(len(sizes)), sizes))
    bands = [Band(center=90e9 + 30e9 * b, width=20e9, name=f"b{b}", NEP=1e-17, knee=1.0) for b in range(len(sizes))]
    dets = Detectors(rng.normal(0, 1e-3, (sum(sizes), 2)), bands, band_index, primary_size=10.0)
    within = np.zeros(dets.n, int)
    for b in range(len(sizes)):
        within[band_index == b] = np.arange(sizes[b])
    want = 1000.0 * band_index + within
    n, T = dets.n, 3
    for lo in range(n):
        for hi in range(lo + 1, n + 1):
            ctx = _RecordingContext(sizes)
            out = simulate_noise(ctx, dets, T, 50.0, 100, device="cpu", det_slice=slice(lo, hi))
            np.testing.assert_array_equal(out.numpy(), np.repeat(want[lo:hi, None], T, axis=1), err_msg=f"{sizes} [{lo}, {hi})")
            for band, row0, n_rows in ctx.calls:
                assert row0 % 2 == 0 and row0 + n_rows <= sizes[band]
                assert (row0 + n_rows) % 2 == 0 or row0 + n_rows == sizes[band]


def test_the_unsharded_call_is_one_draw_per_band():
    from maria_amd.noise import simulate_noise

    sizes = (61, 61)
    ctx = _RecordingContext(sizes)
    simulate_noise(ctx, _dets(sizes), 4, 50.0, 100, device="cpu")
    assert ctx.calls == [(0, 0, 61), (1, 0, 61)]


def test_a_band_that_grows_with_the_loading_needs_it():
    from maria_amd.instrument import Band, Detectors
    from maria_amd.noise import simulate_noise

    bands = [Band(center=90e9, width=20e9, name="b0", NEP=1e-17, knee=1.0, NEP_per_loading=0.1)]
    dets = Detectors(np.zeros((4, 2)), bands, np.zeros(4, int), primary_size=10.0)
    with pytest.raises(ValueError, match="NEP_per_loading"):
        simulate_noise(_RecordingContext((4,)), dets, 4, 50.0, 100, device="cpu")
    assert torch.zeros(1).device.type == "cpu"
