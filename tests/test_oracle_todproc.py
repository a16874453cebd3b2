"""CPU tier for the TOD pre-processing row: the restatement (and the product's host pieces)
against outputs of the reference's own utils/signal functions (tests/golden/leaves.json)."""

import json
import os

import numpy as np

from maria_amd import tod_processing as tp
from oracle import todproc

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "leaves.json")))["signal"]
T_AXIS, DATA = np.array(GOLD["t"]), np.array(GOLD["D"])


def test_bspline_basis_matches_reference():
    ref = np.array(GOLD["bspline_basis_spacing_3_order_3"])
    np.testing.assert_allclose(tp.bspline_basis(T_AXIS, spacing=3.0, order=3), ref, rtol=0, atol=1e-13)
    got = todproc.bspline_basis(T_AXIS, spacing=3.0, order=3)  # scipy's evaluation: the last sample sits on a knot
    np.testing.assert_allclose(got[:, :-1], ref[:, :-1], rtol=0, atol=1e-12)
    assert np.allclose(ref.sum(axis=0), 1.0)  # a partition of unity over the data


def test_slope_filters_and_modes_match_reference():
    np.testing.assert_allclose(todproc.remove_slope(DATA), np.array(GOLD["remove_slope"]), rtol=0, atol=1e-12)
    np.testing.assert_allclose(todproc.bessel(DATA, 1.5, 20.0, 1, "low"), np.array(GOLD["lowpass_fc1.5_order1"]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(todproc.bessel(DATA, 0.2, 20.0, 1, "high"), np.array(GOLD["highpass_fc0.2_order1"]), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(todproc.bessel(DATA, 2.5, 20.0, 2, "low"), np.array(GOLD["lowpass_fc2.5_order2"]), rtol=1e-12, atol=1e-12)
    removed = DATA - todproc.remove_modes(DATA, 2, k=3)
    np.testing.assert_allclose(removed, np.array(GOLD["decompose_k3_first2_modes"]), rtol=0, atol=1e-8 * np.abs(DATA).max())


def test_chunk_matrix_chains_the_recursion():
    """The state transition the device chains its chunks with: filtering in two halves with the
    state carried through M and the zero-state end state equals filtering in one go."""
    import scipy.signal

    rng = np.random.default_rng(0)
    sos = np.concatenate([tp.bessel_sos(1.5, 20.0, 1, "low"), tp.bessel_sos(0.2, 20.0, 1, "high")])
    x = rng.normal(size=300)
    y_full, zf = scipy.signal.sosfilt(sos, x, zi=np.zeros((len(sos), 2)))
    L = 100
    M = tp.chunk_matrix(sos, L)
    state = np.zeros(2 * len(sos))
    for c in range(3):
        _, z_zero = scipy.signal.sosfilt(sos, x[c * L : (c + 1) * L], zi=np.zeros((len(sos), 2)))
        state = M @ state + z_zero.ravel()
    np.testing.assert_allclose(state, zf.ravel(), rtol=1e-10, atol=1e-12)


def test_config_forms():
    cfg = tp.process_operation_kwargs(f_lower=0.1, window="tukey", window_kwargs={"alpha": 0.1}, modes_to_remove=2)
    assert cfg == {"window": {"name": "tukey", "kwargs": {"alpha": 0.1}}, "filter": {"f_lower": 0.1}, "remove_modes": {"modes_to_remove": 2}}
    assert tp.validate_process_config({"filter": {"f_lower": "0.5", "order": 2.0}}) == {"filter": {"f_lower": 0.5, "order": 2}}
    import pytest

    with pytest.raises(ValueError):
        tp.process_operation_kwargs(nonsense=1)
    with pytest.raises(ValueError):
        tp.validate_process_config({"despike": {}})
    with pytest.raises(ValueError):
        tp.validate_process_config({"filter": {"cutoff": 1.0}})
