"""CPU tier for the noise row: the numpy restatement of noise/generation.py and
utils/linalg.py behaves as the reference documents, and the product's host-side basis
equals it (the device synthesis itself is tested in test_gpu_noise.py)."""

import numpy as np
import scipy.signal

from oracle import noise as onoise


def test_oracle_spectrum_white_plus_pink():
    fs, knee = 100.0, 2.0
    x = onoise.generate_noise_with_knee((64, 1 << 15), sample_rate=fs, knee=knee, rng=np.random.default_rng(0))
    f, p = scipy.signal.welch(x, fs=fs, nperseg=4096, detrend=False, axis=-1)
    p = p.mean(axis=0)
    for lo, hi in [(0.1, 0.5), (0.5, 2.0), (2.0, 10.0), (10.0, 45.0)]:
        m = (f >= lo) & (f < hi)
        assert abs(p[m].mean() / onoise.one_sided_psd_model(f[m], fs, knee).mean() - 1) < 0.1
    # knee = 0: white only, variance = sample rate (generation.py:25)
    w = onoise.generate_noise_with_knee((8, 1 << 15), sample_rate=fs, knee=0.0, rng=np.random.default_rng(1))
    assert abs(w.var() / fs - 1) < 0.02


def test_spatial_basis_reproduces_the_matern_kernel():
    """B B^T approximates the Matern-5/2 covariance between detectors (utils/linalg.py:105-126);
    the product's host function is the same computation."""
    from maria_amd import noise as mnoise
    from maria_amd import synthetic

    off = synthetic.hex_pack(217, np.radians(1.0))
    scale = mnoise.diameter(off)
    assert abs(scale - np.sqrt(((off[:, None] - off[None]) ** 2).sum(-1)).max()) < 1e-12
    B = onoise.generate_spatial_basis(off, k=5, n_side=16, scale=scale)
    assert B.shape == (217, 5) and B[:, 0].mean() > 0
    np.testing.assert_allclose(mnoise.spatial_basis(off, k=5, n_side=16, scale=scale), B, atol=1e-12)
    r = np.sqrt(((off[:, None] - off[None]) ** 2).sum(-1)) / scale
    # 5 of 256 modes carry most of a kernel as wide as the focal plane
    assert np.abs(B @ B.T - onoise.matern_five_halves(r)).max() < 0.12


def test_correlated_oracle_covariance():
    """sqrt(c) B modes + sqrt(1-c) pink: low-frequency covariance c B B^T + (1-c) I."""
    rng = np.random.default_rng(5)
    D, T, fs, knee, c = 40, 1 << 14, 50.0, 25.0, 0.6
    B = rng.normal(0, 1, (D, 3)) / np.sqrt(3)
    x = onoise.generate_noise_with_knee((D, T), fs, knee, basis=B, corr_prop=c, rng=rng)
    X = np.fft.rfft(x, axis=1)
    f = np.fft.rfftfreq(T, 1 / fs)
    m = (f > 0.05) & (f < 1.0)
    Xw = X[:, m] / np.sqrt(1 + knee / f[m])
    cov = (Xw @ Xw.conj().T).real / m.sum()
    model = c * B @ B.T + np.mean(((1 - c) * knee / f[m] + 1) / (1 + knee / f[m])) * np.eye(D)
    cov *= np.trace(model) / np.trace(cov)
    assert np.corrcoef(cov.ravel(), model.ravel())[0, 1] > 0.97


def test_sky_transform_stack_known_directions():
    """The host-side horizon -> equatorial rotation (astropy's role in
    coords/coordinates.py:184-236): orthonormal, zenith at (LST, latitude), the north
    point at elevation = latitude on the pole, sidereal rate."""
    from maria_amd.sim import sky_transform_stack
    from oracle import mapsample

    t = 1.7e9 + np.array([0.0, 3600.0, 86164.0905])
    lat, lon = 35.0, -110.0
    M = sky_transform_stack(t, lat, lon)
    np.testing.assert_allclose(M @ np.swapaxes(M, 1, 2), np.broadcast_to(np.eye(3), M.shape), atol=1e-14)
    zen = np.array([0.0, 0.0, 1.0]) @ M  # [3 times, 3]
    np.testing.assert_allclose(np.degrees(np.arcsin(zen[:, 2])), lat, atol=1e-10)
    ra = np.degrees(np.arctan2(zen[:, 1], zen[:, 0])) % 360
    assert abs(((ra[1] - ra[0]) % 360) - 15.0410686) < 1e-4  # one hour of sidereal rotation
    assert abs(((ra[2] - ra[0] + 180) % 360) - 180) < 1e-4     # one sidereal day: back again
    pole = mapsample.phi_theta_to_xyz(0.0, np.radians(lat)).astype(float) @ M[0]
    assert abs(pole[2] - 1) < 1e-7
    # east point on the horizon: declination 0, six hours east of the meridian
    east = mapsample.phi_theta_to_xyz(np.pi / 2, 0.0).astype(float) @ M[0]
    assert abs(east[2]) < 1e-7 and abs(((np.degrees(np.arctan2(east[1], east[0])) - ra[0]) % 360) - 90) < 1e-4


def test_offsets_round_trip_like_the_reference_test():
    """maria/tests/coordinates/test_coordinates.py:7-19 on the restated float32 transforms:
    offsets -> (phi, theta) -> offsets for random centres, mean square error < 1e-5 (here the
    round trip is good to float32 rounding)."""
    from oracle import hotpath, mapsample

    rng = np.random.default_rng(0)
    for cphi in rng.uniform(0, 2 * np.pi, 5):
        for ctheta in rng.uniform(-np.pi / 2, np.pi / 2, 5):
            offsets = np.radians(rng.uniform(-0.5, 0.5, (256, 2)))
            phi, theta = hotpath.offsets_to_phi_theta(offsets[:, 0], offsets[:, 1], cphi, ctheta)
            back = mapsample.phi_theta_to_offsets(phi, theta, cphi, ctheta)
            assert np.mean(np.square(offsets - back)) < 1e-5
            assert np.abs(offsets - back).max() < 5e-6
