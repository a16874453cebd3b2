"""CPU tier: the host side of the covariance-matched screen amplitudes (mrx_screen_amplitudes)."""

import numpy as np

from oracle import functions, screens


def _device_formula(log_first, log_step, log_cov, log_sf, x):
    """radial_correlation of csrc/mrx_screen.hip in numpy: four-point Lagrange in log x, the reference's blend."""
    n = len(log_cov)
    xe = np.maximum(np.abs(x), np.exp(log_first))
    u = (np.log(xe) - log_first) / log_step
    i = np.clip(u.astype(int), 1, n - 3)
    w = u - i
    a, b, c, d = w + 1, w, w - 1, w - 2

    def lag(f):
        return f[i - 1] * (-b * c * d / 6) + f[i] * (a * c * d / 2) + f[i + 1] * (-a * b * d / 2) + f[i + 2] * (a * b * c / 6)

    t = 1 / (1 + xe**2)
    return np.where(x == 0, 1.0, t * (1 - np.exp(lag(log_sf))) + (1 - t) * np.exp(lag(log_cov)))


def test_log_tables_reproduce_the_matern_correlation():
    """maria_amd.pipeline.matern_log_tables feeds the device's radial correlation; the device formula over
    those tables is the exact Matern correlation (functions/__init__.py:30-39, pinned by
    tests/golden/leaves.json) to 1e-11 out to the image cut-off, and within the 1e-5 of the
    reference's own 1024-node approximation (:42-74) of it."""
    from maria_amd.pipeline import matern_log_tables

    for nu in (5 / 6, 1 / 3):
        log_first, log_step, log_cov, log_sf, x_cut = matern_log_tables(nu)
        assert 10 < x_cut < 40
        x = np.concatenate([[0.0], np.geomspace(2e-6, x_cut, 4000)])
        got = _device_formula(log_first, log_step, log_cov, log_sf, x)
        want = np.where(x == 0, 1.0, functions.normalized_matern(np.maximum(x, 1e-300), nu))
        assert np.abs(got - want).max() < 1e-11 and np.abs(got / want - 1).max() < 1e-8
        grid = x >= 1e-4  # a pixel is never a smaller part of the outer scale; below, 1 - rho drowns in rounding
        assert np.abs((1 - got[grid]) / (1 - want[grid]) - 1).max() < 1e-9
        approx = functions.approximate_normalized_matern(x * 700.0, nu=nu, r0=700.0)
        assert np.abs(got[1:] - approx[1:]).max() < 2e-5


def test_covariance_amplitudes_define_a_matern_field():
    """The inverse transform of the squared amplitudes is the image-summed covariance: positive definite
    (nothing clipped) even when the box is a few outer scales wide, and its structure function is Matern's
    from one pixel up -- while the reference's 1e-5 approximation of the correlation, put through the same
    construction, has negative eigenvalues whose clipping inflates the one-pixel structure by half."""
    from itertools import product

    for shape, steps, nu, r0 in (((256, 256), (5.0, 5.0), 5 / 6, 250.0), ((16, 32, 64), (40.0, 5.0, 6.0), 1 / 3, 100.0)):
        rho = screens.periodic_covariance(shape, steps, r0, nu)
        lam = np.fft.fftn(rho).real
        assert lam.min() > 0
        amp, rho0 = screens.covariance_amplitude(shape, steps, r0, nu)
        cov = np.fft.ifftn(amp**2).real
        assert np.abs(cov - rho).max() < 1e-10 and rho0 == rho.reshape(-1)[0] and rho0 >= 1
        lag = np.array([1, 2, 4])
        sf = rho0 - cov[(0,) * (len(shape) - 1) + (lag,)]
        # (the images' curvature: second order in lag / period -- these boxes are only 1.6 ... 5 outer scales wide)
        assert np.abs(sf / (1 - functions.normalized_matern(lag * steps[-1] / r0, nu)) - 1).max() < 1.5e-2
    # the approximation, wrapped or image-summed, is not positive definite on a grid of 1/200 of the outer scale
    shape, steps, nu, r0 = (512, 512), (5.0, 5.0), 5 / 6, 1000.0
    half = [np.arange(n // 2 + 1) * d for n, d in zip(shape, steps)]
    acc = np.zeros((257, 257))
    for ky, kx in product(range(-8, 9), repeat=2):
        r = np.sqrt((half[0][:, None] + ky * 2560.0) ** 2 + (half[1][None, :] + kx * 2560.0) ** 2)
        acc += functions.approximate_normalized_matern(r, nu=nu, r0=r0)
    fold = [np.minimum(np.arange(n), n - np.arange(n)) for n in shape]
    lam = np.fft.fftn(acc[np.ix_(*fold)]).real
    assert lam.min() < 0
    cov = np.fft.ifftn(np.maximum(lam, 0)).real
    assert (cov[0, 0] - cov[0, 1]) / (1 - functions.normalized_matern(5.0 / r0, nu)) > 1.4
