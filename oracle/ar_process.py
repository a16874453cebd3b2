"""The reference's autoregressive turbulence generator.  TEST INFRASTRUCTURE ONLY.

Restates maria/atmosphere/process.py:19-209 with
numpy's global random state, so that ``np.random.seed(s)`` reproduces a
reference-style realisation.  The GPU path replaces this generator by a spectral
one with the same target covariance; this class provides the
statistical target the screens are compared with and reference-style screens for
end-to-end CPU goldens.  The module itself cannot be imported from the reference
(it needs dask): parity unpinned, except for its leaves (Matern callback and
``fast_psd_inverse``), which are pinned in tests/golden/leaves.json.
"""

from __future__ import annotations

import numpy as np

from .functions import approximate_normalized_matern
from .geometry import fast_psd_inverse

COV_MAT_JITTER = 1e-6


class AutoregressiveProcess:
    def __init__(
        self,
        cross_section,
        extrusion,
        callback=approximate_normalized_matern,
        callback_kwargs=None,
        lookback_decay_rate: float = 2,
        jitter: float = 1e-6,
        MIN_SAMPLES_PER_LAYER: int = 4,
    ):
        """process.py:20-109.  ``cross_section`` [C,2] = (x, height), ``extrusion`` [E]."""
        self.cross_section = np.asarray(cross_section, float)
        self.extrusion = np.asarray(extrusion, float)
        self.callback = callback
        self.callback_kwargs = callback_kwargs or {}
        self.jitter = jitter
        self.n_cross_section = len(self.cross_section)
        self.n_extrusion = len(self.extrusion)
        I, J = np.meshgrid(np.arange(self.n_cross_section), np.arange(self.n_extrusion))  # noqa: E741
        points = np.c_[self.extrusion[J][..., None], self.cross_section[I]]

        # lookback stencil: rows {0, 1, 2, 4, ..., E-1} ahead of the live edge (:44-48)
        extrusion_indices = [
            0,
            *(2 ** np.arange(0, np.log(self.n_extrusion) / np.log(2))).astype(int),
            self.n_extrusion - 1,
        ]
        e_idx, c_idx = [], []
        for i, extrusion_index in enumerate(extrusion_indices):
            n_ribbon = np.minimum(
                np.maximum(int(self.n_cross_section * 2.0 ** -(i)), MIN_SAMPLES_PER_LAYER), self.n_cross_section
            )
            cs = np.unique(np.linspace(0, self.n_cross_section - 1, n_ribbon).astype(int))
            c_idx.extend(cs)
            e_idx.extend(np.repeat(extrusion_index, len(cs)))
        self.cross_section_sample_index = np.array(c_idx)
        self.extrusion_sample_index = np.array(e_idx)
        self.sample_points = points[self.extrusion_sample_index, self.cross_section_sample_index]
        self.n_sample = len(self.sample_points)
        live = points[0].copy()
        live[:, 0] -= np.gradient(self.extrusion).mean()  # one row before row 0 (:91)
        self.live_edge_points = live
        self.n_live_edge = len(live)

    def _cov(self, a, b):
        return self.callback(np.sqrt(np.square(a - b).sum(axis=-1)), **self.callback_kwargs)

    def compute_covariance_matrices(self):
        """process.py:111-189."""
        i, j = np.triu_indices(self.n_live_edge, k=1)
        COV_E_E = np.eye(self.n_live_edge) + self.jitter
        COV_E_E[i, j] = self._cov(self.live_edge_points[j], self.live_edge_points[i])
        COV_E_E[j, i] = COV_E_E[i, j]
        COV_E_E += np.diag(COV_MAT_JITTER * np.diag(COV_E_E))
        COV_E_S = self._cov(self.sample_points[None], self.live_edge_points[:, None])
        i, j = np.triu_indices(self.n_sample, k=1)
        COV_S_S = np.eye(self.n_sample) + self.jitter
        COV_S_S[i, j] = self._cov(self.sample_points[j], self.sample_points[i])
        COV_S_S[j, i] = COV_S_S[i, j]
        COV_S_S += np.diag(COV_MAT_JITTER * np.diag(COV_S_S))
        inv = fast_psd_inverse(COV_S_S)
        self.A = COV_E_S @ inv
        if (self.A.sum(axis=-1) > 1.0).any():
            raise ValueError(f"Propagation operator is unstable (A_max = {self.A.sum(axis=-1).max()}).")
        self.B = np.linalg.cholesky(COV_E_E - self.A @ COV_E_S.T)
        self.values = np.zeros((self.n_extrusion, self.n_cross_section))
        initial_slice = self.B @ np.random.standard_normal(self.n_cross_section)
        initial_slice *= np.sqrt(np.diag(COV_E_E) / initial_slice.var())
        self.values[:] = initial_slice

    def run(self):
        """process.py:191-209: the sequential generator loop."""
        if not hasattr(self, "A"):
            self.compute_covariance_matrices()
        n_steps = 2 * self.n_extrusion
        BUFFER = np.random.standard_normal((self.n_extrusion + n_steps, self.n_cross_section))
        for k in np.arange(n_steps)[::-1]:
            BUFFER[k] = self.A @ BUFFER[
                k + self.extrusion_sample_index + 1, self.cross_section_sample_index
            ] + self.B @ np.random.standard_normal(size=self.n_live_edge)
        self.values = BUFFER[: self.n_extrusion]
        return self.values
