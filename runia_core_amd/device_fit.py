"""Setup-time fits on the device (SURVEY 8f "next #1"; opt-in).

``runia_core_amd.config.device_fit = True`` moves

* the covariance of ``MDLatentSpace.setup`` and ``mahalanobis_preprocess`` (``np.cov(X.T, bias=1)`` inside sklearn
  ``EmpiricalCovariance``, reference ``inference/postprocessors.py:217-220`` / ``inference/funcs.py:62-66``) to the f64
  matrix cores (``runia_covariance_*``),
* ``scipy.linalg.pinvh`` to the hand-written Jacobi eigen-solver (``runia_eigh_*``, ``csrc/eigh.hip``) with SciPy's
  cut-off rule and a device matrix product,
* the PCA fit of ``apply_pca_ds_split(..., svd_solver="covariance_eigh" | "full")`` to covariance + the same
  eigen-solver + sklearn's sign convention (``svd_flip(u_based_decision=False)``).  The reference's default
  ``svd_solver="randomized"`` stays the scikit-learn call: its result depends on draws from NumPy's global generator.

No vendor solver is involved.  The default (False) keeps the reference's own host calls, so fitted state is
bit-identical to the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _hip

__all__ = ["empirical_precision_device", "pinvh_device", "pca_fit_device", "FittedPCA"]


def pinvh_device(cov: torch.Tensor) -> torch.Tensor:
    """``scipy.linalg.pinvh(cov)``: eigen-decomposition, eigenvalues with ``|s| <= max|s| * max(M,N) * eps``
    dropped, ``(U / s) @ U^T``."""
    s, u = _hip.eigh(cov)
    cutoff = s.abs().max() * (max(cov.shape) * torch.finfo(cov.dtype).eps)
    keep = s.abs() > cutoff
    u = u[:, keep].contiguous()
    return _hip.matmul_f64((u * (1.0 / s[keep])).contiguous(), u, transpose_b=True)


def empirical_precision_device(x) -> np.ndarray:
    """``EmpiricalCovariance(assume_centered=False).fit(x).precision_`` computed on the GPU -> f64 ndarray."""
    dtype = torch.float32 if getattr(x, "dtype", None) in (np.float32, torch.float32) else torch.float64
    xd = _hip.to_device(x, dtype)
    _, cov = _hip.covariance(xd)
    return pinvh_device(cov).cpu().numpy()


class FittedPCA:
    """What ``apply_pca_ds_split`` returns for a device fit: the public attributes of a fitted sklearn ``PCA`` that
    the path (and user code) reads, plus ``transform``."""

    def __init__(self, components, mean, explained_variance, whiten, n_samples, total_var):
        self.components_ = components
        self.mean_ = mean
        self.explained_variance_ = explained_variance
        self.whiten = whiten
        self.n_components = self.n_components_ = components.shape[0]
        self.n_features_in_ = components.shape[1]
        self.n_samples_ = n_samples
        self.explained_variance_ratio_ = explained_variance / total_var
        self.singular_values_ = np.sqrt(explained_variance * (n_samples - 1))
        self.svd_solver = "covariance_eigh"

    def transform(self, x):
        from .dimensionality_reduction import apply_pca_transform

        return apply_pca_transform(x, self)


def pca_fit_device(samples, n_components: int, whiten: bool = True) -> FittedPCA:
    """sklearn ``PCA(n_components, svd_solver="covariance_eigh").fit`` on the GPU (sklearn ``_pca.py::_fit_full``):
    covariance with ``ddof = 1``, symmetric eigen-decomposition, eigenvalues descending and clamped at 0, components =
    eigenvectors with the sign that makes the largest-magnitude entry of every component positive."""
    x = np.asarray(samples)
    n, d = x.shape
    if not 0 < n_components <= min(n, d):
        raise ValueError(f"n_components={n_components} must be between 0 and min(n_samples, n_features)={min(n, d)}")
    xd = _hip.to_device(x, torch.float32 if x.dtype == np.float32 else torch.float64)
    mean, cov = _hip.covariance(xd)                  # bias = 1 (divided by n)
    cov = cov * (n / (n - 1.0))
    w, v = _hip.eigh(cov)
    w = torch.flip(w, dims=(0,)).clamp_min(0.0)
    vt = torch.flip(v, dims=(1,)).T.contiguous()      # rows = components, descending eigenvalue
    idx = vt.abs().argmax(dim=1, keepdim=True)
    signs = torch.sign(torch.gather(vt, 1, idx))
    signs[signs == 0] = 1.0
    vt = vt * signs
    total = float(w.sum().item())
    return FittedPCA(vt[:n_components].cpu().numpy(), mean.cpu().numpy(), w[:n_components].cpu().numpy(), whiten, n, total)
