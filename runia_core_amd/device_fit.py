"""Setup-time fits on the device (SURVEY 8f "next #1"; opt-in).

``runia_core_amd.config.device_fit = True`` moves the covariance of ``MDLatentSpace.setup`` and
``mahalanobis_preprocess`` (``np.cov(X.T, bias=1)`` inside sklearn ``EmpiricalCovariance``, reference
``inference/postprocessors.py:217-220`` / ``inference/funcs.py:62-66``) to the f64 matrix cores
(``runia_covariance_*``) and ``scipy.linalg.pinvh`` to ``torch.linalg.eigh`` on the GPU with SciPy's cut-off rule.
The default (False) keeps the reference's own host calls, so fitted state is bit-identical to the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _hip

__all__ = ["empirical_precision_device", "pinvh_device"]


def pinvh_device(cov: torch.Tensor) -> torch.Tensor:
    """``scipy.linalg.pinvh(cov)``: eigen-decomposition, eigenvalues with ``|s| <= max|s| * max(M,N) * eps``
    dropped, ``(U / s) @ U^T``."""
    s, u = torch.linalg.eigh(cov)
    cutoff = s.abs().max() * (max(cov.shape) * torch.finfo(cov.dtype).eps)
    keep = s.abs() > cutoff
    u = u[:, keep]
    return (u * (1.0 / s[keep])) @ u.T


def empirical_precision_device(x) -> np.ndarray:
    """``EmpiricalCovariance(assume_centered=False).fit(x).precision_`` computed on the GPU -> f64 ndarray."""
    dtype = torch.float32 if getattr(x, "dtype", None) in (np.float32, torch.float32) else torch.float64
    xd = _hip.to_device(x, dtype)
    _, cov = _hip.covariance(xd)
    return pinvh_device(cov).cpu().numpy()
