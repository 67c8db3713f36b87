"""PCA helpers with the reference's signatures
(``runia_core/dimensionality_reduction.py``: ``apply_pca_ds`` :26-49,
``apply_pca_ds_split`` :52-72, ``apply_pca_transform`` :75-87).

Fitting stays the scikit-learn call the reference makes (setup time);
the per-row transform is the f64-MFMA kernel ``runia_pca_transform_*``.  A fitted
sklearn ``PCA`` is read through its public attributes, so PCA objects created by
user code drop in unchanged.
"""
from __future__ import annotations

import weakref
from typing import Tuple

import numpy as np
import torch

from . import _hip

__all__ = ["apply_pca_ds", "apply_pca_ds_split", "apply_pca_transform", "DevicePCA"]


class DevicePCA:
    """Device-resident copy of a fitted PCA: packed ``components_.T``, ``mean_ @ components_.T``
    and the whitening scale ``max(sqrt(explained_variance_), eps)`` (sklearn ``_BasePCA.transform``)."""

    def __init__(self, components: np.ndarray, mean, explained_variance: np.ndarray, whiten: bool):
        components = np.asarray(components, dtype=np.float64)
        self.n_components, self.n_features = components.shape
        self.whiten = bool(whiten)
        comp_dev = _hip.to_device(np.ascontiguousarray(components), torch.float64)
        # mean_ @ components_.T through the library's own f64 product, not host BLAS: a host gemv adds in an order that
        # follows the alignment of its operands, so two ranks holding the same fitted arrays could derive different last
        # bits (seen on the GPU box: tests/test_distributed_gpu.py); the kernel's order is fixed
        if mean is None:
            self.bias = torch.zeros(self.n_components, dtype=torch.float64, device=comp_dev.device)
        else:
            mean_dev = _hip.to_device(np.asarray(mean, dtype=np.float64).reshape(1, -1), torch.float64)
            self.bias = _hip.matmul_f64(mean_dev, comp_dev, transpose_b=True).reshape(-1).contiguous()
        self.components_host, self.bias_host, self.scale_host = components, self.bias.cpu().numpy(), None
        self.packed_ct = _hip.pack_weights(comp_dev.t().contiguous())
        self.scale = None
        if self.whiten:
            scale = np.sqrt(np.asarray(explained_variance, dtype=np.float64))
            min_scale = np.finfo(scale.dtype).eps
            scale = np.where(scale < min_scale, min_scale, scale)
            self.scale_host = scale
            self.scale = _hip.to_device(scale, torch.float64)

    @classmethod
    def from_sklearn(cls, pca) -> "DevicePCA":
        return cls(pca.components_, getattr(pca, "mean_", None), pca.explained_variance_, getattr(pca, "whiten", False))

    def transform_device(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[1] != self.n_features:
            raise ValueError(
                f"X has {x.shape[1]} features, but PCA is expecting {self.n_features} features as input."
            )
        return _hip.pca_transform(x, self.packed_ct, self.bias, self.scale, self.n_components)


_device_pca_cache: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _pca_fingerprint(pca) -> tuple:
    """Cheap identity of the fitted state a DevicePCA was built from: a PCA object that is refitted (or whose attributes
    are reassigned) gets new arrays, so the cached device copy is rebuilt instead of silently going stale."""
    def ident(a):
        a = np.asarray(a) if a is not None else None
        return None if a is None else (a.__array_interface__["data"][0], a.shape, a.dtype.str)

    return (ident(pca.components_), ident(getattr(pca, "mean_", None)), ident(pca.explained_variance_),
            bool(getattr(pca, "whiten", False)))


def device_pca_for(pca_transform) -> DevicePCA:
    if isinstance(pca_transform, DevicePCA):
        return pca_transform
    fp = _pca_fingerprint(pca_transform)
    try:
        cached = _device_pca_cache.get(pca_transform)
    except TypeError:
        cached = None
    if cached is None or cached[0] != fp:
        cached = (fp, DevicePCA.from_sklearn(pca_transform))
        try:
            _device_pca_cache[pca_transform] = cached
        except TypeError:
            pass
    return cached[1]


def apply_pca_ds(train_samples: np.ndarray, test_samples: np.ndarray, nro_components: int = 16,
                 svd_solver: str = "randomized", whiten: bool = True):
    """Fit on ``train_samples``; return (train reduced, test reduced, fitted PCA)."""
    from sklearn.decomposition import PCA

    from .host_threads import host_compute

    pca_dim_red = PCA(n_components=nro_components, svd_solver=svd_solver, whiten=whiten)
    with host_compute():  # (BLAS pools capped at the container's CPU quota)
        train_ds = pca_dim_red.fit_transform(train_samples)
    test_ds = apply_pca_transform(test_samples, pca_dim_red)
    return train_ds, test_ds, pca_dim_red


def apply_pca_ds_split(samples: np.ndarray, nro_components: int = 16, svd_solver: str = "randomized",
                       whiten: bool = True) -> Tuple[np.ndarray, "PCA"]:  # noqa: F821
    """Fit a PCA on one dataset split; return (reduced samples, fitted sklearn PCA).
    With ``runia_core_amd.config.device_fit`` the fit runs on the GPU - exact solvers through ``device_fit.pca_fit_device``,
    the default ``"randomized"`` through ``device_fit.pca_fit_randomized_device`` (same draws from NumPy's global generator
    as sklearn, same result) - and a ``FittedPCA`` with the same public attributes is returned."""
    from . import config

    if config.use_device_fit() and isinstance(nro_components, int):
        from .device_fit import pca_fit_device, pca_fit_randomized_device

        on_dev = isinstance(samples, torch.Tensor) and samples.is_cuda  # additive: rows already in HBM -> reduced rows stay there
        shape = tuple(samples.shape) if on_dev else np.shape(samples)
        fitted = None
        if svd_solver in ("covariance_eigh", "full"):
            fitted = pca_fit_device(samples, nro_components, whiten)
        elif svd_solver == "randomized" and len(shape) == 2 and shape[0] >= shape[1]:
            fitted = pca_fit_randomized_device(samples, nro_components, whiten)  # consumes np.random like sklearn does
        if fitted is not None:
            if getattr(fitted, "_train_projection", None) is not None:
                # randomized solver: sklearn's fit_transform is U * sqrt(n - 1) (or U * S), see pca_fit_randomized_device
                proj, var = fitted._train_projection
                dp = DevicePCA(proj, fitted.mean_, var, True)
                return (dp.transform_device(samples) if on_dev else apply_pca_transform(samples, dp)), fitted
            return (device_pca_for(fitted).transform_device(samples) if on_dev else apply_pca_transform(samples, fitted)), fitted
    if isinstance(samples, torch.Tensor):
        samples = samples.detach().cpu().numpy()
    from sklearn.decomposition import PCA

    from .host_threads import host_compute

    pca_dim_red = PCA(n_components=nro_components, svd_solver=svd_solver, whiten=whiten)
    with host_compute():  # (BLAS pools capped at the container's CPU quota)
        dataset_dim_red = pca_dim_red.fit_transform(samples)
    return dataset_dim_red, pca_dim_red


def apply_pca_transform(samples: np.ndarray, pca_transform) -> np.ndarray:
    """Project new samples with an already fitted PCA (sklearn ``PCA`` or :class:`DevicePCA`) on the GPU."""
    dp = device_pca_for(pca_transform)
    if isinstance(samples, torch.Tensor):
        x = samples
    else:
        x = np.asarray(samples)
        if x.ndim != 2:
            raise ValueError(f"Expected 2D array, got {x.ndim}D array instead")
    dtype = torch.float32 if (getattr(x, "dtype", None) in (np.float32, torch.float32)) else torch.float64
    xd = _hip.to_device(x, dtype)
    return _hip.to_host(dp.transform_device(xd))
