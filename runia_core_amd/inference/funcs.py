"""Numerical helpers of the logits/features postprocessors with the reference's
names (``runia_core/inference/funcs.py``: ``mahalanobis_preprocess`` :33-66,
``mahalanobis_postprocess`` :69-102, ``normalizer`` :105-115).
Fits (setup time) make the same scikit-learn call as the reference; scoring runs
on the GPU."""
from __future__ import annotations

import warnings
from typing import Dict, Tuple

import numpy as np
import torch

from .. import _hip, config
from ..device_fit import empirical_precision_device

__all__ = ["mahalanobis_preprocess", "mahalanobis_postprocess", "normalizer", "MahalanobisState"]


def mahalanobis_preprocess(ind_data: Dict[str, np.ndarray], num_classes: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per-class means and the pooled precision matrix of the class-centred training features.

    Returns ``(class_mean [C, D], precision [D, D])``; a class without samples warns and
    yields a NaN mean (scored as -inf later), like the reference."""
    from sklearn.covariance import EmpiricalCovariance

    feats, labels = ind_data["train features"], ind_data["train labels"]
    class_mean, centered = [], []
    for c in range(num_classes):
        class_samples = feats[labels == c]
        if len(class_samples) == 0:
            warnings.warn(f"No train examples for class {c}")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            class_mean.append(class_samples.mean(0))
        centered.append(class_samples - class_mean[c].reshape(1, -1))
    class_mean = np.stack(class_mean)
    pooled = np.concatenate(centered).astype(np.float32)
    if config.device_fit:
        return class_mean, empirical_precision_device(pooled)
    estimator = EmpiricalCovariance(assume_centered=False)
    estimator.fit(pooled)
    return class_mean, estimator.precision_


class MahalanobisState:
    """Device-resident fitted state: class means, packed precision and ``class_mean @ P``."""

    def __init__(self, class_mean: np.ndarray, precision: np.ndarray):
        self.num_classes, self.dim = class_mean.shape
        prec = np.asarray(precision, dtype=np.float64)
        self.packed_p = _hip.pack_weights(_hip.to_device(prec, torch.float64))
        with np.errstate(all="ignore"):
            mu_p = np.asarray(class_mean, dtype=np.float64) @ prec
        self.mu_p = _hip.to_device(mu_p, torch.float64)
        self._means = {}
        self._class_mean = class_mean

    def means(self, dtype: torch.dtype) -> torch.Tensor:
        if dtype not in self._means:
            self._means[dtype] = _hip.to_device(self._class_mean, dtype)
        return self._means[dtype]

    def score_device(self, x: torch.Tensor) -> torch.Tensor:
        return _hip.mahalanobis_score(x, self.means(x.dtype), self.packed_p, self.mu_p)


def _maha_dtype(feats, class_mean) -> torch.dtype:
    # NumPy: feats - class_mean[c] is f32 only when both are f32
    f32 = getattr(feats, "dtype", None) in (np.float32, torch.float32) and class_mean.dtype == np.float32
    return torch.float32 if f32 else torch.float64


def mahalanobis_postprocess(feats: np.ndarray, class_mean: np.ndarray, precision: np.ndarray, num_classes: int,
                            _state: MahalanobisState = None) -> np.ndarray:
    """Max over classes of ``-(x - mu_c) P (x - mu_c)^T`` for every row of ``feats`` -> ``(N,)`` f64."""
    state = _state if _state is not None else MahalanobisState(class_mean[:num_classes], precision)
    x = _hip.to_device(feats, _maha_dtype(feats, class_mean))
    return state.score_device(x).cpu().numpy()


def normalizer(x):
    """``x / (||x||_2 + 1e-10)`` along the last axis (f32 on the GPU for f32 input rows)."""
    arr = np.asarray(x)
    if arr.dtype != np.float32:
        # the reference only feeds the result to faiss (f32); wider inputs keep NumPy semantics
        return arr / (np.linalg.norm(arr, ord=2, axis=-1, keepdims=True) + 1e-10)
    flat = arr.reshape(-1, arr.shape[-1])
    out = _hip.l2_normalize(_hip.to_device(flat, torch.float32)).cpu().numpy()
    return out.reshape(arr.shape)
