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

__all__ = ["mahalanobis_preprocess", "mahalanobis_postprocess", "normalizer", "MahalanobisState", "gmm_fit", "GmmState"]


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
        prec = _hip.to_device(np.asarray(precision, dtype=np.float64), torch.float64)
        self.packed_p = _hip.pack_weights(prec)
        # class_mean @ P with the library's f64 product (fixed summation order: every rank of a sharded job derives the
        # same bits, which a host BLAS call does not promise); an empty class's NaN mean stays a NaN row
        self.mu_p = _hip.matmul_f64(_hip.to_device(np.asarray(class_mean, dtype=np.float64), torch.float64), prec)
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
    return _hip.to_host(state.score_device(x))


def normalizer(x):
    """``x / (||x||_2 + 1e-10)`` along the last axis (f32 on the GPU for f32 input rows)."""
    arr = np.asarray(x)
    if arr.dtype != np.float32:
        # the reference only feeds the result to faiss (f32); wider inputs keep NumPy semantics
        return arr / (np.linalg.norm(arr, ord=2, axis=-1, keepdims=True) + 1e-10)
    flat = arr.reshape(-1, arr.shape[-1])
    out = _hip.to_host(_hip.l2_normalize(_hip.to_device(flat, torch.float32)))
    return out.reshape(arr.shape)


_GMM_JITTERS = (0.0,) + tuple(10.0 ** e for e in range(-20, 0))


def gmm_fit(embeddings: torch.Tensor, labels: torch.Tensor, num_classes: int):
    """Class-wise Gaussians behind ``GMMLatentSpace`` / ``DDU`` (what the reference's ``gmm_fit`` returns,
    ``inference/funcs.py:265-344``): per-class mean, per-class covariance of the centred rows divided by
    ``max(n_c, 2) - 1``, classes without samples left out, and the smallest jitter of ``0, 1e-20, ..., 1e-1`` whose
    ``cov + jitter * I`` has a float32 Cholesky factor.

    Setup-time host fit in float32 torch like the reference, restructured: the class means come from one one-hot
    contraction over all classes at once, the class covariances from one ``x_c^T x_c`` per class over the rows grouped by
    label (memory O(N D + C D^2), N D^2 multiply-adds in all - a single ``einsum`` over (rows, classes) would go through an
    (N, C, D) intermediate: 10 GB at CIFAR-100 size), and the jitter ladder is walked with ``torch.linalg.cholesky_ex``
    (status codes instead of exceptions); the factor found is handed to ``MultivariateNormal(scale_tril=...)`` - the same
    factor the reference's ``covariance_matrix=`` construction computes internally.

    Returns ``(MultivariateNormal, jitter)``."""
    with torch.no_grad():
        x = embeddings.to(torch.float32)
        lab = labels.to(torch.long).reshape(-1)
        member = torch.nn.functional.one_hot(lab.clamp(0, num_classes - 1), num_classes).to(x.dtype)
        member = member * ((lab >= 0) & (lab < num_classes)).to(x.dtype).unsqueeze(1)      # (N, C)
        counts = member.sum(dim=0)                                                           # (C,)
        present = counts > 0
        means = (member.t() @ x) / counts.clamp_min(1.0).unsqueeze(1)                        # (C, D)
        valid = (lab >= 0) & (lab < num_classes)
        order = torch.argsort(torch.where(valid, lab, torch.full_like(lab, num_classes)), stable=True)  # rows grouped by class
        starts = torch.cumsum(counts, 0).to(torch.long) - counts.to(torch.long)
        d = x.shape[1]
        covs = torch.zeros((num_classes, d, d), dtype=x.dtype)
        for c in range(num_classes):
            n_c = int(counts[c])
            if n_c:
                rows = x[order[int(starts[c]): int(starts[c]) + n_c]] - means[c]             # the class's rows, centred
                covs[c] = (rows.t() @ rows) / (max(n_c, 2) - 1)
        means, covs = means[present], covs[present]
        eye = torch.eye(covs.shape[-1], dtype=covs.dtype, device=covs.device)
        chosen, factor = _GMM_JITTERS[-1], None
        for jitter in _GMM_JITTERS:
            tril, info = torch.linalg.cholesky_ex(covs + jitter * eye)
            if int(info.abs().max()) == 0 and bool(torch.isfinite(tril).all()):
                chosen, factor = jitter, tril
                break
        if factor is None:  # nothing on the ladder is positive definite: same outcome as the reference's last attempt
            factor = torch.linalg.cholesky(covs + chosen * eye)
        gmm = torch.distributions.MultivariateNormal(loc=means, scale_tril=factor)
    return gmm, (0 if chosen == 0.0 else chosen)


class GmmState:
    """Device-resident form of a fitted class-wise ``MultivariateNormal``: for every component the precision
    ``(L L^T)^-1`` (float32 ``scale_tril`` widened exactly, inverted in f64, packed for the MFMA kernel), the mean and
    ``-0.5 * D * log(2 pi) - sum(log diag L)``.  ``log_prob`` = ``gmm.log_prob(x[:, None, :])`` -> ``(N, C)`` f32."""

    def __init__(self, gmm):
        loc = gmm.loc.detach().cpu()
        tril = gmm.scale_tril.detach().cpu()
        self.n_comp, self.dim = loc.shape
        self.means = [_hip.to_device(loc[c].numpy(), torch.float32) for c in range(self.n_comp)]
        self.packed = []
        for c in range(self.n_comp):
            prec = torch.cholesky_inverse(tril[c].double()).numpy()
            self.packed.append(_hip.pack_weights(_hip.to_device(prec, torch.float64)))
        half_log_det = tril.diagonal(dim1=-2, dim2=-1).log().sum(-1)  # f32, as torch
        self.const = (-0.5 * self.dim * float(np.log(2 * np.pi)) - half_log_det.double()).tolist()

    def log_prob_device(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(torch.float32).contiguous()
        cols = []
        for c in range(self.n_comp):
            neg_m = _hip.md_score(x, self.means[c], self.packed[c])  # -(x-mu)^T P (x-mu), diff in f32 like torch
            cols.append((0.5 * neg_m + self.const[c]).to(torch.float32))
        return torch.stack(cols, dim=1).contiguous()

    def energy_device(self, x: torch.Tensor) -> torch.Tensor:
        lse, _ = _hip.row_lse_msp(self.log_prob_device(x), True, False)
        return lse
