"""Numerical helpers of the logits/features postprocessors with the reference's
names (``runia_core/inference/funcs.py``: ``mahalanobis_preprocess`` :33-66,
``mahalanobis_postprocess`` :69-102, ``normalizer`` :105-115).
Fits (setup time) make the same scikit-learn call as the reference; scoring runs
on the GPU."""
from __future__ import annotations

import warnings
from typing import Dict, Tuple, Union

import numpy as np
import torch

from .. import _hip, config
from ..device_fit import empirical_precision_device

__all__ = [
    "RouteDICE",
    "ash_s_conv_layer",
    "ash_s_linear_layer",
    "gmm_fit",
    "generalized_entropy",
    "get_mcd_pred_uncertainty_score",
    "get_predictive_uncertainty_score",
    "get_dice_feat_mean_react_percentile",
    "mahalanobis_preprocess",
    "mahalanobis_postprocess",
    "normalizer",
    "MahalanobisState",
    "GmmState",
]


def mahalanobis_preprocess(ind_data: Dict[str, np.ndarray], num_classes: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per-class means and the pooled precision matrix of the class-centred training features.

    Returns ``(class_mean [C, D], precision [D, D])``; a class without samples warns and
    yields a NaN mean (scored as -inf later), like the reference."""
    from sklearn.covariance import EmpiricalCovariance

    feats, labels = ind_data["train features"], ind_data["train labels"]
    on_device = config.use_device_fit() and getattr(feats, "dtype", None) in (np.float32, np.float64)
    class_mean, centered = [], []
    for c in range(num_classes):
        class_samples = feats[labels == c]
        if len(class_samples) == 0:
            warnings.warn(f"No train examples for class {c}")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            class_mean.append(class_samples.mean(0))
        if not on_device:
            centered.append(class_samples - class_mean[c].reshape(1, -1))
    class_mean = np.stack(class_mean)
    if on_device and class_mean.dtype == feats.dtype:
        # the centred rows are formed on the device from ONE upload of the features (x - mu_label in the features' dtype, then the
        # reference's cast to float32: the values of `pooled` below, in the rows' own order - the covariance does not care) instead
        # of a second 410 MB host array and its upload (cfg3 size: 0.2 of the fit's 0.25 s)
        from ..device_fit import pinvh_device

        dt = torch.float32 if feats.dtype == np.float32 else torch.float64
        lab = np.asarray(labels).reshape(-1)
        keep = np.flatnonzero((lab >= 0) & (lab < num_classes))
        xd = _hip.to_device(feats, dt)
        idx = _hip.to_device(keep.astype(np.int64), torch.int64)
        mu = _hip.to_device(np.nan_to_num(class_mean), dt)  # (an empty class's NaN row is never indexed)
        z = (xd.index_select(0, idx) - mu.index_select(0, _hip.to_device(lab[keep].astype(np.int64), torch.int64))).to(torch.float32)
        _, cov = _hip.covariance(z)
        return class_mean, _hip.to_host(pinvh_device(cov))
    if on_device:  # (a mean of another dtype than the rows: not a case NumPy produces for float rows)
        centered = [feats[labels == c] - class_mean[c].reshape(1, -1) for c in range(num_classes)]
    pooled = np.concatenate(centered).astype(np.float32)
    if config.use_device_fit():
        return class_mean, empirical_precision_device(pooled)
    from ..host_threads import host_compute

    with host_compute():
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

    The float32 products and factorisations run on torch's intra-op pool capped at the container's CPU quota
    (``host_threads.host_compute``: 716 -> 60 ms per fit of 50 000 x 512 rows on a 16-CPU quota with 256 threads visible).

    With ``config.use_device_fit()`` (the default where a GPU is present) the moments and the factorisations run on the device
    instead (``device_fit.gmm_fit_device``, round 6: 88 -> ~10 ms per fit of 50 000 x 512 rows); the returned distribution then
    carries the device copies of its parameters for ``GmmState`` (``_runia_device_params``).

    Returns ``(MultivariateNormal, jitter)``."""
    from .. import config as _config
    from ..host_threads import host_compute

    if _config.use_device_fit():
        from ..device_fit import gmm_fit_device

        with torch.no_grad():
            loc_d, tril_d, jitter = gmm_fit_device(embeddings, labels, num_classes)
            # (validate_args=False: the factor is lower triangular with a positive diagonal by construction - info == 0 and all finite;
            # torch's constraint check walks the ten 2048 x 2048 factors on the host for 0.11 s)
            gmm = torch.distributions.MultivariateNormal(loc=loc_d.cpu(), scale_tril=tril_d.cpu(), validate_args=False)
        gmm._runia_device_params = (loc_d, tril_d)
        return gmm, jitter
    with torch.no_grad(), host_compute():
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
    """Device-resident form of a fitted class-wise ``MultivariateNormal``: for every component the inverse ``W = L^-1`` of its float32
    ``scale_tril`` (widened exactly, inverted in f64 by ``runia_tril_inverse_f64``, rounded to f32: lower triangular), the mean and
    ``-0.5 * D * log(2 pi) - sum(log diag L)``.  ``log_prob`` = ``gmm.log_prob(x[:, None, :])`` -> ``(N, C)`` f32 from ONE launch over
    all components (``runia_gmm_log_prob_f32``: ``|| W (x - mu) ||^2`` on the f32 matrix cores - torch's own arithmetic, a triangular
    solve in f32 - with the zero half of ``W`` skipped).

    ``dense=True`` keeps rounds 4-5's form instead (the f64 precision ``W^T W`` packed for ``runia_md_score_f32``, one launch per
    component, 2 D^2 multiply-adds per row and component): the comparison leg of the tests and of ``tools/ablate/run_gmm.py``."""

    def __init__(self, gmm, dense: bool = False):
        loc = gmm.loc.detach().cpu()
        tril = gmm.scale_tril.detach().cpu()
        self.n_comp, self.dim = loc.shape
        self.dense = bool(dense)
        dev = getattr(gmm, "_runia_device_params", None)   # a device fit left its parameters on the device
        tril_dev = dev[1].to(torch.float64) if dev is not None else _hip.to_device(tril.double().numpy(), torch.float64)
        w = _hip.tril_inverse(tril_dev)   # all classes in one launch
        half_log_det = tril.diagonal(dim1=-2, dim2=-1).log().sum(-1)  # f32, as torch
        const = -0.5 * self.dim * float(np.log(2 * np.pi)) - half_log_det.double()
        self.const = const.tolist()
        if self.dense:
            self.means = [_hip.to_device(loc[c].numpy(), torch.float32) for c in range(self.n_comp)]
            self.packed = [_hip.pack_weights(_hip.matmul_f64(w[c].t().contiguous(), w[c])) for c in range(self.n_comp)]
        else:
            self.means_dev = dev[0].contiguous() if dev is not None else _hip.to_device(np.ascontiguousarray(loc.numpy()), torch.float32)
            self.w_tril = w.to(torch.float32).contiguous()
            self.const_dev = _hip.to_device(const.numpy(), torch.float64)

    def log_prob_device(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(torch.float32).contiguous()
        if not self.dense:
            return _hip.gmm_log_prob(x, self.means_dev, self.w_tril, self.const_dev, True, False)[0]
        cols = []
        for c in range(self.n_comp):
            neg_m = _hip.md_score(x, self.means[c], self.packed[c])  # -(x-mu)^T P (x-mu), diff in f32 like torch
            cols.append((0.5 * neg_m + self.const[c]).to(torch.float32))
        return torch.stack(cols, dim=1).contiguous()

    def energy_device(self, x: torch.Tensor) -> torch.Tensor:
        if not self.dense:  # logsumexp over the components inside the same call: no (N, C) table
            return _hip.gmm_log_prob(x.to(torch.float32).contiguous(), self.means_dev, self.w_tril, self.const_dev, False, True)[1]
        lse, _ = _hip.row_lse_msp(self.log_prob_device(x), True, False)
        return lse


# ---- the other free functions of the reference's inference/funcs.py (its __all__, funcs.py:19-30) --------------------
class RouteDICE(torch.nn.Linear):
    """Drop-in for the reference's ``RouteDICE`` (``inference/funcs.py:124-189``): a final linear layer whose weights with a
    low mean contribution ``info * W`` (below the ``p``-th percentile) are removed.  Same constructor, attributes
    (``p``, ``info``, ``masked_w``, ``contrib``, ``thresh``) and lazily computed mask; ``forward`` = ``x @ masked_w.T + bias``
    on the f32 matrix-core kernel (``runia_linear_f32``) instead of the reference's (N, C, D) broadcast product."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, p: int = 90, conv1x1: bool = False,
                 info: Union[None, np.ndarray] = None):
        assert 0 < p < 100, "p must be greater than 0 and less than 100"
        if info is not None:
            assert isinstance(info, np.ndarray), "info must be a numpy array or None"
        super().__init__(in_features, out_features, bias)
        if conv1x1:
            self.weight = torch.nn.Parameter(torch.Tensor(out_features, in_features, 1, 1))
        self.p = p
        self.info = info
        self.masked_w = None
        self.contrib = None
        self.thresh = None

    def calculate_mask_weight(self):
        self.contrib = self.info[None, :] * self.weight.data.cpu().numpy()
        self.thresh = np.percentile(self.contrib, self.p)
        mask = torch.Tensor((self.contrib > self.thresh))
        self.masked_w = (self.weight.squeeze().cpu() * mask).to(_hip.require_gpu())

    def forward(self, x):
        if self.masked_w is None:
            self.calculate_mask_weight()
        dev = self.masked_w.device
        xd = x.detach().to(dev, torch.float32).reshape(-1, x.shape[-1])
        w = self.masked_w.to(torch.float32).reshape(self.masked_w.shape[0], -1).contiguous()
        b = None if self.bias is None else self.bias.detach().to(dev, torch.float32).contiguous()
        out = _hip.linear(xd, w, b)
        return out.reshape(*x.shape[:-1], out.shape[-1])


def ash_s_conv_layer(x: torch.Tensor, percentile: int = 65):
    """ASH-S for (B, C, H, W) maps (reference ``inference/funcs.py:194-227``): keep the
    ``k = n - int(np.round(n * percentile / 100))`` largest activations of each sample, zero the rest and multiply the
    sample by ``exp(sum before / sum after)``.  Like the reference - whose ``view`` + ``scatter_`` write through to the
    argument - a contiguous ``x`` is left PRUNED (not scaled); the scaled tensor is returned, on ``x``'s device."""
    assert x.dim() == 4
    assert 0 <= percentile <= 100
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous():
        return _hip.ash_s_conv(x, percentile, True)
    xd = x.detach().to(_hip.require_gpu(), torch.float32).contiguous()
    y = _hip.ash_s_conv(xd, percentile, True)
    if x.is_contiguous():
        with torch.no_grad():
            x.copy_(xd.to(x.device, x.dtype))
    return y.to(x.device, x.dtype)


def ash_s_linear_layer(x: np.ndarray, percentile: int = 85):
    """ASH-S for 2-D activations (reference ``inference/funcs.py:230-261``), NumPy in, NumPy out.  Every kept activation
    stays at its own index (the reference scatters ``np.partition``'s values at ``np.argpartition``'s indices, which
    permutes them within a row when the two orders differ - INTEGRATION.md, known divergences)."""
    assert x.ndim == 2
    assert 0 <= percentile <= 100
    y = _hip.to_host(_hip.ash_s(_hip.to_device(np.ascontiguousarray(x), torch.float32), percentile))
    return y.astype(x.dtype, copy=False) if np.issubdtype(x.dtype, np.floating) else y


def generalized_entropy(probs, gamma, M):
    """``-sum over the M largest probabilities of p^gamma (1 - p)^gamma`` per row (reference
    ``inference/funcs.py:347-375``); computed in f32 on the device, returned in the dtype of ``probs``."""
    arr = probs.detach().cpu().numpy() if isinstance(probs, torch.Tensor) else np.asarray(probs)
    s = _hip.to_host(_hip.gen_entropy(_hip.to_device(np.ascontiguousarray(arr), torch.float32), float(gamma), int(M)))
    return s.astype(arr.dtype, copy=False) if np.issubdtype(arr.dtype, np.floating) else s


def get_predictive_uncertainty_score(input_samples: torch.Tensor, mcd_nro_samples: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Predictive entropy and mutual information of MC-dropout outputs (reference ``inference/funcs.py:430-465``):
    ``input_samples`` (N * n_mc, C) logits, an image's n_mc rows consecutive -> ``(pred_h (N,), mi (N,))`` f32 tensors on
    the input's device.  One row-streaming launch (``runia_mcd_uncertainty_f32``)."""
    assert input_samples.shape[0] % mcd_nro_samples == 0, (
        "Input tensor first dimension must be " "divisible by the mcd_nro_samples"
    )
    x = input_samples.detach().to(_hip.require_gpu(), torch.float32)
    ph, mi, _ = _hip.mcd_uncertainty(x, int(mcd_nro_samples), False)
    return ph.to(input_samples.device), mi.to(input_samples.device)


def get_mcd_pred_uncertainty_score(dnn_model: torch.nn.Module, input_dataloader, mcd_nro_samples: int = 2
                                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """The dataloader form (reference ``inference/funcs.py:378-427``): ``mcd_nro_samples`` forward passes per batch, then
    the same kernel -> ``(softmax samples (N, n_mc, C), pred_h (N,), mi (N,))`` on the device.  As upstream the passes of
    a batch are concatenated in the order they were made, so batches of ONE image give image-major rows."""
    device = _hip.require_gpu()
    outs = []
    with torch.no_grad():
        for image, _ in input_dataloader:
            image = image.to(device)
            for _s in range(mcd_nro_samples):
                outs.append(dnn_model(image))
        logits = torch.cat(outs, dim=0).to(device, torch.float32)
        ph, mi, probs = _hip.mcd_uncertainty(logits, int(mcd_nro_samples), True)
    return probs.reshape(-1, mcd_nro_samples, logits.shape[1]), ph, mi


def get_dice_feat_mean_react_percentile(dnn_model: torch.nn.Module, ind_dataloader, react_percentile: int = 90
                                        ) -> Tuple[np.ndarray, float]:
    """DICE expected activations and the ReAct clipping threshold (reference ``inference/funcs.py:468-495``): the
    model's feature maps are average-pooled per channel on the device (``runia_map_reduce_f32``), their mean over the
    dataset and the ``react_percentile``-th percentile of all pooled activations (NumPy's linear interpolation, as
    upstream) come back as ``(ndarray (C,), float)``."""
    assert 0 < react_percentile < 100, "react_percentile must be greater than 0 and less than 100"
    feat_log = []
    dnn_model.eval()
    assert dnn_model.dice_precompute
    device = _hip.require_gpu()
    with torch.no_grad():
        for inputs, _targets in ind_dataloader:
            outputs = dnn_model(inputs.to(device)).to(torch.float32)
            b, c, h, w = outputs.shape
            rows = _hip.map_reduce(outputs.contiguous(), h, w, "mean")          # (b*c, h): mean over W
            pooled = _hip.map_reduce(rows.contiguous(), 1, h, "mean").reshape(b, c)  # mean over H
            feat_log.append(_hip.to_host(pooled))
    feat_log_array = np.array(feat_log).squeeze()
    return feat_log_array.mean(0), np.percentile(feat_log_array, react_percentile)
