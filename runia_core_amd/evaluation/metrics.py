"""Detection metrics after the scoring path (reference
``runia_core/evaluation/metrics.py:37-100`` ``get_auroc_results``).

``auroc_fpr95_aupr_device`` keeps the whole step on the GPU (``runia_ood_metrics_*``: radix sort of the scores, scan of
the labels and run ends, trapezoid sums - ``csrc/metrics.hip``).  ``get_auroc_results`` / ``auroc_fpr95_aupr`` are the
harness-facing forms: they also return the ROC curve as lists for the results table, which is host data by nature.  The
O(N log N) part - torchmetrics' ``_binary_clf_curve``: sigmoid of scores outside [0, 1], descending sort, cumulative
true / false positives at the end of every run of equal scores - runs on the GPU (``runia_ood_clf_curve_*``); only the
compacted curve comes back, and the float32 ROC / precision-recall points and their trapezoid sums are formed from it
exactly as torchmetrics / sklearn form them (same dtype, same ``torch.trapz`` call).  No CPU fallback: without a GPU the
functions raise, like every other scoring entry point.
"""
from __future__ import annotations

from typing import Dict, Tuple, Union

import numpy as np
import torch

__all__ = ["get_auroc_results", "auroc_fpr95_aupr", "auroc_fpr95_aupr_device"]


def auroc_fpr95_aupr_device(ind_scores, ood_scores, to_host: bool = True):
    """``(auroc, fpr@95, aupr)`` of InD (positive) against OoD scores, computed on the GPU.  Scores may be device tensors
    (no copy) or host arrays (uploaded); f32 scores keep torchmetrics' f32 sigmoid, f64 its f64 sigmoid.
    ``to_host=False`` returns the device tensor ``[3]`` (no synchronisation)."""
    from .. import _hip

    def dev(a):
        if isinstance(a, torch.Tensor):
            return a if a.is_cuda else a.to(_hip.require_gpu())
        a = np.asarray(a)
        return _hip.to_device(a, torch.float32 if a.dtype == np.float32 else torch.float64)

    a, b = dev(ind_scores), dev(ood_scores)
    if a.dtype != b.dtype:
        a, b = a.to(torch.float64), b.to(torch.float64)
    out = _hip.ood_metrics(a, b)
    if not to_host:
        return out
    r = _hip.to_host(out)
    return float(r[0]), float(r[1]), float(r[2])


def _clf_curve_device(ind_scores, ood_scores):
    """torchmetrics' ``_binary_clf_curve`` on the GPU -> host ``(fps, tps)`` int64 arrays, one entry per run of equal
    scores in descending order.  Float32 score sets keep torchmetrics' float32 sigmoid, anything else its float64 one."""
    from .. import _hip

    def dev(a, dt):
        if isinstance(a, torch.Tensor):
            return (a if a.is_cuda else a.to(_hip.require_gpu())).reshape(-1).to(dt)
        return _hip.to_device(np.ravel(np.asarray(a)), dt)

    def is_f32(a):
        if isinstance(a, torch.Tensor):
            return a.dtype == torch.float32
        return np.asarray(a).dtype == np.float32

    dt = torch.float32 if (is_f32(ind_scores) and is_f32(ood_scores)) else torch.float64
    _, tps, fps = _hip.ood_clf_curve(dev(ind_scores, dt), dev(ood_scores, dt))
    return fps, tps


def auroc_fpr95_aupr(ind_scores, ood_scores):
    """InD = positive class.  Returns ``(auroc, fpr@95, aupr, fpr_curve, tpr_curve)``; scores may be host arrays or
    device tensors."""
    fps, tps = _clf_curve_device(ind_scores, ood_scores)
    tps_r = np.concatenate([[0], tps]).astype(np.float32)
    fps_r = np.concatenate([[0], fps]).astype(np.float32)
    fpr = fps_r / fps_r[-1]
    tpr = tps_r / tps_r[-1]
    auroc = float(torch.trapz(torch.from_numpy(tpr), torch.from_numpy(fpr)).item())
    fpr95 = float(fpr[np.where(tpr >= 0.95)[0][0]])
    tps32, fps32 = tps.astype(np.float32), fps.astype(np.float32)
    precision = np.concatenate([(tps32 / (tps32 + fps32))[::-1], np.ones(1, dtype=np.float32)])
    recall = np.concatenate([(tps32 / tps32[-1])[::-1], np.zeros(1, dtype=np.float32)])
    trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
    aupr = float(-trap(precision, recall))
    return auroc, fpr95, aupr, fpr, tpr


def get_auroc_results(detect_exp_name: str, ind_samples_scores: np.ndarray, ood_samples_scores: np.ndarray,
                      return_results_for_mlflow: bool = False) -> Union["pd.DataFrame", Tuple["pd.DataFrame", Dict]]:  # noqa: F821
    """AUROC, FPR@95, AUPR and the ROC curve as a one-row DataFrame named ``detect_exp_name``
    (columns ``auroc, fpr@95, aupr, fpr, tpr``); optionally also a dict with mlflow-safe keys."""
    import pandas as pd

    auroc, fpr95, aupr, fpr, tpr = auroc_fpr95_aupr(ind_samples_scores, ood_samples_scores)
    table = pd.DataFrame.from_dict(
        {detect_exp_name: [auroc, fpr95, aupr, fpr.tolist(), tpr.tolist()]},
        orient="index",
        columns=["auroc", "fpr@95", "aupr", "fpr", "tpr"],
    )
    if not return_results_for_mlflow:
        return table
    results = table.loc[detect_exp_name, ["auroc", "fpr@95", "aupr"]].to_dict()
    results["fpr_95"] = results.pop("fpr@95")
    return table, results
