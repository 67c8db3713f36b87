"""Detection metrics after the scoring path (reference
``runia_core/evaluation/metrics.py:37-100`` ``get_auroc_results``).

``auroc_fpr95_aupr_device`` keeps the whole step on the GPU (``runia_ood_metrics_*``: radix sort of the scores, scan of
the labels and run ends, trapezoid sums - ``csrc/metrics.hip``).  ``get_auroc_results`` / ``auroc_fpr95_aupr`` are the
harness-facing forms: they also return the ROC curve as lists for the results table, which is host data by nature, and
build it with NumPy (binary AUROC, ROC -> FPR@95, precision-recall -> AUPR; torchmetrics' sigmoid of scores outside
[0, 1] and its float32 curves reproduced).
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


def _clf_curve(preds: np.ndarray, target: np.ndarray):
    order = np.argsort(-preds, kind="stable")
    preds, target = preds[order], target[order]
    distinct = np.nonzero(preds[1:] - preds[:-1])[0]
    idx = np.concatenate([distinct, [target.size - 1]])
    tps = np.cumsum(target)[idx]
    fps = 1 + idx - tps
    return fps, tps


def auroc_fpr95_aupr(ind_scores: np.ndarray, ood_scores: np.ndarray):
    """InD = positive class.  Returns ``(auroc, fpr@95, aupr, fpr_curve, tpr_curve)``."""
    # float32 score sets stay float32 through torchmetrics' sigmoid (np.vstack keeps the dtype), anything else is float64
    dt = np.float32 if (np.asarray(ind_scores).dtype == np.float32 and np.asarray(ood_scores).dtype == np.float32) else np.float64
    scores = np.concatenate([np.ravel(ind_scores), np.ravel(ood_scores)]).astype(dt)
    labels = np.concatenate([np.ones(np.size(ind_scores), dtype=np.int64), np.zeros(np.size(ood_scores), dtype=np.int64)])
    if not np.all((scores >= 0) & (scores <= 1)):
        with np.errstate(over="ignore"):
            scores = (dt(1.0) / (dt(1.0) + np.exp(-scores))).astype(dt)
    scores = scores.astype(np.float64)
    fps, tps = _clf_curve(scores, labels)
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
