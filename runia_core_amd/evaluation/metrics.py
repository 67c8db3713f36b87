"""Detection metrics after the scoring path (reference
``runia_core/evaluation/metrics.py:37-100`` ``get_auroc_results``).

Round 1: host NumPy restatement of the torchmetrics curves the reference calls
(binary AUROC, ROC -> FPR@95, precision-recall -> AUPR), kept bit-compatible with the
reference's goldens (including torchmetrics' sigmoid of scores outside [0, 1] and its
float32 curves).  A device sort + scan version is SURVEY 8f "next #2".
"""
from __future__ import annotations

from typing import Dict, Tuple, Union

import numpy as np
import torch

__all__ = ["get_auroc_results", "auroc_fpr95_aupr"]


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
    scores = np.concatenate([np.ravel(ind_scores), np.ravel(ood_scores)]).astype(np.float64)
    labels = np.concatenate([np.ones(np.size(ind_scores), dtype=np.int64), np.zeros(np.size(ood_scores), dtype=np.int64)])
    if not np.all((scores >= 0) & (scores <= 1)):
        with np.errstate(over="ignore"):
            scores = 1.0 / (1.0 + np.exp(-scores))
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
