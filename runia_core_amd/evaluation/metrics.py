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

__all__ = ["get_auroc_results", "auroc_fpr95_aupr", "auroc_fpr95_aupr_device", "log_evaluate_postprocessors",
           "select_and_log_best_larex"]


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


_COLUMNS = ["auroc", "fpr@95", "aupr", "fpr", "tpr"]


def _rows_to_frame(rows: Dict[str, list]):
    import pandas as pd

    return pd.DataFrame.from_dict(rows, orient="index", columns=_COLUMNS) if rows else pd.DataFrame(columns=_COLUMNS)


def _refuse_mlflow(flag: bool) -> None:
    if flag:
        raise NotImplementedError("mlflow logging is outside runia_core_amd (the experiment-tracking side of the harness): "
                                  "pass mlflow_logging=False and log the returned table yourself")


def log_evaluate_postprocessors(ind_dict, ood_dict, ood_datasets_names, experiment_name_extension: str = "",
                                return_density_scores=None, log_step=None, mlflow_logging: bool = False, postprocessors=None,
                                cfg=None, roc_curves: bool = True):
    """The harness loop of the reference (``evaluation/metrics.py:265-380``): every postprocessor of ``postprocessors``
    (default: the whole registry) is instantiated with ``cfg``, set up on ``ind_dict["train latent_space_means"]`` (with
    ``ind_train_labels``), scores the valid split and every OoD set, and each (OoD set, postprocessor) pair becomes the row
    ``f"{ood} {postprocessor}{experiment_name_extension}"`` of the results table (columns ``auroc, fpr@95, aupr, fpr, tpr``).
    Returns ``{"results_df": DataFrame}`` (+ ``"InD"`` / ``"OoD"`` scores of ``return_density_scores``).

    Same calls in the same order as the reference; the arithmetic behind ``setup`` / ``postprocess`` / the metrics is this
    package's (HIP kernels).  Additive: the splits may be device tensors (then nothing is uploaded again - rows stay in HBM
    from the PCA transform to the metrics); ``roc_curves=False`` leaves the ``fpr`` / ``tpr`` columns empty and takes the
    three scalars of every row from the device-only metrics kernel, one read-back for the whole table."""
    from ..inference.postprocessors import postprocessors_dict

    _refuse_mlflow(mlflow_logging)
    if return_density_scores is not None:
        assert return_density_scores in postprocessors_dict.keys()
    if postprocessors is None:
        postprocessors = postprocessors_dict.keys()
    ind_scores_dict, ood_scores_dict = {}, {}

    def score(pp, rows, labels):
        if isinstance(rows, torch.Tensor) and rows.is_cuda and hasattr(pp, "postprocess_device"):
            return pp.postprocess_device(rows)
        return pp.postprocess(rows if not isinstance(rows, torch.Tensor) else rows.cpu().numpy(), pred_labels=labels)

    def host(a):
        return a.cpu().numpy() if isinstance(a, torch.Tensor) else a

    for postprocessor in postprocessors:
        postp_instance = postprocessors_dict[postprocessor](cfg=cfg)
        postp_instance._setup_flag = False
        train_rows = ind_dict["train latent_space_means"]
        if isinstance(train_rows, torch.Tensor) and train_rows.is_cuda and hasattr(postp_instance, "setup_device"):
            # additive: training rows already in HBM (log_evaluate_larex(device_resident=True)) - same fitted state, no host pass
            postp_instance.setup_device(train_rows, ind_train_labels=host(ind_dict["train labels"]))
        else:
            postp_instance.setup(host(train_rows), ind_train_labels=host(ind_dict["train labels"]))
        ind_scores_dict[postprocessor] = score(postp_instance, ind_dict["valid latent_space_means"], ind_dict["valid labels"])
        ood_scores_dict[postprocessor] = {}
        for ood_dataset_name in ood_datasets_names:
            ood_scores_dict[postprocessor][ood_dataset_name] = score(
                postp_instance, ood_dict[f"{ood_dataset_name} latent_space_means"], ood_dict[f"{ood_dataset_name} labels"])

    rows, pending = {}, []
    for ood_dataset_name in ood_datasets_names:
        for postprocessor in postprocessors:
            name = f"{ood_dataset_name} {postprocessor}{experiment_name_extension}"
            ind_s, ood_s = ind_scores_dict[postprocessor], ood_scores_dict[postprocessor][ood_dataset_name]
            if roc_curves:
                auroc, fpr95, aupr, fpr, tpr = auroc_fpr95_aupr(ind_s, ood_s)
                rows[name] = [auroc, fpr95, aupr, fpr.tolist(), tpr.tolist()]
            else:
                rows[name] = None
                pending.append((name, auroc_fpr95_aupr_device(ind_s, ood_s, to_host=False)))
    if pending:
        from .. import _hip

        table = _hip.to_host(torch.stack([t for _, t in pending]))
        for (name, _), r in zip(pending, table):
            rows[name] = [float(r[0]), float(r[1]), float(r[2]), None, None]
    results = {"results_df": _rows_to_frame(rows)}
    if return_density_scores is not None:
        results["InD"] = host(ind_scores_dict[return_density_scores])
        results["OoD"] = {k: host(v) for k, v in ood_scores_dict[return_density_scores].items()}
    return results


def select_and_log_best_larex(overall_metrics_df, n_pca_components_list, postprocessor_name: str,
                              multiple_ood_datasets_flag: bool, log_mlflow: bool = False):
    """Best PCA size of one postprocessor by mean AUROC over the OoD sets (reference ``evaluation/metrics.py:383-462``):
    returns ``(auroc, aupr, fpr@95, n_components)`` of the best configuration, ``n_components = 0`` for the run without
    PCA.  Rows are matched by substring exactly as upstream (so ``"MD"`` also averages the ``"cMD"`` rows)."""
    import pandas as pd

    from ..inference.postprocessors import postprocessors_dict

    _refuse_mlflow(log_mlflow)
    assert postprocessor_name in postprocessors_dict.keys(), f"Got {postprocessor_name}"
    cols = ["auroc", "fpr@95", "aupr"]
    means = {}
    no_pca = [r for r in overall_metrics_df.index if postprocessor_name in r and "anomalies" not in r and "PCA" not in r]
    means[postprocessor_name] = overall_metrics_df.loc[no_pca, cols].astype(float).mean()
    for n_components in n_pca_components_list:
        picked = [r for r in overall_metrics_df.index
                  if postprocessor_name in r and f"PCA {n_components}" in r and r.split(f"PCA {n_components}")[-1] == ""]
        means[f"{postprocessor_name} PCA {n_components}"] = overall_metrics_df.loc[picked, cols].astype(float).mean()
    means_df = pd.DataFrame(means).T[cols]
    best_index = means_df[means_df.auroc == means_df.auroc.max()].index[0]
    best_n_comps = int(best_index.split()[-1]) if "PCA" in best_index else 0
    return (means_df.loc[best_index, "auroc"], means_df.loc[best_index, "aupr"], means_df.loc[best_index, "fpr@95"],
            best_n_comps)
