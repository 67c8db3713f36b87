"""The evaluation harness of the latent-space methods (reference ``runia_core/evaluation/latent_space.py:30-221``
``log_evaluate_larex``): baselines table, postprocessors on the full latent vectors, then - for every size of
``cfg.n_pca_components`` - PCA refit on the training split, transform of the valid / OoD splits, every postprocessor again;
best PCA size per postprocessor by mean AUROC; 95 % InD thresholds of the best configurations.

This is the loop the scoring path exists for: f1 (fits on the device), a3-a9 (PCA + postprocessors) and f2 (metrics on the
device) strung together.  Same calls, same order, same row names (``f"{ood} {postprocessor} PCA {n}"``) as upstream.  What
is NOT here: mlflow logging, the matplotlib plots (score histograms, ROC curves), csv export - experiment tracking and
visualisation, outside the scoring path (``mlflow_logging=True`` / ``save_*=True`` raise).

Additive ``device_resident=True``: every split is uploaded once, the PCA transforms and the postprocessors run on device
rows (``transform_device`` / ``postprocess_device``) and the metrics of the whole table come back in one read (no ROC-curve
columns) - same scores, same table values.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _hip
from ..dimensionality_reduction import apply_pca_ds_split, apply_pca_transform, device_pca_for
from .metrics import _COLUMNS, _refuse_mlflow, _rows_to_frame, get_auroc_results, log_evaluate_postprocessors, \
    select_and_log_best_larex

__all__ = ["log_evaluate_larex", "log_baselines"]


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def log_baselines(baselines_names: List[str], ind_dataset: str, ind_data_dict, ood_baselines_scores, ood_datasets: List[str],
                  overall_metrics_df, mlflow_logging: bool, logs_folder=None):
    """Rows ``f"{ood} {baseline}"`` of precomputed baseline scores (reference :224-322; ``pred_h`` / ``mi`` are negated so that
    InD scores higher).  The score-distribution plots of the reference are not made."""
    _refuse_mlflow(mlflow_logging)
    for baseline in baselines_names:
        for ood_dataset in ood_datasets:
            sign = -1.0 if baseline in ("pred_h", "mi") else 1.0
            name = f"{ood_dataset} {baseline}"
            table = get_auroc_results(name, sign * np.asarray(ind_data_dict[baseline]),
                                      sign * np.asarray(ood_baselines_scores[name]))
            overall_metrics_df.loc[name] = table.loc[name]
    return overall_metrics_df


def _get_best_postprocessors_metrics(baselines_names, overall_metrics_df, postprocessors, n_pca_components, ood_datasets_names):
    """Reference :420-518 without its mlflow calls."""
    multiple = len(ood_datasets_names) > 1
    best = {"best": []}
    for postprocessor in postprocessors:
        auroc, aupr, fpr, best_comp = select_and_log_best_larex(overall_metrics_df, n_pca_components, postprocessor, multiple)
        name = f"{postprocessor}" if best_comp == 0 else f"{postprocessor} PCA {best_comp}"
        best[postprocessor] = {"best_comp": name, "auroc": auroc, "aupr": aupr, "fpr": fpr}
        for ood_dataset in ood_datasets_names:
            best["best"].append(f"{ood_dataset} {name}")
    return best


def _get_best_post_processor_thresholds(postprocessors_names, best_postprocessors_dict, cfg, ind_data, ood_data, ind_dev=None,
                                        ood_dev=None, train_dev=None):
    """Reference :521-605 without the histograms: refit every postprocessor at its best PCA size, threshold = mean - 1.645 std
    of its InD valid scores; the OoD scores of that configuration are attached to ``ood_data`` under ``f"{ood} {best}"``.
    ``ind_dev`` / ``ood_dev`` (device-resident sweep): the same splits as device tensors - fits and scores then run from HBM
    (``setup_device`` / ``postprocess_device``), only the score vectors come back."""
    from ..inference.postprocessors import postprocessors_dict

    on_dev = ind_dev is not None
    thresholds = {}
    for name in postprocessors_names:
        # (the reference copies the splits here; nothing below writes to them)
        train_data = ind_dev["train latent_space_means"] if on_dev else np.asarray(ind_data["train latent_space_means"])
        valid_data = ind_dev["valid latent_space_means"] if on_dev else np.asarray(ind_data["valid latent_space_means"])
        pca_transformation = None
        pp = postprocessors_dict[name](cfg=cfg)
        pp._setup_flag = False
        best_postp = best_postprocessors_dict[name]["best_comp"]
        dev_path = on_dev and hasattr(pp, "setup_device") and hasattr(pp, "postprocess_device")
        if not dev_path and on_dev:
            train_data, valid_data = np.asarray(ind_data["train latent_space_means"]), np.asarray(ind_data["valid latent_space_means"])
        fit_dev = (not dev_path) and train_dev is not None and hasattr(pp, "setup_device")  # host-array mode: the FIT still runs from HBM
        if fit_dev:
            train_data = train_dev
        if "PCA" in best_postp:
            train_data, pca_transformation = apply_pca_ds_split(samples=train_data, nro_components=int(best_postp.split("PCA")[1]))
        if fit_dev:
            pp.setup_device(train_data, ind_train_labels=ind_data["train labels"])
            if "PCA" in best_postp:
                valid_data = apply_pca_transform(valid_data, pca_transformation)
            ind_valid = pp.postprocess(valid_data, pred_labels=ind_data["valid labels"])
        elif dev_path:
            pp.setup_device(train_data, ind_train_labels=ind_data["train labels"])
            dp = device_pca_for(pca_transformation) if "PCA" in best_postp else None
            if dp is not None:
                valid_data = dp.transform_device(valid_data)
            ind_valid = _hip.to_host(pp.postprocess_device(valid_data))
        else:
            pp.setup(train_data, ind_train_labels=ind_data["train labels"])
            if "PCA" in best_postp:
                valid_data = apply_pca_transform(valid_data, pca_transformation)
            ind_valid = pp.postprocess(valid_data, pred_labels=ind_data["valid labels"])
        thresholds[best_postp] = np.mean(ind_valid) - (1.645 * np.std(ind_valid))
        for ood_dataset_name in _cfg_get(cfg, "ood_datasets"):
            if dev_path:
                rows = ood_dev[f"{ood_dataset_name} latent_space_means"]
                rows = dp.transform_device(rows) if dp is not None else rows
                ood_data[f"{ood_dataset_name} {best_postp}"] = _hip.to_host(pp.postprocess_device(rows))
                continue
            ood_dataset = np.asarray(ood_data[f"{ood_dataset_name} latent_space_means"])
            if "PCA" in best_postp:
                ood_dataset = apply_pca_transform(ood_dataset, pca_transformation)
            ood_data[f"{ood_dataset_name} {best_postp}"] = pp.postprocess(ood_dataset, pred_labels=ood_data[f"{ood_dataset_name} labels"])
    return thresholds, ood_data


def log_evaluate_larex(cfg, baselines_names: List[str], ood_baselines_scores: Dict[str, np.ndarray], ind_data_dict, ood_data_dict,
                       mlflow_run_name: str = "", mlflow_logging: bool = False, visualize_score: Optional[str] = None,
                       postprocessors: Optional[List[str]] = None, save_csv: bool = False, save_plots_to_local: bool = False,
                       device_resident: bool = False, thresholds: bool = True):
    """See ``_log_evaluate_larex``; the host-array mode runs inside ``_hip.upload_cache()``: a split (full or PCA-reduced) that five
    postprocessors score in turn is uploaded once, not five times (the arrays are read-only here, as upstream)."""
    import contextlib

    with (contextlib.nullcontext() if device_resident else _hip.upload_cache()):
        return _log_evaluate_larex(cfg, baselines_names, ood_baselines_scores, ind_data_dict, ood_data_dict, mlflow_run_name, mlflow_logging,
                                   visualize_score, postprocessors, save_csv, save_plots_to_local, device_resident, thresholds)


def _log_evaluate_larex(cfg, baselines_names, ood_baselines_scores, ind_data_dict, ood_data_dict, mlflow_run_name, mlflow_logging,
                        visualize_score, postprocessors, save_csv, save_plots_to_local, device_resident, thresholds):
    """``log_evaluate_larex`` of the reference (signature and return value :30-221): returns ``(overall_metrics_df,
    best_postprocessors_dict, postprocessor_thresholds, ood_data_dict)``.  ``cfg`` needs ``ind_dataset``, ``ood_datasets``,
    ``n_pca_components`` (+ what the postprocessors read: ``num_classes``, ``k_neighbors``); a dict works as well.
    ``device_resident`` (additive): see the module docstring.  ``thresholds=False`` (additive) skips the refit of the best
    configurations and returns ``{}`` for the thresholds."""
    from ..inference.postprocessors import postprocessors_dict

    _refuse_mlflow(mlflow_logging)
    if save_csv or save_plots_to_local or visualize_score is not None:
        raise NotImplementedError("csv export / plots are the visualisation side of the harness (outside runia_core_amd)")
    if postprocessors is None:
        postprocessors = list(postprocessors_dict.keys())
    ood_datasets, n_pca_components = list(_cfg_get(cfg, "ood_datasets")), list(_cfg_get(cfg, "n_pca_components"))
    frames = [_rows_to_frame({})]
    if len(baselines_names) > 0:
        frames[0] = log_baselines(baselines_names, _cfg_get(cfg, "ind_dataset"), ind_data_dict, ood_baselines_scores, ood_datasets,
                                  frames[0], mlflow_logging)

    def on_device(a):
        a = np.asarray(a)
        return _hip.to_device(a, torch.float32 if a.dtype == np.float32 else torch.float64)

    from .. import config as _config

    ind_eval, ood_eval = dict(ind_data_dict), dict(ood_data_dict)
    train_rows = ind_data_dict["train latent_space_means"]
    train_on_device = device_resident or (_config.use_device_fit() and isinstance(train_rows, np.ndarray))
    if train_on_device:
        # ONE upload of the training rows (round 6), in BOTH modes: every postprocessor's fit (setup_device) and every PCA refit run
        # from HBM; before, every setup and every refit uploaded the training split again (17 x 205 MB) and formed its host-side
        # statistics with NumPy / CPU torch.  The fitted objects live inside this function only; the host-array mode still scores
        # host arrays through postprocess() and builds the ROC curves, as the reference does.
        train_rows = ind_eval["train latent_space_means"] = on_device(ind_data_dict["train latent_space_means"])
    if device_resident:
        ind_eval["valid latent_space_means"] = on_device(ind_data_dict["valid latent_space_means"])
        for name in ood_datasets:
            ood_eval[f"{name} latent_space_means"] = on_device(ood_data_dict[f"{name} latent_space_means"])
    # the complete latent vectors
    res = log_evaluate_postprocessors(ind_eval, ood_eval, ood_datasets, "", None, None, False, postprocessors, cfg,
                                      roc_curves=not device_resident)
    frames.append(res["results_df"])
    # PCA-reduced vectors
    for n_components in n_pca_components:
        pca_ind_train, pca_transformation = apply_pca_ds_split(samples=train_rows, nro_components=n_components)
        ind_dict_pca = {"train latent_space_means": pca_ind_train}
        ood_dict_pca = {}
        if device_resident:
            dp = device_pca_for(pca_transformation)
            ind_dict_pca["valid latent_space_means"] = dp.transform_device(ind_eval["valid latent_space_means"])
            for name in ood_datasets:
                ood_dict_pca[f"{name} latent_space_means"] = dp.transform_device(ood_eval[f"{name} latent_space_means"])
        else:
            ind_dict_pca["valid latent_space_means"] = apply_pca_transform(ind_data_dict["valid latent_space_means"], pca_transformation)
            for name in ood_datasets:
                ood_dict_pca[f"{name} latent_space_means"] = apply_pca_transform(ood_data_dict[f"{name} latent_space_means"],
                                                                                 pca_transformation)
        for key in ("train labels", "valid labels"):
            if key in ind_data_dict:
                ind_dict_pca[key] = ind_data_dict[key]
        for name in ood_datasets:
            if f"{name} labels" in ood_data_dict:
                ood_dict_pca[f"{name} labels"] = ood_data_dict[f"{name} labels"]
        res = log_evaluate_postprocessors(ind_dict_pca, ood_dict_pca, ood_datasets, f" PCA {n_components}", None, n_components,
                                          False, postprocessors, cfg, roc_curves=not device_resident)
        frames.append(res["results_df"])
    import pandas as pd

    overall_metrics_df = pd.concat([f for f in frames if len(f)]) if any(len(f) for f in frames) else frames[0]
    overall_metrics_df = overall_metrics_df[_COLUMNS]
    best = _get_best_postprocessors_metrics(baselines_names, overall_metrics_df, postprocessors, n_pca_components, ood_datasets)
    if not thresholds:
        return overall_metrics_df, best, {}, ood_data_dict
    postprocessor_thresholds, ood_data_dict = _get_best_post_processor_thresholds(
        postprocessors, best, cfg, ind_data_dict, ood_data_dict, ind_eval if device_resident else None,
        ood_eval if device_resident else None, train_dev=train_rows if train_on_device else None)
    return overall_metrics_df, best, postprocessor_thresholds, ood_data_dict
