"""The baselines harness of the reference (``runia_core/evaluation/baselines.py:37-854``): every features / logits
postprocessor of the registry run over precomputed InD and OoD arrays, scores collected under the reference's keys
(``ind_data_dict[baseline]``, ``ood_baselines_scores[f"{ood} {baseline}"]``).

This is the caller of the a6 / a7 / a8 / f4 kernels in the evaluation flow (BASELINE config 3 is this loop restricted to
``mdist`` + ``energy`` + ``knn``): same function names, same keyword calls into ``setup`` / ``postprocess``, same order -
including ``get_labels_from_logits`` between the logits baselines and ``mdist`` / ``ddu``, which POPS the logits arrays out of
the dictionaries and derives the labels from them, as upstream - and the same ``ValueError`` for ``gen`` with more than 21
classes.  One table (``_BASELINES``) holds what differs between the twelve ``get_*_score_*`` functions of the reference:
constructor arguments, ``setup`` keywords, which split is scored.

Additive ``device_resident=True`` (``calculate_all_baselines`` and every ``get_*`` function): every split - the training
features included - is uploaded ONCE for the whole loop (``_hip.upload_cache``; upstream, and the default here, each ``setup`` and
each ``postprocess`` call moves its rows again: eight fits x 410 MB at cfg3 size, three splits per baseline); the returned scores
are host arrays either way, same bits.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Tuple, Union

import numpy as np

from .. import _hip
from ..inference.postprocessors import ASH, DDU, DICE, GEN, KNN, MSP, DICEReAct, Energy, Mahalanobis, ReAct, ViM

__all__ = ["remove_latent_features", "calculate_all_baselines", "get_labels_from_logits", "baseline_name_dict"]


def _resident(device_resident: bool):
    """``device_resident=True``: every large host array handed to a fit or a scoring call inside the context is uploaded once
    (``_hip.upload_cache``) - the training features by the first of the eight fits that read them, each valid / OoD split by the
    first baseline that scores it."""
    return _hip.upload_cache() if device_resident else contextlib.nullcontext()


def _FEATS_FC(ind, fc):  # setup keywords of the linear-layer family
    return dict(ind_train_data=ind["train features"], valid_feats=ind["valid features"], final_linear_layer_params=fc)


def _FEATS_LABELS(ind, fc):  # ... of the class-conditional fits
    return dict(ind_train_data=ind["train features"], train_labels=ind["train labels"], valid_feats=ind["valid features"])


# name -> (message, constructor(ind, params), setup keywords(ind, fc), input kind, needs the logits at postprocess)
_BASELINES = {
    "dice": ("Calculating DICE score",
             lambda ind, p: DICE(flip_sign=False, dice_percentile=p["percentile"], num_classes=ind["train logits"].shape[1]),
             _FEATS_FC,
             "features", False),
    "react": ("Calculating ReAct score",
              lambda ind, p: ReAct(flip_sign=False, react_percentile=p["percentile"]),
              _FEATS_FC,
              "features", False),
    "dice_react": ("Calculating DICE+ReAct score",
                   lambda ind, p: DICEReAct(flip_sign=False, dice_percentile=p["dice_percentile"], react_percentile=p["react_percentile"],
                                            num_classes=ind["train logits"].shape[1]),
                   _FEATS_FC,
                   "features", False),
    "ash": ("Calculating ash score",
            lambda ind, p: ASH(flip_sign=False, ash_percentile=p["ash_percentile"]),
            _FEATS_FC,
            "features", False),
    "gen": ("Calculating GEN score",
            lambda ind, p: GEN(flip_sign=False, gamma=p["gamma"], num_classes=p["gen_m"]),
            lambda ind, fc: dict(ind_train_data=ind["train logits"]),
            "logits", False),
    "vim": ("Calculating ViM score",
            lambda ind, p: ViM(flip_sign=False),
            lambda ind, fc: dict(ind_train_data=ind["train features"], train_logits=ind["train logits"], valid_feats=ind["valid features"],
                                 valid_logits=ind["valid logits"], final_linear_layer_params=fc),
            "features", True),
    "msp": ("Calculating msp score",
            lambda ind, p: MSP(flip_sign=False),
            lambda ind, fc: dict(ind_train_data=ind["train logits"]),
            "logits", False),
    "energy": ("Calculating energy score",
               lambda ind, p: Energy(flip_sign=False),
               lambda ind, fc: dict(ind_train_data=ind["train logits"]),
               "logits", False),
    "mdist": ("Calculating mahalanobis score",
              lambda ind, p: Mahalanobis(flip_sign=False, num_classes=p["num_classes"]),
              _FEATS_LABELS,
              "features", False),
    "knn": ("Calculating knn score",
            lambda ind, p: KNN(flip_sign=False, k_neighbors=p["k_neighbors"]),
            lambda ind, fc: dict(ind_train_data=ind["train features"], valid_feats=ind["valid features"]),
            "features", False),
    "ddu": ("Calculating ddu score",
            lambda ind, p: DDU(flip_sign=False, num_classes=p["num_classes"]),
            _FEATS_LABELS,
            "features", False),
}


def _run(name: str, fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, params, device_resident=False):
    """One baseline: fit on the training split, score the InD valid split into ``ind_data_dict[name]`` and every OoD set into
    ``ood_baselines_dict[f"{ood} {name}"]`` (keyword calls, as upstream)."""
    message, make, setup_kwargs, kind, with_logits = _BASELINES[name]
    print(message)
    with _resident(device_resident):
        pp = make(ind_data_dict, params)
        pp.setup(**setup_kwargs(ind_data_dict, fc_params))
        extra = (lambda d, prefix: {"logits": d[f"{prefix} logits"]}) if with_logits else (lambda d, prefix: {})  # (ViM)
        ind_data_dict[name] = pp.postprocess(test_data=ind_data_dict[f"valid {kind}"], **extra(ind_data_dict, "valid"))
        for ood_name in ood_names:
            ood_baselines_dict[f"{ood_name} {name}"] = pp.postprocess(test_data=ood_data_dict[f"{ood_name} {kind}"],
                                                                     **extra(ood_data_dict, ood_name))
    return ind_data_dict, ood_baselines_dict


# ---- the reference's per-baseline functions (signatures of baselines.py:37-611; `device_resident` is additive) ----
def get_dice_score_from_features(fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, percentile,
                                 device_resident=False):
    return _run("dice", fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"percentile": percentile},
                device_resident)


def get_react_score_from_features(fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, percentile,
                                  device_resident=False):
    return _run("react", fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"percentile": percentile},
                device_resident)


def get_dice_react_score_from_features(fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, dice_percentile,
                                       react_percentile, device_resident=False):
    return _run("dice_react", fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict,
                {"dice_percentile": dice_percentile, "react_percentile": react_percentile}, device_resident)


def get_ash_score_from_features(fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, ash_percentile,
                                device_resident=False):
    return _run("ash", fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"ash_percentile": ash_percentile}, device_resident)


def get_gen_score_from_logits(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, gamma, gen_m, device_resident=False):
    return _run("gen", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"gamma": gamma, "gen_m": gen_m}, device_resident)


def calculate_vim_score(fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, device_resident=False):
    return _run("vim", fc_params, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {}, device_resident)


def get_msp_score_from_logits(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, device_resident=False):
    return _run("msp", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {}, device_resident)


def get_raw_score_from_logits(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, device_resident=False):
    """``raw`` (reference :395-425): the maximum softmax probability with no postprocessor object around it (no threshold)."""
    print("Calculating raw score")
    pp = MSP(flip_sign=False)
    pp._setup_flag = True  # (scores only: upstream calls scipy's softmax directly)
    with _resident(device_resident):
        ind_data_dict["raw"] = pp.postprocess(test_data=ind_data_dict["valid logits"])
        for ood_name in ood_names:
            ood_baselines_dict[f"{ood_name} raw"] = pp.postprocess(test_data=ood_data_dict[f"{ood_name} logits"])
    return ind_data_dict, ood_baselines_dict


def get_energy_score_from_logits(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, device_resident=False):
    return _run("energy", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {}, device_resident)


def get_mahalanobis_score_from_features(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, num_classes,
                                        device_resident=False):
    return _run("mdist", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"num_classes": num_classes}, device_resident)


def get_knn_score_from_features(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, k_neighbors, device_resident=False):
    return _run("knn", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"k_neighbors": k_neighbors}, device_resident)


def get_ddu_score_from_features(ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, num_classes, device_resident=False):
    return _run("ddu", None, ind_data_dict, ood_data_dict, ood_names, ood_baselines_dict, {"num_classes": num_classes}, device_resident)


def _labels_of(logits):
    """argmax labels of a logits array; a background column (21 or 11 columns: detectors) is dropped first, as upstream."""
    if logits.shape[1] in (21, 11):
        logits = logits[:, :-1]
    return np.argmax(logits, axis=-1)


def get_labels_from_logits(id_data: Dict[str, np.ndarray], ood_data: Dict[str, np.ndarray],
                           ood_names: List[str]) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Predicted labels from the logits entries, which are REMOVED from the dictionaries (reference :614-683): arrays ->
    argmax; a pair of empty lists -> empty label arrays; anything else raises ``NotImplementedError``."""
    empty = lambda: np.asarray([], dtype=int)  # noqa: E731
    is_empty_list = lambda v: isinstance(v, list) and len(v) == 0  # noqa: E731
    train, valid = id_data.pop("train logits", None), id_data.pop("valid logits", None)
    if isinstance(train, np.ndarray) or isinstance(valid, np.ndarray):
        id_data["train labels"] = _labels_of(train) if train is not None else empty()
        id_data["valid labels"] = _labels_of(valid) if valid is not None else empty()
    elif is_empty_list(train) and is_empty_list(valid):
        id_data["train labels"], id_data["valid labels"] = empty(), empty()
    else:
        raise NotImplementedError
    for ood_name in ood_names:
        logits = ood_data.pop(f"{ood_name} logits", None)
        if isinstance(logits, np.ndarray):
            ood_data[f"{ood_name} labels"] = _labels_of(logits)
        elif is_empty_list(logits):
            ood_data[f"{ood_name} labels"] = empty()
        else:
            raise NotImplementedError
    return id_data, ood_data


def remove_latent_features(id_data: Dict[str, np.ndarray], ood_data: Dict[str, np.ndarray], ood_names: List[str]):
    """Drop the feature arrays (missing keys are ignored; reference :686-710)."""
    for key in ("train features", "valid features"):
        id_data.pop(key, None)
    for ood_name in ood_names:
        ood_data.pop(f"{ood_name} features", None)
    return id_data, ood_data


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def calculate_all_baselines(baselines_names: List[str], ind_data_dict: Dict[str, np.ndarray], ood_data_dict: Dict[str,
                            np.ndarray],
                            fc_params: Union[Dict[str, np.ndarray], None], cfg, num_classes: int, device_resident: bool = False):
    """``calculate_all_baselines`` of the reference (:713-854): returns ``(ind_data_dict, ood_data_dict,
    ood_baselines_scores_dict)``.  ``cfg`` needs ``ood_datasets`` and, for the baselines that read them, ``k_neighbors``,
    ``ash_percentile``, ``gen_gamma``, ``react_percentile``, ``dice_percentile`` (an ``omegaconf.DictConfig``, any object with
    those attributes, or a dict).  ``device_resident`` (additive): see the module docstring."""
    if num_classes > 21 and "gen" in baselines_names:
        raise ValueError("Implementation of gen baseline does not yet support num_classes greater than 21. "
                         "Otherwise implement M parameter specification")
    ood_names = list(_cfg_get(cfg, "ood_datasets"))
    scores: Dict[str, np.ndarray] = {}
    common = dict(ind_data_dict=ind_data_dict, ood_data_dict=ood_data_dict, ood_names=ood_names, ood_baselines_dict=scores,
                  device_resident=device_resident)
    want = lambda name: name in baselines_names  # noqa: E731
    if want("vim"):
        calculate_vim_score(fc_params=fc_params, **common)
    if want("msp"):
        get_msp_score_from_logits(**common)
    if want("raw"):
        get_raw_score_from_logits(**common)
    if want("knn"):
        get_knn_score_from_features(k_neighbors=_cfg_get(cfg, "k_neighbors"), **common)
    if want("energy"):
        get_energy_score_from_logits(**common)
    if want("ash"):
        get_ash_score_from_features(fc_params=fc_params, ash_percentile=_cfg_get(cfg, "ash_percentile"), **common)
    if want("gen"):
        get_gen_score_from_logits(gamma=_cfg_get(cfg, "gen_gamma"), gen_m=num_classes, **common)
    if want("react"):
        get_react_score_from_features(fc_params=fc_params, percentile=_cfg_get(cfg, "react_percentile"), **common)
    if want("dice"):
        get_dice_score_from_features(fc_params=fc_params, percentile=_cfg_get(cfg, "dice_percentile"), **common)
    if want("dice_react"):
        get_dice_react_score_from_features(fc_params=fc_params, dice_percentile=_cfg_get(cfg, "dice_percentile"),
                                           react_percentile=_cfg_get(cfg, "react_percentile"), **common)
    # the labels of mdist / ddu are the argmax of the logits, which leave the dictionaries here (as upstream)
    ind_data_dict, ood_data_dict = get_labels_from_logits(id_data=ind_data_dict, ood_data=ood_data_dict, ood_names=ood_names)
    if want("mdist"):
        get_mahalanobis_score_from_features(num_classes=num_classes, **common)
    if want("ddu"):
        get_ddu_score_from_features(num_classes=num_classes, **common)
    return ind_data_dict, ood_data_dict, scores


def _plot_entry(title: str, axis: str, name: str) -> Dict[str, str]:
    return {"plot_title": title, "x_axis": axis, "plot_name": name}


# titles / axis labels / file stems of the score-distribution plots, keyed by baseline (reference :857-928; data only)
baseline_name_dict = {
    "pred_h": _plot_entry("Predictive H distribution", "Predictive H score", "pred_h"),
    "mi": _plot_entry("Predictive MI distribution", "Predictive MI score", "pred_mi"),
    "msp": _plot_entry("Predictive MSP distribution", "Predictive MSP score", "pred_msp"),
    "energy": _plot_entry("Predictive energy score distribution", "Predictive energy score", "pred_energy"),
    "mdist": _plot_entry("Mahalanobis Distance distribution", "Mahalanobis Distance score", "pred_mdist"),
    "knn": _plot_entry("kNN distance distribution", "kNN Distance score", "pred_knn"),
    "ash": _plot_entry("ASH score distribution", "ASH score", "ash_score"),
    "dice": _plot_entry("DICE score distribution", "DICE score", "dice_score"),
    "react": _plot_entry("ReAct score distribution", "ReAct score", "react_score"),
    "dice_react": _plot_entry("DICE + ReAct score distribution", "DICE + ReAct score", "dice_react_score"),
    "vim": _plot_entry("ViM score distribution", "ViM score", "vim_score"),
    "gen": _plot_entry("GEN score distribution", "GEN score", "gen_score"),
    "ddu": _plot_entry("DDU score distribution", "DDU score", "ddu_score"),
    "raw": _plot_entry("Raw predictions", "Raw predictions", "raw_predictions"),
}
