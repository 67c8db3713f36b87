"""Drop-in boundary types of the scoring path.

Mirrors the public contract of the reference's
``runia_core/inference/abstract_classes.py`` (Postprocessor :58-130,
OodPostprocessor :133-211, InferenceModule :217-279,
ProbabilisticInferenceModule :282-321, ObjectDetectionInference :324-370,
get_baselines_thresholds :373-405, get_method_threshold :408-424,
record_time :35-52): same names, argument meaning, attributes and error text, so
that the reference's evaluation harness and tests can use these classes
unchanged.  The numerical work behind ``postprocess`` lives in the HIP library.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from time import monotonic
from typing import Dict, List, Union

import numpy as np
import torch
from numpy import ndarray

__all__ = [
    "record_time",
    "Postprocessor",
    "OodPostprocessor",
    "InferenceModule",
    "ProbabilisticInferenceModule",
    "ObjectDetectionInference",
    "get_baselines_thresholds",
    "get_method_threshold",
]


def record_time(function):
    """Decorator: call ``function`` and return ``(result, elapsed_seconds)`` (monotonic clock)."""

    def wrapper(*args, **kwargs):
        t0 = monotonic()
        result = function(*args, **kwargs)
        return result, monotonic() - t0

    return wrapper


class Postprocessor(ABC):
    """Base of every post-hoc OOD scorer: ``setup`` fits on in-distribution data,
    ``postprocess`` (= ``__call__``) scores new rows.  ``cfg`` is accepted and ignored
    here, exactly as in the reference (subclasses read their own keys)."""

    # attributes that only cache device copies of the fitted state: left out of pickles / broadcasts (rebuilt on first use)
    _device_cache_attrs = ("_dev", "_state", "_wd", "_bd")

    def __init__(self, cfg=None):
        self._setup_flag = False

    def __getstate__(self):
        state = dict(self.__dict__)
        for name in self._device_cache_attrs:
            if name in state:
                state[name] = None
        return state

    @abstractmethod
    def setup(self, ind_train_data: ndarray, **kwargs) -> None:
        raise NotImplementedError

    @abstractmethod
    def postprocess(self, test_data: ndarray, **kwargs) -> ndarray:
        raise NotImplementedError

    def __call__(self, test_data: ndarray, **kwargs) -> ndarray:
        return self.postprocess(test_data, **kwargs)


class OodPostprocessor(Postprocessor):
    """Logits / features family: optional sign flip and a z-score threshold."""

    def __init__(self, flip_sign: bool, cfg=None):
        super().__init__(cfg)
        self.flip_sign = flip_sign
        self.threshold: Union[float, None] = None

    def flip_sign_fn(self, scores: Union[Dict[str, ndarray], ndarray]) -> Union[Dict[str, ndarray], ndarray]:
        """Multiply by -1 when ``flip_sign`` is set.  A dict is updated in place, an
        ndarray yields a new array, anything else is a ``ValueError``."""
        if self.flip_sign:
            if isinstance(scores, dict):
                for method, values in scores.items():
                    scores[method] = values * -1
            elif isinstance(scores, ndarray):
                scores = scores * -1
            else:
                raise ValueError("scores must be a dict or ndarray")
        return scores

    def set_threshold(self, ind_test_scores: ndarray, z_score_percentile: float = 1.645) -> None:
        self.threshold = get_method_threshold(scores=ind_test_scores, z_score_percentile=z_score_percentile)
        self._setup_flag = True

    def setup(self, ind_train_data: ndarray, **kwargs) -> None:
        raise NotImplementedError

    def postprocess(self, test_data: ndarray, **kwargs) -> ndarray:
        raise NotImplementedError


class InferenceModule:
    """Holds a model and a fitted postprocessor; ``device`` is cuda when available."""

    def __init__(self, model, postprocessor):
        self.model = model
        self.postprocessor = postprocessor
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        try:
            self.model.to(self.device)
        except AttributeError:
            pass

    def get_score(self, input_image, *args, **kwargs):
        raise NotImplementedError


class ProbabilisticInferenceModule(InferenceModule):
    """Adds the MC-dropout parameters (DropBlock probability / size, number of samples)."""

    def __init__(self, model, postprocessor, drop_block_prob: float, drop_block_size: int, mcd_samples_nro: int):
        super().__init__(model, postprocessor)
        self.drop_block_prob = drop_block_prob
        self.drop_block_size = drop_block_size
        self.mcd_samples_nro = mcd_samples_nro


class ObjectDetectionInference(InferenceModule):
    """Attribute carrier for detector-based inference (detector glue itself is out of scope)."""

    def __init__(self, model, postprocessor, architecture: str, hooked_layers: List, pca_transform=None,
                 rcnn_extraction_type: str = None):
        super().__init__(model=model, postprocessor=postprocessor)
        self.architecture = architecture
        self.rcnn_extraction_type = rcnn_extraction_type
        self.hooked_layers = hooked_layers
        self.pca_transform = pca_transform


def get_method_threshold(scores: np.ndarray, z_score_percentile: float):
    """``mean - z * std`` (population std); higher score = in-distribution."""
    mean = float(np.mean(scores))
    std = float(np.std(scores))
    return mean - (z_score_percentile * std)


def get_baselines_thresholds(baselines_names: List[str], baselines_scores_dict: Dict[str, np.ndarray],
                             z_score_percentile: float = 1.645) -> Dict[str, float]:
    """Threshold per baseline; ``"raw"`` (no postprocessing) gets 0.0."""
    thresholds = {}
    for name in baselines_names:
        if name == "raw":
            thresholds[name] = 0.0
        else:
            thresholds[name] = get_method_threshold(scores=baselines_scores_dict[name],
                                                    z_score_percentile=z_score_percentile)
    return thresholds
