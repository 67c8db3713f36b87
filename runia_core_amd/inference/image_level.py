"""Per-image inference modules with the reference's constructor and ``get_score``
contract (``runia_core/inference/image_level.py``: ``LaRExInference`` :31-198,
``LaRDInference`` :201-313).  The backbone forward stays PyTorch-ROCm; everything
after the hooked activation runs device-resident through :class:`LaREMPipeline`
and only the final ``(1,)`` score is copied to the host.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .. import _hip
from ..dimensionality_reduction import device_pca_for
from ..feature_extraction.abstract_classes import MCSamplerModule
from ..feature_extraction.utils import Hook, get_mean_or_fullmean_ls_sample
from .abstract_classes import InferenceModule, Postprocessor, ProbabilisticInferenceModule, record_time
from .pipeline import LaREMPipeline

__all__ = ["LaRExInference", "LaRDInference"]


def _score_rows_device(postprocessor, rows: torch.Tensor) -> np.ndarray:
    if hasattr(postprocessor, "postprocess_device"):
        return _hip.to_host(postprocessor.postprocess_device(rows))
    return postprocessor.postprocess(_hip.to_host(rows))


class LaRExInference(ProbabilisticInferenceModule):
    """LaREx (LaRED / LaREM) inference: MC DropBlock sampling of the hooked latent map,
    per-dimension entropy, optional PCA, postprocessor score.

    Args:
        model: trained model
        postprocessor: fitted LaRED (``KDELatentSpace``) or LaREM (``MDLatentSpace``) postprocessor
        drop_block_prob, drop_block_size: DropBlock parameters
        mcd_samples_nro: number of MC samples
        mcd_sampler: sampler class (``MCSamplerModule``)
        pca_transform: optional fitted PCA
        layer_type: ``"Conv"`` or ``"FC"``
    """

    def __init__(self, model, postprocessor, drop_block_prob: float, drop_block_size: int, mcd_samples_nro: int,
                 mcd_sampler: MCSamplerModule, pca_transform=None, layer_type="Conv"):
        super().__init__(model=model, postprocessor=postprocessor, drop_block_prob=drop_block_prob,
                         drop_block_size=drop_block_size, mcd_samples_nro=mcd_samples_nro)
        self.layer_type = layer_type
        self.pca_transform = pca_transform
        self.mc_sampler = mcd_sampler(mc_samples=self.mcd_samples_nro, layer_type=layer_type,
                                      drop_prob=self.drop_block_prob, block_size=self.drop_block_size)
        self.mc_sampler.to(self.device)
        self.mc_sampler.train()
        self._pipeline = None
        self._pipeline_key = None

    def _pipe(self) -> LaREMPipeline:
        # the pipeline is rebuilt when the postprocessor / PCA objects (or the PCA's fitted arrays) are replaced
        pca = self.pca_transform
        key = (id(self.postprocessor), id(pca),
               None if pca is None or not hasattr(pca, "components_") else _hip.array_fingerprint(pca.components_),
               self.mcd_samples_nro, self.drop_block_prob, self.drop_block_size)
        if self._pipeline is None or self._pipeline_key != key:
            self._pipeline = LaREMPipeline(self.postprocessor, pca, self.mcd_samples_nro, self.drop_block_prob,
                                           self.drop_block_size)
            self._pipeline_key = key
        return self._pipeline

    def get_score(self, input_image, layer_hook):
        """LaREx score of one image -> ``(model_output, ndarray (1,))``."""
        with torch.no_grad():
            try:
                input_image = input_image.to(self.device)
            except AttributeError:  # pragma: no cover
                pass
            output = self.model(input_image)
            latent_rep = layer_hook.output
        return output, self.get_scores_from_latents(latent_rep)

    def get_scores_from_latents(self, latents: torch.Tensor, rand: torch.Tensor = None, to_host: bool = True):
        """Additive batched entry point: hooked activations ``(N, C, H, W)`` -> ``(N,)`` scores.
        ``rand`` ``(N, n_mc, H, W)`` supplies the DropBlock draws (default: the sampler's source - the reference's
        CPU-generator stream, or in-kernel counter draws after ``self.mc_sampler.use_counter_draws(seed)``).
        ``to_host=False`` returns the device tensor (stream-ordered, no synchronisation)."""
        x = _hip.to_device(latents, torch.float32)
        active = self.mc_sampler.training and self.drop_block_prob != 0.0
        if rand is None and active:
            rand = self.mc_sampler.next_draws(x.shape[0], x.shape[2], x.shape[3], x.device)
        pipe = self._pipe()
        if pipe._md_state() is not None and self.layer_type == "Conv":
            # LaREM: sampler + entropy and PCA + score as two fused launches (same arithmetic as the stages below)
            s = pipe.score_latents(x, rand if active else None)
            return _hip.to_host(s) if to_host else s
        h = pipe.entropy(self.mc_sampler(x, rand=rand))
        if self.pca_transform:
            h = device_pca_for(self.pca_transform).transform_device(h)
        if not to_host and hasattr(self.postprocessor, "postprocess_device"):
            return self.postprocessor.postprocess_device(h)
        return _score_rows_device(self.postprocessor, h)

    @record_time
    def test_time_inference(self, input_image, layer_hook):  # pragma: no cover
        return self.get_score(input_image, layer_hook)

    @record_time
    def get_layer_mc_samples(self, input_image, layer_hook):  # pragma: no cover
        with torch.no_grad():
            input_image = input_image.to(self.device)
            _ = self.model(input_image)
            latent_rep = layer_hook.output
        return self.mc_sampler(latent_rep)

    @record_time
    def get_mc_samples_full_inference(self, input_image, layer_hook):  # pragma: no cover
        mc_samples = []
        with torch.no_grad():
            for _ in range(self.mcd_samples_nro):
                try:
                    input_image = input_image.to(self.device)
                except AttributeError:
                    pass
                _ = self.model(input_image)
                mc_samples.append(layer_hook.output)
            return _hip.to_host(torch.cat(mc_samples))

    @record_time
    def get_score_full_inference(self, input_image, layer_hook):
        raise NotImplementedError


class LaRDInference(InferenceModule):
    """LaRD inference: reduced latent representation (no MC sampling, no entropy), optional PCA,
    KDE or MD postprocessor score."""

    def __init__(self, model, postprocessor: Postprocessor, pca_transform=None, layer_type="Conv") -> None:
        super().__init__(model, postprocessor)
        self.layer_type = layer_type
        if self.layer_type == "Conv":
            self._reducer = self._reduce_conv_representation
        elif self.layer_type == "FC":
            self._reducer = self._reduce_fc_representation
        else:
            pass
        self.pca_transform = pca_transform

    def get_score(self, input_image: torch.Tensor, layer_hook: Hook) -> Tuple[torch.Tensor, float]:
        with torch.no_grad():
            try:
                input_image = input_image.to(self.device)
            except AttributeError:
                pass
            output = self.model(input_image)
            latent_rep = layer_hook.output
        rows = self._reducer(latent_rep)  # (1, C) device tensor
        if self.pca_transform:
            rows = device_pca_for(self.pca_transform).transform_device(_hip.to_device(rows, torch.float32))
        else:
            rows = _hip.to_device(rows, torch.float32)
        return output, _score_rows_device(self.postprocessor, rows)

    @record_time
    def test_time_inference(self, input_image: torch.Tensor, layer_hook: Hook):  # pragma: no cover
        return self.get_score(input_image, layer_hook)

    @staticmethod
    def _reduce_conv_representation(representation: torch.Tensor) -> torch.Tensor:
        return get_mean_or_fullmean_ls_sample(representation, "fullmean").reshape(1, -1)

    @staticmethod
    def _reduce_fc_representation(representation: torch.Tensor) -> torch.Tensor:
        if representation.ndim > 1:
            return torch.mean(representation, dim=1).reshape(1, -1)
        return representation.reshape(1, -1)
