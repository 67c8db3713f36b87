"""Per-ROI sampling for object-level inference (BASELINE config 4), device-resident: the tail of the reference's
``BoxFeaturesExtractor`` that IS the scoring hot path (``runia_core/feature_extraction/object_level.py``:
``_reduce_features_to_rois`` :254-309, ``_dropblock_rois_get_entropy`` :312-367) with the same signatures.  The detector
glue around it (``BoxFeaturesExtractor`` itself: hooks, NMS, architecture switches) is out of scope.

``roi_align`` -> per-detection ``MCSamplerModule`` -> ``get_dl_h_z(.)[1]`` run as HIP kernels on the hooked feature maps
(``runia_roi_align_f32`` -> ``runia_mc_entropy_f32`` / ``runia_mc_stack_f32`` + ``runia_kl_entropy_per_dim_f32``): nothing
goes through the host between the backbone and the entropies.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor

from .. import _hip
from ..evaluation.entropy import MIN_DIST, neighbors_for
from .abstract_classes import MCSamplerModule

__all__ = ["roi_align", "_reduce_features_to_rois", "_dropblock_rois_get_entropy"]


def roi_align(input: Tensor, boxes, output_size, spatial_scale: float = 1.0, sampling_ratio: int = -1,
              aligned: bool = False) -> Tensor:
    """``torchvision.ops.roi_align`` for the call forms the reference uses: ``boxes`` is a ``Tensor[K, 4]`` (xyxy) or a
    list with one such tensor per image of the batch."""
    x = _hip.to_device(input, torch.float32)
    batch_idx = None
    if isinstance(boxes, (list, tuple)):
        if len(boxes) != x.shape[0]:
            raise ValueError("roi_align: one box tensor per image of the batch is expected")
        batch_idx = torch.cat([torch.full((len(b),), i, dtype=torch.int32) for i, b in enumerate(boxes)])
        boxes = torch.cat([torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4) for b in boxes])
        if x.shape[0] == 1:
            batch_idx = None
    return _hip.roi_align(x, torch.as_tensor(boxes, dtype=torch.float32), output_size, spatial_scale, sampling_ratio,
                          aligned, batch_idx)


def _rois(latent_mcd_sample, output_sizes, boxes, img_shape, sampling_ratio, n_hooked_reps):
    return [
        roi_align(latent_mcd_sample[i], [boxes], output_size=output_sizes[i],
                  spatial_scale=latent_mcd_sample[i].shape[3] / img_shape[1], sampling_ratio=sampling_ratio, aligned=True)
        for i in range(n_hooked_reps)
    ]


def _reduce_features_to_rois(latent_mcd_sample: List[Tensor], output_sizes: Tuple[int], boxes: Tensor,
                             img_shape: Tuple[int, ...], sampling_ratio: int, n_hooked_reps: int,
                             n_detected_objects: int, return_stds: bool = False) -> Tuple[List[Tensor], List[Tensor]]:
    """Means (and optionally standard deviations) of the ROI-aligned activations per detected object, one ``(1, C_total)``
    tensor per object (reference :254-309)."""
    rois = _rois(latent_mcd_sample, output_sizes, boxes, img_shape, sampling_ratio, n_hooked_reps)
    means = torch.cat([r.mean(dim=(2, 3)) for r in rois], dim=1)
    n_objects_means = [means[i].reshape(1, -1) for i in range(n_detected_objects)]
    n_objects_stds = []
    if return_stds:
        stds = torch.cat([r.std(dim=(2, 3)) for r in rois], dim=1)
        n_objects_stds = [stds[i].reshape(1, -1) for i in range(n_detected_objects)]
    return n_objects_means, n_objects_stds


def _dropblock_rois_get_entropy(latent_mcd_sample: List[Tensor], output_sizes: Tuple[int], boxes: Tensor,
                                img_shape: Tuple[int, ...], sampling_ratio: int, n_hooked_reps: int, n_mcd_steps: int,
                                mc_sampler: MCSamplerModule, rand: Tensor = None) -> Tensor:
    """Entropy of the MC-DropBlock means of every detection's ROI-aligned activations: ``(K, C_total)`` (reference
    :312-367; returned as a float32 host ``Tensor`` like ``Tensor(entropies)`` there).  ``rand`` (additive) supplies the
    DropBlock draws ``(K, n_mcd_steps, PH, PW)``; default: the sampler's source, detection after detection, as upstream."""
    assert n_mcd_steps == mc_sampler.mc_samples, "n_mcd_steps must equal the sampler's mc_samples"
    active = mc_sampler.training and mc_sampler.drop_prob != 0.0
    kk = neighbors_for(n_mcd_steps)
    drop = mc_sampler.drop_prob if active else 0.0
    sizes = [(s, s) if isinstance(s, int) else tuple(s) for s in output_sizes[:n_hooked_reps]]
    k = int(torch.as_tensor(boxes).shape[0])
    redraw = isinstance(rand, _hip.CounterDraws) and rand.redraw_dead_layers  # (lives in the keep-flag kernel of the two-call path)
    if (mc_sampler.layer_type == "Conv" and k > 0 and len(set(sizes)) == 1 and not redraw
            and _hip.roi_mc_entropy_supported(sizes[0][0], sizes[0][1], n_mcd_steps, kk, sampling_ratio)
            and all(m[0].numel() * 4 < _hip.ROI_FUSED_MAX_IMAGE_BYTES for m in latent_mcd_sample[:n_hooked_reps])):
        # one pass per hooked layer from the feature map to the entropies: roi_align is folded into the sampler's load
        # (channels-last copy of the map, per-ROI sample table), the (K, C, PH, PW) tensor is never written
        ph, pw = sizes[0]
        dev = _hip.require_gpu()
        if active and rand is None:
            rand = mc_sampler.next_draws(k, ph, pw, dev)
        hs = []
        for i in range(n_hooked_reps):
            x = _hip.to_device(latent_mcd_sample[i], torch.float32)
            hs.append(_hip.roi_mc_entropy(_hip.nchw_to_nhwc(x), torch.as_tensor(boxes, dtype=torch.float32), (ph, pw),
                                          x.shape[3] / img_shape[1], sampling_ratio, True, rand if active else None,
                                          n_mcd_steps, drop, mc_sampler.block_size, kk, MIN_DIST))
        h = torch.cat(hs, dim=1) if len(hs) > 1 else hs[0]
        return h.to(torch.float32).cpu()
    rois = _rois(latent_mcd_sample, output_sizes, boxes, img_shape, sampling_ratio, n_hooked_reps)
    rois = torch.cat(rois, dim=1) if len(rois) > 1 else rois[0]
    k, _, ph, pw = rois.shape
    if active and rand is None:
        rand = mc_sampler.next_draws(k, ph, pw, rois.device)
    if mc_sampler.layer_type == "Conv" and _hip.mc_entropy_supported(ph, pw, n_mcd_steps, kk):
        h = _hip.mc_entropy(rois, rand if active else None, n_mcd_steps, drop, mc_sampler.block_size, kk, MIN_DIST)
    else:
        z = mc_sampler(rois, rand=rand)
        h = _hip.kl_entropy_per_dim(z, n_mcd_steps, kk, MIN_DIST)
    return h.to(torch.float32).cpu()
