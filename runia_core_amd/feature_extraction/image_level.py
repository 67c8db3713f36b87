"""Batched ``FastMCDSamplesExtractor`` (reference ``runia_core/feature_extraction/image_level.py:41-249``): one forward
pass per image through the user's backbone (PyTorch-ROCm), the hooked activation is perturbed ``mcd_nro_samples`` times by
DropBlock and reduced - here the hooked activations of a whole dataloader batch go straight into the sampler kernels, no
per-image, per-sample launches and no host round trip.

Same constructor, ``get_ls_samples(data_loader)`` and result keys as the reference.  The scoring hot path - ONE hooked
convolutional layer, ``reduction_method="fullmean"`` - is one ``runia_mc_stack_f32`` launch per batch.  The other options
of the reference's per-image loop are served by the same kernels: several hooked layers (one DropBlock layer each, results
concatenated per sample), ``reduction_method="mean"`` (``runia_mc_drop_flat_f32`` + ``runia_map_reduce_f32``),
``return_stds`` (std over the rows of the per-row stds, as ``get_std_ls_sample``), and ``layer_type="FC"``
(``torch.nn.Dropout`` on the hooked vector, device generator).

Draw order is the reference's: per image, sample after sample, and inside a sample layer after layer, every DropBlock
call with ``drop_prob > 0`` drawing ``torch.rand(1, H_i, W_i)`` on the CPU default generator.  The CPU uniform kernel
consumes the generator element by element, so one flat ``torch.rand`` per batch, cut in that order, is the same stream
(``tests/test_abi_and_host.py::test_extractor_draws_follow_the_reference_stream``).
"""
from __future__ import annotations

from typing import Dict, List, Union

import torch
from torch import Tensor

from .. import _hip
from .abstract_classes import MCSamplerModule
from .utils import Hook

__all__ = ["FastMCDSamplesExtractor"]


class FastMCDSamplesExtractor:
    def __init__(self, model: torch.nn.Module, hooked_layers: List[Hook], device: torch.device, layer_type: str,
                 reduction_method: str, return_raw_predictions: bool = False, return_stds: bool = False,
                 mcd_nro_samples: int = 1, hook_layer_output: bool = True, dropblock_probs: Union[float, List] = 0.0,
                 dropblock_sizes: Union[int, List] = 0, return_gt_labels: bool = False):
        assert layer_type in ("FC", "Conv"), "Layer type must be either 'FC' or 'Conv'"
        assert reduction_method in ("mean", "fullmean"), "Only mean and fullmean reduction methods supported"
        self.model = model
        self.hooked_layers = hooked_layers
        self.hooked_layer = hooked_layers[0]
        self.device = device
        self.layer_type = layer_type
        self.reduction_method = reduction_method
        self.return_raw_predictions = return_raw_predictions
        self.return_stds = return_stds
        self.mcd_nro_samples = mcd_nro_samples
        self.hook_layer_output = hook_layer_output
        self.return_gt_labels = return_gt_labels
        try:
            self.dropout_n_layers = len(dropblock_probs)
            self.dropblock_probs, self.dropblock_sizes = list(dropblock_probs), list(dropblock_sizes)
        except TypeError:
            self.dropout_n_layers = 1
            self.dropblock_probs, self.dropblock_sizes = [dropblock_probs], [dropblock_sizes]
        if layer_type == "FC" and self.dropout_n_layers != 1:
            raise NotImplementedError  # as the reference (image_level.py:233-235)
        if layer_type == "FC" and return_stds:
            raise NotImplementedError("return_stds is defined for convolutional layers only (the reference's FC branch "
                                      "never assigns the stds it returns)")
        # the reference reuses ONE DropBlock2D layer per hooked layer mcd_nro_samples times; as a sampler that is
        # mc_samples draws per image
        self.samplers = [
            MCSamplerModule(mc_samples=mcd_nro_samples, block_size=max(int(self.dropblock_sizes[i]), 1),
                            drop_prob=float(self.dropblock_probs[i]), layer_type="Conv").train()
            for i in range(self.dropout_n_layers)
        ]
        self.sampler = self.samplers[0]

    # ------------------------------------------------------------------------------------------------------------
    def draw_layers(self, batch: int, shapes, device, generator=None):
        """Uniform draws of every hooked layer for ``batch`` images, ``[(batch, mcd, H_i, W_i) or None]``, in the
        reference's stream order: image, sample, layer; layers with ``drop_prob == 0`` draw nothing (DropBlock2D
        returns its input before touching the generator)."""
        active = [float(p) != 0.0 for p in self.dropblock_probs]
        sizes = [h * w if a else 0 for (h, w), a in zip(shapes, active)]
        total = sum(sizes)
        if total == 0:
            return [None] * len(shapes)
        flat = torch.rand(batch * self.mcd_nro_samples * total, generator=generator).reshape(batch, self.mcd_nro_samples, total)
        out, off = [], 0
        for (h, w), n in zip(shapes, sizes):
            if n == 0:
                out.append(None)
                continue
            out.append(flat[:, :, off:off + n].reshape(batch, self.mcd_nro_samples, h, w).contiguous().to(device))
            off += n
        return out

    def _hooked_latents(self):
        latent = self.hooked_layer.output if self.hook_layer_output else self.hooked_layer.input
        if self.dropout_n_layers == 1:
            if isinstance(latent, (tuple, list)):
                latent = latent[0]
            return [latent]
        # input might be a one-element tuple containing the desired list (reference :190-198)
        if len(latent) == 1 and len(latent[0]) == self.dropout_n_layers:
            latent = latent[0]
        assert len(latent) == self.dropout_n_layers, "Cannot find a suitable latent space sample"
        return list(latent)

    def _samples_of_batch(self, latents) -> Dict[str, Tensor]:
        mcd = self.mcd_nro_samples
        if self.layer_type == "FC":
            x = _hip.to_device(latents[0], torch.float32)
            x = x.reshape(x.shape[0], -1).repeat_interleave(mcd, dim=0)  # image-major: the mcd rows of an image together
            return {"latent_space_means": torch.nn.functional.dropout(x, p=float(self.dropblock_probs[0]), training=True)}
        xs = [_hip.to_device(t, torch.float32) for t in latents]
        if self.dropout_n_layers == 1 and self.reduction_method == "fullmean" and not self.return_stds:
            return {"latent_space_means": self.sampler(xs[0])}  # the hot path: one launch, draws from the sampler's source
        batch = xs[0].shape[0]
        if self.sampler.draw_source == "counter":
            first = self.sampler._next_image
            self.sampler._next_image += batch
            rands = [_hip.CounterDraws(self.sampler.counter_seed + 7919 * i, first, getattr(self.sampler, "redraw_dead_layers", False)) if float(p) != 0.0 else None
                     for i, p in enumerate(self.dropblock_probs)]
        else:
            rands = self.draw_layers(batch, [(t.shape[2], t.shape[3]) for t in xs], xs[0].device)
        means, stds = [], []
        for x, rand, smp in zip(xs, rands, self.samplers):
            _, c, h, w = x.shape
            p = smp.drop_prob if rand is not None else 0.0
            flat = None
            if self.reduction_method == "mean" or self.return_stds:
                flat = _hip.mc_drop_flat(x, rand, mcd, p, smp.block_size)  # (B*mcd, C*H*W): every dropped map
            if self.reduction_method == "fullmean":
                means.append(_hip.mc_stack(x, rand, mcd, p, smp.block_size))
            else:
                means.append(_hip.map_reduce(flat, h, w, "mean").reshape(batch * mcd, c * h))
            if self.return_stds:
                stds.append(_hip.map_reduce(flat, h, w, "std").reshape(batch * mcd, c))
        res = {"latent_space_means": torch.cat(means, dim=1) if len(means) > 1 else means[0]}
        if self.return_stds:
            res["stds"] = torch.cat(stds, dim=1) if len(stds) > 1 else stds[0]
        return res

    def get_ls_samples(self, data_loader, **kwargs) -> Dict[str, Tensor]:
        """Fast MC-DropBlock inference over a dataloader -> ``{"latent_space_means": (N * mcd_nro_samples, D)}`` (+
        ``stds`` / ``raw_preds`` / ``gt_labels`` when requested); the samples stay on the device."""
        results: Dict[str, list] = {"latent_space_means": []}
        if self.return_raw_predictions:
            results["raw_preds"] = []
        if self.return_stds:
            results["stds"] = []
        if self.return_gt_labels:
            results["gt_labels"] = []
        with torch.no_grad():
            for image, gt_labels in data_loader:
                image = image.to(self.device)
                pred = self.model(image, **kwargs)
                # (B, C, H, W) -> (B * mcd, .): every image of the batch gets its own mcd_nro_samples draws, in the
                # reference's order (image after image, sample after sample)
                for key, value in self._samples_of_batch(self._hooked_latents()).items():
                    results[key].append(value)
                if self.return_raw_predictions:
                    results["raw_preds"].append(pred)
                if self.return_gt_labels:
                    results["gt_labels"].append(torch.as_tensor(gt_labels).reshape(-1, 1) if image.shape[0] > 1
                                                else torch.as_tensor(gt_labels).reshape(1, -1))
        out = {k: torch.cat(v, dim=0) for k, v in results.items()}
        print("Latent representation vector size: ", out["latent_space_means"].shape[1])
        return out
