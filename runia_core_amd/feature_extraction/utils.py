"""Model-side helpers kept as plain torch (reference ``runia_core/feature_extraction/utils.py``:
``Hook`` :27-56, ``apply_dropout`` :59-67, ``get_mean_or_fullmean_ls_sample`` :70-92, ``get_variance_ls_sample`` :95-108,
``get_std_ls_sample`` :111-124).  They act on whatever device the model's tensors live on; the batched extractors reach
the same reductions through the HIP kernels (``runia_map_reduce_f32``, the fused sampler)."""
from __future__ import annotations

import torch
from torch import Tensor

__all__ = ["Hook", "apply_dropout", "get_mean_or_fullmean_ls_sample", "get_variance_ls_sample", "get_std_ls_sample"]


class Hook:
    """Catches the input and output of a layer during the forward (or backward) pass."""

    def __init__(self, module: torch.nn.Module, backward: bool = False):
        self.input = None
        self.output = None
        if not backward:
            self.hook = module.register_forward_hook(self.hook_fn)
        else:
            self.hook = module.register_backward_hook(self.hook_fn)

    def hook_fn(self, module, inputs, outputs):
        self.input = inputs
        self.output = outputs

    def close(self):
        self.hook.remove()


def get_mean_or_fullmean_ls_sample(latent_sample: Tensor, method: str = "fullmean") -> Tensor:
    """``"mean"``: average over W; ``"fullmean"``: average over W then H (a C-sized vector)."""
    assert method in ("mean", "fullmean")
    latent_sample = torch.mean(latent_sample, dim=3, keepdim=True)
    if method == "fullmean":
        latent_sample = torch.mean(latent_sample, dim=2, keepdim=True)
    return torch.squeeze(latent_sample)


def apply_dropout(m):
    """``model.apply(apply_dropout)``: puts Dropout / DropBlock layers (anything whose class is named like the
    third-party ``DropBlock2D`` the reference checks for) into training mode so that they sample at inference time."""
    if isinstance(m, torch.nn.Dropout) or type(m).__name__ in ("DropBlock2D", "DropBlock2DTable"):
        m.train()


def get_variance_ls_sample(latent_sample: Tensor) -> Tensor:
    """Variance over W, then the variance of those over H (upstream's two-step reduction), squeezed."""
    latent_sample = torch.var(latent_sample, dim=3, keepdim=True)
    latent_sample = torch.var(latent_sample, dim=2, keepdim=True)
    return torch.squeeze(latent_sample)


def get_std_ls_sample(latent_sample: Tensor) -> Tensor:
    """Standard deviation over W, then the standard deviation of those over H, squeezed."""
    latent_sample = torch.std(latent_sample, dim=3, keepdim=True)
    latent_sample = torch.std(latent_sample, dim=2, keepdim=True)
    return torch.squeeze(latent_sample)
