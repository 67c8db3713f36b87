"""Model-side helpers kept as plain torch (reference ``runia_core/feature_extraction/utils.py``:
``Hook`` :27-56, ``get_mean_or_fullmean_ls_sample`` :70-92)."""
from __future__ import annotations

import torch
from torch import Tensor

__all__ = ["Hook", "get_mean_or_fullmean_ls_sample"]


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
