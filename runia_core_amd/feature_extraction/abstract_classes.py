"""``MCSamplerModule`` with the reference's constructor, attributes and output
(reference ``runia_core/feature_extraction/abstract_classes.py:33-101``).

The n_mc ``DropBlock2D`` layers + ``fullmean`` of the reference collapse into one
kernel (``runia_mc_stack_f32``).  Parity mode keeps the reference's random stream:
one ``torch.rand(1, H, W)`` per drop layer, in ``ModuleList`` order, on the CPU
default generator (what ``dropblock==0.3.0`` does), uploaded as 16*H*W floats.
"""
from __future__ import annotations

import torch

from .. import _hip

__all__ = ["MCSamplerModule", "DropBlockSpec"]


class DropBlockSpec(torch.nn.Module):
    """Parameter-free record of one DropBlock2D layer (``drop_prob``, ``block_size``).
    Identity in eval mode, as the upstream layer."""

    def __init__(self, block_size: int, drop_prob: float):
        super().__init__()
        self.block_size = block_size
        self.drop_prob = drop_prob

    def extra_repr(self) -> str:
        return f"drop_prob={self.drop_prob}, block_size={self.block_size}"


class MCSamplerModule(torch.nn.Module):
    """Monte-Carlo DropBlock sampling of a latent map.

    Args:
        mc_samples: number of MC samples
        block_size: DropBlock size
        drop_prob: DropBlock probability
        layer_type: ``"Conv"``, ``"FC"`` or ``"RPN"``
    """

    def __init__(self, mc_samples: int, block_size: int, drop_prob: float, layer_type: str = "Conv"):
        super().__init__()
        assert layer_type in ("Conv", "FC", "RPN")
        self.layer_type = layer_type
        self.mc_samples = mc_samples
        self.block_size = block_size
        self.drop_prob = drop_prob
        self.drop_blocks = torch.nn.ModuleList(
            [DropBlockSpec(block_size=block_size, drop_prob=drop_prob) for _ in range(self.mc_samples)]
        )
        # Source of the DropBlock draws.  "cpu" (default): the reference's stream - torch.rand on the CPU default
        # generator, uploaded (parity mode).  "counter": Philox4x32-10 inside the keep-flag kernel, keyed by
        # `counter_seed`, image ids advancing with every call (throughput mode: no host RNG, no upload; statistically
        # equivalent, not the same stream) - see use_counter_draws().
        self.draw_source = "cpu"
        self.counter_seed = 0
        self._next_image = 0
        self.redraw_dead_layers = False

    def use_counter_draws(self, seed: int = 0, first_image: int = 0, redraw_dead_layers: bool = False) -> "MCSamplerModule":
        """Switch to in-kernel counter-based draws (additive API; the reference has only the CPU stream).

        ``redraw_dead_layers=True``: a drop layer whose block mask removes the whole map - ``0 * numel / 0 = NaN`` in the
        reference as well, about one image in 200 at 4x4 maps / drop_prob 0.5 / block 2 - draws again from the image's
        next counter block, so a batch never returns a NaN score.  The reference has no redraw; counter mode is a
        different random stream anyway (same statistics: bench.py reports the AUROC gap over several seeds)."""
        self.draw_source, self.counter_seed, self._next_image = "counter", int(seed), int(first_image)
        self.redraw_dead_layers = bool(redraw_dead_layers)
        return self

    def use_cpu_draws(self) -> "MCSamplerModule":
        self.draw_source = "cpu"
        return self

    def next_draws(self, batch: int, h: int, w: int, device):
        """Draws of the next ``batch`` images from the configured source: a device tensor ``(batch, n_mc, h, w)`` in
        "cpu" mode, a ``_hip.CounterDraws`` ticket in "counter" mode (the kernel makes the draws itself)."""
        if self.draw_source == "counter":
            ticket = _hip.CounterDraws(self.counter_seed, self._next_image, self.redraw_dead_layers)
            self._next_image += int(batch)
            return ticket
        return self.draw(batch, h, w, device)

    def draw(self, batch: int, h: int, w: int, device, generator=None) -> torch.Tensor:
        """Uniform draws of the drop layers: ``(batch, mc_samples, h, w)``, image-major, in the reference's
        random stream: upstream every drop layer calls ``torch.rand(1, h, w)`` on the CPU default generator, in
        ``ModuleList`` order, image after image.  The CPU uniform kernel consumes the generator element by element,
        so ONE ``torch.rand(batch * mc_samples, h, w)`` call yields the same values in the same order
        (``tests/test_abi_and_host.py::test_draw_is_the_sequential_cpu_stream``) at 1/60 of the cost of the loop."""
        return torch.rand(batch * self.mc_samples, h, w, generator=generator).reshape(batch, self.mc_samples, h, w).to(device)

    def forward(self, latent_rep: torch.Tensor, rand: torch.Tensor = None) -> torch.Tensor:
        """``(N, C, H, W)`` -> ``(N * mc_samples, C)`` (the reference is called with N = 1).
        ``rand`` optionally supplies the uniform draws (device tensor)."""
        assert latent_rep.dim() == 4, "latent representation must be (N, C, H, W)"
        x = _hip.to_device(latent_rep, torch.float32)
        n, _, h, w = x.shape
        active = self.training and self.drop_prob != 0.0
        if active and rand is None:
            rand = self.next_draws(n, h, w, x.device)
        if self.layer_type != "Conv":
            # "FC" / "RPN": no fullmean, each drop layer's output is flattened (reference :95-99)
            return _hip.mc_drop_flat(x, rand if active else None, self.mc_samples, self.drop_prob if active else 0.0,
                                     self.block_size)
        return _hip.mc_stack(x, rand if active else None, self.mc_samples, self.drop_prob if active else 0.0,
                             self.block_size)
