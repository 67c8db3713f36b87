"""Row sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference has no distributed code; every stage of the scoring path is
row-local given the fitted state, so the test rows are cut into contiguous blocks of
``ceil(N / world)`` rows, the fitted state is replicated (broadcast once after ``setup``),
each rank scores its block with the HIP kernels, and ONE ``all_gather`` of the padded score
shards returns the full ``(N,)`` vector on every rank.  No collective runs
inside the data path.

Backend ``"nccl"`` is RCCL on ROCm; ``"gloo"`` (CPU tensors) is supported for tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "gather_scores", "sharded_scores", "broadcast_fitted", "ShardedPostprocessor"]


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[start, stop)`` of rank ``rank``: blocks of ``ceil(N/world)`` rows, the tail
    ranks may be short or empty."""
    per = -(-n_rows // world) if n_rows > 0 else 0
    start = min(rank * per, n_rows)
    return start, min(start + per, n_rows)


def _world(group) -> Tuple[int, int]:
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def gather_scores(local: torch.Tensor, n_rows: int, group=None) -> torch.Tensor:
    """One all_gather of the per-rank score shards (padded to ``ceil(N/world)``) -> ``(N,)`` on every rank."""
    world, _ = _world(group)
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if local.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"gather_scores: score shards must be float32 or float64 on every rank, got {local.dtype}")
    per = -(-n_rows // world) if n_rows > 0 else 0
    if local.numel() == per and local.is_contiguous():
        buf = local  # even split: no padding copy
    else:
        buf = torch.zeros(per, dtype=local.dtype, device=local.device)
        buf[: local.numel()] = local
    out = torch.empty(world * per, dtype=local.dtype, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, buf, group=group)
    else:  # gloo
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        out = torch.cat(parts)
    return out[:n_rows]


def sharded_scores(score_fn: Callable, rows, group=None) -> torch.Tensor:
    """Score ``rows`` (anything sliceable along dim 0, identical on every rank) with ``score_fn`` applied to this
    rank's block only; returns the full score vector on every rank."""
    world, rank = _world(group)
    n = len(rows)
    a, b = shard_bounds(n, world, rank)
    local = score_fn(rows[a:b])
    if isinstance(local, np.ndarray):
        local = torch.from_numpy(local)
    return gather_scores(local.reshape(-1), n, group)


def broadcast_fitted(obj, src: int = 0, group=None):
    """Replicate a fitted (picklable) state object from ``src`` to every rank (setup-time, once)."""
    world, _ = _world(group)
    if world == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


class ShardedPostprocessor:
    """Wrap a set-up postprocessor: ``postprocess`` scores this rank's block of the rows and gathers.

    All ranks must call with the same ``test_data``.  Extra keyword arguments are forwarded
    (and sliced when they are per-row arrays of the same length, e.g. ``pred_labels``)."""

    def __init__(self, postprocessor, group=None, device: Optional[torch.device] = None):
        self.postprocessor = postprocessor
        self.group = group
        self.device = device

    def __getattr__(self, name):
        return getattr(self.postprocessor, name)

    def postprocess(self, test_data, **kwargs) -> np.ndarray:
        world, rank = _world(self.group)
        n = len(test_data)
        a, b = shard_bounds(n, world, rank)
        kw = {k: (v[a:b] if hasattr(v, "__len__") and not isinstance(v, str) and len(v) == n else v)
              for k, v in kwargs.items()}
        # An empty tail shard still goes through the postprocessor (its kernels return at once for N == 0) so that
        # the shard carries the dtype the scorer returns - f32 for energy / msp / knn / cMD / gen / GMM / ddu: every rank
        # must enter the one all_gather with the same element size.
        local = np.ascontiguousarray(self.postprocessor.postprocess(test_data[a:b], **kw)).reshape(-1)
        local = torch.from_numpy(local)
        backend = dist.get_backend(self.group) if (dist.is_available() and dist.is_initialized()) else None
        if backend == "nccl":
            local = local.to(self.device or torch.device("cuda", torch.cuda.current_device()))
        return gather_scores(local, n, self.group).cpu().numpy()

    __call__ = postprocess
