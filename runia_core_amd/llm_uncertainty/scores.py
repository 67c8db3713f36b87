"""``eigen_score`` with the reference's signature (``runia_core/llm_uncertainty/scores.py:49-66``; Chen et al. 2024).

The reference forms the ``hidden x hidden`` covariance of the ``(n_samples, hidden)`` embedding matrix
(``torch.cov(E.T)``, rank < n_samples) and runs ``np.linalg.svd`` on ``cov + alpha I`` - an O(hidden^3) host SVD (hidden =
4096 for Llama-3.1-8B) whose spectrum is ``hidden - n`` copies of ``alpha`` plus the ``n`` eigenvalues of the centred Gram
matrix shifted by ``alpha``.  Here: Gram matrix ``Ec Ec^T / (n-1)`` on the device (``runia_centred_gram_f32``), its
eigenvalues by the Jacobi solver (``runia_eigh_*``), and

    mean(log(sv)) = [ sum_i log(lambda_i + alpha) + (hidden - n) log(alpha) ] / hidden.

The reference's float32 ``torch.cov`` leaves rounding noise of ~1e-6 alpha in the null space, so the two agree to
~7e-7 on the reference's own golden (its test tolerance is 1e-6); the rest of ``llm_uncertainty`` (generation, NLI
clustering) is model glue and out of scope.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from .. import _hip

__all__ = ["eigen_score"]


def _construct_embedding_matrix(hidden_states: Tuple[torch.Tensor, ...], token_index: int = -1, layer_index: int = 15) -> torch.Tensor:
    """``(seq_length, hidden_size)`` embeddings of one token position and layer of HuggingFace ``outputs.hidden_states``
    (reference ``llm_uncertainty/utils.py:102-117``)."""
    return hidden_states[token_index][layer_index].squeeze()


def eigen_score(hidden_states: Tuple[torch.Tensor, ...], alpha: float = 1e-3) -> float:
    """Mean log singular value of ``cov(embeddings) + alpha I`` (see the module docstring for the Gram form)."""
    e = _construct_embedding_matrix(hidden_states)
    assert e.dim() == 2, "embedding matrix must be (num_samples, hidden)"
    n, hidden = e.shape
    ed = _hip.to_device(e, torch.float32)
    g = _hip.centred_gram(ed, float(n - 1))
    lam, _ = _hip.eigh(g)
    lam = lam.clamp_min(0.0)
    k = min(n, hidden)
    # the Gram matrix carries min(n, hidden) of the covariance's eigenvalues (its largest ones when n > hidden)
    top = torch.flip(lam, dims=(0,))[:k]
    total = float(torch.log(top + alpha).sum().item()) + (hidden - k) * math.log(alpha)
    return total / hidden
