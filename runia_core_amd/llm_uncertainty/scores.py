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
from typing import Dict, List, Tuple

import torch

from .. import _hip

__all__ = ["eigen_score", "normalized_entropy", "semantic_entropy", "perplexity", "generation_entropy"]


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


def _entails_both_ways(model, tokenizer, a: str, b: str) -> bool:
    """Bidirectional NLI check of two texts (reference ``llm_uncertainty/utils.py:11-43``): class 0 = contradiction, 1 =
    neutral, 2 = entailment; equivalent unless either direction contradicts or both are neutral.  Both directions go
    through the model in ONE batched forward pass."""
    enc = tokenizer([a, b], [b, a], return_tensors="pt", padding=True)
    enc = {k: v.to(model.device) for k, v in enc.items()}
    with torch.no_grad():
        verdicts = tuple(int(v) for v in torch.argmax(model(**enc).logits, dim=1).tolist())
    return (0 not in verdicts) and verdicts != (1, 1)


def _semantic_clustering(model, tokenizer, texts: List[str]) -> Dict[int, List[int]]:
    """Greedy clustering of semantically equivalent texts (reference ``llm_uncertainty/utils.py:46-81``): every
    unassigned text opens a cluster and collects the later unassigned texts that are equivalent to it."""
    owner = [-1] * len(texts)
    clusters: Dict[int, List[int]] = {}
    for i in range(len(texts)):
        if owner[i] >= 0:
            continue
        cid = len(clusters)
        owner[i] = cid
        clusters[cid] = [i]
        for j in range(i + 1, len(texts)):
            if owner[j] < 0 and _entails_both_ways(model, tokenizer, texts[i], texts[j]):
                owner[j] = cid
                clusters[cid].append(j)
    return clusters


def semantic_entropy(model, tokenizer, texts: List[str]) -> Tuple[float, Dict[int, List[int]]]:
    """Discrete semantic entropy (reference ``llm_uncertainty/scores.py:88-118``; Kuhn et al. 2023): entropy of the
    cluster-size distribution of the NLI clustering of ``texts``.  Host scalars: the cost is the NLI model's forward passes
    (PyTorch-ROCm), there is no kernel work here.  Returns ``(entropy, clusters)``."""
    clusters = _semantic_clustering(model, tokenizer, texts)
    total = sum(len(v) for v in clusters.values())
    entropy = 0.0
    for members in clusters.values():
        p = len(members) / total
        if p > 0:
            entropy -= p * math.log(p)
    return float(entropy), clusters


# ---- scalar scores of one generation (host-side torch on the few hundred numbers a generation yields; reference
# llm_uncertainty/scores.py:69-85, 121-152) --------------------------------------------------------------------------
def normalized_entropy(log_probs: torch.Tensor) -> float:
    """Length-normalised entropy over sequences of token log-probabilities (``-inf`` = padding)."""
    n = len(log_probs)
    entropy = 0.0
    for seq in log_probs:
        valid = seq != -float("inf")
        entropy += torch.sum(seq[valid]) / torch.sum(valid)
    return (-entropy / n).item()


def perplexity(log_probs) -> float:
    """``-mean(log_probs)`` of one sequence (what the reference calls perplexity)."""
    return -torch.mean(log_probs).item()


def generation_entropy(logits) -> float:
    """Entropy of every generated token's distribution (HuggingFace ``outputs.scores``, first batch element),
    normalised by ``log(vocabulary)`` and averaged over the tokens."""
    entropies = []
    for logit in logits:
        p = torch.softmax(logit[0], dim=-1).cpu()
        log_p = torch.clamp(p, min=1e-12).log()
        entropy = -(p * log_p).sum() / torch.log(torch.tensor(p.shape[-1], dtype=torch.float32))
        entropies.append(entropy.item())
    return float(sum(entropies) / len(entropies)) if entropies else float("nan")
