"""CPU restatement of the reference's evaluation harness loop (TEST INFRASTRUCTURE, like the rest of ``oracle/``).

``larex_eval_sweep`` follows ``log_evaluate_larex`` (/root/reference/runia_core/evaluation/latent_space.py:105-170: full
vectors, then one PCA refit + transform per entry of ``n_pca_components``) and ``log_evaluate_postprocessors``
(evaluation/metrics.py:322-360: instantiate, ``setup`` on the train split, score valid + every OoD split, one
``get_auroc_results`` row per (OoD set, postprocessor)), with the five latent-space postprocessors of the registry
(inference/postprocessors.py:131-492) in NumPy / SciPy / scikit-learn / CPU torch:

* PCA: the reference's own call, ``sklearn.decomposition.PCA(n, svd_solver="randomized", whiten=True)`` (consumes NumPy's
  global generator; dimensionality_reduction.py:70-71);
* KDE (LaRED): the exact log-density (``oracle.kde_score``'s definition) with the pair distances formed as
  |x|^2 + |t|^2 - 2 x.t on centred rows in float64 (BLAS) - the direct-difference form of ``oracle.kde_score`` is the same
  number to ~1e-13 and 50 x slower; tests/test_oracle_goldens.py checks one against the other;
* MD (LaREM): ``oracle.md_setup`` / ``oracle.md_score``;
* cMD: class means and the pooled float32 centred rows as upstream (:299-316), precision rounded to float32 (:314), the
  quadratic forms of the float32 differences accumulated in float64 and rounded to float32 once (upstream accumulates in
  float32 row by row: its result is this one up to its own ~1e-6 rounding);
* KNN: ``normalizer`` + k-th smallest squared distance against the normalised bank in faiss' single-query float32 arithmetic
  (``oracle.knn_kth_score``'s), the bank rows to measure picked by float64 BLAS distances first;
* GMM: ``gmm_fit`` restated in upstream's form (inference/funcs.py:285-344: per-class mean, ``x^T x / (max(n, 2) - 1)``, the
  jitter ladder through exceptions) + ``oracle.gmm_energy``.

``all_baselines`` follows ``calculate_all_baselines`` (/root/reference/runia_core/evaluation/baselines.py:713-854), the loop of
the features / logits postprocessors; pinned by the ten means of /root/reference/tests/unit_test_baselines.py:255-268
(tests/test_baselines_harness.py).
"""
from __future__ import annotations

import time
from typing import Dict, Iterable, List, Sequence

import numpy as np
from scipy.special import logsumexp

from . import hotpath as H

LATENT_POSTPROCESSORS = ("KDE", "MD", "cMD", "KNN", "GMM")


def kde_score_blas(train: np.ndarray, x: np.ndarray, bandwidth: float = 1.0, chunk: int = 2048) -> np.ndarray:
    train, x = np.asarray(train, np.float64), np.asarray(x, np.float64)
    m, d = train.shape
    mu = train.mean(axis=0)
    tc, xc = train - mu, x - mu
    tn = np.einsum("ij,ij->i", tc, tc)
    out = np.empty(x.shape[0])
    log_norm = -np.log(m) - d * np.log(bandwidth) - 0.5 * d * np.log(2 * np.pi)
    for s in range(0, x.shape[0], chunk):
        xs = xc[s: s + chunk]
        d2 = np.maximum(np.einsum("ij,ij->i", xs, xs)[:, None] + tn[None, :] - 2.0 * (xs @ tc.T), 0.0)
        out[s: s + chunk] = logsumexp(-0.5 * d2 / (bandwidth * bandwidth), axis=1) + log_norm
    return out


def knn_kth_blas(bank_normed: np.ndarray, queries: np.ndarray, k: int, chunk: int = 1024) -> np.ndarray:
    """-(k-th smallest squared L2) per query in faiss' single-query arithmetic (float32 ``sum((q - b)^2)``, what
    ``oracle.knn_kth_score`` computes for every bank row) at BLAS cost: the bank rows are ranked by float64 distances from the
    norm expansion, the 2k + 8 nearest are re-measured exactly in float32, the k-th smallest of those is the score (a bank row
    beyond the 2k + 8 nearest would have to move k + 8 ranks through ~1e-7 of rounding to matter)."""
    bank = np.ascontiguousarray(bank_normed, dtype=np.float32)
    m = bank.shape[0]
    if k > m:
        return np.full(queries.shape[0], -np.float32(H.FLT_MAX), dtype=np.float32)
    b64 = bank.astype(np.float64)
    bn = np.einsum("ij,ij->i", b64, b64)
    keep = min(m, 2 * k + 8)
    out = np.empty(queries.shape[0], dtype=np.float32)
    for s in range(0, queries.shape[0], chunk):
        q32 = np.ascontiguousarray(np.asarray(H.normalizer(queries[s: s + chunk])).astype(np.float32))
        q = q32.astype(np.float64)
        d2 = np.einsum("ij,ij->i", q, q)[:, None] + bn[None, :] - 2.0 * (q @ b64.T)
        cand = np.argpartition(d2, keep - 1, axis=1)[:, :keep]
        diff = q32[:, None, :] - bank[cand]                                   # (chunk, keep, D) float32
        exact = (diff * diff).sum(axis=2, dtype=np.float32)
        out[s: s + chunk] = -np.partition(exact, k - 1, axis=1)[:, k - 1]
    return out


def cmd_setup(train: np.ndarray, labels: np.ndarray, num_classes: int):
    """cMDLatentSpace.setup (inference/postprocessors.py:295-316): float32 tensors in, class means in float32."""
    x = np.asarray(train, dtype=np.float32)
    means, centred = [], []
    for c in range(num_classes):
        rows = x[np.asarray(labels) == c]
        mu = rows.mean(axis=0, dtype=np.float32) if len(rows) else np.full(x.shape[1], np.nan, np.float32)
        means.append(mu)
        centred.append(rows - mu[None, :])
    pooled = np.concatenate(centred).astype(np.float32)
    precision = H.empirical_precision(pooled).astype(np.float32)
    return np.stack(means), precision


def cmd_score(x: np.ndarray, class_mean: np.ndarray, precision32: np.ndarray) -> np.ndarray:
    """cMDLatentSpace.postprocess (:335-357): max over classes of -(x - mu_c) P (x - mu_c)^T, NaN (empty class) -> -inf."""
    x = np.asarray(x, dtype=np.float32)
    p = precision32.astype(np.float64)
    best = np.full(x.shape[0], -np.inf)
    for mu in class_mean:
        t = (x - mu[None, :]).astype(np.float64)          # float32 difference, widened exactly
        s = -np.einsum("ij,ij->i", t @ p, t)
        best = np.maximum(best, np.where(np.isnan(s), -np.inf, s))
    return best.astype(np.float32)


def gmm_fit(train: np.ndarray, labels: np.ndarray, num_classes: int):
    """gmm_fit in upstream's form (inference/funcs.py:285-344), CPU torch float32."""
    import torch

    emb, lab = torch.Tensor(np.asarray(train)), torch.Tensor(np.asarray(labels))
    jitters = [0] + [10 ** e for e in range(-20, 0, 1)]

    def centred_cov(v):
        n = max(v.shape[0], 2)
        return 1 / (n - 1) * v.t().mm(v)

    with torch.no_grad():
        means = torch.stack([torch.mean(emb[lab == c], dim=0) for c in range(num_classes)])
        covs = torch.stack([centred_cov(emb[lab == c] - means[c]) for c in range(num_classes)])
        keep = ~torch.any(means.isnan(), dim=1)
        means, covs = means[keep], covs[keep]
        gmm = None
        for eps in jitters:
            try:
                gmm = torch.distributions.MultivariateNormal(loc=means, covariance_matrix=covs + eps * torch.eye(covs.shape[1]).unsqueeze(0))
            except (RuntimeError, ValueError):
                continue
            break
    return gmm, eps


def latent_scores(train, labels, splits: Dict[str, np.ndarray], postprocessors: Iterable[str], num_classes: int, k: int,
                  seconds: Dict[str, float]) -> Dict[str, Dict[str, np.ndarray]]:
    """setup + postprocess of every postprocessor on every split (metrics.py:322-340); per-postprocessor wall time added to
    ``seconds``."""
    out = {}
    for name in postprocessors:
        t0 = time.perf_counter()
        if name == "KDE":
            out[name] = {s: kde_score_blas(train, v) for s, v in splits.items()}
        elif name == "MD":
            mean, _, prec = H.md_setup(np.asarray(train))
            out[name] = {s: H.md_score(v, mean, prec) for s, v in splits.items()}
        elif name == "cMD":
            cm, p32 = cmd_setup(train, labels, num_classes)
            out[name] = {s: cmd_score(v, cm, p32) for s, v in splits.items()}
        elif name == "KNN":
            bank = np.array([H.normalizer(f) for f in np.asarray(train)]).astype(np.float32)  # :395, row by row as upstream
            out[name] = {s: knn_kth_blas(bank, v, k) for s, v in splits.items()}
        elif name == "GMM":
            gmm, _ = gmm_fit(train, labels, num_classes)
            out[name] = {s: H.gmm_energy(gmm, np.asarray(v, dtype=np.float32)).astype(np.float32) for s, v in splits.items()}
        else:
            raise KeyError(name)
        seconds[name] = seconds.get(name, 0.0) + time.perf_counter() - t0
    return out


def larex_eval_sweep(ind: Dict[str, np.ndarray], ood: Dict[str, np.ndarray], ood_names: Sequence[str],
                     n_pca_components: Sequence[int], postprocessors: Sequence[str] = LATENT_POSTPROCESSORS,
                     num_classes: int = 10, k: int = 50):
    """The results table of ``log_evaluate_larex`` as ``{row name: (auroc, fpr@95, aupr)}`` + a wall-time breakdown.
    ``ind``: "train latent_space_means", "valid latent_space_means", "train labels"; ``ood``: f"{name} latent_space_means"."""
    from sklearn.decomposition import PCA

    table, seconds = {}, {}

    def one_config(train, valid, oods: Dict[str, np.ndarray], ext: str):
        splits = {"valid": valid, **oods}
        scores = latent_scores(train, ind["train labels"], splits, postprocessors, num_classes, k, seconds)
        t0 = time.perf_counter()
        for name in ood_names:
            for pp in postprocessors:
                table[f"{name} {pp}{ext}"] = H.auroc_fpr95_aupr(scores[pp]["valid"], scores[pp][name])
        seconds["metrics"] = seconds.get("metrics", 0.0) + time.perf_counter() - t0

    one_config(ind["train latent_space_means"], ind["valid latent_space_means"],
               {n: ood[f"{n} latent_space_means"] for n in ood_names}, "")
    for n_comp in n_pca_components:
        t0 = time.perf_counter()
        pca = PCA(n_components=n_comp, svd_solver="randomized", whiten=True)
        train_red = pca.fit_transform(ind["train latent_space_means"])
        valid_red = pca.transform(ind["valid latent_space_means"])
        oods_red = {n: pca.transform(ood[f"{n} latent_space_means"]) for n in ood_names}
        seconds["pca"] = seconds.get("pca", 0.0) + time.perf_counter() - t0
        one_config(train_red, valid_red, oods_red, f" PCA {n_comp}")
    return table, seconds


BASELINES = ("vim", "msp", "raw", "knn", "energy", "ash", "gen", "react", "dice", "dice_react", "mdist", "ddu")


def all_baselines(names: Iterable[str], ind: Dict[str, np.ndarray], ood: Dict[str, np.ndarray], ood_names: Sequence[str], w, b,
                  num_classes: int, k_neighbors: int, ash_percentile: int, react_percentile: int, dice_percentile: int,
                  gen_gamma: float, seconds: Dict[str, float] | None = None) -> Dict[str, Dict[str, np.ndarray]]:
    """CPU form of ``calculate_all_baselines`` (/root/reference/runia_core/evaluation/baselines.py:713-854): every baseline fit
    on the training split and scored on ``"valid"`` + every OoD set -> ``{baseline: {"valid": scores, ood_name: scores}}``.  The
    labels of ``mdist`` / ``ddu`` are the argmax of the TRAIN LOGITS (``get_labels_from_logits``, :614-683, run before them); the
    postprocessors are the oracle's restatements (``hotpath.py``), the Gaussians of ``ddu`` upstream's ``gmm_fit`` (above).
    ``ind`` / ``ood`` are read only."""
    seconds = {} if seconds is None else seconds
    feats = {"valid": ind["valid features"], **{n: ood[f"{n} features"] for n in ood_names}}
    logits = {"valid": ind["valid logits"], **{n: ood[f"{n} logits"] for n in ood_names}}
    tr_f, tr_l = ind["train features"], ind["train logits"]
    labels = np.argmax(tr_l[:, :-1] if tr_l.shape[1] in (21, 11) else tr_l, axis=-1)
    out: Dict[str, Dict[str, np.ndarray]] = {}
    for name in names:
        t0 = time.perf_counter()
        if name == "vim":
            u, ns, alpha = H.vim_setup(tr_f, tr_l, w, b)
            out[name] = {s: H.vim_score(feats[s], logits[s], u, ns, alpha) for s in feats}
        elif name in ("msp", "raw"):
            out[name] = {s: H.msp_score(v) for s, v in logits.items()}
        elif name == "energy":
            out[name] = {s: H.energy_score(v) for s, v in logits.items()}
        elif name == "gen":
            out[name] = {s: H.gen_score(v, gen_gamma, num_classes) for s, v in logits.items()}
        elif name == "knn":
            bank = np.ascontiguousarray(H.normalizer(tr_f).astype(np.float32))
            out[name] = {s: knn_kth_blas(bank, v, k_neighbors) for s, v in feats.items()}
        elif name == "ash":
            out[name] = {s: H.linear_energy(H.ash_s_defined(np.array(v, copy=True), ash_percentile), w, b) for s, v in feats.items()}
        elif name == "react":
            thr = H.react_threshold(tr_f, react_percentile)
            out[name] = {s: H.react_score(v, w, b, thr) for s, v in feats.items()}
        elif name in ("dice", "dice_react"):
            mw = H.dice_masked_weight(tr_f, w, dice_percentile)
            clip = np.float32(H.react_threshold(tr_f, react_percentile)) if name == "dice_react" else None
            out[name] = {s: logsumexp(H.dice_logits(v if clip is None else v.clip(max=clip), mw, b), axis=1) for s, v in feats.items()}
        elif name == "mdist":
            import warnings

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # (a class without samples warns, as upstream)
                cm, prec = H.mahalanobis_setup(tr_f, labels, num_classes)
            out[name] = {s: H.mahalanobis_score(v, cm, prec, num_classes) for s, v in feats.items()}
        elif name == "ddu":
            gmm, _ = gmm_fit(tr_f, labels, num_classes)
            out[name] = {s: H.gmm_energy(gmm, np.asarray(v, dtype=np.float32)) for s, v in feats.items()}
        else:
            raise KeyError(name)
        seconds[name] = seconds.get(name, 0.0) + time.perf_counter() - t0
    return out
