"""NumPy/SciPy restatement of the runia_core scoring hot path (TEST INFRASTRUCTURE).

Every function cites the reference file:line it follows (paths relative to
``/root/reference/runia_core/`` unless absolute).  Third-party kernels that are
absent from the image are restated from their published algorithm and pinned by
the reference's own golden vectors (see ``oracle/__init__.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np
from scipy.linalg import pinvh
from scipy.spatial import cKDTree
from scipy.special import digamma, logsumexp, softmax

__all__ = [
    "get_h",
    "single_image_entropy_calculation",
    "get_dl_h_z",
    "kl_entropy_per_dim_vectorized",
    "kl_entropy_joint_vectorized",
    "dropblock_block_mask",
    "torch_cpu_sum_lastdim",
    "mc_stack",
    "roi_align",
    "rois_mc_entropy",
    "philox4x32_10",
    "counter_draws",
    "counter_draws_redrawn",
    "pca_transform",
    "empirical_precision",
    "md_setup",
    "md_score",
    "md_score_reference_form",
    "mahalanobis_setup",
    "mahalanobis_score",
    "mahalanobis_score_reference_form",
    "energy_score",
    "msp_score",
    "normalizer",
    "knn_kth_score",
    "kde_score",
    "method_threshold",
    "binary_clf_curve",
    "auroc_fpr95_aupr",
    "larem_pipeline",
    "ash_s_linear_layer",
    "ash_s_defined",
    "ash_s_conv_defined",
    "predictive_uncertainty",
    "kde_score_kernel",
    "linear_energy",
    "react_threshold",
    "react_score",
    "dice_masked_weight",
    "dice_logits",
    "generalized_entropy",
    "gen_score",
    "gmm_energy",
    "vim_setup",
    "vim_score",
    "FLT_MAX",
]

_trap = np.trapezoid if hasattr(np, "trapezoid") else np.trapz
FLT_MAX = float(np.finfo(np.float32).max)  # 3.4028234663852886e38, faiss fill value


# --------------------------------------------------------------------------
# a2  Kozachenko-Leonenko kNN entropy  (evaluation/entropy.py:20-93)
# --------------------------------------------------------------------------
def get_h(x: np.ndarray, k: int = 1, norm: str = "max", min_dist: float = 0.0) -> float:
    """``entropy_estimators.continuous.get_h`` (entropy-estimators==0.0.1,
    /root/reference/requirements.txt:3), call sites evaluation/entropy.py:35,68,79.

    Published algorithm: build a k-d tree on the n samples, query k+1 neighbours
    (the query point itself is in the set) under the chosen norm, take the k-th
    non-self distance, clip to ``min_dist``, and return
    ``psi(n) - psi(k) + log_c_d + (d/n) * sum(log(2*eps))`` in nats with
    ``log_c_d = 0`` for the max norm.
    """
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n, d = x.shape
    if norm == "max":
        p = np.inf
        log_c_d = 0.0
    elif norm == "euclidean":
        p = 2
        log_c_d = (d / 2.0) * math.log(math.pi) - math.lgamma(d / 2.0 + 1)
    else:
        raise NotImplementedError("Variable 'norm' either 'max' or 'euclidean'")
    tree = cKDTree(x)
    distances, _ = tree.query(x, k + 1, eps=0, p=p)
    distances = distances[:, -1].copy()
    distances[distances < min_dist] = min_dist
    sum_log_dist = np.sum(np.log(2 * distances))
    return float(-digamma(k) + digamma(n) + log_c_d + (d / float(n)) * sum_log_dist)


def single_image_entropy_calculation(sample: np.ndarray, neighbors: int) -> np.ndarray:
    """evaluation/entropy.py:20-38 — per-dimension entropy of one image's MC samples."""
    return np.asarray(
        [get_h(sample[:, j], k=neighbors, norm="max", min_dist=1e-5) for j in range(sample.shape[1])]
    )


def _neighbors(mcd_samples_nro: int) -> int:
    # evaluation/entropy.py:66
    return 5 if mcd_samples_nro > 5 else mcd_samples_nro - 1


def get_dl_h_z(dl_z_samples: np.ndarray, mcd_samples_nro: int = 32) -> Tuple[np.ndarray, np.ndarray]:
    """evaluation/entropy.py:41-93 in its reference algorithmic form (one k-d tree
    per (image, dim) plus one joint tree per image).  Returns ``(N,1)`` joint and
    ``(N,D)`` per-dimension entropies, float64."""
    z = np.asarray(dl_z_samples)
    n_img = int(z.shape[0] / mcd_samples_nro)
    blocks = np.split(z, n_img)
    k = _neighbors(mcd_samples_nro)
    h_mvn = np.array([get_h(s, k=k, norm="max", min_dist=1e-5) for s in blocks])
    h_mvn = np.expand_dims(h_mvn, axis=1)
    h_z = np.asarray([single_image_entropy_calculation(s, k) for s in blocks])
    return h_mvn, h_z


def kl_entropy_per_dim_vectorized(z: np.ndarray, n_mc: int, k: int | None = None) -> np.ndarray:
    """Vectorised equivalent of the per-dimension loop (evaluation/entropy.py:77-82):
    1-D k-th-NN distance by sorting the n_mc values of each (image, dim) column.
    Used for sizes where the k-d-tree form takes too long; checked against
    :func:`get_dl_h_z` in tests."""
    z = np.asarray(z, dtype=np.float64)
    n_img = z.shape[0] // n_mc
    d = z.shape[1]
    if k is None:
        k = _neighbors(n_mc)
    v = np.sort(z.reshape(n_img, n_mc, d), axis=1)  # ascending along samples
    # distances |v_i - v_j| for all pairs, k-th smallest non-self per i
    diff = np.abs(v[:, :, None, :] - v[:, None, :, :])  # (N, n, n, D)
    diff.sort(axis=2)
    eps = np.maximum(diff[:, :, k, :], 1e-5)  # index 0 is self (0.0)
    const = digamma(n_mc) - digamma(k)
    return const + np.log(2.0 * eps).sum(axis=1) / n_mc


def kl_entropy_joint_vectorized(z: np.ndarray, n_mc: int, k: int | None = None) -> np.ndarray:
    """Vectorised equivalent of the joint call (evaluation/entropy.py:67-69):
    Chebyshev distance in D dims."""
    z = np.asarray(z, dtype=np.float64)
    n_img = z.shape[0] // n_mc
    d = z.shape[1]
    if k is None:
        k = _neighbors(n_mc)
    v = z.reshape(n_img, n_mc, d)
    out = np.empty((n_img, 1))
    const = digamma(n_mc) - digamma(k)
    for i in range(n_img):
        cheb = np.abs(v[i][:, None, :] - v[i][None, :, :]).max(axis=2)  # (n, n)
        cheb.sort(axis=1)
        eps = np.maximum(cheb[:, k], 1e-5)
        out[i, 0] = const + (d / n_mc) * np.log(2.0 * eps).sum()
    return out


# --------------------------------------------------------------------------
# a1  MC-dropout latent stacking (feature_extraction/abstract_classes.py:81-101)
#     Pinned by tests/golden/ref_sampler.npz: outputs of the reference's own
#     MCSamplerModule.forward (loaded by path, tools/make_goldens_r2.py); the only
#     restated piece is the third-party DropBlock2D layer (dropblock==0.3.0, absent).
# --------------------------------------------------------------------------
def dropblock_block_mask(rand: np.ndarray, drop_prob: float, block_size: int) -> np.ndarray:
    """``dropblock.DropBlock2D`` block mask (dropblock==0.3.0,
    /root/reference/requirements.txt:2; call sites
    feature_extraction/abstract_classes.py:74-79,93).  ``rand`` is the uniform
    draw ``torch.rand(B,H,W)``; returns ``1 - maxpool(rand < gamma)`` with kernel
    ``block_size``, stride 1, padding ``block_size//2`` and the last row/column
    cropped when ``block_size`` is even."""
    rand = np.asarray(rand, dtype=np.float32)
    gamma = np.float32(drop_prob / (block_size**2))
    mask = (rand < gamma).astype(np.float32)
    b, h, w = mask.shape
    pad = block_size // 2
    padded = np.zeros((b, h + 2 * pad, w + 2 * pad), dtype=np.float32)
    padded[:, pad : pad + h, pad : pad + w] = mask
    oh = h + 2 * pad - block_size + 1
    ow = w + 2 * pad - block_size + 1
    pooled = np.zeros((b, oh, ow), dtype=np.float32)
    for dy in range(block_size):
        for dx in range(block_size):
            pooled = np.maximum(pooled, padded[:, dy : dy + oh, dx : dx + ow])
    if block_size % 2 == 0:
        pooled = pooled[:, :-1, :-1]
    return (1.0 - pooled).astype(np.float32)


def torch_cpu_sum_lastdim(a: np.ndarray) -> np.ndarray:
    """f32 sum over the last axis in the order of ``torch.sum`` / ``torch.mean`` on a CPU tensor whose reduced
    dimension is contiguous (ATen ``SumKernel.cpp`` ``cascade_sum``; what ``get_mean_or_fullmean_ls_sample``,
    feature_extraction/utils.py:88-92, runs when the sampler sits on the host as in the reference's tests):

    * n < 8: four interleaved partial sums ``p[k] = a[k] (+ a[4+k] ...)``, the remainder added to ``p[0]``, then
      ``((p0 + p1) + p2) + p3``  -  for 4 <= n < 8 the chain a0, a4, .., a[n-1], a1, a2, a3;
    * n >= 8: 8-lane vector partials (four of them interleaved over the 8-element chunks), the scalar remainder
      summed first, then the 8 lanes added to it one by one.

    Verified against torch itself for every n in 1..69 (tests/test_oracle_goldens.py).  The cascade levels that ATen
    adds for rows of >= 512 elements are not restated."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[-1]
    lead = a.shape[:-1]
    f = np.float32
    if n < 8:
        s = n // 4
        p = [np.zeros(lead, f) for _ in range(4)]
        for k in range(4):
            for j in range(s):
                p[k] = (p[k] + a[..., j * 4 + k]).astype(f)
        for i in range(s * 4, n):
            p[0] = (p[0] + a[..., i]).astype(f)
        for k in range(1, 4):
            p[0] = (p[0] + p[k]).astype(f)
        return p[0]
    vs = n // 8
    vec = a[..., : vs * 8].reshape(lead + (vs, 8))
    p = [np.zeros(lead + (8,), f) for _ in range(4)]
    s = vs // 4
    for k in range(4):
        for j in range(s):
            p[k] = (p[k] + vec[..., j * 4 + k, :]).astype(f)
    for i in range(s * 4, vs):
        p[0] = (p[0] + vec[..., i, :]).astype(f)
    for k in range(1, 4):
        p[0] = (p[0] + p[k]).astype(f)
    acc = np.zeros(lead, f)
    for k in range(vs * 8, n):
        acc = (acc + a[..., k]).astype(f)
    for lane in range(8):
        acc = (acc + p[0][..., lane]).astype(f)
    return acc


def philox4x32_10(counter: np.ndarray, key) -> np.ndarray:
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) on an array of
    counters ``(..., 4)`` uint32 with one key ``(k0, k1)``.  Checker of the build's throughput-mode draws (the reference
    has no counterpart: it draws ``torch.rand`` on the CPU generator); pinned by the published known-answer vectors."""
    c = np.asarray(counter, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = np.uint64(int(key[0]) & 0xFFFFFFFF), np.uint64(int(key[1]) & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        c1, c3, c0, c2 = p1 & mask, p0 & mask, n0 & mask, n2 & mask
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def counter_draws(n: int, n_mc: int, h: int, w: int, seed: int, first_image: int = 0, attempt: int = 0) -> np.ndarray:
    """The draws of ``runia_mc_draws_f32`` / the counter entry points (csrc/philox.hpp): draw i = layer*H*W + position of
    image g is component ``(i // 64) & 3`` of ``philox((g.lo, g.hi, i % 64 + 64 * (i // 256), attempt), seed)``,
    ``u = (bits >> 8) * 2**-24``; ``attempt`` = 0 for the draws proper, 1, 2, ... for the redraws of
    ``redraw_dead_layers``.  Returns ``(n, n_mc, h, w)`` float32."""
    per = n_mc * h * w
    i = np.arange(per, dtype=np.uint64)
    g = (np.arange(n, dtype=np.uint64) + np.uint64(first_image))[:, None]
    ctr = np.empty((n, per, 4), dtype=np.uint64)
    ctr[..., 0] = g & np.uint64(0xFFFFFFFF)
    ctr[..., 1] = g >> np.uint64(32)
    ctr[..., 2] = (i % np.uint64(64) + np.uint64(64) * (i // np.uint64(256)))[None, :]
    ctr[..., 3] = int(attempt)
    seed = int(seed) & (2**64 - 1)
    blocks = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32))
    comp = ((i // np.uint64(64)) & np.uint64(3)).astype(np.int64)
    bits = np.take_along_axis(blocks, np.broadcast_to(comp[None, :, None], (n, per, 1)), axis=2)[..., 0]
    return ((bits >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24)).reshape(n, n_mc, h, w)


def counter_draws_redrawn(n: int, n_mc: int, h: int, w: int, seed: int, first_image: int, drop_prob: float,
                          block_size: int, max_attempts: int = 16) -> np.ndarray:
    """Checker of the build's ``CounterDraws(redraw_dead_layers=True)`` (no reference counterpart): the explicit draws
    that are equivalent to it - every drop layer whose block mask removes the whole map takes the draws of the next
    attempt (fourth counter word 1, 2, ...) until its mask keeps something or ``max_attempts`` is reached."""
    rand = counter_draws(n, n_mc, h, w, seed, first_image)
    for attempt in range(1, max_attempts + 1):
        bm = dropblock_block_mask(rand.reshape(n * n_mc, h, w), drop_prob, block_size).reshape(n, n_mc, h * w)
        dead = bm.sum(axis=2) == 0
        if not dead.any():
            break
        again = counter_draws(n, n_mc, h, w, seed, first_image, attempt)
        rand[dead] = again[dead]
    return rand


def mc_stack(x: np.ndarray, rand: np.ndarray, drop_prob: float, block_size: int, layer_type: str = "Conv") -> np.ndarray:
    """``MCSamplerModule.forward`` (feature_extraction/abstract_classes.py:91-101): per drop layer
    ``y = x * bm * numel / sum`` (dropblock==0.3.0), then for ``layer_type="Conv"`` the ``fullmean``
    (mean over W, then over H; feature_extraction/utils.py:88-92), then ``reshape(1, -1)`` and ``cat``.
    ``x`` is ``(1,C,H,W)`` f32, ``rand`` is ``(n_mc,H,W)`` - the uniform draw of each drop layer in
    ``ModuleList`` order.  Returns ``(n_mc, C)`` f32 for ``"Conv"``, ``(n_mc, C*H*W)`` for ``"FC"`` / ``"RPN"``."""
    x = np.asarray(x, dtype=np.float32)
    assert x.ndim == 4 and x.shape[0] == 1
    n_mc = rand.shape[0]
    _, c, h, w = x.shape
    conv = layer_type == "Conv"
    out = np.empty((n_mc, c if conv else c * h * w), dtype=np.float32)

    def fullmean(y):  # (C,H,W) -> (C,)
        rows = torch_cpu_sum_lastdim(y) / np.float32(w)
        return torch_cpu_sum_lastdim(rows) / np.float32(h)

    if drop_prob == 0.0:
        out[:] = fullmean(x[0]) if conv else x[0].reshape(-1)
        return out
    bm = dropblock_block_mask(rand, drop_prob, block_size)  # (n_mc,H,W)
    for s in range(n_mc):
        # published order: out = x * bm; out = out * bm.numel() / bm.sum()   (all f32)
        with np.errstate(invalid="ignore", divide="ignore"):
            y = (x[0] * bm[s][None]) * np.float32(bm[s].size) / np.float32(bm[s].sum(dtype=np.float32))
        out[s] = fullmean(y) if conv else y.reshape(-1)
    return out


# --------------------------------------------------------------------------
# f3  roi_align in front of the per-ROI sampler (feature_extraction/object_level.py:283-292, 340-349)
#     torchvision is absent from the image: restated from its published algorithm; PARITY UNPINNED for this function
#     (the reference's own per-ROI glue around it is pinned by tests/golden/ref_roi.npz with this restatement plugged in).
# --------------------------------------------------------------------------
def roi_align(x: np.ndarray, boxes: np.ndarray, output_size, spatial_scale: float = 1.0, sampling_ratio: int = -1,
              aligned: bool = False) -> np.ndarray:
    """``torchvision.ops.roi_align(x, [boxes], output_size, spatial_scale, sampling_ratio, aligned)`` for one image
    ``x (1, C, H, W)`` f32, ``boxes (K, 4)`` xyxy -> ``(K, C, PH, PW)`` f32 (float32 arithmetic, samples added row by row)."""
    x = np.asarray(x, dtype=np.float32)
    boxes = np.asarray(boxes, dtype=np.float32)
    _, c, h, w = x.shape
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
    f = np.float32
    out = np.zeros((boxes.shape[0], c, ph, pw), dtype=np.float32)
    off = f(0.5) if aligned else f(0.0)
    sc = f(spatial_scale)

    def bilinear(yy, xx):
        if yy < -1.0 or yy > h or xx < -1.0 or xx > w:
            return np.zeros(c, dtype=np.float32)
        yy = max(yy, f(0.0))
        xx = max(xx, f(0.0))
        yl, xl = int(yy), int(xx)
        if yl >= h - 1:
            yh = yl = h - 1
            yy = f(yl)
        else:
            yh = yl + 1
        if xl >= w - 1:
            xh = xl = w - 1
            xx = f(xl)
        else:
            xh = xl + 1
        ly, lx = f(yy - f(yl)), f(xx - f(xl))
        hy, hx = f(f(1.0) - ly), f(f(1.0) - lx)
        w1, w2, w3, w4 = f(hy * hx), f(hy * lx), f(ly * hx), f(ly * lx)
        return ((w1 * x[0, :, yl, xl] + w2 * x[0, :, yl, xh]) + w3 * x[0, :, yh, xl]) + w4 * x[0, :, yh, xh]

    for k, bx in enumerate(boxes):
        x1, y1, x2, y2 = (f(f(v * sc) - off) for v in bx)
        rw, rh = f(x2 - x1), f(y2 - y1)
        if not aligned:
            rw, rh = max(rw, f(1.0)), max(rh, f(1.0))
        bh, bw = f(rh / f(ph)), f(rw / f(pw))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / ph))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / pw))
        count = f(max(gh * gw, 1))
        for i in range(ph):
            for j in range(pw):
                acc = np.zeros(c, dtype=np.float32)
                for iy in range(gh):
                    yy = f(f(y1 + f(f(i) * bh)) + f(f(f(iy) + f(0.5)) * bh) / f(gh))
                    for ix in range(gw):
                        xx = f(f(x1 + f(f(j) * bw)) + f(f(f(ix) + f(0.5)) * bw) / f(gw))
                        acc = (acc + bilinear(yy, xx)).astype(np.float32)
                out[k, :, i, j] = acc / count
    return out


def rois_mc_entropy(feature_maps, output_sizes, boxes, img_shape, sampling_ratio, rand, drop_prob, block_size) -> np.ndarray:
    """``_dropblock_rois_get_entropy`` (feature_extraction/object_level.py:312-367): roi_align of every hooked layer
    (spatial_scale = W_feat / W_img, aligned), concatenation along channels, ``mc_sampler(roi[None])`` per detection,
    ``get_dl_h_z(...)[1]``.  ``rand`` ``(K, n_mc, PH, PW)``: the DropBlock draws, detection-major.  Returns ``(K, C_total)``."""
    rois = [roi_align(fm, boxes, output_sizes[i], fm.shape[3] / img_shape[1], sampling_ratio, True)
            for i, fm in enumerate(feature_maps)]
    rois = np.concatenate(rois, axis=1) if len(rois) > 1 else rois[0]
    n_mc = rand.shape[1]
    z = np.concatenate([mc_stack(rois[k : k + 1], rand[k], drop_prob, block_size) for k in range(rois.shape[0])])
    return kl_entropy_per_dim_vectorized(z, n_mc)


# --------------------------------------------------------------------------
# a4  PCA transform (dimensionality_reduction.py:75-87 -> sklearn PCA.transform)
# --------------------------------------------------------------------------
def pca_transform(
    x: np.ndarray,
    components: np.ndarray,
    mean: np.ndarray,
    explained_variance: np.ndarray,
    whiten: bool = True,
) -> np.ndarray:
    """sklearn ``_BasePCA.transform`` closed form
    (site-packages/sklearn/decomposition/_base.py:146-164):
    ``Y = X @ C.T - mean @ C.T`` then ``Y /= max(sqrt(var), eps)`` when whitening."""
    x = np.asarray(x, dtype=np.float64)
    y = x @ components.T
    y -= mean.reshape(1, -1) @ components.T
    if whiten:
        scale = np.sqrt(explained_variance)
        min_scale = np.finfo(scale.dtype).eps
        scale = np.where(scale < min_scale, min_scale, scale)
        y /= scale
    return y


# --------------------------------------------------------------------------
# a5  LaREM = MDLatentSpace (inference/postprocessors.py:181-244)
# --------------------------------------------------------------------------
def empirical_precision(x: np.ndarray) -> np.ndarray:
    """``EmpiricalCovariance(assume_centered=False).fit(X).precision_``:
    ``pinvh(np.cov(X.T, bias=1))`` (sklearn/covariance/_empirical_covariance.py)."""
    x = np.asarray(x)
    cov = np.cov(x.T, bias=1)
    cov = np.atleast_2d(cov)
    return pinvh(cov, check_finite=False)


def md_setup(ind_train_data: np.ndarray):
    """inference/postprocessors.py:210-222. Returns (feats_mean (1,D), centered, precision)."""
    assert ind_train_data.ndim == 2
    feats_mean = np.mean(ind_train_data, 0, keepdims=True)
    centered = ind_train_data - feats_mean
    precision = empirical_precision(centered)
    return feats_mean, centered, precision


def md_score_reference_form(test_data, feats_mean, precision) -> np.ndarray:
    """inference/postprocessors.py:241-242 verbatim shape: full N x N product, then diag."""
    diff = test_data - feats_mean
    return -np.diag(np.matmul(np.matmul(diff, precision), np.transpose(diff)))


def md_score(test_data, feats_mean, precision) -> np.ndarray:
    """Row-wise form of the same quadratic form (O(N D^2) instead of O(N^2 D))."""
    diff = np.asarray(test_data - feats_mean, dtype=np.float64)
    return -np.einsum("ij,ij->i", diff @ precision, diff)


# --------------------------------------------------------------------------
# a6  class-conditional Mahalanobis (inference/funcs.py:33-102)
# --------------------------------------------------------------------------
def mahalanobis_setup(train_feats: np.ndarray, train_labels: np.ndarray, num_classes: int):
    """inference/funcs.py:52-66."""
    class_mean = []
    centered = []
    for c in range(num_classes):
        cs = train_feats[train_labels == c]
        class_mean.append(cs.mean(0))
        centered.append(cs - class_mean[c].reshape(1, -1))
    class_mean = np.stack(class_mean)
    precision = empirical_precision(np.concatenate(centered).astype(np.float32))
    return class_mean, precision


def mahalanobis_score_reference_form(feats, class_mean, precision, num_classes) -> np.ndarray:
    """inference/funcs.py:88-100 verbatim shape: Python double loop."""
    out = []
    for f in feats:
        cs = np.zeros((1, num_classes))
        for c in range(num_classes):
            t = f - class_mean[c].reshape(1, -1)
            cs[:, c] = np.diag(-np.matmul(np.matmul(t, precision), t.T))
        cs[np.isnan(cs)] = -np.inf
        out.append(np.max(cs, axis=1))
    return np.concatenate(out)


def mahalanobis_score(feats, class_mean, precision, num_classes) -> np.ndarray:
    """Vectorised over rows; keeps the dtype of ``feats - class_mean[c]``
    (f32 when both are f32, inference/funcs.py:92) before the f64 quadratic form."""
    n = feats.shape[0]
    cs = np.empty((n, num_classes))
    for c in range(num_classes):
        t = feats - class_mean[c].reshape(1, -1)
        cs[:, c] = -np.einsum("ij,ij->i", np.matmul(t, precision), t)
    cs[np.isnan(cs)] = -np.inf
    return cs.max(axis=1)


# --------------------------------------------------------------------------
# a7  Energy / MSP (inference/postprocessors.py:549, 606)
# --------------------------------------------------------------------------
def energy_score(logits: np.ndarray) -> np.ndarray:
    return logsumexp(logits, axis=1)


def msp_score(logits: np.ndarray) -> np.ndarray:
    return np.max(softmax(logits, axis=1), axis=1)


# --------------------------------------------------------------------------
# a8  kNN (inference/postprocessors.py:395-421, 842-880; inference/funcs.py:105-115)
# --------------------------------------------------------------------------
def normalizer(x: np.ndarray) -> np.ndarray:
    """inference/funcs.py:115."""
    return x / (np.linalg.norm(x, ord=2, axis=-1, keepdims=True) + 1e-10)


def knn_kth_score(bank_normed: np.ndarray, queries: np.ndarray, k: int, chunk: int = 256, normalize: bool = True) -> np.ndarray:
    """``faiss.IndexFlatL2.add(bank); search(normalizer(q), k)`` then ``-D[:, -1]``
    (faiss-gpu==1.7.2, /root/reference/requirements.txt:4; call sites
    inference/postprocessors.py:396-397,419,850-851,878).  Published behaviour:
    squared L2 in float32, results sorted ascending, tail filled with FLT_MAX when
    ``k > ntotal``.  The reference searches one query at a time, for which faiss
    accumulates ``sum((q-b)^2)`` directly (no norm expansion).  faiss keeps a max-heap
    initialised with FLT_MAX and inserts a distance only if it compares SMALLER than the
    heap top (``CMax::cmp``), so a NaN or infinite distance is never inserted: it counts
    as the FLT_MAX fill.  ``normalize=False`` skips the reference's ``normalizer`` on the
    queries (checks of the raw ``runia_knn_kth_f32`` entry point)."""
    bank = np.ascontiguousarray(bank_normed, dtype=np.float32)
    m = bank.shape[0]
    n = queries.shape[0]
    out = np.empty(n, dtype=np.float32)
    if k > m:
        out[:] = -np.float32(FLT_MAX)
        return out
    for s in range(0, n, chunk):
        q = normalizer(queries[s : s + chunk]) if normalize else queries[s : s + chunk]
        q = np.ascontiguousarray(np.asarray(q).astype(np.float32))
        with np.errstate(invalid="ignore", over="ignore"):
            d = ((q[:, None, :] - bank[None, :, :]) ** 2).sum(axis=2, dtype=np.float32)
        d = np.where(np.isfinite(d), d, np.float32(FLT_MAX)).astype(np.float32)
        out[s : s + chunk] = -np.partition(d, k - 1, axis=1)[:, k - 1]
    return out


# --------------------------------------------------------------------------
# a9  LaRED = KDELatentSpace (inference/postprocessors.py:78-178)
# --------------------------------------------------------------------------
def kde_score(train: np.ndarray, x: np.ndarray, bandwidth: float = 1.0, chunk: int = 512) -> np.ndarray:
    """sklearn ``KernelDensity(kernel="gaussian", bandwidth=h).fit(train).score_samples(x)``
    closed form (exact because rtol=atol=0):
    ``logsumexp_i(-|x-x_i|^2 / 2h^2) - log(N) - D*log(h) - (D/2)*log(2*pi)``."""
    train = np.asarray(train, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    n_train, d = train.shape
    out = np.empty(x.shape[0])
    log_norm = -math.log(n_train) - d * math.log(bandwidth) - 0.5 * d * math.log(2 * math.pi)
    for s in range(0, x.shape[0], chunk):
        xs = x[s : s + chunk]
        d2 = ((xs[:, None, :] - train[None, :, :]) ** 2).sum(axis=2)
        out[s : s + chunk] = logsumexp(-0.5 * d2 / (bandwidth * bandwidth), axis=1) + log_norm
    return out


# --------------------------------------------------------------------------
# a10 threshold (inference/abstract_classes.py:408-424)
# --------------------------------------------------------------------------
def method_threshold(scores: np.ndarray, z_score_percentile: float = 1.645) -> float:
    mean = float(np.mean(scores))
    std = float(np.std(scores))
    return mean - (z_score_percentile * std)


# --------------------------------------------------------------------------
# f2  metrics after the path (evaluation/metrics.py:37-100; torchmetrics==1.8.2
#     binary auroc / roc / precision_recall_curve, /root/reference/requirements.txt)
# --------------------------------------------------------------------------
def binary_clf_curve(preds: np.ndarray, target: np.ndarray):
    """torchmetrics ``_binary_clf_curve``: sort by score descending, keep the last
    index of each run of equal scores, cumulative true / false positives."""
    order = np.argsort(-preds, kind="stable")
    preds = preds[order]
    target = target[order]
    distinct = np.nonzero(preds[1:] - preds[:-1])[0]
    idx = np.concatenate([distinct, [target.size - 1]])
    tps = np.cumsum(target)[idx]
    fps = 1 + idx - tps
    return fps, tps, preds[idx]


def auroc_fpr95_aupr(ind_scores: np.ndarray, ood_scores: np.ndarray):
    """evaluation/metrics.py:61-81.  InD = positive class.  torchmetrics applies a
    sigmoid when any score is outside [0,1] (monotone, but saturates), and returns
    float32 curves; both are reproduced.  Returns (auroc, fpr@95, aupr) as floats."""
    # np.vstack keeps float32 when both score sets are float32 (energy / msp / knn / ... scores): torchmetrics then
    # applies its sigmoid in float32; every other combination is float64
    both_f32 = np.asarray(ind_scores).dtype == np.float32 and np.asarray(ood_scores).dtype == np.float32
    dt = np.float32 if both_f32 else np.float64
    scores = np.concatenate([np.ravel(ind_scores), np.ravel(ood_scores)]).astype(dt)
    labels = np.concatenate(
        [np.ones(np.size(ind_scores), dtype=np.int64), np.zeros(np.size(ood_scores), dtype=np.int64)]
    )
    if not np.all((scores >= 0) & (scores <= 1)):
        with np.errstate(over="ignore"):
            scores = (dt(1.0) / (dt(1.0) + np.exp(-scores))).astype(dt)
    scores = scores.astype(np.float64)
    fps, tps, _ = binary_clf_curve(scores, labels)
    # roc: leading (0,0), float32 ratios
    tps_r = np.concatenate([[0], tps]).astype(np.float32)
    fps_r = np.concatenate([[0], fps]).astype(np.float32)
    fpr = fps_r / fps_r[-1]
    tpr = tps_r / tps_r[-1]
    # torchmetrics: torch.trapz on the float32 curves (sum order matters at 1e-7)
    import torch

    auroc = float(torch.trapz(torch.from_numpy(tpr), torch.from_numpy(fpr)).item())
    fpr95 = float(fpr[np.where(tpr >= 0.95)[0][0]])
    tps32 = tps.astype(np.float32)
    fps32 = fps.astype(np.float32)
    precision = tps32 / (tps32 + fps32)
    recall = tps32 / tps32[-1]
    precision = np.concatenate([precision[::-1], np.ones(1, dtype=np.float32)])
    recall = np.concatenate([recall[::-1], np.zeros(1, dtype=np.float32)])
    # sklearn.metrics.auc: trapezoid, sign by monotonic direction (recall decreasing)
    aupr = float(-_trap(precision, recall))
    return auroc, fpr95, aupr


# --------------------------------------------------------------------------
# a11 LaREM pipeline on pre-stacked MC samples (inference/image_level.py:115-119)
# --------------------------------------------------------------------------
def larem_pipeline(z, n_mc, pca_components, pca_mean, pca_var, md_mean, md_precision, faithful=False):
    """``get_dl_h_z -> apply_pca_transform -> MDLatentSpace.postprocess`` on
    ``z (N*n_mc, D)``; returns ``(scores (N,), h_z (N,D))``."""
    if faithful:
        _, h = get_dl_h_z(z, n_mc)
    else:
        h = kl_entropy_per_dim_vectorized(z, n_mc)
    y = pca_transform(h, pca_components, pca_mean, pca_var, whiten=True)
    return md_score(y, md_mean, md_precision), h


# --------------------------------------------------------------------------
# f4  remaining logits/features postprocessors (SURVEY 8f "next #4")
# --------------------------------------------------------------------------
def ash_s_linear_layer(x: np.ndarray, percentile: int = 85) -> np.ndarray:
    """inference/funcs.py:234-261: keep the top-k entries per row (k = n - round(n*p/100)), zero the rest,
    scale by exp(sum(row) / sum(kept))."""
    assert x.ndim == 2 and 0 <= percentile <= 100
    s1 = x.sum(axis=1)
    n = x.shape[1]
    k = n - int(np.round(n * percentile / 100.0))
    idx = np.argpartition(x, -k)[:, -k:]
    top_k = np.partition(x, -k)[:, -k:]
    scattered = np.zeros_like(x)
    np.put_along_axis(scattered, indices=idx, values=top_k, axis=1)
    s2 = scattered.sum(axis=1)
    return scattered * np.exp((s1 / s2)[:, None])


def ash_s_defined(x: np.ndarray, percentile: int = 85) -> np.ndarray:
    """ASH-S as defined (each kept activation stays at its own index).  The reference scatters the VALUES of
    ``np.partition`` at the INDICES of ``np.argpartition``; the two selections agree only as sets, so its kept
    values can come out permuted within a row (observed with numpy 2.2 at D = 300) - an implementation artefact
    that no other implementation can reproduce.  Rows where the reference is self-consistent equal this function."""
    assert x.ndim == 2 and 0 <= percentile <= 100
    n = x.shape[1]
    k = n - int(np.round(n * percentile / 100.0))
    if k == 0:
        k = n
    kth = np.partition(x, -k, axis=1)[:, -k][:, None]
    out = np.zeros_like(x)
    for r in range(x.shape[0]):
        gt = np.flatnonzero(x[r] > kth[r, 0])
        eq = np.flatnonzero(x[r] == kth[r, 0])[: k - gt.size]  # ties at the threshold: lowest indices first
        keep = np.concatenate([gt, eq])
        out[r, keep] = x[r, keep]
    s1 = x.sum(axis=1)
    s2 = out.sum(axis=1)
    return out * np.exp((s1 / s2)[:, None])


def linear_energy(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """logsumexp(x @ w.T + b, axis=1) (inference/postprocessors.py:1193-1194, 1441-1442)."""
    return logsumexp(np.matmul(x, w.T) + b, axis=1)


def react_threshold(train_feats: np.ndarray, percentile: int) -> float:
    """inference/postprocessors.py:1437."""
    return np.percentile(train_feats.flatten(), percentile)


def react_score(feats: np.ndarray, w, b, threshold) -> np.ndarray:
    """inference/postprocessors.py:1465-1467 (clip is done in the feature dtype, as numpy==1.26 does)."""
    thr = np.asarray(threshold).astype(feats.dtype)
    return linear_energy(feats.clip(max=thr), w, b)


def dice_masked_weight(train_feats: np.ndarray, w: np.ndarray, p: int) -> np.ndarray:
    """RouteDICE.calculate_mask_weight (inference/funcs.py:173-181) with info = mean feature vector
    (inference/postprocessors.py:1288): weights whose contribution info*w is <= the p-th percentile are zeroed."""
    import torch

    info = torch.Tensor(train_feats).mean(0).numpy()
    contrib = info[None, :] * w
    thresh = np.percentile(contrib, p)
    return (w * (contrib > thresh).astype(np.float32)).astype(np.float32)


def dice_logits(x: np.ndarray, masked_w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """RouteDICE.forward (inference/funcs.py:183-190): vote = x[:, None, :] * masked_w; vote.sum(2) + bias, f32."""
    x = np.asarray(x, dtype=np.float32)
    return (x[:, None, :] * masked_w[None]).sum(axis=2, dtype=np.float32) + b


def generalized_entropy(probs: np.ndarray, gamma: float, M: int) -> np.ndarray:
    """inference/funcs.py:347-375."""
    probs_sorted = np.sort(probs, axis=1)[:, -M:]
    return -np.sum(probs_sorted**gamma * (1 - probs_sorted) ** gamma, axis=1)


def ash_s_conv_defined(x: np.ndarray, percentile: int = 65):
    """ash_s_conv_layer (inference/funcs.py:194-227) on (B, C, H, W) maps: per sample the k = n - round(n * p / 100) largest of
    its C*H*W activations kept (torch.topk + scatter_: each at its own index; ties at the threshold: lowest index first here),
    the sample multiplied by exp(sum before / sum after).  Returns (scaled, pruned): the reference leaves its argument pruned."""
    b = x.shape[0]
    flat = np.asarray(x, dtype=np.float32).reshape(b, -1)
    n = flat.shape[1]
    k = n - int(np.round(n * percentile / 100.0))
    pruned = np.zeros_like(flat)
    if k > 0:
        idx = np.argsort(-flat, axis=1, kind="stable")[:, :k]
        np.put_along_axis(pruned, idx, np.take_along_axis(flat, idx, axis=1), axis=1)
    s1 = flat.sum(axis=1, dtype=np.float32)
    s2 = pruned.sum(axis=1, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        scaled = pruned * np.exp(s1 / s2).astype(np.float32)[:, None]
    return scaled.reshape(x.shape), pruned.reshape(x.shape)


def predictive_uncertainty(logits: np.ndarray, n_mc: int):
    """get_predictive_uncertainty_score (inference/funcs.py:430-465): logits (N * n_mc, C), an image's rows consecutive.
    softmax per row; pred_h = -sum(mean_s p * log(mean_s p)); mi = pred_h - mean_s(-sum(p log p)); float32 as torch (the sums
    here are float64 and rounded once: the reference's float32 sums differ from them by their own rounding, ~1e-7)."""
    x = np.asarray(logits, dtype=np.float32)
    assert x.shape[0] % n_mc == 0
    e = np.exp(x - x.max(axis=1, keepdims=True))
    p = (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
    p3 = p.reshape(-1, n_mc, x.shape[1])
    mean = (p3.sum(axis=1, dtype=np.float32) / np.float32(n_mc)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        pred_h = -(mean * np.log(mean)).sum(axis=1, dtype=np.float64)
        h_s = -(p3 * np.log(p3)).sum(axis=2, dtype=np.float64)
    exp_h = h_s.mean(axis=1)
    return pred_h.astype(np.float32), (pred_h - exp_h).astype(np.float32)


def kde_score_kernel(train: np.ndarray, x: np.ndarray, bandwidth: float, kernel: str) -> np.ndarray:
    """sklearn.neighbors.KernelDensity(kernel=..., bandwidth=h).fit(train).score_samples(x) by its definition (what
    DetectorKDE(kernel=...) forwards to, inference/postprocessors.py:78-128; kernels and normalisation of
    sklearn/neighbors/_binary_tree.pxi.tp): log(sum_i K(|x - t_i| / h)) - log N + log_norm(kernel, d, h); compact kernels
    are 0 from |x - t| >= h on."""
    from scipy.special import gammaln

    train, x = np.asarray(train, np.float64), np.asarray(x, np.float64)
    m, d = train.shape
    h = float(bandwidth)
    dist = np.sqrt(((x[:, None, :] - train[None]) ** 2).sum(-1))
    log_vn = lambda n: 0.5 * n * np.log(np.pi) - gammaln(0.5 * n + 1)
    log_sn = lambda n: np.log(2 * np.pi) + log_vn(n - 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        if kernel == "gaussian":
            kv, factor = np.exp(-0.5 * (dist / h) ** 2), 0.5 * d * np.log(2 * np.pi)
        elif kernel == "tophat":
            kv, factor = (dist < h).astype(np.float64), log_vn(d)
        elif kernel == "epanechnikov":
            kv, factor = np.where(dist < h, 1.0 - (dist * dist) / (h * h), 0.0), log_vn(d) + np.log(2.0 / (d + 2.0))
        elif kernel == "exponential":
            kv, factor = np.exp(-dist / h), log_sn(d - 1) + gammaln(d)
        elif kernel == "linear":
            kv, factor = np.where(dist < h, 1.0 - dist / h, 0.0), log_vn(d) - np.log(d + 1.0)
        elif kernel == "cosine":
            f, tmp = 0.0, 2.0 / np.pi
            for k in range(1, d + 1, 2):
                f += tmp
                tmp *= -(d - k) * (d - k - 1) * (2.0 / np.pi) ** 2
            kv, factor = np.where(dist < h, np.cos(0.5 * np.pi * dist / h), 0.0), np.log(f) + log_sn(d - 1)
        else:
            raise ValueError(kernel)
        return np.log(kv.sum(axis=1)) - np.log(m) - factor - d * np.log(h)


def gen_score(logits: np.ndarray, gamma: float, M: int) -> np.ndarray:
    """GEN.postprocess (inference/postprocessors.py:688-689)."""
    return generalized_entropy(softmax(logits, axis=1), gamma, M)


def vim_setup(train_feats, train_logits, w, b):
    """ViM.setup (inference/postprocessors.py:1044-1066): u = -pinv(W) b; NS = eigenvectors of the
    (assumed-centred) covariance of train - u beyond the DIM largest eigenvalues; alpha matches the scales."""
    u = -np.matmul(np.linalg.pinv(w), b)
    d = train_feats.shape[-1]
    dim = 1000 if d >= 2048 else (512 if d >= 768 else d // 2)
    xc = train_feats - u
    cov = np.dot(xc.T, xc) / xc.shape[0]  # sklearn empirical_covariance(assume_centered=True)
    eig_vals, eig_vecs = np.linalg.eig(cov)
    ns = np.ascontiguousarray((eig_vecs.T[np.argsort(eig_vals * -1)[dim:]]).T)
    vlogit = np.linalg.norm(np.matmul(train_feats - u, ns), axis=-1)
    alpha = train_logits.max(axis=-1).mean() / vlogit.mean()
    return u, ns, alpha


def vim_score(feats, logits, u, ns, alpha):
    """ViM.postprocess (inference/postprocessors.py:1106-1109)."""
    vlogit = np.linalg.norm(np.matmul(feats - u, ns), axis=-1) * alpha
    return -vlogit + logsumexp(logits, axis=-1)


def gmm_energy(gmm, x: np.ndarray) -> np.ndarray:
    """GMMLatentSpace / DDU scoring (inference/postprocessors.py:490-491, 778-779): torch's own
    ``MultivariateNormal.log_prob`` on the host followed by scipy logsumexp over the components."""
    import torch

    return logsumexp(gmm.log_prob(torch.Tensor(np.asarray(x)[:, None, :])).numpy(), axis=1)
