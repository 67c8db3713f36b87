"""Setup-time fits on the device (SURVEY 8f "next #1"; opt-in).

``runia_core_amd.config.device_fit = True`` moves

* the covariance of ``MDLatentSpace.setup`` and ``mahalanobis_preprocess`` (``np.cov(X.T, bias=1)`` inside sklearn
  ``EmpiricalCovariance``, reference ``inference/postprocessors.py:217-220`` / ``inference/funcs.py:62-66``) to the f64
  matrix cores (``runia_covariance_*``),
* ``scipy.linalg.pinvh`` to the hand-written Jacobi eigen-solver (blocked form ``runia_eigh_block_*``, ``csrc/eigh_block.hip``;
  scalar-rotation form ``runia_eigh_*``, ``csrc/eigh.hip``) with SciPy's
  cut-off rule and a device matrix product,
* the PCA fit of ``apply_pca_ds_split`` to the device: ``svd_solver="covariance_eigh" | "full"`` = covariance + the same
  eigen-solver + sklearn's sign convention (``svd_flip(u_based_decision=False)``); the reference's default
  ``svd_solver="randomized"`` = sklearn's randomized range finder carried out in covariance space with the SAME random
  test matrix (drawn from NumPy's global generator exactly as sklearn draws it), see ``pca_fit_randomized_device``.

* round 6: ``gmm_fit`` (class means, class covariances, float32 Cholesky factors under the reference's jitter ladder,
  ``inference/funcs.py:265-344``): rows grouped by label on the device, one ``runia_covariance_f32in`` per class, all classes'
  factorisations in one ``runia_cholesky_f32`` launch per ladder step (``gmm_fit_device``).

No vendor solver is involved.  The default (False) keeps the reference's own host calls, so fitted state is
bit-identical to the reference.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _hip

__all__ = ["empirical_precision_device", "pinvh_device", "pca_fit_device", "pca_fit_randomized_device", "FittedPCA", "gmm_fit_device",
           "vim_null_space_device", "percentile_flat"]


_PINVH_CHOLESKY_FROM = 128      # below: the eigen-decomposition is a few launches anyway
_PINVH_CHOLESKY_MAX_COND = 1e8  # bound on ||A||_F ||A^-1||_F up to which the Cholesky route is taken


def pinvh_device(cov: torch.Tensor) -> torch.Tensor:
    """``scipy.linalg.pinvh(cov)``: eigen-decomposition, eigenvalues with ``|s| <= max|s| * max(M,N) * eps``
    dropped, ``(U / s) @ U^T``.

    A symmetric positive definite matrix whose smallest eigenvalue is far above that cut-off loses no direction: its pseudo-inverse IS
    its inverse, ``W^T W`` with ``W = L^-1`` from the Cholesky factor (``runia_cholesky_f64`` + ``runia_tril_inverse_f64`` + one
    product: 2048 x 2048 in ~5 ms, the Jacobi eigen-decomposition takes 156).  Taken when the factorisation succeeds AND
    ``||A||_F ||A^-1||_F < 1e8`` - an upper bound of the 2-norm condition number, so ``lambda_min > 1e-8 lambda_max``, eight orders
    above SciPy's cut-off; everything else (rank-deficient, indefinite to rounding, badly conditioned) takes the eigen route."""
    n = cov.shape[0]
    if n >= _PINVH_CHOLESKY_FROM:
        sym = ((cov + cov.T) * 0.5).contiguous()
        tril, info = _hip.cholesky(sym)
        w = _hip.tril_inverse(tril.unsqueeze(0))[0]
        inv = _hip.matmul_f64(w.T.contiguous(), w)
        failed, na, ni = torch.stack([info.reshape(-1)[0].to(torch.float64), torch.linalg.norm(sym), torch.linalg.norm(inv)]).tolist()
        if failed == 0 and na * ni < _PINVH_CHOLESKY_MAX_COND:  # (a NaN compares false)
            return inv
    s, u = _hip.eigh(cov)
    cutoff = s.abs().max() * (max(cov.shape) * torch.finfo(cov.dtype).eps)
    keep = s.abs() > cutoff
    u = u[:, keep].contiguous()
    return _hip.matmul_f64((u * (1.0 / s[keep])).contiguous(), u, transpose_b=True)


def empirical_precision_device(x) -> np.ndarray:
    """``EmpiricalCovariance(assume_centered=False).fit(x).precision_`` computed on the GPU -> f64 ndarray."""
    dtype = torch.float32 if getattr(x, "dtype", None) in (np.float32, torch.float32) else torch.float64
    xd = _hip.to_device(x, dtype)
    _, cov = _hip.covariance(xd)
    return _hip.to_host(pinvh_device(cov))


class FittedPCA:
    """What ``apply_pca_ds_split`` returns for a device fit: the public attributes of a fitted sklearn ``PCA`` that
    the path (and user code) reads, plus ``transform``."""

    def __init__(self, components, mean, explained_variance, whiten, n_samples, total_var):
        self.components_ = components
        self.mean_ = mean
        self.explained_variance_ = explained_variance
        self.whiten = whiten
        self.n_components = self.n_components_ = components.shape[0]
        self.n_features_in_ = components.shape[1]
        self.n_samples_ = n_samples
        self.explained_variance_ratio_ = explained_variance / total_var
        self.singular_values_ = np.sqrt(explained_variance * (n_samples - 1))
        self.svd_solver = "covariance_eigh"

    def transform(self, x):
        from .dimensionality_reduction import apply_pca_transform

        return apply_pca_transform(x, self)


def pca_fit_device(samples, n_components: int, whiten: bool = True) -> FittedPCA:
    """sklearn ``PCA(n_components, svd_solver="covariance_eigh").fit`` on the GPU (sklearn ``_pca.py::_fit_full``):
    covariance with ``ddof = 1``, symmetric eigen-decomposition, eigenvalues descending and clamped at 0, components =
    eigenvectors with the sign that makes the largest-magnitude entry of every component positive."""
    on_dev = isinstance(samples, torch.Tensor) and samples.is_cuda
    x = samples if on_dev else np.asarray(samples)
    n, d = x.shape
    if not 0 < n_components <= min(n, d):
        raise ValueError(f"n_components={n_components} must be between 0 and min(n_samples, n_features)={min(n, d)}")
    xd = x if on_dev else _hip.to_device(x, torch.float32 if x.dtype == np.float32 else torch.float64)
    mean, cov = _hip.covariance(xd)                  # bias = 1 (divided by n)
    cov = cov * (n / (n - 1.0))
    w, v = _hip.eigh(cov)
    w = torch.flip(w, dims=(0,)).clamp_min(0.0)
    vt = torch.flip(v, dims=(1,)).T.contiguous()      # rows = components, descending eigenvalue
    idx = vt.abs().argmax(dim=1, keepdim=True)
    signs = torch.sign(torch.gather(vt, 1, idx))
    signs[signs == 0] = 1.0
    vt = vt * signs
    total = float(w.sum().item())
    return FittedPCA(_hip.to_host(vt[:n_components]), _hip.to_host(mean), _hip.to_host(w[:n_components]), whiten, n, total)


def _inv_sqrt_spd(t: torch.Tensor) -> torch.Tensor:
    """``T^(-1/2)`` of a symmetric positive semi-definite matrix through the Jacobi solver; directions below the numerical
    rank are dropped."""
    w, v = _hip.eigh(t)
    floor = w.max() * (t.shape[0] * torch.finfo(t.dtype).eps)
    inv = torch.where(w > floor, 1.0 / torch.sqrt(w.clamp_min(floor)), torch.zeros_like(w))
    return _hip.matmul_f64((v * inv).contiguous(), v, transpose_b=True)


def _orthonormalise(y: torch.Tensor) -> torch.Tensor:
    """A well-conditioned basis of the column span of ``y`` for the next step of the subspace iteration (only the span enters the
    result: see ``pca_fit_randomized_device``).  ``y L^-T`` with the Cholesky factor of the Gram matrix ``y^T y = L L^T`` - one
    factorisation + one triangular inverse (``runia_cholesky_f64`` / ``runia_tril_inverse_f64``) where the symmetric form
    ``y (y^T y)^(-1/2)`` takes a Jacobi eigen-decomposition (266 columns: ~1 ms against 5-8 ms, seven times per fit).  Columns
    orthonormal to ``eps * cond(y)^2``; a Gram matrix without a factor, or with a pivot ratio below 1e-12 (a rank-deficient block:
    constant features), takes the symmetric form, which drops the directions below the numerical rank."""
    g = _hip.matmul_f64(y.T.contiguous(), y)
    tril, info = _hip.cholesky(g)
    diag = torch.diagonal(tril)
    failed, lo, hi = torch.stack([info.reshape(-1)[0].to(torch.float64), diag.min(), diag.max()]).tolist()  # one read-back
    if failed == 0 and lo > 1e-6 * hi:  # (pivots are square roots; a NaN compares false)
        return _hip.matmul_f64(y, _hip.tril_inverse(tril.unsqueeze(0))[0], transpose_b=True)
    return _hip.matmul_f64(y, _inv_sqrt_spd(g))


def pca_fit_randomized_device(samples, n_components: int, whiten: bool = True, n_oversamples: int = 10) -> FittedPCA:
    """sklearn ``PCA(n_components, svd_solver="randomized").fit`` (what ``apply_pca_ds_split`` calls by default,
    reference ``dimensionality_reduction.py:70-71``) on the GPU, reproducing sklearn's result for the same state of NumPy's
    global generator.

    sklearn (``utils/extmath.py::_randomized_svd``, tall data): Q = random normal ``(n_features, k + 10)``; ``n_iter`` times
    Q <- normalise(M Q), Q <- normalise(M^T Q); Q <- qr(M Q); SVD of ``B = Q^T M``.  The normalisations only re-condition a
    basis, so the outcome is a function of ``span(S^n_iter Omega)`` with ``S = M^T M``: every step is carried out on the
    ``n_features x (k + 10)`` side with S alone - ONE pass over the data (the covariance kernel on the matrix cores), then
    small products and Jacobi eigen-decompositions:
        Z <- orth(S Z) x n_iter;  T = Z^T S Z;  C = T^(-1/2) (S Z)^T (S Z) T^(-1/2) = U diag(s^2) U^T;
        Vt = diag(1/s) U^T T^(-1/2) (S Z)^T.
    Agreement with sklearn: components to ~1e-11 on the test spectra (tests/test_api_gpu.py).  Data with fewer samples than
    features (sklearn transposes the problem and draws a different matrix) is not covered: the caller falls back to sklearn."""
    on_dev = isinstance(samples, torch.Tensor) and samples.is_cuda   # rows already in HBM (the harness's device-resident sweep)
    x = samples if on_dev else np.asarray(samples)
    n, d = x.shape
    if not 0 < n_components <= min(n, d):
        raise ValueError(f"n_components={n_components} must be between 0 and min(n_samples, n_features)={min(n, d)}")
    if n < d:
        raise ValueError("pca_fit_randomized_device: n_samples < n_features is fitted by sklearn (transposed problem)")
    size = min(n_components + n_oversamples, d)
    omega = np.random.normal(size=(d, n_components + n_oversamples))[:, :size]  # the draw sklearn makes (random_state=None)
    n_iter = 7 if n_components < 0.1 * min(n, d) else 4
    xd = x if on_dev else _hip.to_device(x, torch.float32 if x.dtype == np.float32 else torch.float64)
    mean, cov = _hip.covariance(xd)           # bias = 1
    s_mat = (cov * float(n)).contiguous()     # S = M^T M of the centred rows
    z = _hip.to_device(omega, torch.float64)
    for _ in range(n_iter):
        z = _orthonormalise(_hip.matmul_f64(s_mat, z))
    sz = _hip.matmul_f64(s_mat, z)                                      # (d, size)
    t_ih = _inv_sqrt_spd(_hip.matmul_f64(z.T.contiguous(), sz))         # (Z^T S Z)^(-1/2)
    szt = _hip.matmul_f64(sz, t_ih)                                     # S Z T^(-1/2)
    w, u = _hip.eigh(_hip.matmul_f64(szt.T.contiguous(), szt))          # = B B^T of sklearn's B = Q^T M
    w = torch.flip(w, dims=(0,)).clamp_min(0.0)
    u = torch.flip(u, dims=(1,)).contiguous()
    sing = torch.sqrt(w)
    vt = _hip.matmul_f64(u.T.contiguous(), szt.T.contiguous())          # rows = s_i * v_i^T
    vt = vt / sing.clamp_min(torch.finfo(torch.float64).tiny).unsqueeze(1)
    idx = vt.abs().argmax(dim=1, keepdim=True)
    signs = torch.sign(torch.gather(vt, 1, idx))
    signs[signs == 0] = 1.0
    vt = vt * signs
    u = (u * signs.reshape(1, -1)).contiguous()  # svd_flip flips the columns of U with the rows of Vt
    total_var = float(torch.diagonal(s_mat).sum().item()) / (n - 1.0)
    fitted = FittedPCA(_hip.to_host(vt[:n_components]), _hip.to_host(mean), _hip.to_host((w[:n_components] / (n - 1.0))),
                       whiten, n, total_var)
    fitted.svd_solver = "randomized"
    # sklearn's fit_transform returns U (scaled), not transform(X): with an approximate SVD the two differ in the trailing
    # components.  U = Q Uhat = M (Z T^(-1/2) Uhat), so the training rows are projected with that matrix instead of V.
    proj = _hip.matmul_f64(_hip.matmul_f64(z, t_ih), u[:, :n_components].contiguous())       # (d, k)
    scale = np.full(n_components, 1.0 / (n - 1.0)) if whiten else 1.0 / np.maximum(_hip.to_host(sing[:n_components]), 1e-300) ** 2
    fitted._train_projection = (_hip.to_host(proj.T.contiguous()), scale)
    return fitted


_PERCENTILE_DEVICE_FROM = 1 << 22  # elements from which the device select pays (NumPy: ~6.5 ns per element)


def _numpy_linear_percentile_plan(n: int, q, dtype):
    """What ``np.percentile(a, q)`` (default ``method="linear"``) does with a sorted 1-D array of ``n`` elements of ``dtype``:
    ``(previous index, next index, finish(a[previous], a[next]) -> result)``, built from NumPy's OWN index arithmetic
    (``numpy.lib._function_base_impl``: the virtual index, its neighbours, gamma and the interpolation run in the array's dtype
    - for float32 the index itself is a float32 product, so beyond 2^24 elements it is not the textbook ``(n - 1) q / 100``).
    Private NumPy names: any surprise (a missing name, another signature, an array-valued q) returns None and the caller takes
    ``np.percentile`` itself - the result is NumPy's either way, only the time differs.  tests/test_abi_and_host.py compares the
    plan with ``np.percentile`` on the installed NumPy."""
    try:
        from numpy.lib import _function_base_impl as F

        methods = F._QuantileMethods["linear"]
        quant = np.asanyarray(np.true_divide(q, dtype(100)))
        if quant.ndim != 0 or not (0.0 <= float(quant) <= 1.0):
            return None
        vi = np.asanyarray(methods["get_virtual_index"](n, quant))
        if np.issubdtype(vi.dtype, np.integer):
            k = int(vi) % n
            return k, k, (lambda a, b: dtype(a))
        prev, nxt = F._get_indexes(np.empty((0,), dtype=dtype), vi, n)
        gamma = F._get_gamma(vi, prev, methods)
        prev, nxt = int(prev) % n, int(nxt) % n

        def finish(a, b):
            return F._lerp(dtype(a), dtype(b), gamma.reshape(()))

        return prev, nxt, finish
    except Exception:  # noqa: BLE001 - see the docstring: NumPy's public call is the fallback
        return None


def percentile_flat(a, q):
    """``np.percentile(np.asarray(a).flatten(), q)`` (ReAct / DICE+ReAct thresholds, reference
    ``inference/postprocessors.py:1441, 1466``).  Large float32 arrays where device fits are on: the two order statistics NumPy
    interpolates between come from a radix select on the device (``_hip.kth_smallest_flat``: 102 M activations 0.68 s -> a few
    ms + the upload), NumPy's own arithmetic does the rest; same value, bit for bit.  Everything else: NumPy's call."""
    from . import config as _config

    arr = np.asarray(a)
    if arr.dtype != np.float32 or arr.size < _PERCENTILE_DEVICE_FROM or arr.size >= (1 << 32) or not _config.use_device_fit():
        return np.percentile(arr.flatten(), q)
    plan = _numpy_linear_percentile_plan(int(arr.size), q, np.float32)
    if plan is None:
        return np.percentile(arr.flatten(), q)
    prev, nxt, finish = plan
    xd = _hip.to_device(arr if arr.flags.c_contiguous else np.ascontiguousarray(arr), torch.float32)
    if bool(torch.isnan(xd).any()):  # NumPy returns NaN (with its own warning): its call decides
        return np.percentile(arr.flatten(), q)
    lo, hi = (_hip.kth_smallest_flat(xd, [prev]) * 2) if prev == nxt else _hip.kth_smallest_flat(xd, [prev, nxt])
    return finish(lo, hi)


def vim_null_space_device(train, u: np.ndarray, dim: int) -> np.ndarray:
    """ViM's residual space (reference ``inference/postprocessors.py:1051-1058``): eigenvectors of the second moment of
    ``train - u`` (sklearn ``EmpiricalCovariance(assume_centered=True)``) beyond the ``dim`` largest eigenvalues, columns in
    descending eigenvalue order -> ``(D, D - dim)`` host array in the features' dtype (float32 rows give a float32 basis, as
    NumPy's ``eig`` of a float32 matrix does upstream).  The moment about ``u`` is ``cov + (mean - u)(mean - u)^T`` with the
    bias-1 covariance and the column means of ``runia_covariance_*`` (f64 accumulation on the matrix cores); the decomposition is
    the symmetric Jacobi solver, so the basis is orthonormal and real by construction (upstream's general solver can return a
    complex one)."""
    f32 = getattr(train, "dtype", None) in (np.float32, torch.float32)
    xd = _hip.to_device(train, torch.float32 if f32 else torch.float64)
    mean, cov = _hip.covariance(xd)
    delta = mean - _hip.to_device(np.asarray(u, dtype=np.float64).reshape(-1), torch.float64)
    vals, vecs = _hip.eigh(cov + torch.outer(delta, delta))          # ascending
    order = torch.argsort(vals, descending=True, stable=True)[dim:]
    ns = _hip.to_host(vecs.index_select(1, order).contiguous())
    return np.ascontiguousarray(ns.astype(np.float32) if f32 else ns)


_GMM_JITTERS = (0.0,) + tuple(10.0 ** e for e in range(-20, 0))


def gmm_fit_device(embeddings, labels, num_classes: int):
    """``gmm_fit`` on the device -> ``(loc [C', D] f32, scale_tril [C', D, D] f32, jitter)`` as DEVICE tensors, classes without
    samples left out (``C'`` present classes in label order).

    The reference (``inference/funcs.py:265-344``) takes, in float32 torch on the host: the class means, the covariance of every
    class's centred rows over ``max(n_c, 2) - 1``, and the smallest jitter of ``0, 1e-20, ..., 1e-1`` for which
    ``MultivariateNormal(covariance_matrix=cov + jitter I)`` can be built, i.e. for which every class has a float32 Cholesky
    factor.  Here: rows grouped by label with one gather, mean and covariance of a class from ``runia_covariance_f32in``
    (accumulated in f64 on the matrix cores, then rounded to f32 - closer to the exact moments than a float32 accumulation), and
    the ladder walked with ``runia_cholesky_f32`` on all classes at once: float32 arithmetic like torch's LAPACK call, so a
    covariance that is indefinite to float32 rounding fails here as it does there (the summation order inside a dot product
    differs: a pivot within rounding of zero can fall the other way; ``tests/test_api_gpu.py`` compares jitter and factors with the
    host fit on the reference-run fixtures)."""
    on_dev = isinstance(embeddings, torch.Tensor) and embeddings.is_cuda
    if not on_dev:
        x = embeddings.detach().cpu().numpy() if isinstance(embeddings, torch.Tensor) else np.asarray(embeddings)
    lab = labels.detach().cpu().numpy() if isinstance(labels, torch.Tensor) else np.asarray(labels)
    lab = lab.reshape(-1).astype(np.int64)
    valid = (lab >= 0) & (lab < num_classes)
    counts = np.bincount(lab[valid], minlength=num_classes)
    order = np.argsort(np.where(valid, lab, num_classes), kind="stable")[: int(valid.sum())]
    xd = embeddings.detach().to(torch.float32) if on_dev else _hip.to_device(np.ascontiguousarray(x, dtype=np.float32), torch.float32)
    xs = xd.index_select(0, _hip.to_device(order, torch.int64).to(xd.device))  # rows grouped by class (a gather, no arithmetic)
    del xd
    means, covs, start = [], [], 0
    for c in range(num_classes):
        n_c = int(counts[c])
        if n_c == 0:
            continue
        mean, cov = _hip.covariance(xs[start: start + n_c])           # f64, cov = sum / n_c
        means.append(mean.to(torch.float32))
        covs.append((cov * (n_c / (max(n_c, 2) - 1.0))).to(torch.float32))
        start += n_c
    loc, cov = torch.stack(means), torch.stack(covs)
    chosen, factor = _GMM_JITTERS[-1], None
    diag = cov.diagonal(dim1=-2, dim2=-1)
    tried = None  # the diagonal of the last matrix that was factorised
    for jitter in _GMM_JITTERS:
        # a jitter below half an ulp of every diagonal entry leaves the float32 matrix as it was (1e-20 ... 1e-8 against entries of
        # 0.1 ... 1): the factorisation would fail at the same pivot again - those ladder steps are skipped, the outcome is the same
        # (cfg3-sized DDU: 15 factorisations of ten 2048 x 2048 matrices -> 2)
        shifted = diag + torch.tensor(jitter, dtype=diag.dtype, device=diag.device)
        if tried is not None and bool((shifted == tried).all()):
            continue
        tried = shifted
        tril, info = _hip.cholesky(cov, jitter)
        if int(info.abs().max()) == 0 and bool(torch.isfinite(tril).all()):
            chosen, factor = jitter, tril
            break
    if factor is None:  # nothing on the ladder is positive definite: the reference's last attempt raises as well
        raise RuntimeError("gmm_fit: no jitter of the ladder makes every class covariance positive definite (cholesky failed)")
    return loc, factor, (0 if chosen == 0.0 else chosen)
