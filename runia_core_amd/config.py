"""Process-wide switches of runia_core_amd (all default to the reference's behaviour)."""

# Where setup() fits run (see runia_core_amd.device_fit): covariance on the f64 matrix cores, pinvh / the PCA solvers /
# ViM's eigen-decomposition on the hand-written Jacobi eigen-solver - or the reference's own host calls (scikit-learn,
# SciPy, NumPy: fitted state bit-identical to the reference's).
#   None (default)  the device when one is there (every reference golden and reference-run fixture holds at the 1e-5
#                   contract under it: tests/test_api_gpu.py runs them under both; cfg3's setup 6.3 s -> 0.5 s, the
#                   harness's PCA refit sweep 7.3 s -> 0.26 s), the host calls on a box without a GPU
#   True / False    always the device (raises without one) / always the host calls (the documented opt-out)
device_fit = None


def use_device_fit() -> bool:
    if device_fit is not None:
        return bool(device_fit)
    import torch

    return bool(torch.cuda.is_available())


# kNN on large problems (>= 512 queries x 4 096 bank rows x 256 features, >= 2^31 multiply-adds): rank the bank rows by
# distances from bf16 piece products on the matrix cores (csrc/knn_bf16.hip) before the exact f32 re-measurement.  The
# scores are the same bits either way; False keeps the f32 matrix-core kernel (by handing the entry point the smaller
# workspace).
knn_bf16_candidates = True

# BASELINE config 1 ("MSP postprocessor, 10k test images on CPU, no GPU"): there is NO silent CPU fallback anywhere in
# this package - scoring without a GPU raises RuniaHipError.  This switch is the explicit exception for the logits
# family only (MSP, Energy): set it to True on a box without a GPU and their scores come from the reference's own host
# calls (scipy.special.softmax / logsumexp, inference/postprocessors.py:549, 606).  Ignored where a GPU is present
# (the HIP kernels always run there); every other postprocessor still raises without one.
host_logits_without_gpu = False

# MDLatentSpace on wide (un-reduced) features, n >= 512: score = -|| W (x - mu) ||^2 with the lower-triangular factor W of the
# precision (precision = W^T W, device Cholesky) instead of -(x - mu) P (x - mu)^T - about half the multiply-adds (the kernel
# skips the zero half of W).  Falls back to the P form by itself when the precision has no such factor (rank-deficient pinvh,
# not symmetric, badly conditioned).  The choice depends on the fitted state alone, never on the batch.  False: always the P form.
md_triangular = True

# ViM.setup: the eigen-decomposition of the training covariance.  False (default): the reference's own host calls (sklearn
# EmpiricalCovariance + np.linalg.eig, float32 for float32 features) - its scores carry that solver's rounding, and the 1e-5
# contract with the reference-run fixtures holds.  True (opt-in): second moment about u on the f64 matrix cores + the Jacobi
# eigen-solver (device_fit.vim_null_space_device): 2048-d features 17.9 s -> 0.3 s per fit; the residual norm then comes from the
# exact covariance's null space and sits up to ~1.5e-5 (relative) from the reference's float32 one on tests/golden/ref_f4.npz -
# INTEGRATION.md, "Known divergences".
vim_device_fit = False
