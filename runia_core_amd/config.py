"""Process-wide switches of runia_core_amd (all default to the reference's behaviour)."""

# Fit covariance / precision matrices of setup() on the GPU (see runia_core_amd.device_fit).
device_fit = False

# kNN on large problems (>= 512 queries x 4 096 bank rows x 256 features, >= 2^31 multiply-adds): rank the bank rows by
# distances from bf16 piece products on the matrix cores (csrc/knn_bf16.hip) before the exact f32 re-measurement.  The
# scores are the same bits either way; False keeps the f32 matrix-core kernel (by handing the entry point the smaller
# workspace).
knn_bf16_candidates = True
