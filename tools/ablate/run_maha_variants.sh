#!/bin/bash
# Mahalanobis (262 144 x 2048 f32 rows, 10 classes) on variants of gemm_f64.hip built with tools/ablate/build_lib_variant.sh:
#   tools/ablate/build_lib_variant.sh gemm_f64.hip maha_noepi -DMAHA_ABLATE=1     (class-term epilogue compiled out: timing only)
#   tools/ablate/build_lib_variant.sh gemm_f64.hip maha_ring0 -DGEMM_RING=0       (one pair of weight look-ahead instead of three)
#   tools/ablate/build_lib_variant.sh gemm_f64.hip maha_rt4 -DMAHA_RT=4           (64-row tiles, one wave per SIMD)
for lib in hip maha_noepi maha_ring0 maha_rt4; do
  echo "== $lib"
  RUNIA_LIB=$GRAFT_REPO_ROOT/runia_core_amd/librunia_$lib.so python tools/ablate/run_maha.py 262144 2048 10
done
