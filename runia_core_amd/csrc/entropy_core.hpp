// Register-level core of the 1-D Kozachenko-Leonenko entropy (shared by entropy.hip and fused.hip):
// bitonic sort of NP floats and the k-th-nearest-neighbour window scan in f64.
#pragma once
#include "common.hpp"

namespace runia_entropy {

constexpr double kInf = __builtin_inf();

template <int NP>
__device__ __forceinline__ void bitonic_sort_asc(float (&v)[NP]) {
#pragma unroll
  for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = ((i & k) == 0);
          const float a = v[i], b = v[l];
          const float lo = fminf(a, b), hi = fmaxf(a, b);
          v[i] = up ? lo : hi;
          v[l] = up ? hi : lo;
        }
      }
    }
  }
}

// sum_i log(2*max(eps_i, min_dist)) for one sorted column (entries >= n are +inf pads)
template <int NP, int K>
__device__ __forceinline__ double column_log_sum(const float (&vs)[NP], int n, double min_dist) {
  double v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = (double)vs[i];
  double mant = 1.0;
  int esum = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i < n) {
      double e = kInf;
#pragma unroll
      for (int j = 0; j <= K; ++j) {
        const int li = i - j, ri = i + (K - j);
        double L, R;
        if (j == 0) L = 0.0; else if (li >= 0) L = v[i] - v[li]; else L = kInf;
        if (K - j == 0) R = 0.0; else if (ri < NP) R = v[ri] - v[i]; else R = kInf;
        e = fmin(e, fmax(L, R));
      }
      e = fmax(e, min_dist);
      int ex;
      const double m = frexp(2.0 * e, &ex);
      mant *= m;
      esum += ex;
    }
  }
  return log(mant) + (double)esum * 0.69314718055994530942;
}


}  // namespace runia_entropy
