// Register-level core of the 1-D Kozachenko-Leonenko entropy (shared by entropy.hip and fused.hip):
// a sorting network over NP floats and the k-th-nearest-neighbour window scan in f64.
#pragma once
#include "common.hpp"

namespace runia_entropy {

constexpr double kInf = __builtin_inf();

// One compare-exchange = v_min_f32 + v_max_f32.  Written as instructions because fminf/fmaxf make the
// compiler canonicalise every operand it cannot prove quiet (72 extra `v_max_f32 v, v, v` per 16-sort).
// A NaN operand is dropped in favour of the other one, as with fminf/fmaxf.
__device__ __forceinline__ void cswap(float& a, float& b) {
  float lo, hi;
  asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  a = lo;
  b = hi;
}
__device__ __forceinline__ void cswap(double& a, double& b) {
  double lo, hi;
  asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
  asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
  a = lo;
  b = hi;
}

// One 3-sorter = v_min3_f32 + v_med3_f32 + v_max3_f32: three instructions order three registers, where compare-exchanges
// need three pairs = six (min / max / med3 all issue at the same ~4.1 cycles on gfx950, tools/microbench/valu_issue.hip).
// A NaN operand is dropped in favour of another operand (v_med3_f32 returns the minimum then), as with cswap: callers
// catch NaN samples before sorting.
__device__ __forceinline__ void sort3(float& a, float& b, float& c) {
  float lo, md, hi;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(a), "v"(b), "v"(c));
  asm("v_med3_f32 %0, %1, %2, %3" : "=v"(md) : "v"(a), "v"(b), "v"(c));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(a), "v"(b), "v"(c));
  a = lo;
  b = md;
  c = hi;
}
__device__ __forceinline__ void sort3(double& a, double& b, double& c) {  // no 3-input f64 min / max on gfx950
  cswap(a, b);
  cswap(b, c);
  cswap(a, b);
}

// Ascending sort of NP (power of two) registers.
// NP = 16: 28 three-sorters = 84 instructions (the best-known compare-exchange network, 60 comparators in 10 layers, is
// 120); NP = 8: 8 three-sorters = 24 instructions (19 comparators = 38).  Both networks were found by
// tools/search/sorter3.cpp (mutate a valid network of 2- and 3-sorters, keep the mutant if it still sorts every 0/1
// input and costs no more) and are re-checked over all 2^NP 0/1 inputs by tests/test_abi_and_host.py (0-1 principle:
// min3 / med3 / max3 are monotone, so it carries over).  NP = 32: two 16-blocks by that network + a searched merging
// network (2 * 84 + 100 = 268 instructions; Batcher's 32-sorter: 191 comparators = 382); NP = 64: the same on both halves
// + Batcher's last merge stage; smaller sizes: Batcher's sort.
template <int OFF, int NP, typename T>
__device__ __forceinline__ void sort16_at(T (&v)[NP]) {
#define RUNIA_S3(a, b, c) sort3(v[OFF + a], v[OFF + b], v[OFF + c]);
  RUNIA_S3(6, 7, 15) RUNIA_S3(10, 12, 13) RUNIA_S3(0, 2, 3) RUNIA_S3(1, 4, 5) RUNIA_S3(9, 11, 14)
  RUNIA_S3(3, 5, 15) RUNIA_S3(8, 13, 14) RUNIA_S3(2, 4, 7) RUNIA_S3(8, 10, 11) RUNIA_S3(9, 10, 12)
  RUNIA_S3(5, 7, 14) RUNIA_S3(0, 1, 6) RUNIA_S3(11, 12, 13) RUNIA_S3(3, 4, 6) RUNIA_S3(1, 2, 3)
  RUNIA_S3(5, 6, 13) RUNIA_S3(0, 8, 9) RUNIA_S3(1, 8, 9) RUNIA_S3(3, 7, 11) RUNIA_S3(13, 14, 15)
  RUNIA_S3(11, 12, 13) RUNIA_S3(4, 8, 11) RUNIA_S3(5, 7, 9) RUNIA_S3(2, 6, 10) RUNIA_S3(6, 7, 8)
  RUNIA_S3(3, 5, 6) RUNIA_S3(2, 3, 4) RUNIA_S3(9, 10, 11)
#undef RUNIA_S3
}

// Merge of two ascending 16-blocks at OFF .. OFF+15 and OFF+16 .. OFF+31: 30 three-sorters + 5 compare-exchanges = 100
// instructions (Batcher's odd-even merge: 65 comparators = 130), found by tools/search/sorter3.cpp in merge mode (valid
// iff it orders every pair of sorted 0/1 halves - 17 x 17 inputs - re-checked by tests/test_abi_and_host.py).
template <int OFF, int NP, typename T>
__device__ __forceinline__ void merge16x2_at(T (&v)[NP]) {
#define RUNIA_S3(a, b, c) sort3(v[OFF + a], v[OFF + b], v[OFF + c]);
#define RUNIA_S2(a, b) cswap(v[OFF + a], v[OFF + b]);
  RUNIA_S2(10, 26) RUNIA_S3(14, 26, 30) RUNIA_S3(12, 26, 28) RUNIA_S3(4, 12, 20) RUNIA_S3(0, 4, 16)
  RUNIA_S3(1, 4, 17) RUNIA_S3(3, 17, 19) RUNIA_S3(2, 10, 18) RUNIA_S3(15, 30, 31) RUNIA_S3(6, 10, 22)
  RUNIA_S3(5, 17, 21) RUNIA_S3(2, 3, 4) RUNIA_S3(13, 21, 29) RUNIA_S3(7, 19, 23) RUNIA_S3(11, 19, 27)
  RUNIA_S3(9, 21, 25) RUNIA_S3(7, 9, 17) RUNIA_S3(8, 20, 24) RUNIA_S3(15, 23, 27) RUNIA_S3(6, 7, 16)
  RUNIA_S3(11, 13, 17) RUNIA_S3(14, 18, 22) RUNIA_S3(8, 12, 16) RUNIA_S3(15, 19, 21) RUNIA_S3(9, 10, 12)
  RUNIA_S2(5, 6) RUNIA_S3(21, 22, 24) RUNIA_S3(18, 19, 20) RUNIA_S3(23, 25, 26) RUNIA_S3(15, 17, 18)
  RUNIA_S3(27, 28, 29) RUNIA_S3(14, 15, 16) RUNIA_S2(23, 24) RUNIA_S2(13, 14) RUNIA_S2(11, 12)
#undef RUNIA_S2
#undef RUNIA_S3
}

template <int NP, typename T>
__device__ __forceinline__ void sort_asc(T (&v)[NP]) {
  static_assert((NP & (NP - 1)) == 0, "power of two");
  if constexpr (NP == 8) {
#define RUNIA_S3(a, b, c) sort3(v[a], v[b], v[c]);
    RUNIA_S3(0, 5, 6) RUNIA_S3(1, 2, 3) RUNIA_S3(4, 6, 7) RUNIA_S3(0, 1, 4) RUNIA_S3(2, 5, 6) RUNIA_S3(3, 6, 7) RUNIA_S3(3, 4, 5) RUNIA_S3(1, 2, 3)
#undef RUNIA_S3
  } else {
    // 16-blocks by the 3-sorter network, then (and for NP < 16 from the start) Batcher's merge stages
    if constexpr (NP >= 16) {
      sort16_at<0>(v);
      if constexpr (NP >= 32) sort16_at<16>(v);
      if constexpr (NP >= 64) { sort16_at<32>(v); sort16_at<48>(v); }
      static_assert(NP <= 64, "add the 16-blocks of a larger sort here");
    }
    if constexpr (NP >= 32) {
      merge16x2_at<0>(v);
      if constexpr (NP >= 64) merge16x2_at<32>(v);
    }
#pragma unroll
    for (int p = (NP >= 32 ? 32 : (NP >= 16 ? 16 : 1)); p < NP; p <<= 1) {
#pragma unroll
      for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
        for (int j = k % p; j + k < NP; j += 2 * k) {
#pragma unroll
          for (int i = 0; i < k; ++i) {
            if (i + j + k < NP && (i + j) / (2 * p) == (i + j + k) / (2 * p)) cswap(v[i + j], v[i + j + k]);
          }
        }
      }
    }
  }
}

// a * b + c with c held in a scalar register pair (one rounding, as fma)
__device__ __forceinline__ double fma_scalar_addend(double a, double b, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
}

// log(f) and the binary exponent e of a positive, finite, normal x = f * 2^e with f in [sqrt(1/2), sqrt(2)):
// log(f) = 2 atanh(s), s = (f-1)/(f+1), |s| <= 0.1716, odd series to s^21 (next term < 3e-17); plain f64 operations,
// ~35 instructions against ~75 of the library log (which carries double-double terms this sum does not need: the
// result enters h as (log(f) + e ln 2) / n_mc, absolute error ~2e-16).
__device__ __forceinline__ double log_mantissa(double x, int& e_out) {
  int e;
  double f = frexp(x, &e);                      // [0.5, 1)
  const bool low = f < 0.70710678118654752440;
  f = low ? f + f : f;
  e_out = low ? e - 1 : e;
  const double num = f - 1.0, den = f + 1.0;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  r = fma(fma(-den, r, 1.0), r, r);
  double q = num * r;
  q = fma(fma(-den, q, num), r, q);             // (f-1)/(f+1) to ~0.5 ulp
  const double t = q * q;
  // Horner steps with the coefficient as a SCALAR operand of v_fma_f64: as C++ literals the compiler turns every step
  // into v_fmac_f64 and first materialises its coefficient with two v_mov_b32 (20 vector instructions per thread); the
  // scalar unit builds the constants for free
  double p = 1.0 / 21.0;
  p = fma_scalar_addend(p, t, 1.0 / 19.0);
  p = fma_scalar_addend(p, t, 1.0 / 17.0);
  p = fma_scalar_addend(p, t, 1.0 / 15.0);
  p = fma_scalar_addend(p, t, 1.0 / 13.0);
  p = fma_scalar_addend(p, t, 1.0 / 11.0);
  p = fma_scalar_addend(p, t, 1.0 / 9.0);
  p = fma_scalar_addend(p, t, 1.0 / 7.0);
  p = fma_scalar_addend(p, t, 1.0 / 5.0);
  p = fma_scalar_addend(p, t, 1.0 / 3.0);
  const double q2 = q + q;
  return fma(q2 * t, p, q2);
}

// sum_i log(2*max(eps_i, min_dist)) over the first n entries of one ascending column (entries >= n are +inf
// pads; FULL promises n == NP and removes every run-time guard).
//   eps_i = k-th-nearest-neighbour distance of rank i = min_j max(v[i]-v[i-j], v[i+K-j]-v[i]),  j = 0..K,
// with the differences of the f32 samples taken in f64 (exact; the reference promotes to f64 before its tree
// query) and each one formed once: gap[a][m-1] = v[a+m] - v[a].  The n logarithms collapse into one:
// groups of four eps are multiplied raw (no overflow: each < 2^129), split by frexp, and the mantissas
// and exponents accumulated separately.
template <int NP, int K, bool FULL = false>
__device__ __forceinline__ double column_log_sum(const float (&vs)[NP], int n, double min_dist) {
  double v[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) v[i] = (double)vs[i];
  double gap[NP][K];
#pragma unroll
  for (int a = 0; a < NP; ++a)
#pragma unroll
    for (int m = 1; m <= K; ++m) gap[a][m - 1] = (a + m < NP) ? v[a + m] - v[a] : kInf;
  double mant = 1.0;
  int esum = 0;
#pragma unroll
  for (int i0 = 0; i0 < NP; i0 += 4) {
    double prod = 1.0;
#pragma unroll
    for (int i = i0; i < i0 + 4 && i < NP; ++i) {
      double e = kInf;
      bool first = true;  // compile-time after unrolling: the first window's candidate is taken as it is (fmin(+inf, x)
                          // cannot be folded by the compiler - x might be NaN - and cost one v_min_f64 per sample)
#pragma unroll
      for (int j = 0; j <= K; ++j) {
        const int li = i - j, ri = i + (K - j);
        if (li < 0 || ri >= NP) continue;           // that window leaves the column
        double cand;
        if (j == 0) cand = gap[i][K - 1];           // gaps of an ascending column are >= 0
        else if (j == K) cand = gap[li][K - 1];
        else cand = fmax(gap[li][j - 1], gap[i][K - j - 1]);
        e = first ? cand : fmin(e, cand);
        first = false;
      }
      e = fmax(e, min_dist);
      if (FULL || i < n) prod *= e;
    }
    int ex;
    const double m = frexp(prod, &ex);
    mant *= m;
    esum += ex;
  }
  int em;
  const double lf = log_mantissa(mant, em);
  return fma((double)(esum + em + (FULL ? NP : n)), 0.69314718055994530942, lf);  // + n: the factors 2 of log(2*eps)
}

}  // namespace runia_entropy
