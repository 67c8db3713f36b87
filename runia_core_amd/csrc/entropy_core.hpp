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

// Ascending sort of NP (power of two) registers.  NP = 16: the 60-comparator, 10-layer network (the best
// known size); other sizes: Batcher's odd-even merge sort (19 / 191 / 543 comparators for 8 / 32 / 64).
// Both were checked with the 0-1 principle (all 2^16 inputs; 2e6 random 0-1 vectors for 32 and 64).
template <int NP, typename T>
__device__ __forceinline__ void sort_asc(T (&v)[NP]) {
  if constexpr (NP == 16) {
#define RUNIA_CX(a, b) cswap(v[a], v[b]);
    RUNIA_CX(0, 13) RUNIA_CX(1, 12) RUNIA_CX(2, 15) RUNIA_CX(3, 14) RUNIA_CX(4, 8) RUNIA_CX(5, 6) RUNIA_CX(7, 11) RUNIA_CX(9, 10)
    RUNIA_CX(0, 5) RUNIA_CX(1, 7) RUNIA_CX(2, 9) RUNIA_CX(3, 4) RUNIA_CX(6, 13) RUNIA_CX(8, 14) RUNIA_CX(10, 15) RUNIA_CX(11, 12)
    RUNIA_CX(0, 1) RUNIA_CX(2, 3) RUNIA_CX(4, 5) RUNIA_CX(6, 8) RUNIA_CX(7, 9) RUNIA_CX(10, 11) RUNIA_CX(12, 13) RUNIA_CX(14, 15)
    RUNIA_CX(0, 2) RUNIA_CX(1, 3) RUNIA_CX(4, 10) RUNIA_CX(5, 11) RUNIA_CX(6, 7) RUNIA_CX(8, 9) RUNIA_CX(12, 14) RUNIA_CX(13, 15)
    RUNIA_CX(1, 2) RUNIA_CX(3, 12) RUNIA_CX(4, 6) RUNIA_CX(5, 7) RUNIA_CX(8, 10) RUNIA_CX(9, 11) RUNIA_CX(13, 14)
    RUNIA_CX(1, 4) RUNIA_CX(2, 6) RUNIA_CX(5, 8) RUNIA_CX(7, 10) RUNIA_CX(9, 13) RUNIA_CX(11, 14)
    RUNIA_CX(2, 4) RUNIA_CX(3, 6) RUNIA_CX(9, 12) RUNIA_CX(11, 13)
    RUNIA_CX(3, 5) RUNIA_CX(6, 8) RUNIA_CX(7, 9) RUNIA_CX(10, 12)
    RUNIA_CX(3, 4) RUNIA_CX(5, 6) RUNIA_CX(7, 8) RUNIA_CX(9, 10) RUNIA_CX(11, 12)
    RUNIA_CX(6, 7) RUNIA_CX(8, 9)
#undef RUNIA_CX
  } else {
#pragma unroll
    for (int p = 1; p < NP; p <<= 1) {
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
