// a2: Kozachenko-Leonenko kNN entropy of MC-dropout samples.
// Replaces the per-image / per-dimension loops of get_dl_h_z
// (reference evaluation/entropy.py:41-93) over entropy_estimators.continuous.get_h
// (k-d tree query of k+1 neighbours, max-norm, min_dist clip, psi(n)-psi(k)+(d/n)*sum log(2 eps)).
//
// 1-D case: the k-th nearest-neighbour distance of the element at sorted rank i is
//   min_{j=0..k} max(v[i]-v[i-j], v[i+k-j]-v[i])      (out of range -> +inf)
// so each (image, dim) column is a register sort of n_mc floats plus a k+1-wide
// window scan; differences are taken in f64 (exact for f32 inputs, as in the
// reference which promotes to f64 before the tree query).  sum_i log(2 eps_i) is
// evaluated as log(prod mantissas) + ln2 * sum exponents: one f64 log per column.
//
// HBM-bound by design: 4*n_mc bytes in, 8 bytes out per column; loads are 16 B per
// lane (4 adjacent dims), 1 KiB contiguous per wave instruction.
#include "common.hpp"
#include "entropy_core.hpp"

namespace {

using namespace runia_entropy;

// VEC adjacent dims per thread (VEC = 4 -> float4 loads, VEC = 1 -> scalar)
template <int NP, int K, int VEC>
__global__ __launch_bounds__(256) void entropy_per_dim_kernel(const float* __restrict__ z,
                                                               double* __restrict__ h, int64_t N, int n,
                                                               int64_t D, double min_dist,
                                                               double const_term, double inv_n) {
  const int64_t DV = D / VEC;
  const int64_t total = N * DV;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
    const int64_t img = g / DV;
    const int64_t c = g - img * DV;
    const float* base = z + (img * n) * D + c * VEC;
    float v[VEC][NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
      if (s < n) {
        if constexpr (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(base + (int64_t)s * D);
          v[0][s] = t.x; v[1][s] = t.y; v[2][s] = t.z; v[3][s] = t.w;
        } else {
          v[0][s] = base[(int64_t)s * D];
        }
      } else {
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q][s] = INFINITY;
      }
    }
    double out[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      // a NaN sample makes the reference's entropy NaN; the min/max sort would drop it, so it is caught first (a sum is
      // NaN iff a term is: the +inf pads of a short column cannot cancel finite samples)
      float probe = v[q][0];
#pragma unroll
      for (int t = 1; t < NP; ++t) probe += v[q][t];
      sort_asc<NP>(v[q]);
      out[q] = const_term + inv_n * column_log_sum<NP, K>(v[q], n, min_dist);
      if (probe != probe) out[q] = NAN;
    }
    double* dst = h + img * D + c * VEC;
    if constexpr (VEC == 4) {
      reinterpret_cast<double2*>(dst)[0] = make_double2(out[0], out[1]);
      reinterpret_cast<double2*>(dst)[1] = make_double2(out[2], out[3]);
    } else {
      dst[0] = out[0];
    }
  }
}

// Any 2 <= n <= 64, any 1 <= k < n: thread-private column in LDS, insertion sort.
__global__ __launch_bounds__(64) void entropy_per_dim_generic_kernel(const float* __restrict__ z,
                                                                      double* __restrict__ h, int64_t N,
                                                                      int n, int64_t D, int k,
                                                                      double min_dist, double const_term,
                                                                      double inv_n) {
  __shared__ float col[64][64];  // [sample][thread]
  const int t = threadIdx.x;
  const int64_t total = N * D;
  for (int64_t g = (int64_t)blockIdx.x * 64 + t; g < total; g += (int64_t)gridDim.x * 64) {
    const int64_t img = g / D;
    const int64_t c = g - img * D;
    const float* base = z + (img * n) * D + c;
    bool has_nan = false;
    for (int s = 0; s < n; ++s) {  // insertion sort while loading
      const float x = base[(int64_t)s * D];
      has_nan = has_nan || (x != x);
      int p = s;
      while (p > 0 && col[p - 1][t] > x) {
        col[p][t] = col[p - 1][t];
        --p;
      }
      col[p][t] = x;
    }
    double mant = 1.0;
    int esum = 0;
    for (int i = 0; i < n; ++i) {
      const double vi = (double)col[i][t];
      double e = kInf;
      for (int j = 0; j <= k; ++j) {
        const int li = i - j, ri = i + (k - j);
        const double L = (j == 0) ? 0.0 : (li >= 0 ? vi - (double)col[li][t] : kInf);
        const double R = (k - j == 0) ? 0.0 : (ri < n ? (double)col[ri][t] - vi : kInf);
        e = fmin(e, fmax(L, R));
      }
      e = fmax(e, min_dist);
      int ex;
      mant *= frexp(2.0 * e, &ex);
      esum += ex;
      if ((i & 15) == 15) {  // keep the mantissa product far from underflow
        int e2;
        mant = frexp(mant, &e2);
        esum += e2;
      }
    }
    h[img * D + c] = has_nan ? NAN : const_term + inv_n * (log(mant) + (double)esum * 0.69314718055994530942);
  }
}

// Joint (D-dimensional, Chebyshev) entropy: one workgroup per image.
//   task = (sample pair a<b, slice of the staged dims): n = 16 gives 120 pairs x 2 slices = 240 busy lanes.
//   The image's samples are staged through LDS in chunks of dims, already promoted to f64 (the reference promotes
//   before its tree query), so the inner step is two 8-byte LDS reads + v_add_f64 + v_max_f64(|.|).
//   Slices of a pair meet in dist[a][b] through an LDS max on the bit pattern (distances are >= 0).
// LDS is sized by n at launch: tile n x (chunk+2) + dist n x (n+1) doubles (18.8 KB at n = 16: 8 images per CU).
constexpr int kJointChunk = 128;  // dims per staged chunk
constexpr int kJointTasks = 8;    // tasks per thread at most (n = 64: 2016 pairs)
__global__ __launch_bounds__(256) void entropy_joint_kernel(const float* __restrict__ z,
                                                             double* __restrict__ h_mvn, int64_t N, int n,
                                                             int64_t D, int k, double min_dist,
                                                             double const_term, double d_over_n) {
  extern __shared__ __attribute__((aligned(16))) double joint_lds[];
  constexpr int TP = kJointChunk + 2;           // even pitch: 16-byte aligned rows, 4-bank row offset
  double* tile = joint_lds;                     // [n][TP]
  double* dist = joint_lds + (size_t)n * TP;    // [n][n + 1]
  const int DP = n + 1;
  const int tid = threadIdx.x;
  const int npairs = n * (n - 1) / 2;
  const int slices = (npairs >= 256) ? 1 : 256 / npairs;
  const int ntasks = npairs * slices;
  // this thread's tasks: t = tid, tid + 256, ...  -> (a, b, slice); fixed for the whole launch
  int ta[kJointTasks], tb[kJointTasks], ts[kJointTasks];
#pragma unroll
  for (int q = 0; q < kJointTasks; ++q) {
    const int t = tid + 256 * q;
    ta[q] = -1; tb[q] = 0; ts[q] = 0;
    if (t < ntasks) {
      const int p = t / slices;
      ts[q] = t - p * slices;
      int a = 0, rem = p;
      while (rem >= n - 1 - a) { rem -= n - 1 - a; ++a; }
      ta[q] = a;
      tb[q] = a + 1 + rem;
    }
  }
  __shared__ int nan_seen;
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    const float* base = z + img * n * D;
    if (tid == 0) nan_seen = 0;  // ordered before the first staging pass by its barrier
    bool my_nan = false;
    double best[kJointTasks];
#pragma unroll
    for (int q = 0; q < kJointTasks; ++q) best[q] = 0.0;
    for (int i = tid; i < n * DP; i += 256) dist[i] = 0.0;
    for (int64_t d0 = 0; d0 < D; d0 += kJointChunk) {
      const int w = (int)((D - d0 < kJointChunk) ? (D - d0) : kJointChunk);
      __syncthreads();
      for (int i = tid; i < n * w; i += 256) {
        const int s = i / w, j = i - s * w;
        const float xv = base[(int64_t)s * D + d0 + j];
        my_nan = my_nan || (xv != xv);
        tile[s * TP + j] = (double)xv;
      }
      __syncthreads();
      const int per = ((w + slices - 1) / slices + 3) & ~3;  // multiple of 4: slices start 32-byte aligned
#pragma unroll
      for (int q = 0; q < kJointTasks; ++q) {
        if (ta[q] >= 0) {
          const double* ra = tile + ta[q] * TP;
          const double* rb = tile + tb[q] * TP;
          const int j0 = ts[q] * per, j1 = (j0 + per < w) ? j0 + per : w;
          double m0 = best[q], m1 = 0.0, m2 = 0.0, m3 = 0.0;
          int j = j0;
          for (; j + 4 <= j1; j += 4) {  // rows are 16-byte aligned (even pitch): two ds_read_b128 per row
            const double2 a01 = *reinterpret_cast<const double2*>(ra + j), a23 = *reinterpret_cast<const double2*>(ra + j + 2);
            const double2 b01 = *reinterpret_cast<const double2*>(rb + j), b23 = *reinterpret_cast<const double2*>(rb + j + 2);
            m0 = fmax(m0, fabs(a01.x - b01.x));
            m1 = fmax(m1, fabs(a01.y - b01.y));
            m2 = fmax(m2, fabs(a23.x - b23.x));
            m3 = fmax(m3, fabs(a23.y - b23.y));
          }
          for (; j < j1; ++j) m0 = fmax(m0, fabs(ra[j] - rb[j]));
          best[q] = fmax(fmax(m0, m1), fmax(m2, m3));
        }
      }
    }
#pragma unroll
    for (int q = 0; q < kJointTasks; ++q) {
      if (ta[q] >= 0) {  // non-negative doubles order like their bit patterns
        const unsigned long long bits = (unsigned long long)__double_as_longlong(best[q]);
        atomicMax(reinterpret_cast<unsigned long long*>(&dist[ta[q] * DP + tb[q]]), bits);
        atomicMax(reinterpret_cast<unsigned long long*>(&dist[tb[q] * DP + ta[q]]), bits);
      }
    }
    if (my_nan) nan_seen = 1;  // a NaN sample makes the reference's joint entropy NaN; fmax would drop it
    __syncthreads();
    // sample i: k-th smallest distance to the others (selection by counting, n <= 64)
    double logsum = 0.0;
    if (tid < n) {
      const int i = tid;
      double kth = kInf;
      for (int c = 0; c < n; ++c) {
        if (c == i) continue;
        const double dc = dist[i * DP + c];
        int less = 0, leq = 0;
        for (int o = 0; o < n; ++o) {
          if (o == i) continue;
          const double d_o = dist[i * DP + o];
          less += (d_o < dc);
          leq += (d_o <= dc);
        }
        if (less < k && k <= leq) kth = dc;  // dc is the k-th order statistic (1-based)
      }
      logsum = log(2.0 * fmax(kth, min_dist));
    }
    // reduce over the first wave (n <= 64 lanes)
    if (tid < 64) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) logsum += shfl_xor_f64(logsum, o);
      if (tid == 0) h_mvn[img] = nan_seen ? NAN : const_term + d_over_n * logsum;
    }
    __syncthreads();
  }
}

// ---- joint entropy, register form (n <= 32, D a multiple of the vector width, aligned rows) --------------------
// One workgroup per image, one thread per VEC adjacent dims: the thread reads its dims of ALL n samples straight from
// global memory (16-byte loads, 1 KB of a sample row per wave-instruction), promotes them to f64 once, and forms the
// partial Chebyshev distance of every sample pair over its own dims in registers (VEC subtractions + VEC-1 maxima per
// pair; the LDS form above reads both rows of a pair from LDS for every pair - 16 bytes of LDS traffic per pair and
// dim, which is what bounded it).  Sixteen pair maxima at a time are then reduced over the wave by a halving exchange:
// lanes i and i^32 split the 16 slots between them (v_permlane32_swap + v_max_f64 per slot kept), then i^16
// (v_permlane16_swap), i^8 and i^4 (DPP row rotations), leaving one slot per lane quad; two quad-permute butterfly
// steps finish it.  ~65 vector instructions per 16 pairs instead of 16 x 6 x 3, none through the LDS crossbar.
// Distances are maxima of exact differences: the same bits as the LDS form whatever the order.  The k-th neighbour of
// a sample is entry k-1 of its sorted row of distances (60-comparator network on f64 registers; the counting selection
// of the LDS form is 2 n^2 dependent LDS reads per sample).
// 10 000 images x 16 samples x 512 dims: 0.477 ms (LDS form) -> 0.132 ms; 8 samples x 2048 dims reads 6.0 TB/s.
// (two dims per thread at n = 16: 0.187 ms; three waves per SIMD forced, 168 registers + 384 B of scratch: 0.245 ms.)
#ifndef JOINT_VEC16
#define JOINT_VEC16 4
#endif
template <int NP> constexpr int joint_vec() { return NP <= 8 ? 4 : (NP <= 16 ? JOINT_VEC16 : 2); }
__host__ __device__ constexpr int joint_pair_a(int np, int p) {
  int a = 0;
  while (p >= np - 1 - a) { p -= np - 1 - a; ++a; }
  return a;
}
__host__ __device__ constexpr int joint_pair_b(int np, int p) {
  int a = 0;
  while (p >= np - 1 - a) { p -= np - 1 - a; ++a; }
  return a + 1 + p;
}

__device__ __forceinline__ double max_f64(double a, double b) {  // IEEE maxNum, no canonicalisation of the operands
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_abs_f64(double m, double d) {  // max(m, |d|)
  double r;
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(m), "v"(d));
  return r;
}
__device__ __forceinline__ double max_abs2_f64(double a, double b) {  // max(|a|, |b|)
  double r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// DPP move of an f64 (two 32-bit moves): lanes whose bank (group of four lanes in a row of 16) is in BANKS take the
// value the pattern CTRL brings them, the others keep `old`.  Vector-ALU latency, no trip through the LDS crossbar.
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E;          // quad_perm:[1,0,3,2], [2,3,0,1]
constexpr int kDppRowRor4 = 0x124, kDppRowRor8 = 0x128, kDppRowRor12 = 0x12C;  // lane i <- lane (i - n) mod 16 of its row
template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, 0xf, BANKS, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, 0xf, BANKS, false);
  return __hiloint2double(hi, lo);
}
// lanes i, i+32: the lower lane ends with max over both of slot `a`, the upper lane with that of slot `b` (in `a`)
__device__ __forceinline__ void halve32(double& a, const double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = max_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
}
__device__ __forceinline__ void halve16(double& a, const double b) {  // the same for lanes i, i+16 (rows of 16 lanes)
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = max_f64(__hiloint2double((int)hi[0], (int)lo[0]), __hiloint2double((int)hi[1], (int)lo[1]));
}
// 16 per-lane values -> their maxima over the wave; lane L ends with slot (L >> 2) & 15 (all four lanes of a quad)
__device__ __forceinline__ double wave_max16(double (&v)[16], int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) halve32(v[j], v[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; ++j) halve16(v[j], v[j + 4]);
  const bool u8 = (lane & 8) != 0, u4 = (lane & 4) != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // lanes i, i ^ 8: a rotation of the row by 8 is that exchange
    const double send = u8 ? v[j] : v[j + 2], keep = u8 ? v[j + 2] : v[j];
    v[j] = max_f64(keep, dpp_f64<kDppRowRor8>(send, send));
  }
  {  // lanes i, i ^ 4: banks 1 and 3 take lane i - 4, banks 0 and 2 lane i + 4 (= i - 12 in the row)
    const double send = u4 ? v[0] : v[1], keep = u4 ? v[1] : v[0];
    double recv = dpp_f64<kDppRowRor4, 0xA>(send, send);
    recv = dpp_f64<kDppRowRor12, 0x5>(recv, send);
    v[0] = max_f64(keep, recv);
  }
  double r = v[0];
  r = max_f64(r, dpp_f64<kDppQuadXor2>(r, r));
  r = max_f64(r, dpp_f64<kDppQuadXor1>(r, r));
  return r;
}

// K1D > 0: the same pass also emits the per-dimension entropies h [N, D] of get_dl_h_z (evaluation/entropy.py:77-82): the thread
// holds every sample of its VEC dims anyway - one register sort + window scan per dim with entropy_per_dim_kernel's own
// column code (same bits), so the (N * n_mc, D) samples are read ONCE for both outputs of the call.
template <int NP, int K1D = 0>
__global__ __launch_bounds__(256) void entropy_joint_reg_kernel(const float* __restrict__ z,
                                                                 double* __restrict__ h_mvn, int64_t N, int n,
                                                                 int64_t D, int k, double min_dist,
                                                                 double const_term, double d_over_n,
                                                                 double* __restrict__ h_dim = nullptr, double inv_n = 0.0) {
  constexpr int VEC = joint_vec<NP>();
  constexpr int NPAIRS = NP * (NP - 1) / 2, NGROUPS = (NPAIRS + 15) / 16;
  constexpr int DP = NP + 1;
  typedef float fvec __attribute__((ext_vector_type(VEC)));
  __shared__ double dist[NP * DP];
  __shared__ double red[4 * NGROUPS * 16];  // per wave: the reduced pair maxima, [group][slot]
  __shared__ int nan_seen;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    const float* base = z + img * n * D;
    for (int i = tid; i < NP * DP; i += blockDim.x) dist[i] = 0.0;
    if (tid == 0) nan_seen = 0;
    bool my_nan = false;
    double acc[NGROUPS];
#pragma unroll
    for (int g = 0; g < NGROUPS; ++g) acc[g] = 0.0;
    for (int64_t d0 = 0; d0 < D; d0 += (int64_t)blockDim.x * VEC) {
      // No predicates: a lane past the end of the row re-reads dims 0.., a sample slot past n re-reads sample n-1.
      // Maxima are idempotent (a dim counted twice changes nothing) and pairs with a slot >= n are never read back.
      const int64_t dd = d0 + (int64_t)tid * VEC;
      const int64_t d = (dd < D) ? dd : 0;  // D % VEC == 0
      double x[NP][VEC];
      if constexpr (K1D > 0) {
        fvec raw[NP];
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          const int sc = (s < n) ? s : n - 1;  // wave-uniform
          raw[s] = *reinterpret_cast<const fvec*>(base + (int64_t)sc * D + d);
        }
        if (dd < D) {  // (a lane past the end of the row re-read dims 0..: it has nothing to write)
          double out[VEC];
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            float col[NP];
#pragma unroll
            for (int s = 0; s < NP; ++s) col[s] = (s < n) ? raw[s][q] : INFINITY;
            float probe = col[0];  // NaN sample -> NaN entropy, as entropy_per_dim_kernel
#pragma unroll
            for (int t = 1; t < NP; ++t) probe += col[t];
            sort_asc<NP>(col);
            out[q] = const_term + inv_n * column_log_sum<NP, K1D>(col, n, min_dist);
            if (probe != probe) out[q] = NAN;
          }
          double* dst = h_dim + img * D + dd;
#pragma unroll
          for (int q = 0; q < VEC; q += 2) *reinterpret_cast<double2*>(dst + q) = make_double2(out[q], out[q + 1]);
        }
#pragma unroll
        for (int s = 0; s < NP; ++s)
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            my_nan = my_nan || (raw[s][q] != raw[s][q]);
            x[s][q] = (double)raw[s][q];
          }
      } else {
#pragma unroll
        for (int s = 0; s < NP; ++s) {
          const int sc = (s < n) ? s : n - 1;  // wave-uniform
          const fvec t = *reinterpret_cast<const fvec*>(base + (int64_t)sc * D + d);
#pragma unroll
          for (int q = 0; q < VEC; ++q) {
            my_nan = my_nan || (t[q] != t[q]);
            x[s][q] = (double)t[q];
          }
        }
      }
      {
        // pairs (a, b), a < b, in row-major order; sixteen at a time go through the wave reduction.  Both loops unroll
        // fully, so the pair counter and every register index are compile-time constants.
        double v[16];
        int p = 0;
#pragma unroll
        for (int a = 0; a < NP - 1; ++a) {
#pragma unroll
          for (int b = a + 1; b < NP; ++b) {
            double m = max_abs2_f64(x[a][0] - x[b][0], x[a][1] - x[b][1]);
#pragma unroll
            for (int q = 2; q < VEC; ++q) m = max_abs_f64(m, x[a][q] - x[b][q]);
            v[p & 15] = m;
            if ((p & 15) == 15 || p == NPAIRS - 1) {
#pragma unroll
              for (int j = (p & 15) + 1; j < 16; ++j) v[j] = 0.0;  // short last group: 0 is neutral (distances >= 0)
              acc[p >> 4] = max_f64(acc[p >> 4], wave_max16(v, lane));
            }
            ++p;
          }
        }
      }
    }
    if ((lane & 3) == 0) {
#pragma unroll
      for (int g = 0; g < NGROUPS; ++g) red[(tid >> 6) * (NGROUPS * 16) + 16 * g + (lane >> 2)] = acc[g];
    }
    __syncthreads();  // dist zeroed, red written
    for (int p = tid; p < NPAIRS; p += blockDim.x) {
      int a = 0, rem = p;
      while (rem >= NP - 1 - a) { rem -= NP - 1 - a; ++a; }
      const int b = a + 1 + rem;
      if (b < n) {
        double m = red[p];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmax(m, red[w * (NGROUPS * 16) + p]);
        dist[a * DP + b] = m;
        dist[b * DP + a] = m;
      }
    }
    if (my_nan) nan_seen = 1;  // a NaN sample makes the reference's joint entropy NaN; the maxima would drop it
    __syncthreads();
    double logsum = 0.0;
    if (tid < NP) {  // sample slot i: its row of distances into registers, sorted there; the k-th smallest is entry k-1
      double r[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) r[c] = (c < n && c != tid) ? dist[tid * DP + c] : kInf;
      sort_asc<NP>(r);
      double kth = r[0];
#pragma unroll
      for (int c = 1; c < NP - 1; ++c) kth = (c == k - 1) ? r[c] : kth;
      if (tid < n) logsum = log(2.0 * fmax(kth, min_dist));
    }
    if (tid < 64) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) logsum += shfl_xor_f64(logsum, o);
      if (tid == 0) h_mvn[img] = nan_seen ? NAN : const_term + d_over_n * logsum;
    }
    __syncthreads();
  }
}

// ---- joint entropy, pair-group form (round 6; 9 <= n <= 16 samples, rows of whole aligned pairs of dims) ------------------------
// The register form above gives every thread ALL 120 sample pairs of its own dims and reduces sixteen pair maxima at a time over
// the wave (no room for 120 f64 accumulators per lane): at 16 samples x 512 dims the reductions are 1 040 of the 4 140 vector
// instructions an image costs (two waves), and the f64 subtract + maximum per (pair, dim) are 1 920.  Here ONE wave takes an
// image and its two halves of 32 lanes split the PAIRS: every lane keeps 60 accumulators over all passes of the row and the
// cross-lane reduction happens once per image.  Both halves run the same instructions on the same register numbers, so the split
// is a property of the DATA: the 60 register pairs are the edges of a self-complementary graph on 16 vertices (four blocks of
// four registers A B C D; edges: inside A, inside D, A-B, B-C, C-D), the lower half loads sample i into register i, the upper half
// sample sigma(i) with sigma: A -> B -> D -> C -> A blockwise - the image of the graph under sigma is its complement, so the two
// halves together visit each of the 120 sample pairs exactly once.  Distances are maxima of exact f64 differences: the same bits
// as the other two forms whatever the order.  10 000 x 16 x 512: 0.1325 -> 0.114 ms.  (With the per-dimension entropies of
// get_dl_h_z formed in the same pass - the single-read kernel, entropy_joint_reg_kernel<16, 5> - this form measured 0.236 ms
// against 0.174: 60 accumulators + a 16-input sort do not fit 256 registers; the single-read call keeps the register form.)
__device__ __forceinline__ int pair16_sigma(int i) { return i < 4 ? i + 4 : (i < 8 ? i + 8 : (i < 12 ? i - 8 : i - 4)); }
__device__ const unsigned char kPair16A[64] = {0, 0, 0, 1, 1, 2, 12, 12, 12, 13, 13, 14, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                               4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10,
                                               11, 11, 11, 11, 0, 0, 0, 0};
__device__ const unsigned char kPair16B[64] = {1, 2, 3, 2, 3, 3, 13, 14, 15, 14, 15, 15, 4, 5, 6, 7, 4, 5, 6, 7, 4, 5, 6, 7, 4, 5, 6, 7,
                                               8, 9, 10, 11, 8, 9, 10, 11, 8, 9, 10, 11, 8, 9, 10, 11, 12, 13, 14, 15, 12, 13, 14, 15,
                                               12, 13, 14, 15, 12, 13, 14, 15, 0, 0, 0, 0};
// 16 per-lane values -> their maxima over the 32 lanes that share lane bit 5; lane L ends with slot
// 8 * bit4(L) + 4 * bit3(L) + 2 * bit2(L) + bit1(L) (both lanes of a pair L, L ^ 1)
__device__ __forceinline__ double half_max16(double (&v)[16], int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) halve16(v[j], v[j + 8]);
  const bool u8 = (lane & 8) != 0, u4 = (lane & 4) != 0, u2 = (lane & 2) != 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {  // lanes i, i ^ 8
    const double send = u8 ? v[j] : v[j + 4], keep = u8 ? v[j + 4] : v[j];
    v[j] = max_f64(keep, dpp_f64<kDppRowRor8>(send, send));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // lanes i, i ^ 4
    const double send = u4 ? v[j] : v[j + 2], keep = u4 ? v[j + 2] : v[j];
    double recv = dpp_f64<kDppRowRor4, 0xA>(send, send);
    recv = dpp_f64<kDppRowRor12, 0x5>(recv, send);
    v[j] = max_f64(keep, recv);
  }
  {  // lanes i, i ^ 2
    const double send = u2 ? v[0] : v[1], keep = u2 ? v[1] : v[0];
    v[0] = max_f64(keep, dpp_f64<kDppQuadXor2>(send, send));
  }
  return max_f64(v[0], dpp_f64<kDppQuadXor1>(v[0], v[0]));
}

__global__ __launch_bounds__(64) void entropy_joint_pair16_kernel(const float* __restrict__ z, double* __restrict__ h_mvn, int64_t N,
                                                                  int n, int64_t D, int k, double min_dist, double const_term,
                                                                  double d_over_n) {
  constexpr int NP = 16, DP = NP + 1;
  __shared__ double dist[NP * DP];
  const int lane = threadIdx.x, grp = lane >> 5, l = lane & 31;
  // register i of this lane holds sample src(i) (clamped to the last real sample: pairs with a slot >= n are never read back)
  unsigned roff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int src = grp ? pair16_sigma(i) : i;
    roff[i] = (unsigned)((src < n ? src : n - 1) * D);
  }
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    const float* base = z + img * n * D;
    for (int i = lane; i < NP * DP; i += 64) dist[i] = 0.0;
    double acc[64];
#pragma unroll
    for (int p = 0; p < 64; ++p) acc[p] = 0.0;
    bool my_nan = false;
    int pass = 0;
    for (int64_t d0 = 0; d0 < D; d0 += 64, ++pass) {
      const int64_t dd = d0 + 2 * l;
      const int64_t d = (dd < D) ? dd : 0;  // (a lane past the end of the row re-reads dims 0, 1: maxima are idempotent)
      float2 raw[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) raw[i] = *reinterpret_cast<const float2*>(base + roff[i] + d);
      double x[NP][2];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        my_nan = my_nan || (raw[i].x != raw[i].x) || (raw[i].y != raw[i].y);
        x[i][0] = (double)raw[i].x;
        x[i][1] = (double)raw[i].y;
      }
      // the 60 register pairs (edges of the self-complementary graph), compile-time indices
      int p = 0;
#define RUNIA_PAIR(a, b)                                                                         \
  acc[p] = max_f64(acc[p], max_abs2_f64(x[a][0] - x[b][0], x[a][1] - x[b][1]));                 \
  ++p;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = a + 1; b < 4; ++b) { RUNIA_PAIR(a, b) }
#pragma unroll
      for (int a = 12; a < 15; ++a)
#pragma unroll
        for (int b = a + 1; b < 16; ++b) { RUNIA_PAIR(a, b) }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 4; b < 8; ++b) { RUNIA_PAIR(a, b) }
#pragma unroll
      for (int a = 4; a < 8; ++a)
#pragma unroll
        for (int b = 8; b < 12; ++b) { RUNIA_PAIR(a, b) }
#pragma unroll
      for (int a = 8; a < 12; ++a)
#pragma unroll
        for (int b = 12; b < 16; ++b) { RUNIA_PAIR(a, b) }
#undef RUNIA_PAIR
    }
    // once per image: the accumulators over the 32 lanes of the half, sixteen slots at a time
    const int slot_in_group = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // dist is zeroed
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      double v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = acc[16 * g + j];
      const double m = half_max16(v, lane);
      const int pidx = 16 * g + slot_in_group;
      if (pidx < 60 && (lane & 1) == 0) {
        const int ia = kPair16A[pidx], ib = kPair16B[pidx];
        const int a = grp ? pair16_sigma(ia) : ia, b = grp ? pair16_sigma(ib) : ib;
        if (a < n && b < n) {
          dist[a * DP + b] = m;
          dist[b * DP + a] = m;
        }
      }
    }
    const bool any_nan = __ballot(my_nan) != 0ull;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    double logsum = 0.0;
    if (lane < NP) {  // sample slot i: its row of distances into registers, sorted there; the k-th smallest is entry k-1
      double r[NP];
#pragma unroll
      for (int c = 0; c < NP; ++c) r[c] = (c < n && c != lane) ? dist[lane * DP + c] : kInf;
      sort_asc<NP>(r);
      double kth = r[0];
#pragma unroll
      for (int c = 1; c < NP - 1; ++c) kth = (c == k - 1) ? r[c] : kth;
      if (lane < n) logsum = log(2.0 * fmax(kth, min_dist));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) logsum += shfl_xor_f64(logsum, o);
    if (lane == 0) h_mvn[img] = any_nan ? NAN : const_term + d_over_n * logsum;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();  // dist is rewritten by the next image
  }
}

int next_pow2(int n) {
  int p = 4;
  while (p < n) p <<= 1;
  return p;
}

double digamma_diff(int n, int k) {  // psi(n) - psi(k) = sum_{j=k}^{n-1} 1/j for integers
  double s = 0.0;
  for (int j = n - 1; j >= k; --j) s += 1.0 / (double)j;
  return s;
}

template <int NP, int K>
void launch_per_dim(const float* z, double* h, int64_t N, int n, int64_t D, double min_dist, double ct,
                    double inv_n, hipStream_t s) {
  const bool vec = (NP <= 16) && ((D & 3) == 0) && ((((uintptr_t)z) & 15) == 0) && ((((uintptr_t)h) & 15) == 0);
  if constexpr (NP <= 16) {
    if (vec) {
      entropy_per_dim_kernel<NP, K, 4><<<runia_stream_grid(N * (D / 4), 256), 256, 0, s>>>(
          z, h, N, n, D, min_dist, ct, inv_n);
      return;
    }
  }
  entropy_per_dim_kernel<NP, K, 1><<<runia_stream_grid(N * D, 256), 256, 0, s>>>(z, h, N, n, D, min_dist,
                                                                                  ct, inv_n);
}

}  // namespace

extern "C" int runia_kl_entropy_per_dim_f32(const float* z, double* h, int64_t N, int n_mc, int64_t D, int k,
                                            double min_dist, runia_stream_t stream) {
  if (N < 0 || D <= 0 || n_mc < 2 || n_mc > 64 || k < 1 || k >= n_mc || (N > 0 && (!z || !h)))
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  hipStream_t s = as_stream(stream);
  const double ct = digamma_diff(n_mc, k);
  const double inv_n = 1.0 / (double)n_mc;
  const int np = next_pow2(n_mc);
  bool done = true;
  if (np == 4 && k == 1) launch_per_dim<4, 1>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 4 && k == 2) launch_per_dim<4, 2>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 4 && k == 3) launch_per_dim<4, 3>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 8 && k == 4) launch_per_dim<8, 4>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 8 && k == 5) launch_per_dim<8, 5>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 16 && k == 5) launch_per_dim<16, 5>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 32 && k == 5) launch_per_dim<32, 5>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else if (np == 64 && k == 5) launch_per_dim<64, 5>(z, h, N, n_mc, D, min_dist, ct, inv_n, s);
  else done = false;
  if (!done)
    entropy_per_dim_generic_kernel<<<runia_stream_grid(N * D, 64), 64, 0, s>>>(z, h, N, n_mc, D, k, min_dist,
                                                                               ct, inv_n);
  return runia_check_launch();
}

extern "C" int runia_kl_entropy_joint_f32(const float* z, double* h_mvn, int64_t N, int n_mc, int64_t D,
                                          int k, double min_dist, runia_stream_t stream) {
  if (N < 0 || D <= 0 || n_mc < 2 || n_mc > 64 || k < 1 || k >= n_mc || (N > 0 && (!z || !h_mvn)))
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  const double ct = digamma_diff(n_mc, k);
  const double d_over_n = (double)D / (double)n_mc;
  hipStream_t s = as_stream(stream);
#ifndef JOINT_PAIR16
#define JOINT_PAIR16 1
#endif
  if (JOINT_PAIR16 && n_mc > 8 && n_mc <= 16 && (D & 1) == 0 && (((uintptr_t)z) & 7) == 0) {  // pair-group form (round 6)
    entropy_joint_pair16_kernel<<<(unsigned)(N < 0x7fffffffll ? N : 0x7fffffffll), 64, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
    return runia_check_launch();
  }
  {
    // register form: rows of whole, aligned vectors (the LDS form below takes everything else)
    const int vec = n_mc <= 8 ? 4 : (n_mc <= 16 ? JOINT_VEC16 : 2);
    if (n_mc <= 32 && D % vec == 0 && ((uintptr_t)z) % (4 * vec) == 0) {
      const int64_t lanes = (D / vec + 63) / 64 * 64;
      const unsigned threads = (unsigned)(lanes < 256 ? lanes : 256);
      const unsigned grid = (unsigned)(N < 0x7fffffffll ? N : 0x7fffffffll);
      if (n_mc <= 8) entropy_joint_reg_kernel<8><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
      else if (n_mc <= 16) entropy_joint_reg_kernel<16><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
      else entropy_joint_reg_kernel<32><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
      return runia_check_launch();
    }
  }
  const size_t lds = ((size_t)n_mc * (kJointChunk + 2) + (size_t)n_mc * (n_mc + 1)) * sizeof(double);  // <= 99 KB
  static std::atomic<uint64_t> lds_ok{0};
  if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(entropy_joint_kernel), 100 * 1024, lds_ok) != RUNIA_OK)
    return RUNIA_E_LAUNCH;
  const unsigned grid = (unsigned)(N < 65535 ? N : 65535);
  entropy_joint_kernel<<<grid, 256, lds, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
  return runia_check_launch();
}

// get_dl_h_z's two outputs from ONE pass over the samples (reference evaluation/entropy.py:67-84 computes both per image):
// h_mvn [N] = runia_kl_entropy_joint_f32's bits, h [N, D] = runia_kl_entropy_per_dim_f32's bits.  The single-read kernel exists
// for 5 <= n_mc <= 32 with k = 5 (and n_mc = 5 ... 8 with k = 4), rows of whole aligned vectors; `runia_kl_entropy_both_fused`
// says whether a shape takes it - every other shape runs the two kernels one after the other (same results, two reads).
extern "C" int runia_kl_entropy_both_fused(int n_mc, int64_t D, int k) {
  if (n_mc < 5 || n_mc > 32) return 0;
  const int vec = n_mc <= 8 ? 4 : (n_mc <= 16 ? JOINT_VEC16 : 2);
  if (D <= 0 || D % vec != 0) return 0;
  if (n_mc <= 8) return k == 4 || k == 5;
  return k == 5;
}

extern "C" int runia_kl_entropy_both_f32(const float* z, double* h_mvn, double* h, int64_t N, int n_mc, int64_t D, int k,
                                         double min_dist, runia_stream_t stream) {
  if (N < 0 || D <= 0 || n_mc < 2 || n_mc > 64 || k < 1 || k >= n_mc || (N > 0 && (!z || !h_mvn || !h)))
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  const int vec = n_mc <= 8 ? 4 : (n_mc <= 16 ? JOINT_VEC16 : 2);
  if (!runia_kl_entropy_both_fused(n_mc, D, k) || ((uintptr_t)z) % (4 * vec) != 0 || ((uintptr_t)h) % 16 != 0) {
    if (int rc = runia_kl_entropy_joint_f32(z, h_mvn, N, n_mc, D, k, min_dist, stream)) return rc;
    return runia_kl_entropy_per_dim_f32(z, h, N, n_mc, D, k, min_dist, stream);
  }
  const double ct = digamma_diff(n_mc, k);
  const double d_over_n = (double)D / (double)n_mc, inv_n = 1.0 / (double)n_mc;
  hipStream_t s = as_stream(stream);
  const int64_t lanes = (D / vec + 63) / 64 * 64;
  const unsigned threads = (unsigned)(lanes < 256 ? lanes : 256);
  const unsigned grid = (unsigned)(N < 0x7fffffffll ? N : 0x7fffffffll);
  if (n_mc <= 8 && k == 4) entropy_joint_reg_kernel<8, 4><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n, h, inv_n);
  else if (n_mc <= 8) entropy_joint_reg_kernel<8, 5><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n, h, inv_n);
  else if (n_mc <= 16) entropy_joint_reg_kernel<16, 5><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n, h, inv_n);
  else entropy_joint_reg_kernel<32, 5><<<grid, threads, 0, s>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n, h, inv_n);
  return runia_check_launch();
}
