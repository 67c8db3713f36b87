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
  const size_t lds = ((size_t)n_mc * (kJointChunk + 2) + (size_t)n_mc * (n_mc + 1)) * sizeof(double);  // <= 99 KB
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(entropy_joint_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
      return RUNIA_E_LAUNCH;
    attr_set = true;
  }
  const unsigned grid = (unsigned)(N < 65535 ? N : 65535);
  entropy_joint_kernel<<<grid, 256, lds, as_stream(stream)>>>(z, h_mvn, N, n_mc, D, k, min_dist, ct, d_over_n);
  return runia_check_launch();
}
