// Pairwise-distance postprocessors.
//   a8  kNN   : faiss.IndexFlatL2.search(q, k) -> -(k-th smallest squared L2)
//               (reference inference/postprocessors.py:417-421, 873-880; faiss-gpu==1.7.2)
//   a9  LaRED : sklearn KernelDensity(gaussian).score_samples
//               (reference inference/postprocessors.py:118-128)
//
// Round-1 form: exact-difference distances on the vector ALUs (faiss's own path for
// the reference's one-query-at-a-time search accumulates sum((q-b)^2) directly, without
// the norm expansion), one workgroup per query row, bank streamed from L2/HBM.
// The k-th order statistic is found by an 8-bit radix select over the distance bits
// (distances are >= 0, so unsigned integer order == float order) - no sort, no top-k list.
#include "common.hpp"

namespace {

constexpr float kFltMax = 3.4028234663852886e38f;

// squared L2 between the workgroup's query (in LDS) and every bank row -> dist[M] (global)
__device__ __forceinline__ void distances_to_bank(const float* __restrict__ qs, const float* __restrict__ bank,
                                                   float* __restrict__ dist, int64_t M, int64_t D, bool vec) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t m = wave; m < M; m += 4) {
    const float* b = bank + m * D;
    float acc = 0.f;
    if (vec) {
      const float4* b4 = reinterpret_cast<const float4*>(b);
      const float4* q4 = reinterpret_cast<const float4*>(qs);
      for (int64_t i = lane; i < (D >> 2); i += 64) {
        const float4 bv = b4[i], qv = q4[i];
        const float d0 = qv.x - bv.x, d1 = qv.y - bv.y, d2 = qv.z - bv.z, d3 = qv.w - bv.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      }
    } else {
      for (int64_t i = lane; i < D; i += 64) {
        const float d = qs[i] - b[i];
        acc += d * d;
      }
    }
    acc = wave_sum_f32(acc);
    if (lane == 0) dist[m] = acc;
  }
}

__global__ __launch_bounds__(256) void knn_kth_kernel(const float* __restrict__ q, const float* __restrict__ bank,
                                                       float* __restrict__ score, float* __restrict__ work,
                                                       int64_t N, int64_t M, int64_t D, int k) {
  extern __shared__ float qs[];  // D floats (16-byte aligned by construction)
  __shared__ unsigned hist[256];
  __shared__ unsigned sel_prefix, sel_rank;
  const int tid = threadIdx.x;
  float* dist = work + (int64_t)blockIdx.x * M;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)bank) & 15) == 0) && ((((uintptr_t)q) & 15) == 0);
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    __syncthreads();
    for (int64_t i = tid; i < D; i += 256) qs[i] = q[row * D + i];
    __syncthreads();
    distances_to_bank(qs, bank, dist, M, D, vec);
    __threadfence_block();
    __syncthreads();
    // radix select of the k-th smallest (1-based) among dist[0..M)
    if (tid == 0) { sel_prefix = 0u; sel_rank = (unsigned)k; }
    const unsigned* bits = reinterpret_cast<const unsigned*>(dist);
    for (int pass = 3; pass >= 0; --pass) {
      hist[tid] = 0u;
      __syncthreads();
      const unsigned shift = 8u * pass;
      const unsigned himask = (pass == 3) ? 0u : (0xFFFFFFFFu << (shift + 8));
      const unsigned prefix = sel_prefix;
      for (int64_t m = tid; m < M; m += 256) {
        const unsigned u = bits[m];
        if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned r = sel_rank, b = 0;
        for (; b < 256; ++b) {
          const unsigned c = hist[b];
          if (r <= c) break;
          r -= c;
        }
        sel_rank = r;
        sel_prefix = prefix | (b << shift);
      }
      __syncthreads();
    }
    if (tid == 0) score[row] = -__uint_as_float(sel_prefix);
  }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// Gaussian KDE log-density, f64: online logsumexp over the training rows, one workgroup per query.
__global__ __launch_bounds__(256) void kde_kernel(const double* __restrict__ train, const double* __restrict__ x,
                                                   double* __restrict__ score, int64_t M, int64_t N, int64_t D,
                                                   double neg_half_inv_h2, double log_norm) {
  extern __shared__ double xs[];  // D doubles
  __shared__ double wm[4], wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    __syncthreads();
    for (int64_t i = tid; i < D; i += 256) xs[i] = x[row * D + i];
    __syncthreads();
    double mx = -kInfD(), s = 0.0;  // running max / sum of exp(. - mx), identical on all lanes of a wave
    for (int64_t m = wave; m < M; m += 4) {
      const double* t = train + m * D;
      double acc = 0.0;
      for (int64_t i = lane; i < D; i += 64) {
        const double d = xs[i] - t[i];
        acc += d * d;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
      const double v = acc * neg_half_inv_h2;
      if (v > mx) {
        s = s * exp(mx - v) + 1.0;
        mx = v;
      } else {
        s += exp(v - mx);
      }
    }
    if (lane == 0) { wm[wave] = mx; wsum[wave] = s; }
    __syncthreads();
    if (tid == 0) {
      double gm = fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]));
      double gs = 0.0;
      for (int w = 0; w < 4; ++w)
        if (wsum[w] > 0.0) gs += wsum[w] * exp(wm[w] - gm);
      score[row] = log(gs) + gm + log_norm;
    }
  }
}

}  // namespace

extern "C" size_t runia_knn_workspace_bytes(int64_t N, int64_t M, int64_t D, int k) {
  (void)D; (void)k;
  if (N <= 0 || M <= 0) return 0;
  const int64_t slots = N < 1024 ? N : 1024;  // one distance row per resident workgroup
  return (size_t)(slots * M) * sizeof(float);
}

extern "C" int runia_knn_kth_f32(const float* q, const float* bank, float* score, void* workspace,
                                 size_t workspace_bytes, int64_t N, int64_t M, int64_t D, int k,
                                 runia_stream_t stream) {
  if (N < 0 || M < 0 || D <= 0 || k < 1 || (N > 0 && (!q || !score)) || (M > 0 && !bank)) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  hipStream_t s = as_stream(stream);
  if (k > M) {  // faiss fills missing neighbours with FLT_MAX
    fill_kernel<<<runia_stream_grid(N, 256), 256, 0, s>>>(score, N, -kFltMax);
    return runia_check_launch();
  }
  int64_t slots = (int64_t)(workspace_bytes / ((size_t)M * sizeof(float)));
  if (!workspace || slots < 1) return RUNIA_E_WORKSPACE;
  if (slots > N) slots = N;
  if (slots > 1024) slots = 1024;
  const size_t shmem = (size_t)((D + 3) / 4 * 4) * sizeof(float);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  knn_kth_kernel<<<(unsigned)slots, 256, shmem, s>>>(q, bank, score, reinterpret_cast<float*>(workspace), N, M, D, k);
  return runia_check_launch();
}

extern "C" int runia_kde_score_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                                   int64_t D, double bandwidth, runia_stream_t stream) {
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0) || !train || (N > 0 && (!x || !score))) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  const size_t shmem = (size_t)D * sizeof(double);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  const double log_norm = -log((double)M) - (double)D * log(bandwidth) - 0.5 * (double)D * log(2.0 * M_PI);
  kde_kernel<<<runia_stream_grid(N, 1), 256, shmem, as_stream(stream)>>>(
      train, x, score, M, N, D, -0.5 / (bandwidth * bandwidth), log_norm);
  return runia_check_launch();
}
