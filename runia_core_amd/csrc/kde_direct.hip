// a9  LaRED, direct form: sklearn KernelDensity(kernel).score_samples by its definition, one difference at a time
//     (reference inference/postprocessors.py:78-128).  The matrix-core form of the Gaussian kernel (|x|^2 + |t|^2 - 2 x.t with an
//     online logsumexp) lives in gemm_f64.hip; this file keeps the direct f64 kernels: low dimensions, the five non-Gaussian
//     kernels of sklearn, and the reference for the matrix-core form's tests.
#include "common.hpp"

namespace {

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

// The other kernels sklearn's KernelDensity offers (DetectorKDE(kernel=...) forwards any of them, reference
// inference/postprocessors.py:78-128): tophat, epanechnikov, exponential, linear, cosine - values in [0, 1], so the
// density is a plain f64 sum over the training rows (one wave per training row, fixed order), then one log.
// KIND: 1 tophat [d < h], 2 epanechnikov 1 - d^2 / h^2, 3 exponential exp(-d / h), 4 linear 1 - d / h, 5 cosine
// cos(pi d / 2 h); compact kernels are 0 from d >= h on (sklearn's strict d < h).  No training row in range: log(0) = -inf.
template <int KIND>
__global__ __launch_bounds__(256) void kde_other_kernel(const double* __restrict__ train, const double* __restrict__ x,
                                                         double* __restrict__ score, int64_t M, int64_t N, int64_t D, double h,
                                                         double log_norm) {
  extern __shared__ double xs[];  // D doubles
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    __syncthreads();
    for (int64_t i = tid; i < D; i += 256) xs[i] = x[row * D + i];
    __syncthreads();
    double s = 0.0;
    for (int64_t m = wave; m < M; m += 4) {
      const double* t = train + m * D;
      double acc = 0.0;
      for (int64_t i = lane; i < D; i += 64) {
        const double d = xs[i] - t[i];
        acc += d * d;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
      const double dist = sqrt(acc);
      double kv;
      if (KIND == 1) kv = dist < h ? 1.0 : 0.0;
      else if (KIND == 2) kv = dist < h ? 1.0 - (dist * dist) / (h * h) : 0.0;
      else if (KIND == 3) kv = exp(-dist / h);
      else if (KIND == 4) kv = dist < h ? 1.0 - dist / h : 0.0;
      else kv = dist < h ? cos(0.5 * M_PI * dist / h) : 0.0;
      s += kv;
    }
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) score[row] = log((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + log_norm;
  }
}

// Gaussian KDE for D <= 64 (the regime where the reference's tree evaluation is converged): one thread
// owns one query with its D coordinates in registers; training rows are staged through LDS and read as broadcasts;
// the four waves of a workgroup take a quarter of every staged tile each and merge their (max, sum) pairs at the end.
// Per (query, train row): 2*D f64 ops + one f64 exp; logsumexp is kept online per group of 8 rows.
// Any D <= DP: training rows staged through LDS and read as broadcasts; NW waves share 64 queries and split every
// staged tile.  NW = 16 (8 at DP = 64: register budget) when the batch has too few 64-query groups to fill the chip.
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void kde_small_kernel(const double* __restrict__ train,
                                                             const double* __restrict__ x,
                                                             double* __restrict__ score, int64_t M, int64_t N, int D,
                                                             double neg_half_inv_h2, double log_norm) {
  constexpr int TM = (NW == 4) ? 64 : ((DP <= 32) ? 128 : 64);  // staged training rows (<= 32 KB of LDS)
  constexpr int RPW = TM / NW;                                   // rows per wave and tile
  constexpr int GS = (RPW < 8) ? RPW : 8;                        // rows per online-logsumexp group
  __shared__ double tile[TM][DP];
  __shared__ double pm[NW][64], ps[NW][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t qrow = (int64_t)blockIdx.x * 64 + lane;
  double xq[DP];
#pragma unroll
  for (int i = 0; i < DP; ++i) xq[i] = (qrow < N && i < D) ? x[qrow * D + i] : 0.0;
  double mx = -kInfD(), sum = 0.0;
  for (int64_t t0 = 0; t0 < M; t0 += TM) {
    __syncthreads();
    for (int i = tid; i < TM * DP; i += 64 * NW) {
      const int r = i / DP, c = i - r * DP;
      tile[r][c] = (t0 + r < M && c < D) ? train[(t0 + r) * D + c] : 0.0;
    }
    __syncthreads();
    const int rows = (int)((M - t0 < TM) ? (M - t0) : TM);
#pragma unroll
    for (int g = 0; g < RPW / GS; ++g) {  // this wave's share of the tile, GS rows at a time
      const int r0 = wave * RPW + g * GS;
      double v[GS];
      double gmax = -kInfD();
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          const double d = xq[i] - tile[r0 + j][i];
          acc = fma(d, d, acc);
        }
        v[j] = (r0 + j < rows) ? acc * neg_half_inv_h2 : -kInfD();
        gmax = fmax(gmax, v[j]);
      }
      if (gmax > -kInfD()) {
        const double mnew = fmax(mx, gmax);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < GS; ++j) part += exp(v[j] - mnew);
        sum = sum * exp(mx - mnew) + part;
        mx = mnew;
      }
    }
  }
  pm[wave][lane] = mx;
  ps[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && qrow < N) {
    double gm = pm[0][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) gm = fmax(gm, pm[w][lane]);
    double gs = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (ps[w][lane] > 0.0) gs += ps[w][lane] * exp(pm[w][lane] - gm);
    score[qrow] = log(gs) + gm + log_norm;
  }
}

template <int DP>
void launch_kde_small(const double* train, const double* x, double* score, int64_t M, int64_t N, int D, double nh,
                      double log_norm, hipStream_t s) {
  const unsigned qblocks = (unsigned)((N + 63) / 64);
  const bool few = (int64_t)qblocks < 2 * runia_cu_count();
  // (training rows through the scalar cache instead of LDS - s_load_dwordx16, SGPR operands - measured 0.355 / 3.81 /
  //  2.60 ms against 0.365 / 2.90 / 1.84 ms at D = 16 / 32 / 64: not kept)
  if (few) {  // 16 waves leave 128 VGPRs per lane: enough for D <= 32, not for a 64-wide query -> 8 waves there
    if constexpr (DP <= 32) kde_small_kernel<DP, 16><<<qblocks, 1024, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
    else kde_small_kernel<DP, 8><<<qblocks, 512, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
  } else {
    kde_small_kernel<DP, 4><<<qblocks, 256, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
  }
}

}  // namespace

extern "C" int runia_kde_score_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                                   int64_t D, double bandwidth, runia_stream_t stream) {
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0)) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!train || !x || !score) return RUNIA_E_INVALID;
  const double log_norm = -log((double)M) - (double)D * log(bandwidth) - 0.5 * (double)D * log(2.0 * M_PI);
  const double nh = -0.5 / (bandwidth * bandwidth);
  hipStream_t s = as_stream(stream);
  if (D <= 8) launch_kde_small<8>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 16) launch_kde_small<16>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 32) launch_kde_small<32>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 64) launch_kde_small<64>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else {
    const size_t shmem = (size_t)D * sizeof(double);
    if (shmem > 64 * 1024) return RUNIA_E_INVALID;
    kde_kernel<<<runia_stream_grid(N, 1), 256, shmem, s>>>(train, x, score, M, N, D, nh, log_norm);
  }
  return runia_check_launch();
}

// log of sklearn's kernel normalisation (neighbors/_binary_tree.pxi.tp, _log_kernel_norm): -factor - d log h
static double kde_log_norm(int kind, int64_t D, double h) {
  const double d = (double)D, log_pi = log(M_PI), log_2pi = log(2.0 * M_PI);
  auto logVn = [&](double n) { return 0.5 * n * log_pi - lgamma(0.5 * n + 1.0); };  // volume of the unit n-ball
  auto logSn = [&](double n) { return log_2pi + logVn(n - 1.0); };                   // surface of the unit n-sphere
  double factor = 0.0;
  switch (kind) {
    case 0: factor = 0.5 * d * log_2pi; break;
    case 1: factor = logVn(d); break;
    case 2: factor = logVn(d) + log(2.0 / (d + 2.0)); break;
    case 3: factor = logSn(d - 1.0) + lgamma(d); break;
    case 4: factor = logVn(d) - log(d + 1.0); break;
    default: {
      double tmp = 2.0 / M_PI;
      for (int64_t k = 1; k < D + 1; k += 2) {
        factor += tmp;
        tmp *= -(d - (double)k) * (d - (double)k - 1.0) * (2.0 / M_PI) * (2.0 / M_PI);
      }
      factor = log(factor) + logSn(d - 1.0);
    }
  }
  return -factor - d * log(h);
}

extern "C" int runia_kde_score_kernel_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                                          int64_t D, double bandwidth, int kind, runia_stream_t stream) {
  if (kind == 0) return runia_kde_score_f64(train, x, score, M, N, D, bandwidth, stream);
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0) || kind < 0 || kind > 5) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!train || !x || !score) return RUNIA_E_INVALID;
  const size_t shmem = (size_t)D * sizeof(double);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  const double log_norm = -log((double)M) + kde_log_norm(kind, D, bandwidth);
  hipStream_t s = as_stream(stream);
  const unsigned grid = runia_stream_grid(N, 1);
  switch (kind) {
    case 1: kde_other_kernel<1><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 2: kde_other_kernel<2><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 3: kde_other_kernel<3><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 4: kde_other_kernel<4><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    default: kde_other_kernel<5><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
  }
  return runia_check_launch();
}
