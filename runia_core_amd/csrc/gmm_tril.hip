// Class-wise Gaussian log densities (GMM / DDU postprocessors; reference inference/postprocessors.py:490-491, 778-779 evaluate
// gmm.log_prob(x[:, None, :]) of the torch MultivariateNormal fitted by inference/funcs.py:265-344, then logsumexp over the classes).
// torch keeps the float32 Cholesky factor L_c of every class covariance and evaluates
//     log_prob[n, c] = -0.5 (D log 2 pi + || L_c^-1 (x_n - mu_c) ||^2) - sum log diag L_c        in float32.
// Rounds 4-5 scored every class with the dense f64 quadratic form x P_c x^T (2 D^2 multiply-adds per (row, class), 353 ms per
// 262 144 x 2048 x 10).  Here the whitened form: y = (x - mu_c) W_c^T with W_c = L_c^-1 LOWER TRIANGULAR, on the f32 matrix cores
// (the reference's own arithmetic).  Column tile j of W_c needs k < 128 (j + 1) only: the zero half is never multiplied, D^2
// flop per (row, class).  One launch scores every class:
//   K_a gmm_whiten_kernel : workgroup = 128 rows x 128 columns of ONE class (4 waves x 64 x 64 of v_mfma_f32_32x32x2_f32, the
//        tile engine of nt_tile_f32.hpp: 32-k chunks through LDS, next chunk's loads in flight), x - mu_c formed in f32 while the
//        chunk is staged (torch's `diff`), K loop cut at the diagonal block; epilogue: sum over the tile's 128 columns of y^2 in
//        f64 -> partial[class tile][row].  Tiles are walked in XCD-aware super-tiles (8 column tiles x 8 row tiles per XCD turn).
//   K_b gmm_finish_kernel : thread = row: M_c = sum_j partial (fixed order), log_prob = const_c - M_c / 2 -> f32, optional
//        logsumexp over the classes (no second launch, no (N, C) table unless the caller asks for it).
#include "nt_tile_f32.hpp"

namespace {

// mean chunk of the class: the 16 floats beside this thread's 16 operands (rows share them: L1 hits)
__device__ __forceinline__ void load_mean(const float* __restrict__ mu, int64_t D, int64_t k0, float (&m)[16], int tid, bool vec) {
  const int half = tid & 1;
  const float* p = mu + k0 + half * 16;
  if (vec && k0 + half * 16 + 16 <= D) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = reinterpret_cast<const float4*>(p)[j];
      m[4 * j] = t.x; m[4 * j + 1] = t.y; m[4 * j + 2] = t.z; m[4 * j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) m[j] = (k0 + half * 16 + j < D) ? p[j] : 0.f;
  }
}

__device__ __forceinline__ void store_diff(const float (&v)[16], const float (&m)[16], float (*dst)[KP], int tid) {
  const int row = tid >> 1, half = tid & 1;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    *reinterpret_cast<float2*>(&dst[row][half * 16 + 2 * j]) = make_float2(v[2 * j] - m[2 * j], v[2 * j + 1] - m[2 * j + 1]);
}

// 32 per-lane values -> their totals over the 32 lanes that share lane bit 5: lane ends with the total of slot (lane & 31).
// Halving exchange: at distance d the lanes with bit d clear keep the lower half of the slots and receive their partner's.
template <int D_, int N_>
__device__ __forceinline__ void halve_step(double (&v)[32], int lane) {
  const bool up = (lane & D_) != 0;
#pragma unroll
  for (int u = 0; u < N_; ++u) {
    const double send = up ? v[u] : v[u + N_], keep = up ? v[u + N_] : v[u];
    v[u] = keep + shfl_xor_f64(send, D_);
  }
}
__device__ __forceinline__ double halve32(double (&v)[32], int lane) {
  halve_step<16, 16>(v, lane);
  halve_step<8, 8>(v, lane);
  halve_step<4, 4>(v, lane);
  halve_step<2, 2>(v, lane);
  halve_step<1, 1>(v, lane);
  return v[0];
}

// x [N, D] f32, mu [C, D] f32, w [C, D, D] f32 row-major lower triangular (exact zeros above the diagonal),
// partial [C * nct][N] f64 with nct = ceil(D / 128).
__global__ __launch_bounds__(256) void gmm_whiten_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ w, double* __restrict__ partial,
                                                          int64_t N, int64_t D, int C) {
  __shared__ __attribute__((aligned(16))) float As[TQ][KP];
  __shared__ __attribute__((aligned(16))) float Bs[TB][KP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wb = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t nct = (D + TB - 1) / TB;
  int64_t q0, bt;
  {
    // the tile order of knn_dist_kernel: ids i, i + 8, ... (one XCD's share) walk a super-tile of kSuperB column tiles x
    // sq row tiles, so a 16 KB slice of a row tile is fetched into that L2 once and read by 8 column tiles
    const int64_t nbt = nct * C, nqt = (N + TQ - 1) / TQ;
    const int sq = knn_super_q(nqt), wps = kSuperB * sq;
    const int64_t nqg = (nqt + sq - 1) / sq;
    const int64_t l = blockIdx.x >> 3;
    const int64_t st = (l / wps) * 8 + (blockIdx.x & 7);
    const int r = (int)(l % wps);
    bt = (st / nqg) * kSuperB + r % kSuperB;
    const int64_t qt = (st % nqg) * sq + r / kSuperB;
    if (bt >= nbt || qt >= nqt) return;
    q0 = qt * TQ;
  }
#ifndef GMM_ORDER
#define GMM_ORDER 1
#endif
  // which (component, column tile) the tile index stands for.  GMM_ORDER 1: the component runs fastest and the column tiles come
  // longest first - the eight tiles a super-tile puts side by side on an XCD are the SAME column tile of eight components: they
  // share the row slices, run the same number of chunks (they sweep K in step, so a slice is fetched into that L2 once), and the
  // launch ends on its shortest tiles.  0: component-major (eight consecutive column tiles of one component: K ranges 1 : 8).
#if GMM_ORDER
  const int c = (int)(bt % C);
  const int64_t jt = nct - 1 - bt / C;
#else
  const int c = (int)(bt / nct);
  const int64_t jt = bt - (int64_t)c * nct;
#endif
  const int64_t m0 = jt * TB;                             // first column (= row of W_c) of the tile
  const int64_t kend = (m0 + TB < D) ? m0 + TB : D;      // W_c[m, k] = 0 for k > m: nothing right of the diagonal block
  const float* wc = w + (int64_t)c * D * D;
  const float* muc = mu + (int64_t)c * D;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)mu)) & 15) == 0;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float ra[16], rb[16], rm[16];
  load_chunk(x, q0, N, D, 0, ra, tid, vec);
  load_mean(muc, D, 0, rm, tid, vec);
  load_chunk(wc, m0, D, D, 0, rb, tid, vec);
  for (int64_t k0 = 0; k0 < kend; k0 += KCH) {
    __syncthreads();
    store_diff(ra, rm, As, tid);   // zero-filled k >= D: 0 - 0
    store_chunk<false>(rb, Bs, tid);
    __syncthreads();
    if (k0 + KCH < kend) {
      load_chunk(x, q0, N, D, k0 + KCH, ra, tid, vec);
      load_mean(muc, D, k0 + KCH, rm, tid, vec);
      load_chunk(wc, m0, D, D, k0 + KCH, rb, tid, vec);
    }
#pragma unroll
    for (int s = 0; s < KCH / 4; ++s) {
      float2 av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = *reinterpret_cast<const float2*>(&As[wq * 64 + a * 32 + li][4 * s + 2 * lh]);
#pragma unroll
      for (int b = 0; b < 2; ++b) bv[b] = *reinterpret_cast<const float2*>(&Bs[wb * 64 + b * 32 + li][4 * s + 2 * lh]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].x, bv[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a].y, bv[b].y, acc[a][b], 0, 0, 0);
        }
    }
  }
  // epilogue: C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]; sum of squares over the tile's columns in f64
  double sq[32];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const double y0 = (double)acc[a][0][r], y1 = (double)acc[a][1][r];
      sq[a * 16 + r] = fma(y1, y1, y0 * y0);   // columns beyond D are zero rows of W: they add 0
    }
  const double tot = halve32(sq, lane);  // slot li: a = li >> 4, r = li & 15
  __syncthreads();                       // the last chunk's LDS reads are done: As is free
  double* red = reinterpret_cast<double*>(&As[0][0]);  // [2 column waves][128 rows]
  const int slot_row = wq * 64 + (li >> 4) * 32 + (li & 3) + 8 * ((li & 15) >> 2) + 4 * lh;
  red[wb * TQ + slot_row] = tot;
  __syncthreads();
  if (tid < TQ && q0 + tid < N) partial[((int64_t)c * nct + jt) * N + q0 + tid] = red[tid] + red[TQ + tid];
}

// log_prob[n, c] = consts[c] - 0.5 sum_j partial[c nct + j][n]; lse[n] = logsumexp_c of the f32 log_probs
__global__ __launch_bounds__(256) void gmm_finish_kernel(const double* __restrict__ partial, const double* __restrict__ consts,
                                                          float* __restrict__ log_prob, float* __restrict__ lse, int64_t N,
                                                          int64_t nct, int C) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  // two passes over the classes (the maximum first) would read the partial sums twice; a running (max, sum) pair reads them once
  double mx = -kInfD(), sum = 0.0;
  bool nan = false;
  for (int c = 0; c < C; ++c) {
    double m = 0.0;
    for (int64_t j = 0; j < nct; ++j) m += partial[((int64_t)c * nct + j) * N + n];
    const float lp = (float)(consts[c] - 0.5 * m);
    if (log_prob) log_prob[n * C + c] = lp;
    const double v = (double)lp;
    if (v != v) nan = true;
    if (v > mx) {
      sum = (mx == -kInfD()) ? 1.0 : fma(sum, exp(mx - v), 1.0);
      mx = v;
    } else if (v == v && v > -kInfD()) {
      sum += exp(v - mx);
    }
  }
  if (lse) {
    // scipy.special.logsumexp: NaN in -> NaN; every term -inf -> -inf; a +inf term -> +inf
    float out;
    if (nan) out = __builtin_nanf("");
    else if (mx == -kInfD() || mx == kInfD()) out = (float)mx;
    else out = (float)(mx + log(sum));
    lse[n] = out;
  }
}

}  // namespace

extern "C" size_t runia_gmm_log_prob_workspace_bytes(int64_t N, int64_t D, int C) {
  if (N <= 0 || D <= 0 || C <= 0) return 0;
  return (size_t)(((D + TB - 1) / TB) * C) * (size_t)N * sizeof(double);
}

extern "C" int runia_gmm_log_prob_f32(const float* x, const float* means, const float* w_tril, const double* consts,
                                      float* log_prob, float* lse, void* workspace, size_t workspace_bytes, int64_t N,
                                      int64_t D, int C, runia_stream_t stream) {
  if (N < 0 || D <= 0 || C <= 0 || C > 65535) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !means || !w_tril || !consts || (!log_prob && !lse)) return RUNIA_E_INVALID;
  const int64_t nct = (D + TB - 1) / TB;
  if (!workspace || (((uintptr_t)workspace) & 7) != 0) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  // rows in chunks the workspace holds (any size >= one row tile's partial sums works)
  const int64_t per_row = nct * C * (int64_t)sizeof(double);
  int64_t chunk = (int64_t)(workspace_bytes / (size_t)per_row);
  if (chunk >= N) chunk = N;
  else chunk = (chunk / TQ) * TQ;
  if (chunk <= 0) return RUNIA_E_WORKSPACE;
  double* partial = static_cast<double*>(workspace);
  for (int64_t r0 = 0; r0 < N; r0 += chunk) {
    const int64_t rows = (N - r0 < chunk) ? (N - r0) : chunk;
    gmm_whiten_kernel<<<knn_dist_grid(rows, nct * C * TB), 256, 0, s>>>(x + r0 * D, means, w_tril, partial, rows, D, C);
    gmm_finish_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, s>>>(partial, consts, log_prob ? log_prob + r0 * C : nullptr,
                                                                      lse ? lse + r0 : nullptr, rows, nct, C);
  }
  return runia_check_launch();
}
