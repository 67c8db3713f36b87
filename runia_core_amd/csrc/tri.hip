// Setup-time helper of the class-wise Gaussians (GMM / DDU, reference inference/postprocessors.py:426-492, 694-786):
// torch's MultivariateNormal keeps the Cholesky factor L of every class covariance and evaluates the Mahalanobis term as
// || L^-1 (x - mu) ||^2.  The device state of these postprocessors is the precision (L L^T)^-1 = W^T W with W = L^-1, packed
// for the matrix cores.  W used to come from torch.cholesky_inverse on the host (0.8 s of a 4.4 s harness sweep: ten classes x
// nine PCA sizes); here every column of W is one thread's forward substitution - threads of a wave hold neighbouring columns, so
// the row of W they read back is one contiguous run and the element of L they share is one broadcast load.
#include "common.hpp"

namespace {

// W [B][D][D] (row-major) = inverse of the lower-triangular L [B][D][D]; the strict upper triangle of W is zeroed.
__global__ __launch_bounds__(64) void tril_inverse_kernel(const double* __restrict__ L, double* __restrict__ W, int64_t D) {
  const int64_t b = blockIdx.x;
  const int64_t j = (int64_t)blockIdx.y * 64 + threadIdx.x;
  const double* l = L + b * D * D;
  double* w = W + b * D * D;
  // every lane walks all the rows (the wave stays converged: the loads of L are uniform); rows above its column are zeros
  const int64_t j0 = (int64_t)blockIdx.y * 64;  // first column of the wave
  for (int64_t i = 0; i < D; ++i) {
    double v = 0.0;
    if (j < D) {
      if (i == j) {
        v = 1.0 / l[i * D + i];
      } else if (i > j) {
        double acc = 0.0;
        for (int64_t m = j0; m < i; ++m) acc = fma(l[i * D + m], (m >= j) ? w[m * D + j] : 0.0, acc);
        v = -acc / l[i * D + i];
      }
      w[i * D + j] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the row just written is read back by this wave's later rows
  }
}

// ---- factors of kTrilBlockedFrom columns or more: block forward substitution in 64 x 64 blocks.  The kernel above walks D rows per wave with a
// dependent inner loop (ten 2048 x 2048 factors: 494 ms of a 0.8 s DDU fit).  With W and L cut into blocks,
//     W[J][J] = L[J][J]^-1,      W[I][J] = -W[I][I] (sum_{K = J}^{I - 1} L[I][K] W[K][J])   for I > J,
// so: tril_diag_blocks_kernel inverts every diagonal block (the kernel above on 64 rows, one wave per block), then
// tril_block_columns_kernel takes one block column J per workgroup and walks down its block rows: the sum as a 64 x 64 tile product
// over k in [c0, i0) staged through LDS (one fma chain per element, k ascending), then the product with the inverted diagonal block.
// Block columns are independent of each other; what a workgroup reads of W it wrote itself (behind a barrier).
#ifndef TRIL_BLOCKED_FROM
#define TRIL_BLOCKED_FROM 128  // (ten matrices of 128 / 256 / 512 / 700 columns: panel forms faster from the first size - tools/ablate/run_gmm_fit.py)
#endif
constexpr int64_t kTrilBlockedFrom = TRIL_BLOCKED_FROM;
constexpr int kTB = 64, kTK = 16;

__global__ __launch_bounds__(64) void tril_diag_blocks_kernel(const double* __restrict__ L, double* __restrict__ W, int64_t D) {
  const int64_t b = blockIdx.x, c0 = (int64_t)blockIdx.y * kTB;
  const int64_t j = c0 + threadIdx.x;
  const int64_t rows = (D - c0 < kTB) ? D - c0 : kTB;
  const double* l = L + b * D * D;
  double* w = W + b * D * D;
  for (int64_t r = 0; r < rows; ++r) {
    const int64_t i = c0 + r;
    if (j < D) {
      double v = 0.0;
      if (i == j) {
        v = 1.0 / l[i * D + i];
      } else if (i > j) {
        double acc = 0.0;
        for (int64_t m = c0; m < i; ++m) acc = fma(l[i * D + m], (m >= j) ? w[m * D + j] : 0.0, acc);
        v = -acc / l[i * D + i];
      }
      w[i * D + j] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the row just written is read back by this wave's later rows
  }
}

__global__ __launch_bounds__(256) void tril_block_columns_kernel(const double* __restrict__ L, double* __restrict__ W, int64_t D) {
  const int64_t b = blockIdx.x, c0 = (int64_t)blockIdx.y * kTB;
  const double* l = L + b * D * D;
  double* w = W + b * D * D;
  __shared__ double As[kTB][kTK + 1], Bs[kTK][kTB + 1], Ss[kTB][kTB + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  // rows above the block column's diagonal block are zeros
  for (int64_t e = tid; e < c0 * kTB; e += 256) {
    const int64_t i = e / kTB, c = c0 + e % kTB;
    if (c < D) w[i * D + c] = 0.0;
  }
  for (int64_t i0 = c0 + kTB; i0 < D; i0 += kTB) {
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
    for (int64_t k0 = c0; k0 < i0; k0 += kTK) {  // (c0 and i0 are multiples of 64: whole chunks)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int idx = tid + 256 * e;
        const int ra = idx >> 4, ka = idx & 15;   // As: 64 rows x 16 k
        const int kb = idx >> 6, cb = idx & 63;   // Bs: 16 k x 64 columns
        As[ra][ka] = (i0 + ra < D) ? l[(i0 + ra) * D + k0 + ka] : 0.0;
        Bs[kb][cb] = (c0 + cb < D) ? w[(k0 + kb) * D + c0 + cb] : 0.0;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < kTK; ++kk) {
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = As[ty * 4 + r][kk];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = Bs[kk][tx * 4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) Ss[ty * 4 + r][tx * 4 + c] = acc[r][c];
    __syncthreads();
    // W[I][J] = -W[I][I] S, W[I][I] lower triangular (its zeros are skipped)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = ty * 4 + r;
      const int64_t i = i0 + rr;
      if (i >= D) continue;
      double out[4] = {0.0, 0.0, 0.0, 0.0};
      const double* wii = w + i * D + i0;
      for (int k = 0; k <= rr; ++k) {
        const double f = wii[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = fma(f, Ss[k][tx * 4 + c], out[c]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c0 + tx * 4 + c < D) w[i * D + c0 + tx * 4 + c] = -out[c];
    }
    __syncthreads();  // the block just written is an operand of the next block row's sum
  }
}

}  // namespace

extern "C" int runia_tril_inverse_f64(const double* tril, double* inv, int64_t batch, int64_t D, runia_stream_t stream) {
  if (batch < 0 || D <= 0 || D > 16384) return RUNIA_E_INVALID;
  if (batch == 0) return RUNIA_OK;
  if (!tril || !inv || batch > 65535) return RUNIA_E_INVALID;
  const dim3 grid((unsigned)batch, (unsigned)((D + 63) / 64));
  if (D < kTrilBlockedFrom) {
    tril_inverse_kernel<<<grid, 64, 0, as_stream(stream)>>>(tril, inv, D);
    return runia_check_launch();
  }
  tril_diag_blocks_kernel<<<grid, 64, 0, as_stream(stream)>>>(tril, inv, D);
  tril_block_columns_kernel<<<grid, 256, 0, as_stream(stream)>>>(tril, inv, D);
  return runia_check_launch();
}
