// Setup-time Cholesky factorisation on the device, batched over classes.
//   f32: the jitter ladder of gmm_fit (reference inference/funcs.py:310-342: torch's MultivariateNormal(covariance_matrix = cov + jitter I)
//        factorises every class covariance in float32 and the ladder moves on when a factorisation fails) - the host's
//        torch.linalg.cholesky_ex was 0.36 s of a 2.1 s harness sweep (ten classes x ten PCA sizes);
//   f64: the triangular factor of a positive-definite precision matrix (MDLatentSpace on un-reduced features: -|| W d ||^2 with
//        W lower triangular costs half the multiply-adds of -d P d^T, see runia_md_score_tril_*).
// One workgroup (16 waves) per matrix, left-looking by columns, in place on the lower triangle: column j of L needs the dot
// products of rows j.. with row j over the j columns already final; a wave takes four rows at a time (lanes over k, one fma chain
// per lane and row, then a fixed-order wave sum), lane 0 leaves A[i][j] - dot in place; after a barrier the pivot is tested - not
// positive (or NaN) ends the factorisation with info = j + 1, LAPACK's convention - and the column is scaled.  Every sum has one
// fixed order, so the factor is the same bits from run to run and on every rank.
#include "common.hpp"

namespace {

constexpr int kCholWaves = 16;
constexpr int kCholRows = 4;  // rows a wave works on at once (their loads overlap)

template <typename T>
__device__ __forceinline__ T wave_sum_fixed(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T>
__global__ __launch_bounds__(64 * kCholWaves) void cholesky_kernel(T* __restrict__ a, int64_t D, T jitter, int* __restrict__ info) {
  T* m = a + (int64_t)blockIdx.x * D * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int failed;
  if (tid == 0) failed = 0;
  __syncthreads();
  for (int64_t j = 0; j < D; ++j) {
    const T* rj = m + j * D;
    // rows j + (wave * kCholRows + r) + t * kCholWaves * kCholRows
    for (int64_t i0 = j + (int64_t)wave * kCholRows; i0 < D; i0 += kCholWaves * kCholRows) {
      T acc[kCholRows];
#pragma unroll
      for (int r = 0; r < kCholRows; ++r) acc[r] = (T)0;
      for (int64_t k = lane; k < j; k += 64) {
        const T lj = rj[k];
#pragma unroll
        for (int r = 0; r < kCholRows; ++r) {
          const int64_t i = (i0 + r < D) ? i0 + r : D - 1;  // (a row past the end re-reads the last one; never written)
          acc[r] = fma(m[i * D + k], lj, acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < kCholRows; ++r) {
        const T s = wave_sum_fixed(acc[r]);
        const int64_t i = i0 + r;
        if (lane == 0 && i < D) m[i * D + j] = (m[i * D + j] + ((i == j) ? jitter : (T)0)) - s;
      }
    }
    __syncthreads();
    const T p = m[j * D + j];
    if (!(p > (T)0)) {  // LAPACK's potrf: "the leading minor of order j + 1 is not positive"; NaN pivots fail as well
      if (tid == 0) { failed = 1; info[blockIdx.x] = (int)(j + 1); }
      break;  // uniform: every thread read the same pivot
    }
    const T d = sqrt(p);
    __syncthreads();  // every thread has read the pivot before it is overwritten
    for (int64_t i = j + tid; i < D; i += 64 * kCholWaves) m[i * D + j] = (i == j) ? d : m[i * D + j] / d;
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0 && !failed) info[blockIdx.x] = 0;
  // the strict upper triangle reads as zeros (torch.linalg.cholesky's output)
  for (int64_t e = tid; e < D * D; e += 64 * kCholWaves) {
    const int64_t i = e / D, c = e - i * D;
    if (c > i) m[e] = (T)0;
  }
}

// ---- matrices of kCholBlockedFrom columns or more: the same left-looking factorisation in panels of kCholPanel columns.  The one-workgroup
// kernel above reads D^3 / 3 elements through one compute unit and meets 3 barriers per column with dot products of up to D terms
// between them (ten 2048 x 2048 float32 matrices: 173 ms, and a jitter ladder calls it several times).  Per panel [j0, j1):
//   chol_update_kernel  rows >= j0, columns of the panel: A[i][c] = (A[i][c] + jitter on the diagonal) - sum_{k < j0} L[i][k] L[c][k]
//                       as 64 x 64 tiles on the whole chip (one fma chain over k per element, k ascending);
//   chol_diag_kernel    the panel's 64 x 64 diagonal block, one wave per matrix in LDS (pivot test here);
//   chol_trsm_kernel    the rows below it, one thread per row (dot products over k in [j0, j): < kCholPanel terms, one chain each).
// A failed pivot writes info = j + 1 (LAPACK's convention) and every later launch of that matrix returns at its first instruction.
// Every sum has one fixed order: same bits from run to run and on every rank (not the bits of the one-workgroup kernel: the
// sums are grouped by panel).
constexpr int kCholPanel = 64;
#ifndef CHOL_BLOCKED_FROM
#define CHOL_BLOCKED_FROM 128  // (ten matrices of 128 / 256 / 512 / 700 columns: panel forms faster from the first size - tools/ablate/run_gmm_fit.py)
#endif
constexpr int64_t kCholBlockedFrom = CHOL_BLOCKED_FROM;

template <typename T>
__global__ __launch_bounds__(256) void chol_update_kernel(T* __restrict__ a, int64_t D, int64_t j0, int64_t j1, T jitter,
                                                           const int* __restrict__ info) {
  const int64_t b = blockIdx.y;
  if (info[b] != 0) return;  // (uniform)
  T* m = a + b * D * D;
  const int64_t r0 = j0 + (int64_t)blockIdx.x * 64;
  __shared__ T As[64][33], Bs[64][33];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  T acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = (T)0;
  for (int64_t k0 = 0; k0 < j0; k0 += 32) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = tid + 256 * e, row = idx >> 5, kk = idx & 31;
      const int64_t k = k0 + kk, ia = r0 + row, ib = j0 + row;
      As[row][kk] = (ia < D && k < j0) ? m[ia * D + k] : (T)0;
      Bs[row][kk] = (ib < j1 && k < j0) ? m[ib * D + k] : (T)0;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      T av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = As[ty * 4 + r][kk];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = Bs[tx * 4 + c][kk];
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
    for (int c = 0; c < 4; ++c) {
      const int64_t i = r0 + ty * 4 + r, cc = j0 + tx * 4 + c;
      if (i < D && cc < j1 && cc <= i) m[i * D + cc] = (m[i * D + cc] + ((i == cc) ? jitter : (T)0)) - acc[r][c];
    }
}

// The panel itself, after the update: (1) its diagonal block is factorised by one wave per matrix in LDS (column by column: dot
// products over k in [j0, j), pivot test, scaling - the arithmetic of the one-workgroup kernel on a 64 x 64 block); (2) the rows below
// are independent of each other given that block: L[i][panel] = A[i][panel] L_diag^-T, one thread per row with its 64 panel entries
// in registers and the block's entries as scalar operands (wave-uniform loads through the constant address space: no LDS, no
// barrier), on the whole chip.  (A first form did both in one workgroup per matrix, thread per row with the row re-read from
// memory for every column: 1.07 ms per panel of ten 2048 x 2048 matrices - 8 KB-strided rows thrash the L1; profiles/README.md.)
template <typename T>
__global__ __launch_bounds__(64) void chol_diag_kernel(T* __restrict__ a, int64_t D, int64_t j0, int64_t j1, int* __restrict__ info) {
  const int64_t b = blockIdx.x;
  if (info[b] != 0) return;  // (uniform)
  T* m = a + b * D * D;
  const int w = (int)(j1 - j0), r = threadIdx.x;
  __shared__ T ld[kCholPanel][kCholPanel + 1];
  for (int c = 0; c < w; ++c) ld[r][c] = (r < w) ? m[(j0 + r) * D + j0 + c] : (T)0;
  __builtin_amdgcn_wave_barrier();
  for (int c = 0; c < w; ++c) {
    T acc = (T)0;
    for (int k = 0; k < c; ++k) acc = fma(ld[r][k], ld[c][k], acc);
    const T v = ld[r][c] - acc;
    __builtin_amdgcn_wave_barrier();
    const T p = __shfl(v, c, 64);  // the pivot: row c's value
    if (!(p > (T)0)) {             // not positive, or NaN (uniform)
      if (r == 0) info[b] = (int)(j0 + c + 1);
      return;
    }
    const T d = sqrt(p);
    if (r >= c && r < w) ld[r][c] = (r == c) ? d : v / d;
    __builtin_amdgcn_wave_barrier();
  }
  if (r < w)
    for (int c = 0; c <= r; ++c) m[(j0 + r) * D + j0 + c] = ld[r][c];
}

template <typename T>
__global__ __launch_bounds__(256) void chol_trsm_kernel(T* __restrict__ a, int64_t D, int64_t j0, int64_t j1, const int* __restrict__ info) {
  const int64_t b = blockIdx.y;
  if (info[b] != 0) return;  // (uniform: the diagonal block of this panel - or an earlier one - had no factor)
  T* m = a + b * D * D;
  const int64_t i = j1 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int w = (int)(j1 - j0);
  typedef const __attribute__((address_space(4))) T* cptr;  // wave-uniform reads: scalar loads
  cptr ld = (cptr)(m + j0 * D + j0);
  if (i >= D) return;
  T* ri = m + i * D + j0;
  T x[kCholPanel];
#pragma unroll
  for (int c = 0; c < kCholPanel; ++c) x[c] = (c < w) ? ri[c] : (T)0;
#pragma unroll
  for (int c = 0; c < kCholPanel; ++c) {
    if (c < w) {  // (uniform; the last panel of a ragged matrix is narrower)
      T acc = (T)0;
#pragma unroll
      for (int k = 0; k < c; ++k) acc = fma(x[k], ld[(int64_t)c * D + k], acc);
      x[c] = (x[c] - acc) / ld[(int64_t)c * D + c];
    }
  }
#pragma unroll
  for (int c = 0; c < kCholPanel; ++c)
    if (c < w) ri[c] = x[c];
}

template <typename T>
__global__ __launch_bounds__(256) void chol_zero_upper_kernel(T* __restrict__ a, int64_t D) {
  T* m = a + (int64_t)blockIdx.y * D * D;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < D * D; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / D, c = e - i * D;
    if (c > i) m[e] = (T)0;
  }
}

template <typename T>
int launch_cholesky(T* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  if (batch < 0 || D <= 0 || D > 16384 || batch > 65535) return RUNIA_E_INVALID;
  if (batch == 0) return RUNIA_OK;
  if (!a || !info) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (D < kCholBlockedFrom) {
    cholesky_kernel<T><<<(unsigned)batch, 64 * kCholWaves, 0, s>>>(a, D, (T)jitter, info);
    return runia_check_launch();
  }
  if (hipMemsetAsync(info, 0, (size_t)batch * sizeof(int), s) != hipSuccess) return RUNIA_E_LAUNCH;
  for (int64_t j0 = 0; j0 < D; j0 += kCholPanel) {
    const int64_t j1 = (j0 + kCholPanel < D) ? j0 + kCholPanel : D;
    const dim3 grid((unsigned)((D - j0 + 63) / 64), (unsigned)batch);
    chol_update_kernel<T><<<grid, 256, 0, s>>>(a, D, j0, j1, (T)jitter, info);
    chol_diag_kernel<T><<<(unsigned)batch, 64, 0, s>>>(a, D, j0, j1, info);
    if (j1 < D) {
      const dim3 tgrid((unsigned)((D - j1 + 255) / 256), (unsigned)batch);
      chol_trsm_kernel<T><<<tgrid, 256, 0, s>>>(a, D, j0, j1, info);
    }
  }
  const dim3 zgrid((unsigned)runia_stream_grid(D * D, 256) < 1024u ? (unsigned)runia_stream_grid(D * D, 256) : 1024u, (unsigned)batch);
  chol_zero_upper_kernel<T><<<zgrid, 256, 0, s>>>(a, D);
  return runia_check_launch();
}

}  // namespace

extern "C" int runia_cholesky_f32(float* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  return launch_cholesky<float>(a, info, batch, D, jitter, stream);
}
extern "C" int runia_cholesky_f64(double* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  return launch_cholesky<double>(a, info, batch, D, jitter, stream);
}
