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

template <typename T>
int launch_cholesky(T* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  if (batch < 0 || D <= 0 || D > 16384 || batch > 65535) return RUNIA_E_INVALID;
  if (batch == 0) return RUNIA_OK;
  if (!a || !info) return RUNIA_E_INVALID;
  cholesky_kernel<T><<<(unsigned)batch, 64 * kCholWaves, 0, as_stream(stream)>>>(a, D, (T)jitter, info);
  return runia_check_launch();
}

}  // namespace

extern "C" int runia_cholesky_f32(float* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  return launch_cholesky<float>(a, info, batch, D, jitter, stream);
}
extern "C" int runia_cholesky_f64(double* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream) {
  return launch_cholesky<double>(a, info, batch, D, jitter, stream);
}
