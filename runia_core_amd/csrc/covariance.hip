// f1 (SURVEY 8f "next #1"): the covariance of setup() on the device.
// Replaces np.cov(X.T, bias=1) inside sklearn EmpiricalCovariance.fit
// (reference inference/postprocessors.py:217-220, inference/funcs.py:62-66):
//     mu = mean(X, 0);  Xc = X - mu;  cov = (Xc^T Xc) * (1/N)          all in f64 (f32 rows are promoted first)
// The Gram matrix is a SYRK-shaped contraction over the N rows on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64, A[i][k] = Xc[k][i], B[k][j] = Xc[k][j]).  A workgroup owns a 64x64 output tile and a
// slice of the rows (split-K); partial tiles are summed in a fixed order by a second kernel, so the result is
// bit-reproducible run to run.  The eigen-decomposition behind pinvh stays a library call (host SciPy or
// torch.linalg.eigh) - see runia_core_amd/device_fit.py.
#include "common.hpp"

namespace {

constexpr int TI = 64;          // output tile edge
constexpr int RK = 32;          // rows (k) staged per step
constexpr int PITCH = 80;       // doubles per staged row: == 16 mod 32 -> the two k rows of a 32-lane group miss each other

template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                       int64_t N, int64_t D, int64_t rows_per_block) {
  // grid = (ceil(D/64), nblocks_rows); thread = (column lane, row phase)
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + lane;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  double s = 0.0;
  if (col < D)
    for (int64_t r = r0 + phase; r < r1; r += 4) s += (double)x[r * D + col];
  red[phase][lane] = s;
  __syncthreads();
  if (phase == 0 && col < D) partial[(int64_t)blockIdx.y * D + col] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

__global__ void col_mean_finish_kernel(const double* __restrict__ partial, double* __restrict__ mean, int64_t D,
                                       int64_t nblocks, double inv_n) {
  const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= D) return;
  double s = 0.0;
  for (int64_t b = 0; b < nblocks; ++b) s += partial[b * D + col];
  mean[col] = s * inv_n;
}

template <typename T>
__global__ __launch_bounds__(256) void gram_kernel(const T* __restrict__ x, const double* __restrict__ mean,
                                                    double* __restrict__ partial, int64_t N, int64_t D,
                                                    int64_t rows_per_split) {
  __shared__ double sa[RK][PITCH];
  __shared__ double sb[RK][PITCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.y * TI, j0 = (int64_t)blockIdx.x * TI;
  const int64_t r_begin = (int64_t)blockIdx.z * rows_per_split;
  const int64_t r_end = (r_begin + rows_per_split < N) ? r_begin + rows_per_split : N;
  d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int64_t r0 = r_begin; r0 < r_end; r0 += RK) {
    __syncthreads();
    for (int t = tid; t < RK * TI; t += 256) {
      const int k = t >> 6, c = t & 63;
      const int64_t r = r0 + k;
      double va = 0.0, vb = 0.0;
      if (r < r_end) {
        if (i0 + c < D) va = (double)x[r * D + i0 + c] - mean[i0 + c];
        if (j0 + c < D) vb = (double)x[r * D + j0 + c] - mean[j0 + c];
      }
      sa[k][c] = va;
      sb[k][c] = vb;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RK / 4; ++s) {
      const double a = sa[4 * s + lg][wave * 16 + li];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double b = sb[4 * s + lg][c * 16 + li];
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
      }
    }
  }
  double* out = partial + (int64_t)blockIdx.z * D * D;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t i = i0 + wave * 16 + lg + 4 * r, j = j0 + c * 16 + li;
      if (i < D && j < D) out[i * D + j] = acc[c][r];
    }
}

__global__ void gram_finish_kernel(const double* __restrict__ partial, double* __restrict__ cov, int64_t DD,
                                   int64_t splits, double inv_n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < DD; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int64_t z = 0; z < splits; ++z) s += partial[z * DD + i];
    cov[i] = s * inv_n;
  }
}

int64_t splits_for(int64_t N, int64_t D) {
  const int64_t tiles = ((D + TI - 1) / TI) * ((D + TI - 1) / TI);
  int64_t splits = (2048 + tiles - 1) / tiles;           // aim at ~2048 workgroups
  const int64_t max_by_rows = (N + 255) / 256;            // at least 256 rows per split
  if (splits > max_by_rows) splits = max_by_rows;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  return splits;
}

template <typename T>
int covariance_impl(const T* x, double* mean, double* cov, void* workspace, size_t workspace_bytes, int64_t N,
                    int64_t D, runia_stream_t stream) {
  if (N <= 0 || D <= 0 || !x || !mean || !cov) return RUNIA_E_INVALID;
  const int64_t splits = splits_for(N, D);
  const int64_t mean_blocks = (N + 4095) / 4096 < 256 ? (N + 4095) / 4096 : 256;
  const size_t need = (size_t)(splits * D * D + mean_blocks * D) * sizeof(double);
  if (!workspace || workspace_bytes < need) return RUNIA_E_WORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  double* mpart = part + splits * D * D;
  hipStream_t s = as_stream(stream);
  const int64_t rows_per_block = (N + mean_blocks - 1) / mean_blocks;
  col_sum_kernel<T><<<dim3((unsigned)((D + 63) / 64), (unsigned)mean_blocks), 256, 0, s>>>(x, mpart, N, D, rows_per_block);
  col_mean_finish_kernel<<<(unsigned)((D + 255) / 256), 256, 0, s>>>(mpart, mean, D, mean_blocks, 1.0 / (double)N);
  const int64_t rows_per_split = ((N + splits - 1) / splits + RK - 1) / RK * RK;
  const unsigned t = (unsigned)((D + TI - 1) / TI);
  gram_kernel<T><<<dim3(t, t, (unsigned)splits), 256, 0, s>>>(x, mean, part, N, D, rows_per_split);
  gram_finish_kernel<<<runia_stream_grid(D * D, 256), 256, 0, s>>>(part, cov, D * D, splits, 1.0 / (double)N);
  return runia_check_launch();
}

}  // namespace

extern "C" size_t runia_covariance_workspace_bytes(int64_t N, int64_t D) {
  if (N <= 0 || D <= 0) return 0;
  const int64_t mean_blocks = (N + 4095) / 4096 < 256 ? (N + 4095) / 4096 : 256;
  return (size_t)(splits_for(N, D) * D * D + mean_blocks * D) * sizeof(double);
}

extern "C" int runia_covariance_f64(const double* x, double* mean, double* cov, void* workspace,
                                    size_t workspace_bytes, int64_t N, int64_t D, runia_stream_t stream) {
  return covariance_impl<double>(x, mean, cov, workspace, workspace_bytes, N, D, stream);
}

extern "C" int runia_covariance_f32in(const float* x, double* mean, double* cov, void* workspace,
                                      size_t workspace_bytes, int64_t N, int64_t D, runia_stream_t stream) {
  return covariance_impl<float>(x, mean, cov, workspace, workspace_bytes, N, D, stream);
}
