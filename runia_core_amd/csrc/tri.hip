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

}  // namespace

extern "C" int runia_tril_inverse_f64(const double* tril, double* inv, int64_t batch, int64_t D, runia_stream_t stream) {
  if (batch < 0 || D <= 0 || D > 16384) return RUNIA_E_INVALID;
  if (batch == 0) return RUNIA_OK;
  if (!tril || !inv || batch > 65535) return RUNIA_E_INVALID;
  const dim3 grid((unsigned)batch, (unsigned)((D + 63) / 64));
  tril_inverse_kernel<<<grid, 64, 0, as_stream(stream)>>>(tril, inv, D);
  return runia_check_launch();
}
