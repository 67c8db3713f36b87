// f1 / f4: symmetric eigen-decomposition without a vendor solver - what the setup-time fits of the path need:
//   scipy.linalg.pinvh(cov)  inside sklearn EmpiricalCovariance.fit  (inference/postprocessors.py:213-220, inference/funcs.py:52-66)
//   PCA(svd_solver="covariance_eigh" / "full").fit                    (dimensionality_reduction.py:70-71)
//   np.linalg.svd(cov + alpha I) of eigen_score                        (llm_uncertainty/scores.py:49-66, on the n x n Gram form)
//
// Two-sided cyclic Jacobi in f64 with a parallel (round-robin tournament) ordering: every step rotates n/2 disjoint index
// pairs at once, A <- J^T A J, V <- V J.  A step is two launches: the rotation angles of the step's pairs (read from the
// matrix as it stands), then one thread per 2 x 2 block {p,q} x {p',q'} updating the block in place - the blocks of a
// step are disjoint, so no double buffering and no atomics; only the upper triangle of block pairs is computed and
// mirrored, which keeps A exactly symmetric.  Jacobi is used for its accuracy (small eigenvalues come out with high
// relative accuracy, which the pinvh cut-off relies on); cost O(n^3) per sweep, ~8-10 sweeps.
// One C call = one sweep (n-1 steps); the caller loops until the sweep reports no rotation (the call itself never
// synchronises).
#include "common.hpp"

namespace {

// pair k (0 <= k < m/2) of round t (0 <= t < m-1) of the circle method on m (even) players
__device__ __forceinline__ void tournament_pair(int m, int t, int k, int& p, int& q) {
  if (k == 0) {
    p = m - 1;
    q = t;
  } else {
    p = (t + k) % (m - 1);
    q = (t - k + (m - 1)) % (m - 1);
  }
  if (p > q) { const int s = p; p = q; q = s; }
}

struct Rot { double c, s; };  // J = [[c, s], [-s, c]] in the (p, q) plane; identity = (1, 0)

__global__ __launch_bounds__(256) void jacobi_angles_kernel(const double* __restrict__ A, int n, int m, int t,
                                                            const double* __restrict__ anorm, Rot* __restrict__ rot,
                                                            unsigned* __restrict__ rotations) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= m / 2) return;
  int p, q;
  tournament_pair(m, t, k, p, q);
  Rot r{1.0, 0.0};
  bool rotated = false;
  if (q < n) {
    const double app = A[(int64_t)p * n + p], aqq = A[(int64_t)q * n + q], apq = A[(int64_t)p * n + q];
    const double thr = fmax(1e-19 * anorm[0], 1e-17 * sqrt(fabs(app * aqq)));
    if (fabs(apq) > thr) {
      const double theta = (aqq - app) / (2.0 * apq);
      const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      r.c = 1.0 / sqrt(tt * tt + 1.0);
      r.s = tt * r.c;
      rotated = true;
    }
  }
  rot[k] = r;
  // the sweep's rotation count (convergence test of the host loop): one atomic per wave, not per rotation - atomics on
  // one word retire at ~11 ns each, 1 024 of them per step were half of a step's time at n = 2048
  const unsigned long long any = __ballot(rotated);
  if (any && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true))) atomicAdd(rotations, (unsigned)__popcll(any));
}

__global__ __launch_bounds__(256) void jacobi_apply_kernel(double* __restrict__ A, double* __restrict__ V, int n, int m,
                                                           int t, const Rot* __restrict__ rot) {
  const int half = m / 2;
  const int ka = blockIdx.y, kb = blockIdx.x * 256 + threadIdx.x;
  if (kb >= half) return;
  int pa, qa, pb, qb;
  tournament_pair(m, t, ka, pa, qa);
  tournament_pair(m, t, kb, pb, qb);
  const Rot ra = rot[ka], rb = rot[kb];
  // eigenvector update: row ka' of V ... every (row i, column pair kb): rows are spread over blockIdx.y too
  // (two rows per ka: the rows pa and qa themselves, which covers all rows exactly once)
  if (rb.s != 0.0) {
    const int rows[2] = {pa, qa};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = rows[e];
      if (i < n && qb < n) {
        const double vp = V[(int64_t)i * n + pb], vq = V[(int64_t)i * n + qb];
        V[(int64_t)i * n + pb] = vp * rb.c - vq * rb.s;
        V[(int64_t)i * n + qb] = vp * rb.s + vq * rb.c;
      }
    }
  }
  if (ka > kb || (ra.s == 0.0 && rb.s == 0.0)) return;
  const bool qa_ok = qa < n, qb_ok = qb < n;  // a "bye" index (odd n) leaves its partner untouched (rotation = identity)
  const double b00 = A[(int64_t)pa * n + pb];
  const double b01 = qb_ok ? A[(int64_t)pa * n + qb] : 0.0;
  const double b10 = qa_ok ? A[(int64_t)qa * n + pb] : 0.0;
  const double b11 = (qa_ok && qb_ok) ? A[(int64_t)qa * n + qb] : 0.0;
  // J_a^T B
  const double t00 = ra.c * b00 - ra.s * b10, t01 = ra.c * b01 - ra.s * b11;
  const double t10 = ra.s * b00 + ra.c * b10, t11 = ra.s * b01 + ra.c * b11;
  // (J_a^T B) J_b
  double n00 = t00 * rb.c - t01 * rb.s, n01 = t00 * rb.s + t01 * rb.c;
  double n10 = t10 * rb.c - t11 * rb.s, n11 = t10 * rb.s + t11 * rb.c;
  if (ka == kb) { n01 = 0.0; n10 = 0.0; }  // the rotated pair is annihilated by construction
  A[(int64_t)pa * n + pb] = n00;
  A[(int64_t)pb * n + pa] = n00;
  if (qb_ok) { A[(int64_t)pa * n + qb] = n01; A[(int64_t)qb * n + pa] = n01; }
  if (qa_ok) { A[(int64_t)qa * n + pb] = n10; A[(int64_t)pb * n + qa] = n10; }
  if (qa_ok && qb_ok) { A[(int64_t)qa * n + qb] = n11; A[(int64_t)qb * n + qa] = n11; }
}

__global__ __launch_bounds__(256) void eigh_init_kernel(const double* __restrict__ A, double* __restrict__ V, int n,
                                                        double* __restrict__ anorm) {
  // V = I; anorm = Frobenius norm of A (one workgroup; setup-time)
  __shared__ double part[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < (int64_t)n * n; i += 256) {
    const double a = A[i];
    s += a * a;
    V[i] = (i / n == i % n) ? 1.0 : 0.0;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) anorm[0] = sqrt(part[0]);
}

// C [M, N] = A [M, K] * B [K, N] (row-major f64, plain tiled product; setup-time sizes only)
__global__ __launch_bounds__(256) void matmul_f64_kernel(const double* __restrict__ A, const double* __restrict__ B,
                                                         double* __restrict__ C, int64_t M, int64_t N, int64_t K,
                                                         int transpose_b) {
  __shared__ double As[16][17], Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t row = (int64_t)blockIdx.y * 16 + ty, col = (int64_t)blockIdx.x * 16 + tx;
  double acc = 0.0;
  for (int64_t k0 = 0; k0 < K; k0 += 16) {
    As[ty][tx] = (row < M && k0 + tx < K) ? A[row * K + k0 + tx] : 0.0;
    const int64_t bc = (int64_t)blockIdx.x * 16 + ty;  // for the transposed read: B is [N, K]
    if (transpose_b) Bs[tx][ty] = (bc < N && k0 + tx < K) ? B[bc * K + k0 + tx] : 0.0;
    else Bs[ty][tx] = (k0 + ty < K && col < N) ? B[(k0 + ty) * N + col] : 0.0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fma(As[ty][k], Bs[k][tx], acc);
    __syncthreads();
  }
  if (row < M && col < N) C[row * N + col] = acc;
}

// Gram matrix of the column-centred rows: G [n, n] = Ec Ec^T / denom, Ec = E - mean over the n rows (E [n, H] f32)
__global__ __launch_bounds__(256) void centred_gram_kernel(const float* __restrict__ E, double* __restrict__ G, int n,
                                                           int64_t H, double denom) {
  // one workgroup per (i, j), i <= j
  __shared__ double part[256];
  const int i = blockIdx.y, j = blockIdx.x;
  if (i > j) return;
  double s = 0.0;
  for (int64_t h = threadIdx.x; h < H; h += 256) {
    double mean = 0.0;
    for (int r = 0; r < n; ++r) mean += (double)E[(int64_t)r * H + h];
    mean /= (double)n;
    s += ((double)E[(int64_t)i * H + h] - mean) * ((double)E[(int64_t)j * H + h] - mean);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    G[(int64_t)i * n + j] = part[0] / denom;
    G[(int64_t)j * n + i] = part[0] / denom;
  }
}

}  // namespace

extern "C" size_t runia_eigh_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t m = (n + 1) & ~1ll;
  return (size_t)(m / 2) * sizeof(Rot) + 64;  // rotations of one step + norm + counter
}

// V = I and the matrix norm; call once before the sweeps.
extern "C" int runia_eigh_init_f64(const double* A, double* V, int64_t n, void* workspace, size_t workspace_bytes,
                                   runia_stream_t stream) {
  if (n <= 0 || n > 32768 || !A || !V) return RUNIA_E_INVALID;
  if (!workspace || workspace_bytes < runia_eigh_workspace_bytes(n) || (((uintptr_t)workspace) & 15) != 0)
    return RUNIA_E_WORKSPACE;
  double* anorm = reinterpret_cast<double*>(workspace);
  eigh_init_kernel<<<1, 256, 0, as_stream(stream)>>>(A, V, (int)n, anorm);
  return runia_check_launch();
}

// One Jacobi sweep in place: A -> J^T A J (tends to diag(eigenvalues)), V -> V J (columns tend to the eigenvectors).
// rotations (device, one unsigned) is incremented by the number of rotations applied: zero added = converged.
extern "C" int runia_eigh_sweep_f64(double* A, double* V, int64_t n, void* workspace, size_t workspace_bytes,
                                    unsigned* rotations, runia_stream_t stream) {
  if (n <= 0 || n > 32768 || !A || !V || !rotations) return RUNIA_E_INVALID;
  if (!workspace || workspace_bytes < runia_eigh_workspace_bytes(n) || (((uintptr_t)workspace) & 15) != 0)
    return RUNIA_E_WORKSPACE;
  if (n == 1) return RUNIA_OK;
  const int m = (int)((n + 1) & ~1ll);
  double* anorm = reinterpret_cast<double*>(workspace);
  Rot* rot = reinterpret_cast<Rot*>(reinterpret_cast<char*>(workspace) + 64);
  hipStream_t s = as_stream(stream);
  const int half = m / 2;
  for (int t = 0; t < m - 1; ++t) {
    jacobi_angles_kernel<<<(half + 255) / 256, 256, 0, s>>>(A, (int)n, m, t, anorm, rot, rotations);
    jacobi_apply_kernel<<<dim3((half + 255) / 256, half), 256, 0, s>>>(A, V, (int)n, m, t, rot);
  }
  return runia_check_launch();
}

extern "C" int runia_matmul_f64(const double* A, const double* B, double* C, int64_t M, int64_t N, int64_t K,
                                int transpose_b, runia_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return RUNIA_E_INVALID;
  if ((M + 15) / 16 > 65535) return RUNIA_E_INVALID;
  matmul_f64_kernel<<<dim3((unsigned)((N + 15) / 16), (unsigned)((M + 15) / 16)), 256, 0, as_stream(stream)>>>(
      A, B, C, M, N, K, transpose_b);
  return runia_check_launch();
}

extern "C" int runia_centred_gram_f32(const float* E, double* G, int64_t n, int64_t H, double denom,
                                      runia_stream_t stream) {
  if (n <= 0 || n > 4096 || H <= 0 || !E || !G || !(denom > 0.0)) return RUNIA_E_INVALID;
  centred_gram_kernel<<<dim3((unsigned)n, (unsigned)n), 256, 0, as_stream(stream)>>>(E, G, (int)n, H, denom);
  return runia_check_launch();
}
