// f1: symmetric eigen-decomposition, blocked - the same two-sided cyclic Jacobi as eigh.hip, reorganised so that a sweep
// is O(n / 32) launches of matrix-core work instead of 2 (n - 1) launches of 2 x 2 rotations (n = 512: 1 022 launches per
// sweep at ~8.7 us each were 89 ms of launch latency; n = 2048: 1.04 s against LAPACK's 0.43 s on the host).
//
// Columns are cut into blocks of 32; a step pairs the blocks by the round-robin tournament (nb / 2 disjoint pairs (I, J)):
//   block_solve   one workgroup per pair: the 64 x 64 sub-matrix S = A[IJ, IJ] goes to LDS, ONE cyclic Jacobi sweep over
//                 its 64 indices runs there (63 inner steps of 32 disjoint rotations, S updated in place, the rotations
//                 accumulated in R = J_1 J_2 ...), and R (64 x 64, orthogonal) is written out;
//   block_update  A[IJ_a, IJ_b] <- R_a^T A[IJ_a, IJ_b] R_b for every pair of pairs a <= b (mirrored, so A stays exactly
//                 symmetric) and V[:, IJ_b] <- V[:, IJ_b] R_b: 64 x 64 x 64 products on v_mfma_f64_16x16x4_f64.
// Mathematically this IS a cyclic Jacobi method (the rotations of an inner sweep applied to the whole matrix at once), so
// the convergence theory and the accuracy are those of eigh.hip; a sweep is (nb - 1) steps x 2 launches.
// Used by scipy.linalg.pinvh / PCA covariance_eigh restatements (device_fit.py), eigen_score, and the folding of the LaREM
// weights (inference/pipeline.py) - reference call sites: inference/postprocessors.py:213-220, inference/funcs.py:52-66,
// dimensionality_reduction.py:70-71, llm_uncertainty/scores.py:49-66.
#include "common.hpp"

namespace {

constexpr int BS = 32;   // columns per block
constexpr int SB = 64;   // sub-problem size (a pair of blocks)
constexpr int SP = 65;   // LDS pitch of the solve kernel (odd: rows p and q of a rotation never share a bank)
constexpr int UP = 66;   // LDS pitch of the update kernel (== 2 mod 32: conflict-free 16-row MFMA operand reads)

__device__ __forceinline__ void round_robin_pair(int m, int t, int k, int& p, int& q) {
  if (k == 0) {
    p = m - 1;
    q = t;
  } else {
    p = (t + k) % (m - 1);
    q = (t - k + (m - 1)) % (m - 1);
  }
  if (p > q) { const int s = p; p = q; q = s; }
}

// global index of local index e (0..63) of the block pair (I, J)
__device__ __forceinline__ int64_t gidx(int I, int J, int e) { return (int64_t)(e < BS ? I : J) * BS + (e & (BS - 1)); }

__global__ __launch_bounds__(256) void block_solve_kernel(const double* __restrict__ A, int64_t N, int nb, int t,
                                                          const double* __restrict__ anorm, double* __restrict__ Rg,
                                                          unsigned* __restrict__ rotations) {
  extern __shared__ double lds_solve[];
  double* S = lds_solve;                 // [64][SP]
  double* R = S + SB * SP;               // [64][SP]
  double* rc = R + SB * SP;              // [32] cosines of the inner step
  double* rs = rc + 32;                  // [32] sines
  unsigned short* lut = reinterpret_cast<unsigned short*>(rs + 32);  // [528] upper triangle of the 32 x 32 grid of
                                                                     // rotation pairs: (ka << 8) | kb
  unsigned char* pp = reinterpret_cast<unsigned char*>(lut + 528);   // [32] first index of every rotation pair
  unsigned char* qq = pp + 32;                                       // [32] second index
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int I, J;
  round_robin_pair(nb, t, blockIdx.x, I, J);
  for (int i = tid; i < SB * SB; i += 256) {
    const int r = i >> 6, c = i & 63;
    S[r * SP + c] = A[gidx(I, J, r) * N + gidx(I, J, c)];
    R[r * SP + c] = (r == c) ? 1.0 : 0.0;
  }
  if (tid < 32) {
    const int off = tid * 32 - tid * (tid - 1) / 2;
    for (int kb = tid; kb < 32; ++kb) lut[off + kb - tid] = (unsigned short)((tid << 8) | kb);
  }
  const double a_norm = anorm[0];
  unsigned applied = 0;
  __syncthreads();
  // Static work assignment (an inner step is latency-bound: LDS round trips and two barriers, so every dependent trip
  // counts - 170 us per call with the indices looked up inside the step, see profiles/README.md):
  //   S update: this thread's rotation-pair blocks (ka, kb), ka <= kb, are fixed for the whole sweep (528 blocks, <= 3 each)
  //   R update: rotation k = tid & 31 on the rows (tid >> 5) + 8 j, j = 0..7
  int bka[3], bkb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = tid + 256 * j;
    const unsigned short e = (i < 528) ? lut[i] : (unsigned short)0xFFFF;
    bka[j] = (i < 528) ? (e >> 8) : -1;
    bkb[j] = e & 255;
  }
  const int rk = tid & 31, rrow0 = tid >> 5;
  for (int ts = 0; ts < SB - 1; ++ts) {
    if (wave == 0) {
      bool rotated = false;
      if (lane < 32) {
        int p, q;
        round_robin_pair(SB, ts, lane, p, q);
        const double app = S[p * SP + p], aqq = S[q * SP + q], apq = S[p * SP + q];
        double c = 1.0, s = 0.0;
        // rotate when |apq| > max(1e-19 |A|, 2e-15 sqrt|app aqq|), compared squared (no square root on the serial path).
        // The relative bound sits ABOVE the rounding noise of the blocked update: a large-angle rotation inside a
        // cluster of (nearly) equal eigenvalues leaves (R^T S R)_pq = O(eps) sqrt|app aqq| behind - the product does not
        // return the exact zero a scalar rotation writes - and with the scalar form's 1e-17 that residue was rotated again
        // sweep after sweep (a PCA-whitened precision matrix, all eigenvalues ~ 1, did not settle in 30 sweeps).
        // Small-angle rotations leave no such residue (their product terms are O(angle) small), so 2e-15 costs no accuracy
        // where eigenvalues are separated: measured |w - LAPACK| <= 8e-12 |A| up to n = 2048.
        if (apq * apq > fmax(1e-38 * a_norm * a_norm, 4e-30 * fabs(app * aqq))) {
          // tan of the rotation angle: t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq);
          // with x = aqq - app, y = 2 apq:  t = sign(x y) |y| / (|x| + sqrt(x^2 + y^2)) - one division instead of two,
          // reciprocal and reciprocal square root from the hardware seeds + Newton steps (1-2 ulp: a rotation only has to
          // be orthogonal to rounding, c^2 + s^2 = 1, which the construction c = rsqrt(1 + t^2), s = t c gives)
          const double x = aqq - app, y = apq + apq;
          const double hyp2 = fma(x, x, y * y);
          double r = __builtin_amdgcn_rsq(hyp2);
          r = r * fma(-0.5 * hyp2 * r, r, 1.5);
          r = r * fma(-0.5 * hyp2 * r, r, 1.5);
          const double den = fabs(x) + hyp2 * r;                 // |x| + sqrt(x^2 + y^2) >= |y| > 0
          double inv = __builtin_amdgcn_rcp(den);
          inv = fma(fma(-den, inv, 1.0), inv, inv);
          inv = fma(fma(-den, inv, 1.0), inv, inv);
          const double tt = ((x == 0.0 || (x > 0.0) == (y > 0.0)) ? fabs(y) : -fabs(y)) * inv;  // theta = +-0 counts as +
          const double w = fma(tt, tt, 1.0);
          double rw = __builtin_amdgcn_rsq(w);
          rw = rw * fma(-0.5 * w * rw, rw, 1.5);
          rw = rw * fma(-0.5 * w * rw, rw, 1.5);
          c = rw;
          s = tt * rw;
          rotated = true;
        }
        rc[lane] = c; rs[lane] = s;
        pp[lane] = (unsigned char)p; qq[lane] = (unsigned char)q;
      }
      applied += (unsigned)__popcll(__ballot(rotated));
    }
    __syncthreads();
    // every operand of the step is requested before anything is computed (two LDS round trips in all)
    double ca[3], sa[3], cb[3], sb[3];
    int pa[3], qa[3], pb[3], qb[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ka = bka[j] < 0 ? 0 : bka[j], kb = bkb[j] & 31;
      ca[j] = rc[ka]; sa[j] = rs[ka]; cb[j] = rc[kb]; sb[j] = rs[kb];
      pa[j] = pp[ka]; qa[j] = qq[ka]; pb[j] = pp[kb]; qb[j] = qq[kb];
    }
    const double c_r = rc[rk], s_r = rs[rk];
    const int p_r = pp[rk], q_r = qq[rk];
    double b00[3], b01[3], b10[3], b11[3], rp[8], rq[8];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      b00[j] = S[pa[j] * SP + pb[j]]; b01[j] = S[pa[j] * SP + qb[j]];
      b10[j] = S[qa[j] * SP + pb[j]]; b11[j] = S[qa[j] * SP + qb[j]];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      rp[j] = R[(rrow0 + 8 * j) * SP + p_r];
      rq[j] = R[(rrow0 + 8 * j) * SP + q_r];
    }
    // S <- J^T S J on the upper triangle of rotation-pair blocks (disjoint 2 x 2 blocks, mirrored)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (bka[j] < 0 || (sa[j] == 0.0 && sb[j] == 0.0)) continue;
      const double t00 = ca[j] * b00[j] - sa[j] * b10[j], t01 = ca[j] * b01[j] - sa[j] * b11[j];
      const double t10 = sa[j] * b00[j] + ca[j] * b10[j], t11 = sa[j] * b01[j] + ca[j] * b11[j];
      double n00 = t00 * cb[j] - t01 * sb[j], n01 = t00 * sb[j] + t01 * cb[j];
      double n10 = t10 * cb[j] - t11 * sb[j], n11 = t10 * sb[j] + t11 * cb[j];
      if (bka[j] == (bkb[j] & 31)) { n01 = 0.0; n10 = 0.0; }  // the rotated pair is annihilated by construction
      S[pa[j] * SP + pb[j]] = n00; S[pb[j] * SP + pa[j]] = n00;
      S[pa[j] * SP + qb[j]] = n01; S[qb[j] * SP + pa[j]] = n01;
      S[qa[j] * SP + pb[j]] = n10; S[pb[j] * SP + qa[j]] = n10;
      S[qa[j] * SP + qb[j]] = n11; S[qb[j] * SP + qa[j]] = n11;
    }
    // R <- R J
    if (s_r != 0.0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        R[(rrow0 + 8 * j) * SP + p_r] = rp[j] * c_r - rq[j] * s_r;
        R[(rrow0 + 8 * j) * SP + q_r] = rp[j] * s_r + rq[j] * c_r;
      }
    }
    __syncthreads();
  }
  double* out = Rg + (int64_t)blockIdx.x * SB * SB;
  for (int i = tid; i < SB * SB; i += 256) out[i] = R[(i >> 6) * SP + (i & 63)];
  if (tid == 0 && applied) atomicAdd(rotations, applied);
}

// 16 rows x 64 columns of X * Rm (X: LDS [rows][UP], this wave's rows x0..; Rm: LDS [64][UP]) -> acc[ct] (C layout)
__device__ __forceinline__ void rows_times(const double* X, const double* Rm, int li, int lg, d4 (&acc)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k = 0; k < SB; k += 4) {
    const double av = X[li * UP + k + lg];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Rm[(k + lg) * UP + 16 * c + li], acc[c], 0, 0, 0);
  }
}

// grid (np, np + row tiles of V): y < np: A[IJ_a, IJ_b] <- R_a^T A[IJ_a, IJ_b] R_b (a = y <= b = x, mirrored);
//                                 y >= np: V[rows, IJ_b] <- V[rows, IJ_b] R_b for the 64-row tile y - np
__global__ __launch_bounds__(256) void block_update_kernel(double* __restrict__ A, double* __restrict__ V, int64_t N,
                                                           int nb, int t, const double* __restrict__ Rg) {
  extern __shared__ double lds[];
  double* Bm = lds;                 // [64][UP]: the block; then T = R_a^T B (each wave its own 16 rows)
  double* Ra = lds + SB * UP;       // [64][UP]
  double* Rb = lds + 2 * SB * UP;   // [64][UP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int np = nb / 2;
  const int b = blockIdx.x;
  int Ib, Jb;
  round_robin_pair(nb, t, b, Ib, Jb);
  const double* Rbg = Rg + (int64_t)b * SB * SB;
  if ((int)blockIdx.y >= np) {  // ---- eigenvector rows
    const int64_t r0 = (int64_t)(blockIdx.y - np) * SB;
    for (int i = tid; i < SB * SB; i += 256) {
      const int r = i >> 6, c = i & 63;
      Bm[r * UP + c] = V[(r0 + r) * N + gidx(Ib, Jb, c)];
      Rb[r * UP + c] = Rbg[i];
    }
    __syncthreads();
    d4 acc[4];
    rows_times(Bm + 16 * wave * UP, Rb, li, lg, acc);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        V[(r0 + 16 * wave + lg + 4 * r) * N + gidx(Ib, Jb, 16 * c + li)] = acc[c][r];
    return;
  }
  const int a = blockIdx.y;
  if (a > b) return;
  int Ia, Ja;
  round_robin_pair(nb, t, a, Ia, Ja);
  const double* Rag = Rg + (int64_t)a * SB * SB;
  for (int i = tid; i < SB * SB; i += 256) {
    const int r = i >> 6, c = i & 63;
    Bm[r * UP + c] = A[gidx(Ia, Ja, r) * N + gidx(Ib, Jb, c)];
    Ra[r * UP + c] = Rag[i];
    Rb[r * UP + c] = Rbg[i];
  }
  __syncthreads();
  // T[16w .., :] = (R_a^T B)[16w .., :]: A operand = R_a^T, i.e. element [i][k] = Ra[k][i]
  d4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k = 0; k < SB; k += 4) {
    const double av = Ra[(k + lg) * UP + 16 * wave + li];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bm[(k + lg) * UP + 16 * c + li], acc[c], 0, 0, 0);
  }
  __syncthreads();  // every wave has read all of B: its rows 16w .. now hold this wave's rows of T
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) Bm[(16 * wave + lg + 4 * r) * UP + 16 * c + li] = acc[c][r];
  __syncthreads();
  rows_times(Bm + 16 * wave * UP, Rb, li, lg, acc);
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * wave + lg + 4 * r, j = 16 * c + li;
      const int64_t gi = gidx(Ia, Ja, i), gj = gidx(Ib, Jb, j);
      if (a == b) {
        if (i > j) continue;  // the diagonal block: upper triangle mirrored -> exactly symmetric
        A[gi * N + gj] = acc[c][r];
        A[gj * N + gi] = acc[c][r];
      } else {
        A[gi * N + gj] = acc[c][r];
        A[gj * N + gi] = acc[c][r];
      }
    }
}

// V = I and the partial sums of squares of A, many workgroups; then one workgroup adds the partials in a fixed order
// (the norm enters the rotation threshold: it must come out the same on every run and every rank)
__global__ __launch_bounds__(256) void block_init_kernel(const double* __restrict__ A, double* __restrict__ V, int64_t N,
                                                         double* __restrict__ partial) {
  __shared__ double part[256];
  double s = 0.0;
  const int64_t total = N * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const double a = A[i];
    s += a * a;
    V[i] = (i / N == i % N) ? 1.0 : 0.0;
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(256) void block_norm_kernel(const double* __restrict__ partial, int count,
                                                         double* __restrict__ anorm) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += partial[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) anorm[0] = sqrt(part[0]);
}

}  // namespace

// Padded dimension of the blocked solver: a whole, EVEN number of 32-column blocks (zero rows / columns stay decoupled:
// a rotation needs a non-zero off-diagonal entry).
extern "C" int64_t runia_eigh_block_padded(int64_t n) {
  if (n <= 0) return 0;
  const int64_t nb = (n + BS - 1) / BS;
  return ((nb + 1) & ~1ll) * BS;
}

extern "C" size_t runia_eigh_block_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  const int64_t np = runia_eigh_block_padded(n) / (2 * BS);
  return 64 + (size_t)np * SB * SB * sizeof(double);  // norm (runia_eigh_init_f64's slot) + one R per block pair
}

// V = I (N x N) and the Frobenius norm of the padded A into the workspace; call once before the sweeps.
extern "C" int runia_eigh_block_init_f64(const double* A, double* V, int64_t N, void* workspace, size_t workspace_bytes,
                                         runia_stream_t stream) {
  if (N <= 0 || N > 32768 || N % (2 * BS) != 0 || !A || !V) return RUNIA_E_INVALID;
  if (!workspace || workspace_bytes < runia_eigh_block_workspace_bytes(N) || (((uintptr_t)workspace) & 15) != 0)
    return RUNIA_E_WORKSPACE;
  double* anorm = reinterpret_cast<double*>(workspace);
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 64);  // the R buffer, not yet in use
  const int64_t total = N * N;
  const int grid = (int)((total / 256 < 1024) ? (total / 256 > 0 ? total / 256 : 1) : 1024);  // <= 4096 doubles of R
  hipStream_t s = as_stream(stream);
  block_init_kernel<<<grid, 256, 0, s>>>(A, V, N, partial);
  block_norm_kernel<<<1, 256, 0, s>>>(partial, grid, anorm);
  return runia_check_launch();
}

// One blocked Jacobi sweep in place on the PADDED matrices (N = runia_eigh_block_padded(n); A zero outside its n x n
// corner, V = I and the norm from runia_eigh_block_init_f64): A -> J^T A J, V -> V J.  `rotations`
// (device) is incremented by the number of rotations applied; a sweep that adds zero has converged.  No synchronisation.
extern "C" int runia_eigh_block_sweep_f64(double* A, double* V, int64_t N, void* workspace, size_t workspace_bytes,
                                          unsigned* rotations, runia_stream_t stream) {
  if (N <= 0 || N > 32768 || N % (2 * BS) != 0 || !A || !V || !rotations) return RUNIA_E_INVALID;
  if (!workspace || workspace_bytes < runia_eigh_block_workspace_bytes(N) || (((uintptr_t)workspace) & 15) != 0)
    return RUNIA_E_WORKSPACE;
  const int nb = (int)(N / BS), np = nb / 2;
  const double* anorm = reinterpret_cast<const double*>(workspace);
  double* Rg = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 64);
  hipStream_t s = as_stream(stream);
  const size_t lds = (size_t)3 * SB * UP * sizeof(double);  // 101 376 bytes
  const size_t lds_s = (size_t)(2 * SB * SP + 64) * sizeof(double) + 528 * sizeof(unsigned short) + 64;  // 68 192 bytes
  static std::atomic<uint64_t> lds_ok{0}, lds_ok_s{0};
  if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(block_update_kernel), 104 * 1024, lds_ok) != RUNIA_OK)
    return RUNIA_E_LAUNCH;
  if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(block_solve_kernel), 72 * 1024, lds_ok_s) != RUNIA_OK)
    return RUNIA_E_LAUNCH;
  const int steps = nb > 2 ? nb - 1 : 1;
  for (int t = 0; t < steps; ++t) {
    block_solve_kernel<<<np, 256, lds_s, s>>>(A, N, nb, t, anorm, Rg, rotations);
    block_update_kernel<<<dim3(np, np + (unsigned)(N / SB)), 256, lds, s>>>(A, V, N, nb, t, Rg);
  }
  return runia_check_launch();
}
