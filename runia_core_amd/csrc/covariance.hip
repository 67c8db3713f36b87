// f1 (SURVEY 8f "next #1"): the covariance of setup() on the device.
// Replaces np.cov(X.T, bias=1) inside sklearn EmpiricalCovariance.fit
// (reference inference/postprocessors.py:217-220, inference/funcs.py:62-66):
//     mu = mean(X, 0);  Xc = X - mu;  cov = (Xc^T Xc) * (1/N)          all in f64 (f32 rows are promoted first)
// The Gram matrix is a SYRK-shaped contraction over the N rows on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64, A[i][k] = Xc[k][i], B[k][j] = Xc[k][j]).  A workgroup owns a 128x128 output tile of the upper
// triangle and a slice of the rows (split-K); partial tiles are summed in a fixed order by a second kernel, which also
// mirrors them, so the result is exactly symmetric and bit-reproducible run to run.  The eigen-decomposition behind pinvh stays a library call (host SciPy or
// torch.linalg.eigh) - see runia_core_amd/device_fit.py.
#include "common.hpp"

namespace {

constexpr int GT = 128;         // output tile edge (4 waves x 64 x 64)
constexpr int GK = 16;          // rows (k) staged per step
constexpr int GP = 144;         // doubles per staged row: == 16 mod 32 -> the two k rows of a 32-lane group miss each other

// Column sums of a slice of rows: a wave covers 256 columns of a row with one 16-byte load per lane (f32; two for f64), the
// four waves of a workgroup take every fourth row.  (One 4-byte load per lane and 13 slices: 0.42 ms for the 410 MB of
// 50 000 x 2048 f32, 1 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void col_sum_kernel(const T* __restrict__ x, double* __restrict__ partial,
                                                       int64_t N, int64_t D, int64_t rows_per_block) {
  // grid = (ceil(D/256), nblocks_rows); thread = (4 columns, row phase)
  __shared__ double red[4][256];
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 256 + 4 * lane;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < N) ? r0 + rows_per_block : N;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0) && col + 3 < D;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (vec) {
    for (int64_t r = r0 + phase; r < r1; r += 4) {
      if constexpr (sizeof(T) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(x + r * D + col);
        s[0] += (double)t.x; s[1] += (double)t.y; s[2] += (double)t.z; s[3] += (double)t.w;
      } else {
        const double2 t0 = *reinterpret_cast<const double2*>(x + r * D + col), t1 = *reinterpret_cast<const double2*>(x + r * D + col + 2);
        s[0] += t0.x; s[1] += t0.y; s[2] += t1.x; s[3] += t1.y;
      }
    }
  } else {
    for (int64_t r = r0 + phase; r < r1; r += 4)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < D) s[e] += (double)x[r * D + col + e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[phase][4 * lane + e] = s[e];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c < D) {
    const int t = threadIdx.x;
    partial[(int64_t)blockIdx.y * D + c] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  }
}

// mean = (sum of the slices' partial sums) / N: 64 columns per workgroup, the four waves take every fourth slice (fixed order)
__global__ __launch_bounds__(256) void col_mean_finish_kernel(const double* __restrict__ partial, double* __restrict__ mean,
                                                               int64_t D, int64_t nblocks, double inv_n) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + lane;
  double s = 0.0;
  if (col < D)
    for (int64_t b = phase; b < nblocks; b += 4) s += partial[b * D + col];
  red[phase][lane] = s;
  __syncthreads();
  if (phase == 0 && col < D) mean[col] = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) * inv_n;
}

// four consecutive columns c0 .. c0 + 3 of row r AS LOADED (zero outside the matrix / the slice's rows); gram_centre
// widens and centres them.  Two steps on purpose: the loads are issued in front of a step's matrix instructions and their
// values first touched behind them - converted at the load, the wave waited for its loads before it multiplied.
template <typename T>
struct GramRaw { T v[4]; };
// Branch-free: the address is clamped into the matrix (row N - 1, column 0 / D - 1) and whatever lies outside is zeroed when
// the values are centred - with `if (inside) load` per thread the loop was a chain of exec-mask branches.
template <typename T, bool VEC>
__device__ __forceinline__ GramRaw<T> gram_fetch(const T* __restrict__ x, int64_t r, int64_t N, int64_t c0, int64_t D) {
  GramRaw<T> g;
  const T* row = x + (r < N ? r : N - 1) * D;
  if constexpr (VEC) {  // D % 4 == 0, 16-byte aligned rows: c0 is inside or the whole quad is outside
    const T* p = row + (c0 < D ? c0 : 0);
    if constexpr (sizeof(T) == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      g.v[0] = t.x; g.v[1] = t.y; g.v[2] = t.z; g.v[3] = t.w;
    } else {
      const double2 t0 = *reinterpret_cast<const double2*>(p), t1 = *reinterpret_cast<const double2*>(p + 2);
      g.v[0] = t0.x; g.v[1] = t0.y; g.v[2] = t1.x; g.v[3] = t1.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) g.v[e] = row[c0 + e < D ? c0 + e : D - 1];
  }
  return g;
}

// The Gram matrix is symmetric: only the tile pairs (ti <= tj) are computed - blockIdx.x walks the upper triangle of
// 128 x 128 tiles row by row -, the finishing kernel mirrors them.  A workgroup = one tile pair x one slice of the rows:
// 4 waves of 64 x 64 (4 x 4 MFMA tiles, 128 accumulator registers), 16 rows staged per step into one of two LDS buffers
// (the next step's rows in flight - global -> registers - under this step's matrix instructions, one barrier per step); a diagonal pair stages one strip and reads both operands from it.
// (Round 5: 64 x 64 tiles over the full square, single-buffered: 12.5 ms at 50 000 x 2048, 0.43 of the f64 matrix peak.)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void gram_kernel(const T* __restrict__ x, const double* __restrict__ mean,
                                                    double* __restrict__ partial, int64_t N, int64_t D,
                                                    int64_t rows_per_split, int ntile) {
  extern __shared__ double gram_lds[];  // sa [2][GK][GP] | sb [2][GK][GP] (73 728 bytes)
  double* sa = gram_lds;
  double* sb = gram_lds + 2 * GK * GP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int li = lane & 15, lg = lane >> 4;
  int ti = 0, tj;
  {
    int p = (int)blockIdx.x, len = ntile;  // row ti of the triangle holds ntile - ti pairs
    while (p >= len) { p -= len; --len; ++ti; }
    tj = ti + p;
  }
  const bool diag = ti == tj;  // (uniform)
  const int64_t i0 = (int64_t)ti * GT, j0 = (int64_t)tj * GT;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r_end = (r_begin + rows_per_split < N) ? r_begin + rows_per_split : N;
  const int srow = tid >> 5, scol = (tid & 31) * 4;  // this thread's share of a staged strip: 4 columns of one row
  double ma[4], mb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ma[e] = (i0 + scol + e < D) ? mean[i0 + scol + e] : 0.0;
    mb[e] = (j0 + scol + e < D) ? mean[j0 + scol + e] : 0.0;
  }
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
  // A step = 16 rows = 64 matrix instructions per wave (~1.7 us): the next step's rows are requested before them and stored
  // to the other LDS buffer after them.  (8 rows per step left ~0.85 us between a load and its use: 5.6 ms at 50 000 x 2048,
  // 0.51 of the matrix peak, the waves waiting on their loads at the end of every step.  Two register sets taking turns
  // over a loop body of two steps doubled the accumulators - 432 registers, one wave per SIMD: 7.6 ms.)
  GramRaw<T> va[GK / 8], vb[GK / 8];
  bool inside[GK / 8];  // the fetched row exists (a row past the slice is staged as zeros, not as -mean)
  auto fetch = [&](int64_t row0) {
#pragma unroll
    for (int h = 0; h < GK / 8; ++h) {
      inside[h] = row0 + 8 * h + srow < r_end;
      va[h] = gram_fetch<T, VEC>(x, row0 + 8 * h + srow, N, i0 + scol, D);
      if (!diag) vb[h] = gram_fetch<T, VEC>(x, row0 + 8 * h + srow, N, j0 + scol, D);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int h = 0; h < GK / 8; ++h) {
      double c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = (inside[h] && i0 + scol + e < D) ? (double)va[h].v[e] - ma[e] : 0.0;
      double* da = sa + ((buf * GK + 8 * h + srow) * GP + scol);
      *reinterpret_cast<double2*>(da) = make_double2(c[0], c[1]);
      *reinterpret_cast<double2*>(da + 2) = make_double2(c[2], c[3]);
      if (!diag) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = (inside[h] && j0 + scol + e < D) ? (double)vb[h].v[e] - mb[e] : 0.0;
        double* db = sb + ((buf * GK + 8 * h + srow) * GP + scol);
        *reinterpret_cast<double2*>(db) = make_double2(c[0], c[1]);
        *reinterpret_cast<double2*>(db + 2) = make_double2(c[2], c[3]);
      }
    }
  };
  fetch(r_begin);
  stage_store(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += GK) {
    const bool more = r0 + GK < r_end;  // (uniform)
#if !defined(GRAM_ABLATE)  // (timing experiments only: 1 = no further loads, 2 = no further loads or LDS stores)
    if (more) fetch(r0 + GK);
#endif
    const double* pa = sa + buf * GK * GP;
    const double* pb = (diag ? sa : sb) + buf * GK * GP;
#pragma unroll
    for (int s = 0; s < GK / 4; ++s) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = pa[(4 * s + lg) * GP + wi * 64 + t * 16 + li];
        b[t] = pb[(4 * s + lg) * GP + wj * 64 + t * 16 + li];
      }
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc[ta][tb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the loaded values are first touched here)
#if !(defined(GRAM_ABLATE) && GRAM_ABLATE >= 2)
    if (more) stage_store(buf ^ 1);
#endif
    __syncthreads();
    buf ^= 1;
  }
  double* out = partial + (int64_t)blockIdx.y * D * D;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = i0 + wi * 64 + ta * 16 + lg + 4 * r, j = j0 + wj * 64 + tb * 16 + li;
        if (i < D && j < D) out[i * D + j] = acc[ta][tb][r];
      }
}

// cov = (sum of the splits' partial tiles) / N over the upper triangle of 32 x 32 blocks, mirrored through LDS into the
// lower one (the result is exactly symmetric; inside a diagonal block the entries above the diagonal are the ones kept)
__global__ __launch_bounds__(256) void gram_finish_kernel(const double* __restrict__ partial, double* __restrict__ cov, int64_t D,
                                                           int64_t splits, double inv_n, int nblk) {
  __shared__ double t[32][33];
  int bi = 0, bj;
  {
    int p = (int)blockIdx.x, len = nblk;
    while (p >= len) { p -= len; --len; ++bi; }
    bj = bi + p;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t DD = D * D;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int li = ty + 8 * q;
    const int64_t i = (int64_t)bi * 32 + li, j = (int64_t)bj * 32 + tx;
    double s = 0.0;
    if (i < D && j < D) {
      for (int64_t z = 0; z < splits; ++z) s += partial[z * DD + i * D + j];
      s *= inv_n;
      if (bi != bj || i <= j) cov[i * D + j] = s;
    }
    t[li][tx] = s;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int lj = ty + 8 * q;  // row of the mirrored block = column of the source
    const int64_t j = (int64_t)bj * 32 + lj, i = (int64_t)bi * 32 + tx;
    if (i < D && j < D && (bi != bj || i < j)) cov[j * D + i] = t[tx][lj];
  }
}

// Row slices per tile pair.  The workgroups are equal (one pair x one slice) and two fit a compute unit (LDS, registers),
// so the launch runs in rounds of 2 x CUs workgroups and its time goes as ceil(pairs * s / slots) / s: the slice count with
// the fullest last round is taken (D = 2048: 136 pairs; 8 slices = 2.1 rounds run as 3, 5.7 ms; 15 slices = 3.98 rounds).
// At least 256 rows per slice, at most 64 slices; of near-equal candidates (2 %) the smallest (workspace = s * D^2 doubles).
int64_t splits_for(int64_t N, int64_t D) {
  const int64_t nt = (D + GT - 1) / GT, pairs = nt * (nt + 1) / 2, slots = 2 * runia_cu_count();
  int64_t max_s = (N + 255) / 256;
  if (max_s > 64) max_s = 64;
  if (max_s < 1) max_s = 1;
  int64_t best = 1;
  double best_cost = 1e300;
  for (int64_t sp = 1; sp <= max_s; ++sp) {
    const double cost = (double)((pairs * sp + slots - 1) / slots) / (double)sp;
    if (cost < best_cost * 0.98) { best_cost = cost; best = sp; }
  }
  return best;
}

template <typename T>
int covariance_impl(const T* x, double* mean, double* cov, void* workspace, size_t workspace_bytes, int64_t N,
                    int64_t D, runia_stream_t stream) {
  if (N <= 0 || D <= 0 || !x || !mean || !cov) return RUNIA_E_INVALID;
  const int64_t splits = splits_for(N, D);
  const int64_t mean_blocks = (N + 255) / 256 < 256 ? (N + 255) / 256 : 256;
  const size_t need = (size_t)(splits * D * D + mean_blocks * D) * sizeof(double);
  if (!workspace || workspace_bytes < need) return RUNIA_E_WORKSPACE;
  double* part = reinterpret_cast<double*>(workspace);
  double* mpart = part + splits * D * D;
  hipStream_t s = as_stream(stream);
  const int64_t rows_per_block = (N + mean_blocks - 1) / mean_blocks;
  col_sum_kernel<T><<<dim3((unsigned)((D + 255) / 256), (unsigned)mean_blocks), 256, 0, s>>>(x, mpart, N, D, rows_per_block);
  col_mean_finish_kernel<<<(unsigned)((D + 63) / 64), 256, 0, s>>>(mpart, mean, D, mean_blocks, 1.0 / (double)N);
  const int64_t rows_per_split = ((N + splits - 1) / splits + GK - 1) / GK * GK;
  const int nt = (int)((D + GT - 1) / GT), nblk = (int)((D + 31) / 32);
  constexpr int kGramLds = 2 * 2 * GK * GP * (int)sizeof(double);
  static std::atomic<uint64_t> lds_ok{0};
  static std::atomic<uint64_t> lds_ok_v{0};
  const dim3 ggrid((unsigned)(nt * (nt + 1) / 2), (unsigned)splits);
  if (((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0)) {
    if (int rc = runia_allow_dynamic_lds(reinterpret_cast<const void*>(gram_kernel<T, true>), kGramLds, lds_ok_v)) return rc;
    gram_kernel<T, true><<<ggrid, 256, kGramLds, s>>>(x, mean, part, N, D, rows_per_split, nt);
  } else {
    if (int rc = runia_allow_dynamic_lds(reinterpret_cast<const void*>(gram_kernel<T, false>), kGramLds, lds_ok)) return rc;
    gram_kernel<T, false><<<ggrid, 256, kGramLds, s>>>(x, mean, part, N, D, rows_per_split, nt);
  }
  gram_finish_kernel<<<(unsigned)((int64_t)nblk * (nblk + 1) / 2), 256, 0, s>>>(part, cov, D, splits, 1.0 / (double)N, nblk);
  return runia_check_launch();
}

}  // namespace

extern "C" size_t runia_covariance_workspace_bytes(int64_t N, int64_t D) {
  if (N <= 0 || D <= 0) return 0;
  const int64_t mean_blocks = (N + 255) / 256 < 256 ? (N + 255) / 256 : 256;
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
