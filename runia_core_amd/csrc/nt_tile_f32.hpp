// 128 x 128 NT tile engine on the f32 matrix cores (v_mfma_f32_32x32x2_f32, exact-f32 fma chains), shared by the kNN distance
// kernel (knn_f32.hip: d2[q, m] = |q|^2 + |b_m|^2 - 2 q.b_m) and the final linear layer of the logits / features postprocessors
// (linear.hip: out[q, m] = min(x_q, clip).w_m + bias_m); the triangular whitening kernel of the class-wise Gaussians (gmm_tril.hip)
// uses its staging helpers and tile order.  Included by exactly those three translation units.
//
// A workgroup owns a 128-row x 128-row tile of two K-contiguous operands (4 waves x 64x64, i.e. 2x2 MFMA tiles of 32x32 per
// wave), staged 32 k at a time into LDS with a 34-float pitch (conflict-free ds_read_b64: one 8-byte read feeds two MFMA
// k-steps, A and B use the same k permutation).  The tile grid is walked in XCD-aware super-tiles (see knn_dist_kernel).
#pragma once
#include "common.hpp"

namespace {

constexpr float kFltMax = 3.4028234663852886e38f;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TQ = 128, TB = 128, KCH = 32, KP = 34;
#ifndef KNN_SB
#define KNN_SB 8
#define KNN_SQ 8
#endif
constexpr int kSuperB = KNN_SB, kSuperQ = KNN_SQ;  // super-tile of workgroups that share L2 lines (knn_dist_kernel)
// Query tiles of a super-tile: kSuperQ, or all of them when the batch has fewer (the super-tile is then kSuperB x nqt
// workgroups, every one of them with a tile).  Workgroups go to compute units in a fixed rotation: with a batch of one
// query tile in 8 x 8 super-tiles only every eighth workgroup had a tile, and those landed on one eighth of the compute
// units (100 queries against 50 000 x 2048: 0.45 ms, as much as 1 000).
__host__ __device__ inline int knn_super_q(int64_t nqt) { return nqt < kSuperQ ? (int)nqt : kSuperQ; }

// squared row norms; max_bits (optional): running maximum of the norms as an unsigned bit pattern (norms are >= 0)
__global__ __launch_bounds__(64 * kRowWaves) void row_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                    int64_t N, int64_t D, unsigned* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  for (int64_t row = (int64_t)blockIdx.x * kRowWaves + wave; row < N; row += (int64_t)gridDim.x * kRowWaves) {
    const float* p = x + row * D;
    float s = 0.f;
    if (vec) {
      // 16-byte loads, four independent partial sums
      const float4* p4 = reinterpret_cast<const float4*>(p);
      const int64_t n4 = D >> 2;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int64_t i = lane;
      for (; i + 192 < n4; i += 256) {
        const float4 a = p4[i], b = p4[i + 64], c = p4[i + 128], d = p4[i + 192];
        s0 = fmaf(a.x, a.x, s0); s0 = fmaf(a.y, a.y, s0); s0 = fmaf(a.z, a.z, s0); s0 = fmaf(a.w, a.w, s0);
        s1 = fmaf(b.x, b.x, s1); s1 = fmaf(b.y, b.y, s1); s1 = fmaf(b.z, b.z, s1); s1 = fmaf(b.w, b.w, s1);
        s2 = fmaf(c.x, c.x, s2); s2 = fmaf(c.y, c.y, s2); s2 = fmaf(c.z, c.z, s2); s2 = fmaf(c.w, c.w, s2);
        s3 = fmaf(d.x, d.x, s3); s3 = fmaf(d.y, d.y, s3); s3 = fmaf(d.z, d.z, s3); s3 = fmaf(d.w, d.w, s3);
      }
      for (; i < n4; i += 64) {
        const float4 a = p4[i];
        s0 = fmaf(a.x, a.x, s0); s0 = fmaf(a.y, a.y, s0); s0 = fmaf(a.z, a.z, s0); s0 = fmaf(a.w, a.w, s0);
      }
      s = (s0 + s1) + (s2 + s3);
    } else {
      for (int64_t i = lane; i < D; i += 64) s = fmaf(p[i], p[i], s);
    }
    s = wave_sum_f32(s);
    if (lane == 0) {
      out[row] = s;
      // Running maximum of the norms (NaN / inf rows do not set the range).  One returning atomic per row on ONE word
      // serialises at ~88 per microsecond: 50 000 bank rows took 0.58 ms for 0.08 ms of reading.  The atomic is only
      // issued when the value beats what the word already holds (a stale read only costs a redundant atomic).
      if (max_bits && s < INFINITY) {
        const unsigned b = __float_as_uint(s);
        if (b > __atomic_load_n(max_bits, __ATOMIC_RELAXED)) atomicMax(max_bits, b);
      }
    }
  }
}

// stage a [rows x 32] K-chunk of a K-contiguous matrix into LDS (zero filled outside the matrix), in two halves so that
// the global loads of chunk i+1 are in flight while the matrix cores work on chunk i:
//   load_chunk : global -> 16 registers per thread (128 rows x 2 halves of 16 floats)
//   store_chunk: registers -> LDS
__device__ __forceinline__ void load_chunk(const float* __restrict__ src, int64_t row0, int64_t nrows, int64_t D,
                                           int64_t k0, float (&v)[16], int tid, bool vec) {
  const int row = tid >> 1, half = tid & 1;
  const int64_t gr = row0 + row;
  const float* p = src + gr * D + k0 + half * 16;
  if (gr < nrows && vec && k0 + half * 16 + 16 <= D) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = reinterpret_cast<const float4*>(p)[j];
      v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (gr < nrows && k0 + half * 16 + j < D) ? p[j] : 0.f;
  }
}

template <bool CLIP>
__device__ __forceinline__ void store_chunk(const float (&v)[16], float (*dst)[KP], int tid, float clip_max = INFINITY) {
  const int row = tid >> 1, half = tid & 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // the clip (ReAct) costs two vector instructions per element: only the instantiation that clips pays for it
    // np.clip keeps a NaN activation (fminf would return the other operand and hide it)
    const float a = (CLIP && v[2 * j] > clip_max) ? clip_max : v[2 * j];
    const float b = (CLIP && v[2 * j + 1] > clip_max) ? clip_max : v[2 * j + 1];
    *reinterpret_cast<float2*>(&dst[row][half * 16 + 2 * j]) = make_float2(a, b);
  }
}

// EPI_DIST  : out[q, m] = max(0, |q|^2 + |b_m|^2 - 2 q.b_m)                  (kNN distances; qn = |q|^2, bn = |b|^2)
// EPI_LINEAR: out[q, m] = min(x_q, clip).w_m + bias_m                          (final linear layer; bn = bias, qn unused)
enum NtEpilogue { EPI_DIST = 0, EPI_LINEAR = 1 };

template <int EPI>
__global__ __launch_bounds__(256) void knn_dist_kernel(const float* __restrict__ q, const float* __restrict__ bank,
                                                        const float* __restrict__ qn, const float* __restrict__ bn,
                                                        float* __restrict__ dist, int64_t Q, int64_t M, int64_t D,
                                                        float clip_max) {
  __shared__ __attribute__((aligned(16))) float As[TQ][KP];
  __shared__ __attribute__((aligned(16))) float Bs[TB][KP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wb = wave & 1;   // 2 x 2 waves
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order of the tile grid.  Consecutive workgroup ids go round-robin over the 8 XCDs; here ids i, i + 8,
  // i + 16, ... (one XCD's share) walk a super-tile of kSuperB bank tiles x kSuperQ query tiles row by row, and the
  // super-tiles themselves are dealt round-robin to the XCDs.  The 64 workgroups of a super-tile are resident on one
  // XCD at the same time and sweep K in step, so every 16 KB slice of a query or bank tile is fetched into that L2
  // once and read 8 times (with the plain (bank tile, query tile) grid a bank tile was fetched again for every query
  // tile: 64 x 410 MB per 8 192-query chunk).
  int64_t q0, m0;
  {
    const int64_t nbt = (M + TB - 1) / TB, nqt = (Q + TQ - 1) / TQ;
    const int sq = knn_super_q(nqt), wps = kSuperB * sq;  // workgroups per super-tile (see knn_super_q)
    const int64_t nqg = (nqt + sq - 1) / sq;
    const int64_t l = blockIdx.x >> 3;
    const int64_t st = (l / wps) * 8 + (blockIdx.x & 7);  // super-tile of this workgroup
    const int r = (int)(l % wps);
    const int64_t bt = (st / nqg) * kSuperB + r % kSuperB, qt = (st % nqg) * sq + r / kSuperB;
    if (bt >= nbt || qt >= nqt) return;  // padding of the grid (uniform over the workgroup)
    q0 = qt * TQ;
    m0 = bt * TB;
  }
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)q) & 15) == 0) && ((((uintptr_t)bank) & 15) == 0);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float ra[16], rb[16];
  load_chunk(q, q0, Q, D, 0, ra, tid, vec);
  load_chunk(bank, m0, M, D, 0, rb, tid, vec);
  for (int64_t k0 = 0; k0 < D; k0 += KCH) {
    __syncthreads();  // every wave has finished reading the previous chunk
    store_chunk<EPI == EPI_LINEAR>(ra, As, tid, clip_max);
    store_chunk<false>(rb, Bs, tid);
    __syncthreads();
    if (k0 + KCH < D) {  // next chunk's loads fly while the matrix cores consume this one
      load_chunk(q, q0, Q, D, k0 + KCH, ra, tid, vec);
      load_chunk(bank, m0, M, D, k0 + KCH, rb, tid, vec);
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
  // epilogue: C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t col = m0 + wb * 64 + b * 32 + li;
      const float bnv = (col < M && bn) ? bn[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = q0 + wq * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < Q && col < M) {
          if constexpr (EPI == EPI_DIST) {
            const float d = (qn[row] + bnv) - 2.0f * acc[a][b][r];
            // faiss keeps a max-heap initialised with FLT_MAX and inserts a distance only if it compares smaller: a NaN
            // or infinite distance is never inserted, i.e. it counts as FLT_MAX (fmaxf alone would turn NaN into 0)
            dist[row * M + col] = (d == d) ? fminf(fmaxf(d, 0.f), kFltMax) : kFltMax;
          } else {
            dist[row * M + col] = acc[a][b][r] + bnv;
          }
        }
      }
    }
}

// 1-D grid of knn_dist_kernel for Q x M: whole super-tiles, a multiple of 8 of them
static inline unsigned knn_dist_grid(int64_t Q, int64_t M) {
  const int64_t nbt = (M + TB - 1) / TB, nqt = (Q + TQ - 1) / TQ;
  const int sq = knn_super_q(nqt);
  const int64_t st = ((nbt + kSuperB - 1) / kSuperB) * ((nqt + sq - 1) / sq);
  return (unsigned)(((st + 7) / 8) * 8 * kSuperB * sq);
}

}  // namespace
