// Pairwise-distance postprocessors.
//   a8  kNN   : faiss.IndexFlatL2.search(q, k) -> -(k-th smallest squared L2)
//               (reference inference/postprocessors.py:417-421, 873-880; faiss-gpu==1.7.2)
//   a9  LaRED : sklearn KernelDensity(gaussian).score_samples
//               (reference inference/postprocessors.py:118-128)
//
// kNN = the one f32 dense contraction of the path, on the matrix cores:
//   d2[q, m] = |q|^2 + |b_m|^2 - 2 q.b_m          (v_mfma_f32_32x32x2_f32, exact-f32 fma chain)
// A workgroup owns a 128-query x 128-bank-row tile (4 waves x 64x64, i.e. 2x2 MFMA tiles of 32x32 per wave);
// both operands are K-contiguous rows, staged 32 k at a time into LDS with a 34-float pitch (conflict-free
// ds_read_b64: one 8-byte read feeds two MFMA k-steps, A and B use the same k permutation).  Distances go to a
// row-chunked workspace [Qc, M]; the k-th order statistic of each row is then found by an 8-bit radix select over
// the distance bits (distances are clamped >= 0, so unsigned integer order == float order) - no sort, no top-k list.
// The tile grid is walked in XCD-aware super-tiles (see knn_dist_kernel).
#include "common.hpp"
#include "knn_perm.hpp"

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

// k-th smallest (1-based) of each row of dist [Q, M]: histogram select, one workgroup per row, followed by an exact
// refinement.  The norm-expansion distances carry ~K*eps cancellation error (4e-6 at D = 2048), while
// faiss's one-query path accumulates sum((q-b)^2) directly; so every bank row whose approximate distance lies within
// +-kRefineDelta of the selected value is re-measured with exact f32 differences and the k-th order statistic is
// re-taken among them (rows below the window keep their rank; a copied bank row gives exactly 0 again).
constexpr float kRefineRel = 5e-6f;  // refinement half-window relative to the row's range bound (2e-5 for unit vectors)
constexpr int kMaxCand = 512;

// Selection in two histogram passes.  The raw float bits of a row of distances share their top 8-9 bits (L2-normalised
// features: everything lies in [0, 4]), so an 8-bit radix select on the bits hammers one or two LDS counters in its
// first passes (measured: 1.26 ms per 8192 x 32768 chunk, five reads of the row).  Here a distance is mapped
// linearly onto a 24-bit key over the row's range bound (sqrt|q|^2 + sqrt(max|b|^2))^2, 12 bits per pass, 4096
// counters (1.03 ms, three reads).  A key bin is range / 2^24 wide; the refinement window (never narrower than two
// bins) then restores the exact k-th value.
// Fast path (one read of the row instead of three): the k-th smallest of ANY subset of the row is an upper bound of the
// row's k-th smallest, so the k-th smallest of a spread sample of 2048 entries (selected in LDS) gives a threshold tau;
// one pass over the row collects the entries <= tau (about k * M / 2048 of them) into LDS and the selection and the
// refinement window are taken among those.  Whenever that cannot be exact - more candidates than the LDS list holds
// (ties, a sample that does not represent the row), a window reaching beyond tau - the three-read path runs instead.
// Every row is written on every path: -FLT_MAX for a query whose norm is not finite (all its distances are
// incomparable, faiss returns its FLT_MAX fill), the approximate value before the refinement starts, the exact value
// after it.  A window holding more than kMaxCand candidates (duplicated bank rows, distances crowded into a few key
// bins by an outlier norm) takes the exact slow path: candidates are re-measured in place in the distance row (marked
// by the sign bit) and the wanted order statistic is taken by an 8-bit radix select over their bit patterns.
constexpr int kSample = 2048;    // strided sample of the row (fast path)
constexpr int kFastCap = 2048;   // candidates <= tau kept in LDS (the sample shares the buffer)

// 24-bit linear key of a distance
__device__ __forceinline__ unsigned dist_key(float d, float scale) {
  const float kf = d * scale;
  return (kf >= 16777215.0f) ? 16777215u : (unsigned)kf;
}

// rank-th smallest (1-based) 24-bit key among load(0..count-1): two 12-bit histogram passes, whole workgroup.
// `vec4` (optional): the same values as a 16-byte aligned array of count/4 float4 (count % 4 == 0), read 16 bytes per
// lane with two loads in flight.
template <class Load>
__device__ __forceinline__ unsigned hist_select24(Load load, int64_t count, unsigned rank, float scale, unsigned* hist,
                                                  unsigned* part, unsigned* sel, int tid, const float4* vec4 = nullptr) {
  unsigned bin1 = 0;
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < 16; ++j) hist[j * 256 + tid] = 0u;  // (consecutive lanes, consecutive words)
    __syncthreads();
    auto tally = [&](float d) {
      const unsigned key = dist_key(d, scale);
      if (pass == 0) atomicAdd(&hist[key >> 12], 1u);
      else if ((key >> 12) == bin1) atomicAdd(&hist[key & 4095u], 1u);
    };
    if (vec4) {
      const int64_t n4 = count >> 2;
      int64_t m = tid;
      for (; m + 256 < n4; m += 512) {
        const float4 a = vec4[m], b = vec4[m + 256];
        tally(a.x); tally(a.y); tally(a.z); tally(a.w);
        tally(b.x); tally(b.y); tally(b.z); tally(b.w);
      }
      for (; m < n4; m += 256) {
        const float4 a = vec4[m];
        tally(a.x); tally(a.y); tally(a.z); tally(a.w);
      }
    } else {
      for (int64_t m = tid; m < count; m += 256) tally(load(m));
    }
    __syncthreads();
    // Thread t owns bins 16 t .. 16 t + 15.  The group that holds the wanted rank is found by a parallel prefix sum over
    // the 256 group totals (wave scan + four wave totals) and its bin by that one thread from registers.  (One thread
    // walking the 256 totals and then 16 bins - every step a dependent LDS read - took ~25 000 cycles per pass, four
    // passes per row: half of the select kernel's time.)
    unsigned hb[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 v = reinterpret_cast<const uint4*>(hist + tid * 16)[j];
      hb[4 * j] = v.x; hb[4 * j + 1] = v.y; hb[4 * j + 2] = v.z; hb[4 * j + 3] = v.w;
    }
    unsigned ssum = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) ssum += hb[j];
    unsigned incl = ssum;  // inclusive prefix within the wave
    const int lane = tid & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) part[tid >> 6] = incl;
    __syncthreads();
    unsigned before = incl - ssum;  // entries in the groups below this one
    for (int w = 0; w < (tid >> 6); ++w) before += part[w];
    // exactly one owner: before < rank <= before + ssum; the last group takes whatever is left (as the serial walk did)
    if ((before < rank && rank <= before + ssum && tid < 255) || (tid == 255 && before < rank)) {
      unsigned r = rank - before, b = 0;
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        if (b == (unsigned)j && r > hb[j]) { r -= hb[j]; b = j + 1; }
      }
      sel[0] = tid * 16 + b;
      sel[1] = r;
    }
    __syncthreads();
    if (pass == 0) bin1 = sel[0];
    rank = sel[1];
    __syncthreads();
  }
  return (bin1 << 12) | sel[0];
}

// Exact squared L2 distance of two rows by one wave: lane l accumulates the elements l, l + 64, ... in that order (one
// fma chain per lane, as ever - the bits of the re-measured distances do not change), eight loads of each row in flight
// per lane instead of one dependent pair per trip (the refinement of a chunk of 8 192 rows: 0.21 -> 0.07 ms).
__device__ __forceinline__ float exact_sqdist_wave(const float* __restrict__ qr, const float* __restrict__ br, int64_t D,
                                                   int lane) {
  float acc = 0.f;
  int64_t i = lane;
  for (; i + 448 < D; i += 512) {
    float qa[8], ba[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { qa[u] = qr[i + 64 * u]; ba[u] = br[i + 64 * u]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float df = qa[u] - ba[u];
      acc = fmaf(df, df, acc);
    }
  }
  for (; i < D; i += 64) {
    const float df = qr[i] - br[i];
    acc = fmaf(df, df, acc);
  }
  return wave_sum_f32(acc);
}

// A handful of queries (serving: one image at a time): the 128-row tiles of the matrix-core kernels would compute 120+
// rows of padding per bank tile (one query against 50 000 x 2048: 0.6 ms).  Here one wave takes one bank row, reads it
// ONCE and measures it against every query with exact f32 differences - the same per-lane fma chain as
// exact_sqdist_wave, so the distances are the values the large paths arrive at by re-measurement and a row scores the
// same bits alone as inside a batch - and adds up the row's squared norm on the way (the selection's range bound).
constexpr int kSmallQ = 8;
__global__ __launch_bounds__(64 * kRowWaves) void knn_small_dist_kernel(const float* __restrict__ q,
                                                                        const float* __restrict__ bank,
                                                                        float* __restrict__ dist,
                                                                        unsigned* __restrict__ bn_max_bits, int Q, int64_t M,
                                                                        int64_t D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t m = (int64_t)blockIdx.x * kRowWaves + wave; m < M; m += (int64_t)gridDim.x * kRowWaves) {
    const float* br = bank + m * D;
    float acc[kSmallQ], bsq = 0.f;
#pragma unroll
    for (int j = 0; j < kSmallQ; ++j) acc[j] = 0.f;
    int64_t i = lane;
    for (; i + 448 < D; i += 512) {
      float ba[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ba[u] = br[i + 64 * u];
#pragma unroll
      for (int j = 0; j < kSmallQ; ++j) {
        if (j < Q) {  // (uniform)
          const float* qr = q + (int64_t)j * D;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float df = qr[i + 64 * u] - ba[u];
            acc[j] = fmaf(df, df, acc[j]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) bsq = fmaf(ba[u], ba[u], bsq);
    }
    for (; i < D; i += 64) {
      const float b = br[i];
#pragma unroll
      for (int j = 0; j < kSmallQ; ++j) {
        if (j < Q) {
          const float df = q[(int64_t)j * D + i] - b;
          acc[j] = fmaf(df, df, acc[j]);
        }
      }
      bsq = fmaf(b, b, bsq);
    }
#pragma unroll
    for (int j = 0; j < kSmallQ; ++j) {
      if (j < Q) {
        const float d = wave_sum_f32(acc[j]);
        // as the matrix-core kernels' epilogues: a NaN or infinite distance counts as faiss's FLT_MAX fill
        if (lane == 0) dist[(int64_t)j * M + m] = (d == d) ? fminf(d, kFltMax) : kFltMax;
      }
    }
    bsq = wave_sum_f32(bsq);
    if (lane == 0 && bsq < INFINITY) {  // NaN / inf rows do not set the range (as row_sqnorm_kernel)
      const unsigned b = __float_as_uint(bsq);
      if (b > __atomic_load_n(bn_max_bits, __ATOMIC_RELAXED)) atomicMax(bn_max_bits, b);
    }
  }
}

// `pm`: column m of a distance row is bank row knn_perm_row(m) (the bf16 kernel's piece order; identity otherwise).
// `row_map` / `n_rows` / `row_first` (the dense fallback of the candidate filter): distance row r belongs to query
// row_map[row_first + r], and only min(Q, *n_rows - row_first) rows exist.
__global__ __launch_bounds__(256) void kth_select_range_kernel(float* __restrict__ dist, const float* __restrict__ q,
                                                                const float* __restrict__ bank,
                                                                const float* __restrict__ qn,
                                                                const unsigned* __restrict__ bn_max_bits,
                                                                float* __restrict__ score, int64_t Q, int64_t M,
                                                                int64_t D, int k, float refine_rel, KnnPerm pm,
                                                                const int* __restrict__ row_map,
                                                                const int* __restrict__ n_rows, int row_first) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned part[256];
  __shared__ unsigned sel[2];
  __shared__ unsigned n_below, n_cand, n_fast;
  __shared__ int cand_idx[kMaxCand];
  __shared__ float cand_d[kMaxCand];
  __shared__ float fast_d[kFastCap];  // first the strided sample (kSample <= kFastCap), then the candidates <= tau
  __shared__ int fast_i[kFastCap];
  static_assert(kSample <= kFastCap, "the sample lives in the candidate buffer");
  float* samp = fast_d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float bmax = __uint_as_float(*bn_max_bits);
  const bool fast_ok = (M >= 4 * (int64_t)kSample) && (k <= kSample / 4);
  if (n_rows) {
    const int64_t have = (int64_t)*n_rows - row_first;
    if (have < Q) Q = have;
  }
  for (int64_t drow_i = blockIdx.x; drow_i < Q; drow_i += gridDim.x) {
    float* drow = dist + drow_i * M;
    const int64_t row = row_map ? (int64_t)row_map[row_first + drow_i] : drow_i;
    const float qnorm = qn[row];
    if (!(qnorm < INFINITY)) {  // NaN or infinite query: every distance is FLT_MAX (wave-uniform branch)
      if (tid == 0) score[row] = -kFltMax;
      continue;
    }
    const float sq = sqrtf(qnorm) + sqrtf(bmax);
    const float range = sq * sq * 1.000001f + 1e-30f;
    const float scale = 16777216.0f / range;
    const float delta = fmaxf(refine_rel * range, 2.0f / scale);  // refinement half-window, never narrower than two bins
    const float4* drow4 = (((M & 3) == 0) && ((((uintptr_t)drow) & 15) == 0)) ? reinterpret_cast<const float4*>(drow) : nullptr;
    __syncthreads();
    if (tid == 0) { n_below = 0u; n_cand = 0u; n_fast = 0u; }
    bool windowed = false;
    float approx = 0.f, lo = 0.f, hi = 0.f;
    if (fast_ok) {
      // the sample: 128 runs of 16 consecutive entries spread evenly over the row (any subset gives a valid bound; a
      // stride-24 sample of single entries touched every cache line of the row, i.e. read it a second time)
      const int64_t run_stride = M / (kSample / 16);
#pragma unroll
      for (int j = 0; j < kSample / 256; ++j) {
        const int sidx = tid + 256 * j;
        samp[sidx] = drow[(int64_t)(sidx >> 4) * run_stride + (sidx & 15)];
      }
      __syncthreads();
      const unsigned key_s = hist_select24([&](int64_t m) { return samp[m]; }, kSample, (unsigned)k, scale, hist, part, sel, tid);
      if (key_s < 16777214u) {
        const float tau = ((float)key_s + 1.0f) / scale * 1.000001f;  // upper edge of the bin: >= k sample entries are <= tau
        __syncthreads();  // the sample (in fast_d) has been consumed
        auto keep = [&](float d, int64_t m) {
          if (d <= tau) {
            const unsigned slot = atomicAdd(&n_fast, 1u);
            if (slot < (unsigned)kFastCap) { fast_d[slot] = d; fast_i[slot] = (int)m; }
          }
        };
        if (drow4) {
          const int64_t n4 = M >> 2;
          int64_t m = tid;
          for (; m + 768 < n4; m += 1024) {  // four 16-byte loads in flight per lane
            const float4 a = drow4[m], b = drow4[m + 256], c = drow4[m + 512], e = drow4[m + 768];
            keep(a.x, 4 * m); keep(a.y, 4 * m + 1); keep(a.z, 4 * m + 2); keep(a.w, 4 * m + 3);
            keep(b.x, 4 * (m + 256)); keep(b.y, 4 * (m + 256) + 1); keep(b.z, 4 * (m + 256) + 2); keep(b.w, 4 * (m + 256) + 3);
            keep(c.x, 4 * (m + 512)); keep(c.y, 4 * (m + 512) + 1); keep(c.z, 4 * (m + 512) + 2); keep(c.w, 4 * (m + 512) + 3);
            keep(e.x, 4 * (m + 768)); keep(e.y, 4 * (m + 768) + 1); keep(e.z, 4 * (m + 768) + 2); keep(e.w, 4 * (m + 768) + 3);
          }
          for (; m < n4; m += 256) {
            const float4 a = drow4[m];
            keep(a.x, 4 * m); keep(a.y, 4 * m + 1); keep(a.z, 4 * m + 2); keep(a.w, 4 * m + 3);
          }
        } else {
          for (int64_t m = tid; m < M; m += 256) keep(drow[m], m);
        }
        __syncthreads();
        const unsigned nf = n_fast;
        if (nf >= (unsigned)k && nf <= (unsigned)kFastCap) {
          const unsigned key_f = hist_select24([&](int64_t m) { return fast_d[m]; }, nf, (unsigned)k, scale, hist, part, sel, tid);
          approx = ((float)key_f + 0.5f) / scale;
          lo = approx - delta;
          hi = approx + delta;
          if (key_f < 16777215u && hi <= tau) {  // every entry of the row inside the window is among the candidates
            unsigned below = 0;
            for (unsigned c = tid; c < nf; c += 256) {
              const float d = fast_d[c];
              if (d < lo) {
                ++below;
              } else if (d <= hi) {
                const unsigned slot = atomicAdd(&n_cand, 1u);
                if (slot < (unsigned)kMaxCand) cand_idx[slot] = (int)knn_perm_row(fast_i[c], pm);
              }
            }
            atomicAdd(&n_below, below);
            windowed = true;
            if (tid == 0) score[row] = -approx;
          }
        }
      }
      __syncthreads();
    }
    if (!windowed) {  // three reads of the row
      if (tid == 0) { n_below = 0u; n_cand = 0u; }
      const unsigned key_sel = hist_select24([&](int64_t m) { return drow[m]; }, M, (unsigned)k, scale, hist, part, sel, tid, drow4);
      if (key_sel == 16777215u) {  // the k-th distance is a FLT_MAX fill (fewer than k comparable bank rows)
        if (tid == 0) score[row] = -kFltMax;
        continue;
      }
      approx = ((float)key_sel + 0.5f) / scale;
      if (tid == 0) score[row] = -approx;  // never left unwritten; replaced by the exact value below
      lo = approx - delta;
      hi = approx + delta;
      unsigned below = 0;
      for (int64_t m = tid; m < M; m += 256) {
        const float d = drow[m];
        if (d < lo) {
          ++below;
        } else if (d <= hi) {
          const unsigned slot = atomicAdd(&n_cand, 1u);
          if (slot < (unsigned)kMaxCand) cand_idx[slot] = (int)knn_perm_row(m, pm);
        }
      }
      atomicAdd(&n_below, below);
    }
    __syncthreads();
    const unsigned nc = n_cand;
    const int want = k - (int)n_below;  // 1-based rank inside the window; 1 <= want <= nc by construction
    if (want < 1 || (unsigned)want > nc) continue;  // (cannot happen; the approximate value stays)
    const float* qr = q + row * D;
    if (nc > (unsigned)kMaxCand) {
      // exact slow path: re-measure every candidate in place (sign bit = "exact"), then radix-select among them
      for (int64_t m = wave; m < M; m += 4) {
        const float d = drow[m];  // wave-uniform
        if (d >= lo && d <= hi) {
          const float acc = exact_sqdist_wave(qr, bank + knn_perm_row(m, pm) * D, D, lane);
          if (lane == 0) drow[m] = __uint_as_float(__float_as_uint(acc) | 0x80000000u);
        }
      }
      __threadfence_block();
      __syncthreads();
      unsigned prefix = 0u, r = (unsigned)want;
      for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int64_t m = tid; m < M; m += 256) {
          const unsigned b = __float_as_uint(drow[m]);
          if ((b & 0x80000000u) && (((b & 0x7FFFFFFFu) & himask) == prefix))
            atomicAdd(&hist[((b & 0x7FFFFFFFu) >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned d8 = 0;
          for (; d8 < 255; ++d8) {
            if (r <= hist[d8]) break;
            r -= hist[d8];
          }
          sel[0] = d8;
          sel[1] = r;
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        r = sel[1];
        __syncthreads();
      }
      if (tid == 0) score[row] = -__uint_as_float(prefix);
      continue;
    }
    for (unsigned c = wave; c < nc; c += 4) {
      const float acc = exact_sqdist_wave(qr, bank + (int64_t)cand_idx[c] * D, D, lane);
      if (lane == 0) cand_d[c] = acc;
    }
    __syncthreads();
    for (unsigned c = tid; c < nc; c += 256) {
      const float dc = cand_d[c];
      int less = 0, leq = 0;
      for (unsigned o = 0; o < nc; ++o) {
        less += (cand_d[o] < dc);
        leq += (cand_d[o] <= dc);
      }
      if (less < want && want <= leq) score[row] = -dc;  // all writers hold the same value
    }
  }
}

// ---- candidate filter (bf16 kernel, FILTER epilogue): the selection without the Q x M matrix -------------------------
// (1) knn_tau_kernel: the k-th smallest of a row's distances to the SAMPLE (the first S piece rows of the bank, dense
//     [Q, S]) is an upper bound tau of the row's k-th smallest distance; thr = tau + 2 windows is what the main pass lets
//     through.  (2) knn_dist_bf16_kernel<true> over the other piece rows.  (3) kth_select_lists_kernel: the row's list +
//     its sample entries <= thr hold EVERY bank row at or below thr, so the k-th smallest key among them is the row's
//     k-th smallest, the window around it is complete, and the rows below the window are counted - the same selection,
//     window and exact re-measurement as kth_select_range_kernel's fast path, i.e. the same bits.  A row where that
//     cannot be done (more hits than a list holds, fewer than k comparable sample entries, a crowded window) is put on
//     the overflow list and takes the dense kernels afterwards (knn_gather_rows_kernel + the dense epilogue +
//     kth_select_range_kernel over however many rows overflowed).
constexpr int kListCap = 2048;  // entries a row's list holds (expected: k * M / S ~ 1 250 +- 180 at the S chosen below)
static_assert(kListCap <= kFastCap, "a full list fits the selection's LDS buffer");

__global__ __launch_bounds__(256) void knn_tau_kernel(const float* __restrict__ samp, const float* __restrict__ qn,
                                                       const unsigned* __restrict__ bn_max_bits, float* __restrict__ thr,
                                                       unsigned* __restrict__ counts, int* __restrict__ n_ov, int64_t Q,
                                                       int S, int k, float refine_rel) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned part[256];
  __shared__ unsigned sel[2];
  const int tid = threadIdx.x;
  const float bmax = __uint_as_float(*bn_max_bits);
  if (blockIdx.x == 0 && tid == 0) *n_ov = 0;
  for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
    const float qnorm = qn[row];
    if (tid == 0) counts[row] = 0u;
    if (!(qnorm < INFINITY)) {  // (uniform) nothing passes; the selection writes -FLT_MAX
      if (tid == 0) thr[row] = -INFINITY;
      continue;
    }
    const float sq = sqrtf(qnorm) + sqrtf(bmax);
    const float range = sq * sq * 1.000001f + 1e-30f;
    const float scale = 16777216.0f / range;
    const float delta = fmaxf(refine_rel * range, 2.0f / scale);
    const float* srow = samp + row * (int64_t)S;
    const unsigned key_s = hist_select24([&](int64_t m) { return srow[m]; }, S, (unsigned)k, scale, hist, part, sel, tid,
                                         reinterpret_cast<const float4*>(srow));
    // fewer than k comparable sample rows: everything passes, the row overflows and takes the dense kernels
    if (tid == 0) thr[row] = (key_s < 16777214u) ? ((float)key_s + 1.0f) / scale * 1.000001f + 2.0f * delta : kFltMax;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void kth_select_lists_kernel(const float* __restrict__ samp, int S,
                                                                const uint2* __restrict__ lists,
                                                                const unsigned* __restrict__ counts,
                                                                const float* __restrict__ thr, const float* __restrict__ q,
                                                                const float* __restrict__ bank, const float* __restrict__ qn,
                                                                const unsigned* __restrict__ bn_max_bits,
                                                                float* __restrict__ score, int* __restrict__ ov_list,
                                                                int* __restrict__ n_ov, int64_t Q, int64_t D, int k,
                                                                float refine_rel, KnnPerm pm) {
  __shared__ unsigned hist[4096];
  __shared__ unsigned part[256];
  __shared__ unsigned sel[2];
  __shared__ unsigned n_below, n_cand, n_fast;
  __shared__ int cand_idx[kMaxCand];
  __shared__ float cand_d[kMaxCand];
  __shared__ float fast_d[kFastCap];
  __shared__ int fast_i[kFastCap];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float bmax = __uint_as_float(*bn_max_bits);
  for (int64_t row = blockIdx.x; row < Q; row += gridDim.x) {
    const float qnorm = qn[row];
    if (!(qnorm < INFINITY)) {
      if (tid == 0) score[row] = -kFltMax;
      continue;
    }
    const float sq = sqrtf(qnorm) + sqrtf(bmax);
    const float range = sq * sq * 1.000001f + 1e-30f;
    const float scale = 16777216.0f / range;
    const float delta = fmaxf(refine_rel * range, 2.0f / scale);
    __syncthreads();
    if (tid == 0) { n_below = 0u; n_cand = 0u; n_fast = 0u; }
    __syncthreads();
    const float t = thr[row];
    const unsigned cnt = counts[row];
    bool dense = cnt > (unsigned)kListCap || !(t < kFltMax);  // (uniform)
    unsigned nf = 0;
    if (!dense) {
      const uint2* lrow = lists + row * (int64_t)kListCap;
      for (unsigned c = tid; c < cnt; c += 256) {
        const uint2 e = lrow[c];
        fast_d[c] = __uint_as_float(e.x);
        fast_i[c] = (int)e.y;
      }
      const float4* srow4 = reinterpret_cast<const float4*>(samp + row * (int64_t)S);
      auto keep = [&](float d, int m) {
        if (d <= t) {
          const unsigned slot = cnt + atomicAdd(&n_fast, 1u);
          if (slot < (unsigned)kFastCap) { fast_d[slot] = d; fast_i[slot] = m; }
        }
      };
      for (int m = tid; m < S / 4; m += 256) {
        const float4 a = srow4[m];
        keep(a.x, 4 * m); keep(a.y, 4 * m + 1); keep(a.z, 4 * m + 2); keep(a.w, 4 * m + 3);
      }
      __syncthreads();
      nf = cnt + n_fast;
      dense = nf > (unsigned)kFastCap || nf < (unsigned)k;
    }
    float lo = 0.f, hi = 0.f;
    if (!dense) {
      const unsigned key_f = hist_select24([&](int64_t m) { return fast_d[m]; }, nf, (unsigned)k, scale, hist, part, sel, tid);
      const float approx = ((float)key_f + 0.5f) / scale;
      lo = approx - delta;
      hi = approx + delta;
      if (key_f < 16777215u && hi <= t) {  // (hi <= t by construction: the k-th key is at most the sample's)
        unsigned below = 0;
        for (unsigned c = tid; c < nf; c += 256) {
          const float d = fast_d[c];
          if (d < lo) {
            ++below;
          } else if (d <= hi) {
            const unsigned slot = atomicAdd(&n_cand, 1u);
            if (slot < (unsigned)kMaxCand) cand_idx[slot] = (int)knn_perm_row(fast_i[c], pm);
          }
        }
        atomicAdd(&n_below, below);
        __syncthreads();
        const int want = k - (int)n_below;
        dense = n_cand > (unsigned)kMaxCand || want < 1 || (unsigned)want > n_cand;
      } else {
        dense = true;
      }
    }
    if (dense) {
      if (tid == 0) ov_list[atomicAdd(n_ov, 1)] = (int)row;
      continue;
    }
    const unsigned nc = n_cand;
    const int want = k - (int)n_below;
    const float* qr = q + row * D;
    for (unsigned c = wave; c < nc; c += 4) {
      const float acc = exact_sqdist_wave(qr, bank + (int64_t)cand_idx[c] * D, D, lane);
      if (lane == 0) cand_d[c] = acc;
    }
    __syncthreads();
    for (unsigned c = tid; c < nc; c += 256) {
      const float dc = cand_d[c];
      int less = 0, leq = 0;
      for (unsigned o = 0; o < nc; ++o) {
        less += (cand_d[o] < dc);
        leq += (cand_d[o] <= dc);
      }
      if (less < want && want <= leq) score[row] = -dc;  // all writers hold the same value
    }
  }
}

// the overflowed rows' pieces and norms, packed for the dense kernel: row r of the output = query ov_list[first + r]
__global__ __launch_bounds__(256) void knn_gather_rows_kernel(const uint4* __restrict__ planes, const float* __restrict__ qn,
                                                               const int* __restrict__ ov_list, const int* __restrict__ n_ov,
                                                               int first, int max_rows, uint4* __restrict__ planes_out,
                                                               float* __restrict__ qn_out, int64_t row_u4) {
  int have = *n_ov - first;
  if (have > max_rows) have = max_rows;  // (this round's share)
  for (int r = blockIdx.x; r < have; r += gridDim.x) {
    const int64_t src = ov_list[first + r];
    for (int64_t i = threadIdx.x; i < row_u4; i += 256) planes_out[(int64_t)r * row_u4 + i] = planes[src * row_u4 + i];
    if (threadIdx.x == 0) qn_out[r] = qn[src];
  }
}

__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// Gaussian KDE log-density, f64: online logsumexp over the training rows, one workgroup per query.
__global__ __launch_bounds__(256) void kde_kernel(const double* __restrict__ train, const double* __restrict__ x,
                                                   double* __restrict__ score, int64_t M, int64_t N, int64_t D,
                                                   double neg_half_inv_h2, double log_norm) {
  extern __shared__ double xs[];  // D doubles
  __shared__ double wm[4], wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    __syncthreads();
    for (int64_t i = tid; i < D; i += 256) xs[i] = x[row * D + i];
    __syncthreads();
    double mx = -kInfD(), s = 0.0;  // running max / sum of exp(. - mx), identical on all lanes of a wave
    for (int64_t m = wave; m < M; m += 4) {
      const double* t = train + m * D;
      double acc = 0.0;
      for (int64_t i = lane; i < D; i += 64) {
        const double d = xs[i] - t[i];
        acc += d * d;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
      const double v = acc * neg_half_inv_h2;
      if (v > mx) {
        s = s * exp(mx - v) + 1.0;
        mx = v;
      } else {
        s += exp(v - mx);
      }
    }
    if (lane == 0) { wm[wave] = mx; wsum[wave] = s; }
    __syncthreads();
    if (tid == 0) {
      double gm = fmax(fmax(wm[0], wm[1]), fmax(wm[2], wm[3]));
      double gs = 0.0;
      for (int w = 0; w < 4; ++w)
        if (wsum[w] > 0.0) gs += wsum[w] * exp(wm[w] - gm);
      score[row] = log(gs) + gm + log_norm;
    }
  }
}

// The other kernels sklearn's KernelDensity offers (DetectorKDE(kernel=...) forwards any of them, reference
// inference/postprocessors.py:78-128): tophat, epanechnikov, exponential, linear, cosine - values in [0, 1], so the
// density is a plain f64 sum over the training rows (one wave per training row, fixed order), then one log.
// KIND: 1 tophat [d < h], 2 epanechnikov 1 - d^2 / h^2, 3 exponential exp(-d / h), 4 linear 1 - d / h, 5 cosine
// cos(pi d / 2 h); compact kernels are 0 from d >= h on (sklearn's strict d < h).  No training row in range: log(0) = -inf.
template <int KIND>
__global__ __launch_bounds__(256) void kde_other_kernel(const double* __restrict__ train, const double* __restrict__ x,
                                                         double* __restrict__ score, int64_t M, int64_t N, int64_t D, double h,
                                                         double log_norm) {
  extern __shared__ double xs[];  // D doubles
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    __syncthreads();
    for (int64_t i = tid; i < D; i += 256) xs[i] = x[row * D + i];
    __syncthreads();
    double s = 0.0;
    for (int64_t m = wave; m < M; m += 4) {
      const double* t = train + m * D;
      double acc = 0.0;
      for (int64_t i = lane; i < D; i += 64) {
        const double d = xs[i] - t[i];
        acc += d * d;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
      const double dist = sqrt(acc);
      double kv;
      if (KIND == 1) kv = dist < h ? 1.0 : 0.0;
      else if (KIND == 2) kv = dist < h ? 1.0 - (dist * dist) / (h * h) : 0.0;
      else if (KIND == 3) kv = exp(-dist / h);
      else if (KIND == 4) kv = dist < h ? 1.0 - dist / h : 0.0;
      else kv = dist < h ? cos(0.5 * M_PI * dist / h) : 0.0;
      s += kv;
    }
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) score[row] = log((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + log_norm;
  }
}

// Gaussian KDE for D <= 64 (the regime where the reference's tree evaluation is converged): one thread
// owns one query with its D coordinates in registers; training rows are staged through LDS and read as broadcasts;
// the four waves of a workgroup take a quarter of every staged tile each and merge their (max, sum) pairs at the end.
// Per (query, train row): 2*D f64 ops + one f64 exp; logsumexp is kept online per group of 8 rows.
// Any D <= DP: training rows staged through LDS and read as broadcasts; NW waves share 64 queries and split every
// staged tile.  NW = 16 (8 at DP = 64: register budget) when the batch has too few 64-query groups to fill the chip.
template <int DP, int NW>
__global__ __launch_bounds__(64 * NW) void kde_small_kernel(const double* __restrict__ train,
                                                             const double* __restrict__ x,
                                                             double* __restrict__ score, int64_t M, int64_t N, int D,
                                                             double neg_half_inv_h2, double log_norm) {
  constexpr int TM = (NW == 4) ? 64 : ((DP <= 32) ? 128 : 64);  // staged training rows (<= 32 KB of LDS)
  constexpr int RPW = TM / NW;                                   // rows per wave and tile
  constexpr int GS = (RPW < 8) ? RPW : 8;                        // rows per online-logsumexp group
  __shared__ double tile[TM][DP];
  __shared__ double pm[NW][64], ps[NW][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t qrow = (int64_t)blockIdx.x * 64 + lane;
  double xq[DP];
#pragma unroll
  for (int i = 0; i < DP; ++i) xq[i] = (qrow < N && i < D) ? x[qrow * D + i] : 0.0;
  double mx = -kInfD(), sum = 0.0;
  for (int64_t t0 = 0; t0 < M; t0 += TM) {
    __syncthreads();
    for (int i = tid; i < TM * DP; i += 64 * NW) {
      const int r = i / DP, c = i - r * DP;
      tile[r][c] = (t0 + r < M && c < D) ? train[(t0 + r) * D + c] : 0.0;
    }
    __syncthreads();
    const int rows = (int)((M - t0 < TM) ? (M - t0) : TM);
#pragma unroll
    for (int g = 0; g < RPW / GS; ++g) {  // this wave's share of the tile, GS rows at a time
      const int r0 = wave * RPW + g * GS;
      double v[GS];
      double gmax = -kInfD();
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          const double d = xq[i] - tile[r0 + j][i];
          acc = fma(d, d, acc);
        }
        v[j] = (r0 + j < rows) ? acc * neg_half_inv_h2 : -kInfD();
        gmax = fmax(gmax, v[j]);
      }
      if (gmax > -kInfD()) {
        const double mnew = fmax(mx, gmax);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < GS; ++j) part += exp(v[j] - mnew);
        sum = sum * exp(mx - mnew) + part;
        mx = mnew;
      }
    }
  }
  pm[wave][lane] = mx;
  ps[wave][lane] = sum;
  __syncthreads();
  if (wave == 0 && qrow < N) {
    double gm = pm[0][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) gm = fmax(gm, pm[w][lane]);
    double gs = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (ps[w][lane] > 0.0) gs += ps[w][lane] * exp(pm[w][lane] - gm);
    score[qrow] = log(gs) + gm + log_norm;
  }
}

template <int DP>
void launch_kde_small(const double* train, const double* x, double* score, int64_t M, int64_t N, int D, double nh,
                      double log_norm, hipStream_t s) {
  const unsigned qblocks = (unsigned)((N + 63) / 64);
  const bool few = (int64_t)qblocks < 2 * runia_cu_count();
  // (training rows through the scalar cache instead of LDS - s_load_dwordx16, SGPR operands - measured 0.355 / 3.81 /
  //  2.60 ms against 0.365 / 2.90 / 1.84 ms at D = 16 / 32 / 64: not kept)
  if (few) {  // 16 waves leave 128 VGPRs per lane: enough for D <= 32, not for a 64-wide query -> 8 waves there
    if constexpr (DP <= 32) kde_small_kernel<DP, 16><<<qblocks, 1024, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
    else kde_small_kernel<DP, 8><<<qblocks, 512, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
  } else {
    kde_small_kernel<DP, 4><<<qblocks, 256, 0, s>>>(train, x, score, M, N, D, nh, log_norm);
  }
}

constexpr int64_t kQueryChunk = 8192;  // query rows per distance-workspace pass

}  // namespace

namespace {
// ---- final linear layer with a small head (C <= 16: CIFAR-10-sized ReAct / DICE / ASH / ViM logits) ---------------------
// The 128 x 128 matrix-core tile spends 118 of its 128 columns on padding there (1 M x 512 -> 10: 4.3 ms, 0.5 TB/s of
// rows).  Here a wave takes one row at a time: lane = four adjacent features per 256-feature stripe (16-byte loads), the
// head's weights come from LDS, the C partial dot products of a row are summed over the wave by the halving exchange of
// the joint-entropy kernel (v_permlane32_swap / v_permlane16_swap / DPP: ~40 instructions for up to 16 sums) and lane
// quad c writes logit c.  Row-streaming: 1 M x 512 -> 10 in 0.71 ms (2.9 TB/s of rows).
constexpr int kSkinnyMaxC = 16;
constexpr int kSkinnyMaxFloats = 24576;  // C * D floats of weights in LDS (96 KB)
#ifndef SKINNY_RPW
#define SKINNY_RPW 16
#endif
constexpr int kSkinnyRowsPerWave = SKINNY_RPW;   // consecutive rows a wave works through (amortises the weight staging)

template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANKS, false));
}
// 16 per-lane partial sums -> their totals over the wave; lane L ends with slot (L >> 2) & 15
__device__ __forceinline__ float wave_sum16(float (&v)[16], int lane) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 8]), false, false);
    v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[j]), __float_as_uint(v[j + 4]), false, false);
    v[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
  const bool u8 = (lane & 8) != 0, u4 = (lane & 4) != 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {  // lanes i, i ^ 8
    const float send = u8 ? v[j] : v[j + 2], keep = u8 ? v[j + 2] : v[j];
    v[j] = keep + dpp_f32<0x128>(send, send);  // row_ror:8
  }
  {  // lanes i, i ^ 4
    const float send = u4 ? v[0] : v[1], keep = u4 ? v[1] : v[0];
    float recv = dpp_f32<0x124, 0xA>(send, send);  // row_ror:4 into banks 1, 3
    recv = dpp_f32<0x12C, 0x5>(recv, send);        // row_ror:12 into banks 0, 2
    v[0] = keep + recv;
  }
  float r = v[0];
  r += dpp_f32<0x4E>(r, r);  // quad_perm:[2,3,0,1]
  r += dpp_f32<0xB1>(r, r);  // quad_perm:[1,0,3,2]
  return r;
}

template <int CT>  // CT = classes rounded up to 4, 8, 12 or 16 (accumulator registers)
__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int64_t N, int D, int C, float clip_max) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [C][D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C * D / 4; i += 256) reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
  const int n4 = D >> 2;                   // float4 per row
  const int stripes = (n4 + 63) / 64;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * kSkinnyRowsPerWave;
  for (int rr = 0; rr < kSkinnyRowsPerWave; ++rr) {  // (next row's loads issued ahead of this row's sums: 0.71 -> 0.79 ms)
    const int64_t row = row0 + rr;
    if (row >= N) break;  // wave-uniform
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (int st = 0; st < stripes; ++st) {
      const int i4 = lane + 64 * st;
      if (i4 < n4) {
        float4 v = xr[i4];
        // x > clip ? clip : x keeps a NaN activation, as np.clip and the matmul that follows do upstream
        v.x = v.x > clip_max ? clip_max : v.x; v.y = v.y > clip_max ? clip_max : v.y;
        v.z = v.z > clip_max ? clip_max : v.z; v.w = v.w > clip_max ? clip_max : v.w;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (c < C) {
            const float4 ww = reinterpret_cast<const float4*>(wl + c * D)[i4];
            acc[c] = fmaf(v.x, ww.x, acc[c]);
            acc[c] = fmaf(v.y, ww.y, acc[c]);
            acc[c] = fmaf(v.z, ww.z, acc[c]);
            acc[c] = fmaf(v.w, ww.w, acc[c]);
          }
        }
      }
    }
    const float total = wave_sum16(acc, lane);
    const int c = (lane >> 2) & 15;
    if ((lane & 3) == 0 && c < C) out[row * C + c] = total + (bias ? bias[c] : 0.f);
  }
}
}  // namespace

// Final linear layer for a handful of rows (serving one image at a time: 1 row x 2048 -> 1000 took 0.22 ms on the
// 128 x 128 tiles of the matrix-core kernel, 127 rows of every tile padding): one thread per (row, class) walks the
// class's weights with ONE f32 fma chain in the k order of the matrix-core kernel - its v_mfma_f32_32x32x2_f32 pairs
// multiply k = 4s, 4s + 2 and then 4s + 1, 4s + 3 of every four, each instruction an exact fma chain - so a row gets the
// same bits alone as inside a batch (tests).
namespace {
constexpr int kLinearFewRows = 8;
// (a workgroup of 64 classes x 4 rows with the weights staged 32 k at a time through LDS measured slower - 115 us against
// 89 for one row x 2048 -> 1000: 16 workgroups, two barriers and one exposed load latency per chunk)
__global__ __launch_bounds__(256) void linear_few_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int64_t N, int64_t D, int64_t C, float clip_max) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * C) return;
  const int64_t row = idx / C, c = idx - row * C;
  const float* xr = x + row * D;
  const float* wr = w + c * D;
  float acc = 0.f;
  int64_t k0 = 0;
  if (((D & 3) == 0) && ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0) {
    // 16-byte loads, four groups of four in flight
    const float4* x4 = reinterpret_cast<const float4*>(xr);
    const float4* w4 = reinterpret_cast<const float4*>(wr);
    const int64_t n4 = D >> 2;
    int64_t g4 = 0;
    for (; g4 + 4 <= n4; g4 += 4) {
      float4 xa[4], wa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { xa[u] = x4[g4 + u]; wa[u] = w4[g4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a0 = (xa[u].x > clip_max) ? clip_max : xa[u].x, a1 = (xa[u].y > clip_max) ? clip_max : xa[u].y;
        const float a2 = (xa[u].z > clip_max) ? clip_max : xa[u].z, a3 = (xa[u].w > clip_max) ? clip_max : xa[u].w;
        acc = fmaf(a0, wa[u].x, acc);
        acc = fmaf(a2, wa[u].z, acc);
        acc = fmaf(a1, wa[u].y, acc);
        acc = fmaf(a3, wa[u].w, acc);
      }
    }
    k0 = g4 * 4;
  }
  for (; k0 + 4 <= D; k0 += 4) {
    float xv[4], wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float v = xr[k0 + j];
      xv[j] = (v > clip_max) ? clip_max : v;  // np.clip keeps a NaN activation
      wv[j] = wr[k0 + j];
    }
    acc = fmaf(xv[0], wv[0], acc);
    acc = fmaf(xv[2], wv[2], acc);
    acc = fmaf(xv[1], wv[1], acc);
    acc = fmaf(xv[3], wv[3], acc);
  }
  if (k0 < D) {  // the last, partial group of four: the missing k are zeros in the matrix-core kernel's staging
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, wv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; k0 + j < D; ++j) {
      const float v = xr[k0 + j];
      xv[j] = (v > clip_max) ? clip_max : v;
      wv[j] = wr[k0 + j];
    }
    acc = fmaf(xv[0], wv[0], acc);
    acc = fmaf(xv[2], wv[2], acc);
    acc = fmaf(xv[1], wv[1], acc);
    acc = fmaf(xv[3], wv[3], acc);
  }
  if (D % KCH) acc = fmaf(0.f, 0.f, acc);  // the zero padding of the last 32-chunk (only turns a -0 into +0)
  out[idx] = acc + (bias ? bias[c] : 0.f);
}

// Some tens to hundreds of rows (9 ... 512: a batch of a service, the proposals of an image): the matrix-core kernel has one
// 128-row tile against C / 128 column tiles - 8 workgroups for 1000 classes, each walking all of K alone (64 ... 512 rows x
// 2048 -> 1000: 203-226 us).  Here a workgroup takes 64 rows x 16 classes (63 workgroups per 64 rows at 1000 classes): rows
// and weights staged 64 k at a time through LDS (two buffers, the next chunk's loads in flight during the arithmetic), a
// thread = one row x four classes, and every (row, class) is again ONE f32 fma chain in the matrix-core kernel's k order -
// same bits as inside a large batch.  (A wave per 64 classes x 8 rows with the weights streamed per lane - no LDS - ran
// 249-270 us: one wave per compute unit and a latency chain of 512 load groups.)
constexpr int kLinearMidRows = 512;
constexpr int kMidRows = 64, kMidCls = 16, kMidK = 64, kMidPitch = kMidK + 4;  // pitch 68 floats: rows 16-byte aligned, 68 mod 32 = 4
__global__ __launch_bounds__(256) void linear_mid_rows_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int64_t N, int64_t D, int64_t C, float clip_max) {
  __shared__ __attribute__((aligned(16))) float xs[2][kMidRows][kMidPitch];
  __shared__ __attribute__((aligned(16))) float ws[2][kMidCls][kMidPitch];
  const int tid = threadIdx.x;
  const int row_l = tid & 63, cq = tid >> 6;  // thread: row row_l, classes 4 cq .. 4 cq + 3 of the tile (a wave shares its classes)
  const int64_t r0 = (int64_t)blockIdx.y * kMidRows, c0 = (int64_t)blockIdx.x * kMidCls;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0;
  // staging: 64 rows x 64 k = 1024 float4 (4 per thread), 16 classes x 64 k = 256 float4 (1 per thread); zeros beyond D / N / C
  auto fetch = [&](int64_t k0, float4 (&xr)[4], float4& wr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u, rr = idx >> 4, kq = (idx & 15) * 4;
      const int64_t row = r0 + rr, k = k0 + kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N) {
        const float* p = x + row * D + k;
        if (vec && k + 4 <= D) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k < D) v.x = p[0];
          if (k + 1 < D) v.y = p[1];
          if (k + 2 < D) v.z = p[2];
          if (k + 3 < D) v.w = p[3];
        }
      }
      xr[u] = v;
    }
    {
      const int cc = tid >> 4, kq = (tid & 15) * 4;
      const int64_t cls = c0 + cc, k = k0 + kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cls < C) {
        const float* p = w + cls * D + k;
        if (vec && k + 4 <= D) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k < D) v.x = p[0];
          if (k + 1 < D) v.y = p[1];
          if (k + 2 < D) v.z = p[2];
          if (k + 3 < D) v.w = p[3];
        }
      }
      wr = v;
    }
  };
  auto clipf = [&](float v) { return (v > clip_max) ? clip_max : v; };  // np.clip keeps a NaN activation
  auto stash = [&](int buf, const float4 (&xr)[4], const float4& wr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u, rr = idx >> 4, kq = (idx & 15) * 4;
      *reinterpret_cast<float4*>(&xs[buf][rr][kq]) = make_float4(clipf(xr[u].x), clipf(xr[u].y), clipf(xr[u].z), clipf(xr[u].w));
    }
    *reinterpret_cast<float4*>(&ws[buf][tid >> 4][(tid & 15) * 4]) = wr;
  };
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t nchunks = (D + kMidK - 1) / kMidK;
  float4 xr[4], wr;
  fetch(0, xr, wr);
  int buf = 0;
  for (int64_t ch = 0; ch < nchunks; ++ch) {
    stash(buf, xr, wr);
    __syncthreads();
    if (ch + 1 < nchunks) fetch((ch + 1) * kMidK, xr, wr);
    // groups of four k that start beyond D do not exist in the matrix-core kernel's chain (its zero padding ends at the
    // 32-chunk; see the fma(0, 0, acc) below); a partial group's missing k are zeros, as staged
    const int64_t kleft = D - ch * kMidK;
    const int groups = (int)((kleft >= kMidK) ? kMidK / 4 : (kleft + 3) / 4);
    for (int g = 0; g < groups; ++g) {
      const float4 xv = *reinterpret_cast<const float4*>(&xs[buf][row_l][4 * g]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(&ws[buf][4 * cq + j][4 * g]);  // (one address per wave: broadcast)
        acc[j] = fmaf(xv.x, wv.x, acc[j]);
        acc[j] = fmaf(xv.z, wv.z, acc[j]);
        acc[j] = fmaf(xv.y, wv.y, acc[j]);
        acc[j] = fmaf(xv.w, wv.w, acc[j]);
      }
    }
    buf ^= 1;  // (the next stash goes to the other buffer; the barrier of the next trip orders it against this trip's reads)
  }
  const int64_t row = r0 + row_l;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t cls = c0 + 4 * cq + j;
    if (row < N && cls < C) {
      float a = acc[j];
      if (D % KCH) a = fmaf(0.f, 0.f, a);  // the zero padding of the last 32-chunk (only turns a -0 into +0)
      out[row * C + cls] = a + (bias ? bias[cls] : 0.f);
    }
  }
}
}  // namespace

// 1-D grid of knn_dist_kernel for Q x M: whole super-tiles, a multiple of 8 of them
static inline unsigned knn_dist_grid(int64_t Q, int64_t M) {
  const int64_t nbt = (M + TB - 1) / TB, nqt = (Q + TQ - 1) / TQ;
  const int sq = knn_super_q(nqt);
  const int64_t st = ((nbt + kSuperB - 1) / kSuperB) * ((nqt + sq - 1) / sq);
  return (unsigned)(((st + 7) / 8) * 8 * kSuperB * sq);
}

// knn_bf16.hip: candidate distances of large problems from bf16 piece products (6/16 of the f32 kernel's matrix-pipe time)
int64_t runia_knn16_padded_rows(int64_t rows);
int64_t runia_knn16_padded_width(int64_t D);
size_t runia_knn16_plane_bytes(int64_t rows, int64_t D);
bool runia_knn16_fits(int64_t rows, int64_t D);
int runia_knn16_split(const float* x, uint16_t* planes, int64_t R, int64_t D, hipStream_t s);
int runia_knn16_split_bank(const float* x, uint16_t* planes, const float* bn, float* bn_p, int64_t R, int64_t D, hipStream_t s);
float runia_knn16_refine_rel(int64_t D);
int64_t runia_knn16_max_width();
int runia_knn16_dist(const uint16_t* qp, const uint16_t* bp, const float* qn, const float* bn, float* dist, int64_t Q,
                     int64_t M, int64_t D, int64_t q_planes_rows, int64_t b_planes_rows, const int* q_count, int q_first,
                     hipStream_t s);
int runia_knn16_filter(const uint16_t* qp, const uint16_t* bp, const float* qn, const float* bn, const float* thr,
                       unsigned* counts, void* lists, int cap, int col0, int64_t Q, int64_t M, int64_t D,
                       int64_t q_planes_rows, int64_t b_planes_rows, hipStream_t s);

#ifndef KNN_BF16
#define KNN_BF16 1
#endif
#ifndef KNN16_MIN_ROWS
#define KNN16_MIN_ROWS 512  // (256 queries x 50 000 x 2048: 0.99 ms against 0.75 on the f32 kernel; 512: 0.97 against 1.12; 1 000 x 20 000: 0.51 against 0.95)
#endif
#ifndef KNN16_FILTER
#define KNN16_FILTER 1
#endif
#ifndef KNN16_FILTER_MIN_ROWS
#define KNN16_FILTER_MIN_ROWS 1024  // queries of a call from which the candidate filter is taken (fewer: the dense form)
#endif
#ifndef KNN16_DENSE_ROWS
#define KNN16_DENSE_ROWS 8192  // rows of the dense fallback matrix of the candidate filter (one round of three launches per that many overflowed rows)
#endif
// the bank side of the bf16 kernel: some thousand rows, wide (but not wider than its window allows) features
static bool knn16_bank_ok(int64_t M, int64_t D) {
// (round 4, with the candidate filter the feature width no longer has to pay for a distance matrix: 65 536 x 50 000 at
// D = 8 / 16 / 32 / 64 / 128: 5.6 / 5.6 / 5.7 / 6.0 / 6.7 ms against 9.1 / 9.0 / 9.0 / 10.2 / 13.1 for the f32 kernel, same
// bits, tools/ablate/run_knn_low_d.py; the limit was 256 while the bf16 kernel wrote the matrix too)
#ifndef KNN16_MIN_D
#define KNN16_MIN_D 8
#endif
  return KNN_BF16 && M >= 4096 && D >= KNN16_MIN_D && D <= runia_knn16_max_width() && runia_knn16_fits(M, D) &&
         runia_knn16_fits(kQueryChunk, D) && 4 * M * 256 <= ((int64_t)1 << 31);  // (chunks of >= 256 queries: whole tiles)
}
// Worth the two split passes and the 256 x 256 tiles: a bank of some thousand rows, wide features, a batch of queries, 2^31
// multiply-adds, 512 queries (tools/ablate/run_knn_paths.py: 1 024 x 4 096 x 256 is 8 % slower on the bf16 kernel, 1 024 x 4 096 x 2048
// 1.46 x faster, 8 192 x 50 000 x 2048 2.67 x)
static bool knn16_wanted(int64_t N, int64_t M, int64_t D) {
  return knn16_bank_ok(M, D) && N >= KNN16_MIN_ROWS && N * M >= ((int64_t)1 << 31) / D;
}
int runia_knn16_terms();
extern "C" int runia_knn_piece_products(int64_t N, int64_t M, int64_t D) { return knn16_wanted(N, M, D) ? runia_knn16_terms() : 0; }
static size_t knn_f32_words(int64_t qc, int64_t M) { return (size_t)(qc * M + qc + M + 4); }  // distances, |q|^2, |b|^2, max |b|^2

// query rows per pass: up to 8 192, fewer for very large banks so that the distance tile stays near 2 GiB (a 1 M-row bank
// would otherwise ask for 32 GB)
static int64_t knn_chunk_rows(int64_t N, int64_t M) {
  int64_t qc = N < kQueryChunk ? N : kQueryChunk;
  const int64_t by_size = ((int64_t)1 << 31) / (4 * M);
  if (qc > by_size) qc = by_size < 256 ? (N < 256 ? N : 256) : by_size;
  return qc;
}
static size_t knn16_head_bytes(int64_t qc, int64_t M) { return (knn_f32_words(qc, M) * sizeof(float) + 255) / 256 * 256; }

// Sample rows of the candidate filter (0 = the dense form): the expected number of bank rows at or below a row's
// threshold is k * M / S, with a spread of ~ sqrt(k) * M / S; S is the multiple of 256 that puts the mean at 0.6 of a
// list (k = 50, M = 50 000: S = 2 048, 1 220 +- 170 entries of 2 048).  Not worth it when the sample would be a
// quarter of the bank or more (large k), or for a call of a few hundred queries.
static int64_t knn16_sample_rows(int64_t N, int64_t M, int k) {
  if (!KNN16_FILTER || N < KNN16_FILTER_MIN_ROWS) return 0;
  int64_t S = ((int64_t)k * M * 10 / (6 * kListCap) + 255) / 256 * 256;
  if (S < 1024) S = 1024;
  if (S < 4 * (int64_t)k) S = (4 * (int64_t)k + 255) / 256 * 256;
  return (4 * S <= M) ? S : 0;
}

static size_t align256(size_t b) { return (b + 255) / 256 * 256; }
// Workspace of one call of the bf16 kernel after the head (dense rows, |q|^2, |b|^2, max |b|^2) and outside the pieces:
//   bn_p [Mpad] (not for a prepared bank: the state holds it) | samp [qc, S] | thr [qc] | counts [qc] | ov_list [qc] |
//   n_ov [64] | lists [qc, kListCap] x 8 B | pieces of the overflowed rows [dense rows, padded] | their |q|^2 | |q|^2 [qc]
struct Knn16Filter {
  size_t bn_p, samp, thr, counts, ov_list, n_ov, lists, ov_planes, ov_qn, qn, end;
};
static Knn16Filter knn16_filter_layout(size_t at, int64_t qc, int64_t dense_rows, int64_t M, int64_t D, int64_t S, bool own_bn_p) {
  Knn16Filter f;
  const size_t on = S > 0 ? 1 : 0;  // (the dense form keeps only bn_p and |q|^2)
  f.bn_p = at;       at += own_bn_p ? align256((size_t)runia_knn16_padded_rows(M) * 4) : 0;
  f.samp = at;       at += align256((size_t)(qc * S) * 4) * on;
  f.thr = at;        at += align256((size_t)qc * 4) * on;
  f.counts = at;     at += align256((size_t)qc * 4) * on;
  f.ov_list = at;    at += align256((size_t)qc * 4) * on;
  f.n_ov = at;       at += 256 * on;
  f.lists = at;      at += align256((size_t)qc * kListCap * 8) * on;
  f.ov_planes = at;  at += S ? runia_knn16_plane_bytes(dense_rows, D) : 0;
  f.ov_qn = at;      at += align256((size_t)runia_knn16_padded_rows(dense_rows) * 4) * on;
  f.qn = at;         at += align256((size_t)qc * 4);  // |q|^2 of the chunk
  f.end = at;
  return f;
}
// Query rows per pass of the bf16 kernel.  Dense form: the f32 kernel's chunk.  With the candidate filter no Q x M matrix
// limits it, and the main pass - (query tiles) x (bank tiles behind the sample) workgroups, one per compute unit at a
// time - should END on a full round: among 32 .. 64 query tiles (8 192 .. 16 384 rows) the count whose last round of
// workgroups is fullest is taken (50 000 bank rows, 2 048 sample rows: 188 bank tiles x 64 = 47.0 rounds of 256; x 32 =
// 23.5, i.e. 2 % of the pass spent on a half-empty chip).
static int64_t knn16_chunk_rows(int64_t N, int64_t M, int64_t S) {
  if (S <= 0) return knn_chunk_rows(N, M);
  const int64_t nbt = (M - S + 255) / 256, cus = runia_cu_count();
  int64_t best = 32;
  double best_fill = 0.0;
  for (int64_t nqt = 32; nqt <= 64; ++nqt) {
    const int64_t wgs = nqt * nbt, rounds = (wgs + cus - 1) / cus;
    const double fill = (double)wgs / (double)(rounds * cus);
    if (fill > best_fill + 1e-9 || (fill > best_fill - 1e-9 && nqt > best)) { best_fill = fill; best = nqt; }
  }
  const int64_t qc = best * 256;
  return N < qc ? N : qc;
}
// rows of the dense matrix a bf16 call keeps: the whole chunk in the dense form; with the filter the share of one overflow
// round (three launches that end at their first instruction when nothing overflowed: 16 us per round)
static int64_t knn16_dense_rows(int64_t qc, int64_t M, int64_t S) {
  if (S <= 0) return qc;
  int64_t dr = knn_chunk_rows(qc, M);
  if (dr > KNN16_DENSE_ROWS) dr = KNN16_DENSE_ROWS;
  return dr;
}

extern "C" size_t runia_knn_workspace_bytes(int64_t N, int64_t M, int64_t D, int k) {
  if (N <= 0 || M <= 0) return 0;
  // + the bf16 pieces of the bank and of one chunk of queries when the bf16 kernel will be taken; the entry point works
  // with whatever it is given (>= 1 row of distances), but takes the bf16 kernel only with at least this much
  if (knn16_wanted(N, M, D)) {
    const int64_t S = knn16_sample_rows(N, M, k), qc = knn16_chunk_rows(N, M, S), dr = knn16_dense_rows(qc, M, S);
    const Knn16Filter f = knn16_filter_layout(knn16_head_bytes(dr, M), qc, dr, M, D, S, true);
    return f.end + runia_knn16_plane_bytes(M, D) + runia_knn16_plane_bytes(qc, D);
  }
  return knn_f32_words(knn_chunk_rows(N, M), M) * sizeof(float);
}

// The passes of a kNN call once the bank's |b|^2, their maximum and (use16) its bf16 pieces exist - computed by the call
// itself into its workspace, or once per bank by runia_knn_prepare_bank_f32.
struct Knn16Run {            // (use16 only)
  const uint16_t* bank_planes;
  uint16_t* q_planes;
  const float* bn_p;         // |b|^2 in piece order
  int64_t S, dense_rows;
  char* ws;                  // base of the workspace `f` is laid out in
  Knn16Filter f;
};
static int knn_scan(const float* q, const float* bank, float* score, float* dist, float* qn, const float* bn,
                    unsigned* bn_max, const Knn16Run* r16, int64_t qc, int64_t N, int64_t M, int64_t D, int k,
                    hipStream_t s, int64_t min_rows16 = 256) {
  const KnnPerm pm = r16 ? knn_perm_for(M) : knn_perm_identity();
  const int64_t Mpad = runia_knn16_padded_rows(M), qpad = runia_knn16_padded_rows(qc);
  for (int64_t r0 = 0; r0 < N; r0 += qc) {
    const int64_t rows = (N - r0 < qc) ? (N - r0) : qc;
    row_sqnorm_kernel<<<runia_rows_grid(rows), 64 * kRowWaves, 0, s>>>(q + r0 * D, qn, rows, D, nullptr);
    int rc = RUNIA_OK;
    const bool now16 = r16 && rows >= min_rows16;
    const unsigned sel_grid = (unsigned)(rows < 4096 ? rows : 4096);
    if (now16 && r16->S > 0) {  // candidate filter: no Q x M matrix
      const int64_t S = r16->S;
      const float rel = runia_knn16_refine_rel(D);
      float* samp = reinterpret_cast<float*>(r16->ws + r16->f.samp);
      float* thr = reinterpret_cast<float*>(r16->ws + r16->f.thr);
      unsigned* counts = reinterpret_cast<unsigned*>(r16->ws + r16->f.counts);
      int* ov_list = reinterpret_cast<int*>(r16->ws + r16->f.ov_list);
      int* n_ov = reinterpret_cast<int*>(r16->ws + r16->f.n_ov);
      void* lists = r16->ws + r16->f.lists;
      uint16_t* ov_planes = reinterpret_cast<uint16_t*>(r16->ws + r16->f.ov_planes);
      float* ov_qn = reinterpret_cast<float*>(r16->ws + r16->f.ov_qn);
      rc = runia_knn16_split(q + r0 * D, r16->q_planes, rows, D, s);
      if (rc == RUNIA_OK) rc = runia_knn16_dist(r16->q_planes, r16->bank_planes, qn, r16->bn_p, samp, rows, S, D, qpad, Mpad, nullptr, 0, s);
      if (rc != RUNIA_OK) return rc;
      knn_tau_kernel<<<sel_grid, 256, 0, s>>>(samp, qn, bn_max, thr, counts, n_ov, rows, (int)S, k, rel);
      const int64_t Dp2 = 2 * runia_knn16_padded_width(D);  // uint16 per piece row
      rc = runia_knn16_filter(r16->q_planes, r16->bank_planes + S * Dp2, qn, r16->bn_p + S, thr, counts, lists, kListCap,
                              (int)S, rows, M - S, D, qpad, Mpad - S, s);
      if (rc != RUNIA_OK) return rc;
      kth_select_lists_kernel<<<sel_grid, 256, 0, s>>>(samp, (int)S, reinterpret_cast<const uint2*>(lists), counts, thr,
                                                       q + r0 * D, bank, qn, bn_max, score + r0, ov_list, n_ov, rows, D, k, rel, pm);
      // the rows that overflowed (none, as a rule: the launches below then end at their first instruction)
      const int64_t dr = r16->dense_rows;
      for (int64_t first = 0; first < rows; first += dr) {
        const int64_t cap_rows = (rows - first < dr) ? (rows - first) : dr;
        knn_gather_rows_kernel<<<(unsigned)(cap_rows < 256 ? cap_rows : 256), 256, 0, s>>>(
            reinterpret_cast<const uint4*>(r16->q_planes), qn, ov_list, n_ov, (int)first, (int)cap_rows, reinterpret_cast<uint4*>(ov_planes),
            ov_qn, Dp2 / 8);
        rc = runia_knn16_dist(ov_planes, r16->bank_planes, ov_qn, r16->bn_p, dist, cap_rows, M, D,
                              runia_knn16_padded_rows(dr), Mpad, n_ov, (int)first, s);
        if (rc != RUNIA_OK) return rc;
        kth_select_range_kernel<<<(unsigned)(cap_rows < 1024 ? cap_rows : 1024), 256, 0, s>>>(
            dist, q + r0 * D, bank, qn, bn_max, score + r0, cap_rows, M, D, k, rel, pm, ov_list, n_ov, (int)first);
      }
      rc = runia_check_launch();
      if (rc != RUNIA_OK) return rc;
      continue;
    }
    if (now16) {
      rc = runia_knn16_split(q + r0 * D, r16->q_planes, rows, D, s);
      if (rc == RUNIA_OK) rc = runia_knn16_dist(r16->q_planes, r16->bank_planes, qn, r16->bn_p, dist, rows, M, D, qpad, Mpad, nullptr, 0, s);
      if (rc != RUNIA_OK) return rc;
    } else {
      knn_dist_kernel<EPI_DIST><<<knn_dist_grid(rows, M), 256, 0, s>>>(q + r0 * D, bank, qn, bn, dist, rows, M, D, INFINITY);
    }
    kth_select_range_kernel<<<sel_grid, 256, 0, s>>>(dist, q + r0 * D, bank, qn, bn_max, score + r0, rows, M, D, k,
                                                     now16 ? runia_knn16_refine_rel(D) : kRefineRel,
                                                     now16 ? pm : knn_perm_identity(), nullptr, nullptr, 0);
    rc = runia_check_launch();
    if (rc != RUNIA_OK) return rc;
  }
  return RUNIA_OK;
}

static int knn_small_path(const float* q, const float* bank, float* score, float* dist, float* qn, unsigned* bn_max,
                          int64_t N, int64_t M, int64_t D, int k, hipStream_t s) {
  row_sqnorm_kernel<<<runia_rows_grid(N), 64 * kRowWaves, 0, s>>>(q, qn, N, D, nullptr);
  knn_small_dist_kernel<<<runia_rows_grid(M), 64 * kRowWaves, 0, s>>>(q, bank, dist, bn_max, (int)N, M, D);
  kth_select_range_kernel<<<(unsigned)N, 256, 0, s>>>(dist, q, bank, qn, bn_max, score, N, M, D, k, kRefineRel,
                                                      knn_perm_identity(), nullptr, nullptr, 0);
  return runia_check_launch();
}

extern "C" int runia_knn_kth_f32(const float* q, const float* bank, float* score, void* workspace,
                                 size_t workspace_bytes, int64_t N, int64_t M, int64_t D, int k,
                                 runia_stream_t stream) {
  if (N < 0 || M < 0 || D <= 0 || k < 1) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!q || !score || (M > 0 && !bank)) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (k > M) {  // faiss fills missing neighbours with FLT_MAX
    fill_kernel<<<runia_stream_grid(N, 256), 256, 0, s>>>(score, N, -kFltMax);
    return runia_check_launch();
  }
  if (!workspace || workspace_bytes < (size_t)(2 * M + 5) * sizeof(float)) return RUNIA_E_WORKSPACE;
  int64_t qc = (int64_t)((workspace_bytes / sizeof(float) - (size_t)M - 4) / (size_t)(M + 1));
  if (qc < 1) return RUNIA_E_WORKSPACE;
  if (qc > N) qc = N;
  if (qc > kQueryChunk) qc = kQueryChunk;
  // bf16 candidate distances exactly when the caller hands over the workspace runia_knn_workspace_bytes asks for (a
  // smaller one - e.g. the f32 kernel's (chunk * M + chunk + M + 4) floats - keeps the f32 kernel: the caller's switch)
  const bool use16 = knn16_wanted(N, M, D) && workspace_bytes >= runia_knn_workspace_bytes(N, M, D, k);
  Knn16Run r16{};
  int64_t dense_rows = qc;
  if (use16) {
    r16.S = knn16_sample_rows(N, M, k);
    qc = knn16_chunk_rows(N, M, r16.S);
    r16.dense_rows = dense_rows = knn16_dense_rows(qc, M, r16.S);
    r16.ws = reinterpret_cast<char*>(workspace);
    r16.f = knn16_filter_layout(knn16_head_bytes(dense_rows, M), qc, dense_rows, M, D, r16.S, true);
    uint16_t* planes = reinterpret_cast<uint16_t*>(r16.ws + r16.f.end);
    r16.bank_planes = planes;
    r16.q_planes = planes + runia_knn16_plane_bytes(M, D) / sizeof(uint16_t);
    r16.bn_p = reinterpret_cast<const float*>(r16.ws + r16.f.bn_p);
  }
  float* dist = reinterpret_cast<float*>(workspace);
  float* qn = dist + dense_rows * M;  // (bf16 kernel: the head keeps `dense_rows` slots; the chunk's |q|^2 have their own block)
  float* bn = qn + dense_rows;
  unsigned* bn_max = reinterpret_cast<unsigned*>(bn + M);
  if (use16) qn = reinterpret_cast<float*>(r16.ws + r16.f.qn);
  if (hipMemsetAsync(bn_max, 0, sizeof(unsigned), s) != hipSuccess) return RUNIA_E_LAUNCH;
  if (N <= kSmallQ && M >= 1024 && qc >= N)  // a handful of queries: one pass over the bank, exact distances
    return knn_small_path(q, bank, score, dist, qn, bn_max, N, M, D, k, s);
  row_sqnorm_kernel<<<runia_rows_grid(M), 64 * kRowWaves, 0, s>>>(bank, bn, M, D, bn_max);
  int rc = runia_check_launch();
  if (rc != RUNIA_OK) return rc;
  if (use16) {
    rc = runia_knn16_split_bank(bank, const_cast<uint16_t*>(r16.bank_planes), bn, const_cast<float*>(r16.bn_p), M, D, s);
    if (rc != RUNIA_OK) return rc;
  }
  return knn_scan(q, bank, score, dist, qn, bn, bn_max, use16 ? &r16 : nullptr, qc, N, M, D, k, s);
}

// ---- a bank prepared once (the index of a deployed postprocessor): |b|^2, their maximum and - for banks the bf16 kernel
// can take - |b|^2 in piece order and the bf16 pieces.  A call against a prepared bank skips those passes (50 000 x 2048:
// 0.13 ms of norms + 0.25 ms of splitting, more than the scan itself for some hundred queries); the scores are the same bits.
#ifndef KNN16_MIN_ROWS_PREPARED
#define KNN16_MIN_ROWS_PREPARED 64
#endif
// with the bank's pieces already there, the bf16 kernel pays from far fewer queries (its 256-row query tile is padded)
static bool knn16_prepared_wanted(int64_t N, int64_t M, int64_t D) {
  return knn16_bank_ok(M, D) && N >= KNN16_MIN_ROWS_PREPARED && N * M >= ((int64_t)1 << 29) / D &&
         (M * D >= ((int64_t)1 << 23) || N >= 1000);  // (tools/ablate/run_knn_prepared.py: small banks stay on the f32 kernel)
}
static size_t knn_state_head_bytes(int64_t M) { return (((size_t)M + 1) * sizeof(float) + 255) / 256 * 256; }
static size_t knn_state_bnp_bytes(int64_t M) { return align256((size_t)runia_knn16_padded_rows(M) * sizeof(float)); }
extern "C" size_t runia_knn_bank_state_bytes(int64_t M, int64_t D) {
  if (M <= 0 || D <= 0) return 0;
  return knn_state_head_bytes(M) + (knn16_bank_ok(M, D) ? knn_state_bnp_bytes(M) + runia_knn16_plane_bytes(M, D) : 0);
}
extern "C" int runia_knn_prepare_bank_f32(const float* bank, void* state, size_t state_bytes, int64_t M, int64_t D,
                                          runia_stream_t stream) {
  if (M <= 0 || D <= 0 || !bank || !state) return RUNIA_E_INVALID;
  if ((((uintptr_t)state) & 15) != 0 || state_bytes < knn_state_head_bytes(M)) return RUNIA_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* bn = reinterpret_cast<float*>(state);
  unsigned* bn_max = reinterpret_cast<unsigned*>(bn + M);
  if (hipMemsetAsync(bn_max, 0, sizeof(unsigned), s) != hipSuccess) return RUNIA_E_LAUNCH;
  row_sqnorm_kernel<<<runia_rows_grid(M), 64 * kRowWaves, 0, s>>>(bank, bn, M, D, bn_max);
  int rc = runia_check_launch();
  if (rc != RUNIA_OK) return rc;
  if (knn16_bank_ok(M, D) && state_bytes >= runia_knn_bank_state_bytes(M, D)) {
    char* st = reinterpret_cast<char*>(state);
    rc = runia_knn16_split_bank(bank, reinterpret_cast<uint16_t*>(st + knn_state_head_bytes(M) + knn_state_bnp_bytes(M)), bn,
                                reinterpret_cast<float*>(st + knn_state_head_bytes(M)), M, D, s);
  }
  return rc;
}
// workspace of a call against a prepared bank: the dense rows, |q|^2 and (bf16 kernel) the filter's buffers and the chunk's pieces
extern "C" size_t runia_knn_prepared_workspace_bytes(int64_t N, int64_t M, int64_t D, int k) {
  if (N <= 0 || M <= 0) return 0;
  if (!knn16_prepared_wanted(N, M, D)) {
    const int64_t qc = knn_chunk_rows(N, M);
    return align256((size_t)(qc * M + qc) * sizeof(float));
  }
  const int64_t S = knn16_sample_rows(N, M, k), qc = knn16_chunk_rows(N, M, S), dr = knn16_dense_rows(qc, M, S);
  const Knn16Filter f = knn16_filter_layout(align256((size_t)(dr * M) * sizeof(float)), qc, dr, M, D, S, false);
  return f.end + runia_knn16_plane_bytes(qc, D);
}
extern "C" int runia_knn_kth_prepared_f32(const float* q, const float* bank, const void* state, size_t state_bytes,
                                          float* score, void* workspace, size_t workspace_bytes, int64_t N, int64_t M,
                                          int64_t D, int k, runia_stream_t stream) {
  if (N < 0 || M <= 0 || D <= 0 || k < 1) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!q || !score || !bank || !state) return RUNIA_E_INVALID;
  if (state_bytes < knn_state_head_bytes(M)) return RUNIA_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  if (k > M) {
    fill_kernel<<<runia_stream_grid(N, 256), 256, 0, s>>>(score, N, -kFltMax);
    return runia_check_launch();
  }
  if (!workspace || workspace_bytes < (size_t)(M + 1) * sizeof(float)) return RUNIA_E_WORKSPACE;
  const float* bn = reinterpret_cast<const float*>(state);
  unsigned* bn_max = const_cast<unsigned*>(reinterpret_cast<const unsigned*>(bn + M));  // (only ever re-written with its own value)
  // the bf16 kernel when the state holds the pieces and the workspace is the one asked for; else the f32 kernel with as
  // many query rows per pass as the workspace holds
  const bool use16 = knn16_prepared_wanted(N, M, D) && state_bytes >= runia_knn_bank_state_bytes(M, D) &&
                     workspace_bytes >= runia_knn_prepared_workspace_bytes(N, M, D, k);
  const int64_t S16 = use16 ? knn16_sample_rows(N, M, k) : 0;
  int64_t qc = use16 ? knn16_chunk_rows(N, M, S16) : (int64_t)(workspace_bytes / sizeof(float) / (size_t)(M + 1));
  if (qc < 1) return RUNIA_E_WORKSPACE;
  if (qc > N) qc = N;
  if (!use16 && qc > kQueryChunk) qc = kQueryChunk;
  float* dist = reinterpret_cast<float*>(workspace);
  float* qn = dist + qc * M;
  if (N <= kSmallQ && M >= 1024 && qc >= N)  // a handful of queries: one pass over the bank, exact distances
    return knn_small_path(q, bank, score, dist, qn, bn_max, N, M, D, k, s);
  Knn16Run r16{};
  if (use16) {
    const char* st = reinterpret_cast<const char*>(state);
    r16.S = S16;
    r16.dense_rows = knn16_dense_rows(qc, M, r16.S);
    r16.ws = reinterpret_cast<char*>(workspace);
    r16.f = knn16_filter_layout(align256((size_t)(r16.dense_rows * M) * sizeof(float)), qc, r16.dense_rows, M, D, r16.S, false);
    r16.bn_p = reinterpret_cast<const float*>(st + knn_state_head_bytes(M));
    r16.bank_planes = reinterpret_cast<const uint16_t*>(st + knn_state_head_bytes(M) + knn_state_bnp_bytes(M));
    qn = reinterpret_cast<float*>(r16.ws + r16.f.qn);
    r16.q_planes = reinterpret_cast<uint16_t*>(r16.ws + r16.f.end);
  }
  return knn_scan(q, bank, score, dist, qn, bn, bn_max, use16 ? &r16 : nullptr, qc, N, M, D, k, s, 1);
}

extern "C" int runia_linear_f32(const float* x, const float* w, const float* bias, float* out, int64_t N, int64_t D,
                                int64_t C, float clip_max, runia_stream_t stream) {
  if (N < 0 || D <= 0 || C <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !w || !out) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (C <= kSkinnyMaxC && C * D <= kSkinnyMaxFloats && (D & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0) {
    const size_t lds = (size_t)C * D * sizeof(float);
    const int64_t per_wg = 4 * kSkinnyRowsPerWave;
    const unsigned grid = (unsigned)((N + per_wg - 1) / per_wg);
#define RUNIA_SKINNY(CT)                                                                                          \
  {                                                                                                               \
    static std::atomic<uint64_t> lds_ok{0};                                                                       \
    if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(linear_skinny_kernel<CT>), 96 * 1024, lds_ok) !=    \
        RUNIA_OK)                                                                                                 \
      return RUNIA_E_LAUNCH;                                                                                      \
    linear_skinny_kernel<CT><<<grid, 256, lds, s>>>(x, w, bias, out, N, (int)D, (int)C, clip_max);               \
  }
    if (C <= 4) RUNIA_SKINNY(4)
    else if (C <= 8) RUNIA_SKINNY(8)
    else if (C <= 12) RUNIA_SKINNY(12)
    else RUNIA_SKINNY(16)
#undef RUNIA_SKINNY
    return runia_check_launch();
  }
  if (N <= kLinearFewRows) {
    linear_few_rows_kernel<<<(unsigned)((N * C + 255) / 256), 256, 0, s>>>(x, w, bias, out, N, D, C, clip_max);
    return runia_check_launch();
  }
  if (N <= kLinearMidRows) {
    const dim3 grid((unsigned)((C + kMidCls - 1) / kMidCls), (unsigned)((N + kMidRows - 1) / kMidRows));
    linear_mid_rows_kernel<<<grid, 256, 0, s>>>(x, w, bias, out, N, D, C, clip_max);
    return runia_check_launch();
  }
  const int64_t qt = (N + TQ - 1) / TQ;
  for (int64_t t0 = 0; t0 < qt; t0 += 65535) {  // grid.y limit
    const int64_t tiles = (qt - t0 < 65535) ? (qt - t0) : 65535;
    const int64_t r0 = t0 * TQ;
    const int64_t rows = (N - r0 < tiles * TQ) ? (N - r0) : tiles * TQ;
    knn_dist_kernel<EPI_LINEAR><<<knn_dist_grid(rows, C), 256, 0, s>>>(x + r0 * D, w, nullptr, bias, out + r0 * C, rows, C, D, clip_max);
  }
  return runia_check_launch();
}

extern "C" int runia_kde_score_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                                   int64_t D, double bandwidth, runia_stream_t stream) {
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0)) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!train || !x || !score) return RUNIA_E_INVALID;
  const double log_norm = -log((double)M) - (double)D * log(bandwidth) - 0.5 * (double)D * log(2.0 * M_PI);
  const double nh = -0.5 / (bandwidth * bandwidth);
  hipStream_t s = as_stream(stream);
  if (D <= 8) launch_kde_small<8>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 16) launch_kde_small<16>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 32) launch_kde_small<32>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else if (D <= 64) launch_kde_small<64>(train, x, score, M, N, (int)D, nh, log_norm, s);
  else {
    const size_t shmem = (size_t)D * sizeof(double);
    if (shmem > 64 * 1024) return RUNIA_E_INVALID;
    kde_kernel<<<runia_stream_grid(N, 1), 256, shmem, s>>>(train, x, score, M, N, D, nh, log_norm);
  }
  return runia_check_launch();
}

// log of sklearn's kernel normalisation (neighbors/_binary_tree.pxi.tp, _log_kernel_norm): -factor - d log h
static double kde_log_norm(int kind, int64_t D, double h) {
  const double d = (double)D, log_pi = log(M_PI), log_2pi = log(2.0 * M_PI);
  auto logVn = [&](double n) { return 0.5 * n * log_pi - lgamma(0.5 * n + 1.0); };  // volume of the unit n-ball
  auto logSn = [&](double n) { return log_2pi + logVn(n - 1.0); };                   // surface of the unit n-sphere
  double factor = 0.0;
  switch (kind) {
    case 0: factor = 0.5 * d * log_2pi; break;
    case 1: factor = logVn(d); break;
    case 2: factor = logVn(d) + log(2.0 / (d + 2.0)); break;
    case 3: factor = logSn(d - 1.0) + lgamma(d); break;
    case 4: factor = logVn(d) - log(d + 1.0); break;
    default: {
      double tmp = 2.0 / M_PI;
      for (int64_t k = 1; k < D + 1; k += 2) {
        factor += tmp;
        tmp *= -(d - (double)k) * (d - (double)k - 1.0) * (2.0 / M_PI) * (2.0 / M_PI);
      }
      factor = log(factor) + logSn(d - 1.0);
    }
  }
  return -factor - d * log(h);
}

extern "C" int runia_kde_score_kernel_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                                          int64_t D, double bandwidth, int kind, runia_stream_t stream) {
  if (kind == 0) return runia_kde_score_f64(train, x, score, M, N, D, bandwidth, stream);
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0) || kind < 0 || kind > 5) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!train || !x || !score) return RUNIA_E_INVALID;
  const size_t shmem = (size_t)D * sizeof(double);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  const double log_norm = -log((double)M) + kde_log_norm(kind, D, bandwidth);
  hipStream_t s = as_stream(stream);
  const unsigned grid = runia_stream_grid(N, 1);
  switch (kind) {
    case 1: kde_other_kernel<1><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 2: kde_other_kernel<2><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 3: kde_other_kernel<3><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    case 4: kde_other_kernel<4><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
    default: kde_other_kernel<5><<<grid, 256, shmem, s>>>(train, x, score, M, N, D, bandwidth, log_norm); break;
  }
  return runia_check_launch();
}
