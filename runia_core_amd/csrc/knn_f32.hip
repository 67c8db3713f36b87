// a8  kNN : faiss.IndexFlatL2.search(q, k) -> -(k-th smallest squared L2)
//            (reference inference/postprocessors.py:417-421, 873-880; faiss-gpu==1.7.2)
//
// kNN = the one f32 dense contraction of the path, on the matrix cores (nt_tile_f32.hpp):
//   d2[q, m] = |q|^2 + |b_m|^2 - 2 q.b_m
// Distances go to a row-chunked workspace [Qc, M]; the k-th order statistic of each row is then found by a histogram select
// over the distance bits and an exact re-measurement of the candidates around it - no sort, no top-k list.  Large problems
// rank the bank rows by bf16 piece products first (knn_bf16.hip) and never write the Q x M matrix (candidate lists).
// This file: the selection kernels, the handful-of-queries path, the dispatch and the C entry points.
#include "nt_tile_f32.hpp"
#include "knn_perm.hpp"

namespace {

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
// One or two queries: one bank row per wave, the queries read through the L1 (8 KB each) - 73 us of kernel time for one query
// against 50 000 x 2048 (the LDS form below: 83).
constexpr int kOneQ = 2;
__global__ __launch_bounds__(64 * kRowWaves) void knn_one_dist_kernel(const float* __restrict__ q,
                                                                        const float* __restrict__ bank,
                                                                        float* __restrict__ dist, float* __restrict__ qn,
                                                                        unsigned* __restrict__ bn_max_bits, int Q, int64_t M,
                                                                        int64_t D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && wave < Q) {  // |q|^2 of the selection's range bound (was a launch of its own in front of this one)
    float sq = 0.f;
    for (int64_t i = lane; i < D; i += 64) sq = fmaf(q[(int64_t)wave * D + i], q[(int64_t)wave * D + i], sq);
    sq = wave_sum_f32(sq);
    if (lane == 0) qn[wave] = sq;
  }
  for (int64_t m = (int64_t)blockIdx.x * kRowWaves + wave; m < M; m += (int64_t)gridDim.x * kRowWaves) {
    const float* br = bank + m * D;
    float acc[kOneQ], bsq = 0.f;
#pragma unroll
    for (int j = 0; j < kOneQ; ++j) acc[j] = 0.f;
    int64_t i = lane;
    for (; i + 448 < D; i += 512) {
      float ba[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ba[u] = br[i + 64 * u];
#pragma unroll
      for (int j = 0; j < kOneQ; ++j) {
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
      for (int j = 0; j < kOneQ; ++j) {
        if (j < Q) {
          const float df = q[(int64_t)j * D + i] - b;
          acc[j] = fmaf(df, df, acc[j]);
        }
      }
      bsq = fmaf(b, b, bsq);
    }
#pragma unroll
    for (int j = 0; j < kOneQ; ++j) {
      if (j < Q) {
        const float d = wave_sum_f32(acc[j]);
        // as the matrix-core kernels' epilogues: a NaN or infinite distance counts as faiss's FLT_MAX fill
        if (lane == 0) dist[(int64_t)j * M + m] = (d == d) ? fminf(d, kFltMax) : kFltMax;
      }
    }
    bsq = wave_sum_f32(bsq);
    if (lane == 0 && bsq < INFINITY) {  // NaN / inf rows do not set the range (as row_sqnorm_kernel)
      const unsigned b = __float_as_uint(bsq);
      if (bn_max_bits && b > __atomic_load_n(bn_max_bits, __ATOMIC_RELAXED)) atomicMax(bn_max_bits, b);  // (null: a prepared bank state already holds it)
    }
  }
}

// Round 5: the queries sit in LDS (staged once per workgroup, which then walks its share of the bank), two bank rows per wave
// and trip, 16 waves per workgroup.  With the queries read from global memory for every bank row the Q x 32 query loads per
// row (L1 / L2 latency in front of every group of fmas) outweighed the row's own 32: 8 queries took 150 us of kernel time
// against 73 for one.  3 to 12 queries whose rows fit 144 KB of LDS take this path (12 x 2048: 148 us of kernel time; from ~14 the padded
// matrix-core tiles - 164 us whatever the count - are faster).
constexpr int kSmallQ = 12;
constexpr int kSmallRows = 2;    // bank rows per wave and trip
constexpr int kSmallWaves = 16;  // waves per workgroup
constexpr int64_t kSmallLdsBytes = 144 * 1024;
__global__ __launch_bounds__(64 * kSmallWaves) void knn_small_dist_kernel(const float* __restrict__ q,
                                                                          const float* __restrict__ bank,
                                                                          float* __restrict__ dist, float* __restrict__ qn,
                                                                          unsigned* __restrict__ bn_max_bits, int Q, int64_t M,
                                                                          int64_t D) {
  extern __shared__ float q_lds[];  // [Q][D]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int64_t total = (int64_t)Q * D;
    if ((total & 3) == 0 && ((((uintptr_t)q) & 15) == 0)) {
      for (int64_t t = tid; t < (total >> 2); t += 64 * kSmallWaves)
        reinterpret_cast<float4*>(q_lds)[t] = reinterpret_cast<const float4*>(q)[t];
    } else {
      for (int64_t t = tid; t < total; t += 64 * kSmallWaves) q_lds[t] = q[t];
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && wave < Q) {  // |q|^2 of the selection's range bound (was a launch of its own in front of this one)
    float sq = 0.f;
    for (int64_t i = lane; i < D; i += 64) sq = fmaf(q_lds[(int64_t)wave * D + i], q_lds[(int64_t)wave * D + i], sq);
    sq = wave_sum_f32(sq);
    if (lane == 0) qn[wave] = sq;
  }
  for (int64_t m0 = ((int64_t)blockIdx.x * kSmallWaves + wave) * kSmallRows; m0 < M; m0 += (int64_t)gridDim.x * kSmallWaves * kSmallRows) {
    const float* br[kSmallRows];
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) br[r] = bank + ((m0 + r < M) ? m0 + r : M - 1) * D;  // (rows past the bank: the last row again, not stored)
    float acc[kSmallQ][kSmallRows], bsq[kSmallRows];
#pragma unroll
    for (int j = 0; j < kSmallQ; ++j)
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) bsq[r] = 0.f;
    int64_t i = lane;
    for (; i + 448 < D; i += 512) {
      float ba[kSmallRows][8];
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r)
#pragma unroll
        for (int u = 0; u < 8; ++u) ba[r][u] = br[r][i + 64 * u];
#pragma unroll
      for (int j = 0; j < kSmallQ; ++j) {
        if (j < Q) {  // (uniform)
          const float* qr = q_lds + (int64_t)j * D + i;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float qv = qr[64 * u];
#pragma unroll
            for (int r = 0; r < kSmallRows; ++r) {
              const float df = qv - ba[r][u];
              acc[j][r] = fmaf(df, df, acc[j][r]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r)
#pragma unroll
        for (int u = 0; u < 8; ++u) bsq[r] = fmaf(ba[r][u], ba[r][u], bsq[r]);
    }
    for (; i < D; i += 64) {
      float b[kSmallRows];
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r) b[r] = br[r][i];
#pragma unroll
      for (int j = 0; j < kSmallQ; ++j) {
        if (j < Q) {
          const float qv = q_lds[(int64_t)j * D + i];
#pragma unroll
          for (int r = 0; r < kSmallRows; ++r) {
            const float df = qv - b[r];
            acc[j][r] = fmaf(df, df, acc[j][r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kSmallRows; ++r) bsq[r] = fmaf(b[r], b[r], bsq[r]);
    }
#pragma unroll
    for (int j = 0; j < kSmallQ; ++j) {
      if (j < Q) {
#pragma unroll
        for (int r = 0; r < kSmallRows; ++r) {
          const float d = wave_sum_f32(acc[j][r]);
          // as the matrix-core kernels' epilogues: a NaN or infinite distance counts as faiss's FLT_MAX fill
          if (lane == 0 && m0 + r < M) dist[(int64_t)j * M + m0 + r] = (d == d) ? fminf(d, kFltMax) : kFltMax;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kSmallRows; ++r) {
      const float bs = wave_sum_f32(bsq[r]);
      if (lane == 0 && m0 + r < M && bs < INFINITY) {  // NaN / inf rows do not set the range (as row_sqnorm_kernel)
        const unsigned b = __float_as_uint(bs);
        if (bn_max_bits && b > __atomic_load_n(bn_max_bits, __ATOMIC_RELAXED)) atomicMax(bn_max_bits, b);  // (null: a prepared bank state already holds it)
      }
    }
  }
}
static bool knn_small_fits(int64_t N, int64_t M, int64_t D) { return N <= kSmallQ && M >= 1024 && N * D * 4 <= kSmallLdsBytes; }

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

constexpr int64_t kQueryChunk = 8192;  // query rows per distance-workspace pass

}  // namespace

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
// (round 6: from D = 2 - a call of >= 2^27 pairs on 2 ... 7 features, e.g. KNNLatentSpace behind a PCA to 2 or 4 components in the
// harness sweep, wrote and re-read its whole chunk x bank matrix through the f32 kernel: 10 000 x 50 000 at D = 4 / 3 / 2:
// 1.52 / 1.48 / 1.62 -> 1.01 / 1.02 / 1.36 ms, 65 536 x 50 000 at D = 4: 8.79 -> 5.62 ms, same bits)
// (one feature stays on the f32 kernel: normalised 1-d rows are +-1, every distance is 0 or 4 and half the bank ties at the k-th
// place - 58.8 ms per 10 000 x 50 000 there, 72.7 ms through the pieces; a degenerate case either way)
#ifndef KNN16_MIN_D
#define KNN16_MIN_D 2
#endif
  return KNN_BF16 && M >= 4096 && D >= KNN16_MIN_D && D <= runia_knn16_max_width() && runia_knn16_fits(M, D) &&
         runia_knn16_fits(kQueryChunk, D) && 4 * M * 256 <= ((int64_t)1 << 31);  // (chunks of >= 256 queries: whole tiles)
}
// Worth the two split passes and the 256 x 256 tiles: a bank of some thousand rows, wide features, a batch of queries, 2^31
// multiply-adds, 512 queries (tools/ablate/run_knn_paths.py: 1 024 x 4 096 x 256 is 8 % slower on the bf16 kernel, 1 024 x 4 096 x 2048
// 1.46 x faster, 8 192 x 50 000 x 2048 2.67 x)
#ifndef KNN16_WORK_MIN_D
#define KNN16_WORK_MIN_D 16  // width the work threshold counts at least: narrow rows cost the f32 kernel its distance matrix, not
#endif                       // multiply-adds (4 096 x 20 000 at D = 2 stays there: 0.36 against 0.39 ms; 10 000 x 50 000 does not)
static bool knn16_wanted(int64_t N, int64_t M, int64_t D) {
  const int64_t dw = D < KNN16_WORK_MIN_D ? KNN16_WORK_MIN_D : D;
  return knn16_bank_ok(M, D) && N >= KNN16_MIN_ROWS && N * M >= ((int64_t)1 << 31) / dw;
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
                    const unsigned* bn_max, const Knn16Run* r16, int64_t qc, int64_t N, int64_t M, int64_t D, int k,
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

// bn_max_w: where the distance kernels leave max |b|^2 (a plain call's workspace word, cleared by the caller), or null when
// bn_max already holds it (a prepared bank state: read-only, shareable between concurrent scoring calls).
// Returns 1 when the few-query kernel's LDS limit cannot be raised: the caller then takes the scan.
static int knn_small_path(const float* q, const float* bank, float* score, float* dist, float* qn, const unsigned* bn_max,
                          unsigned* bn_max_w, int64_t N, int64_t M, int64_t D, int k, hipStream_t s) {
  const int lds = (int)(N * D * 4);
  static std::atomic<uint64_t> lds_ok{0};
  if (N > kOneQ && runia_allow_dynamic_lds(reinterpret_cast<const void*>(knn_small_dist_kernel), (int)kSmallLdsBytes, lds_ok) != RUNIA_OK)
    return 1;
  // one workgroup per compute unit (92 registers x 16 waves), each walking its share of the bank: the queries are staged CUs times
  int64_t grid = runia_cu_count();
  const int64_t trips = (M + kSmallWaves * kSmallRows - 1) / (kSmallWaves * kSmallRows);
  if (grid > trips) grid = trips;
  if (N <= kOneQ) knn_one_dist_kernel<<<runia_rows_grid(M), 64 * kRowWaves, 0, s>>>(q, bank, dist, qn, bn_max_w, (int)N, M, D);
  else knn_small_dist_kernel<<<(unsigned)grid, 64 * kSmallWaves, lds, s>>>(q, bank, dist, qn, bn_max_w, (int)N, M, D);
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
  if (knn_small_fits(N, M, D) && qc >= N) {  // a handful of queries: one pass over the bank, exact distances
    const int rc_small = knn_small_path(q, bank, score, dist, qn, bn_max, bn_max, N, M, D, k, s);
    if (rc_small <= 0) return rc_small;
  }
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
#define KNN16_MIN_ROWS_PREPARED 13
#endif
// with the bank's pieces already there, the bf16 kernel pays from far fewer queries (its 256-row query tile is padded): from
// the first count the few-query path does not take (50 000 x 2048: 16 queries 373 us on the f32 tiles, 225 us here)
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
  const unsigned* bn_max = reinterpret_cast<const unsigned*>(bn + M);  // max |b|^2 of the prepared bank: never written by a scoring call
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
  if (knn_small_fits(N, M, D) && qc >= N) {  // a handful of queries: one pass over the bank, exact distances
    const int rc_small = knn_small_path(q, bank, score, dist, qn, bn_max, nullptr, N, M, D, k, s);
    if (rc_small <= 0) return rc_small;
  }
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
