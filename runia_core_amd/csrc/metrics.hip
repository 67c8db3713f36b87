// f2: the metrics step after the scoring path - AUROC / FPR@95 / AUPR of an in-distribution (positive) and an
// out-of-distribution (negative) score set, as get_auroc_results computes them through torchmetrics' binary
// auroc / roc / precision_recall_curve and sklearn.metrics.auc (reference evaluation/metrics.py:37-100; torchmetrics'
// _binary_clf_curve: descending sort, one curve point per run of equal scores, cumulative true / false positives).
// Everything stays on the device: the sort of the 64-bit keys (round 4: one split into 4 096 buckets that are linear in the
// score + a sort per bucket, 10 launches; the stable LSD radix sort of rounds 2-3, 8 passes of 8 bits and ~31 launches, is
// kept behind -DMETRICS_MSD=0), a scan of the labels and of the run ends, and the trapezoid sums.  Round 6: the three scalars
// alone take six launches (four without the sketch) - buckets by splitter keys, bucket sort and curve terms in one launch
// (msd_keys_split_kernel, msd_sort_curve_kernel below); the curve API keeps the eight-launch chain.
//
// Reference behaviour that is reproduced on purpose (the oracle pins it with the reference's goldens):
//   * if any score lies outside [0, 1] (or is NaN) every score goes through a sigmoid first - in the dtype of the
//     scores (f64 scores: f64 sigmoid; f32 scores: f32 sigmoid).  It is monotone but saturates: f64 scores below -745
//     all become 0.0 and tie;
//   * curve points are float32 (tps, fps converted to f32 and divided in f32), the trapezoid terms are formed in f32.
//     The reference then adds them in f32; here they are accumulated in f64 (differences ~1e-7, the test tolerance).
#include "common.hpp"

namespace {

#ifndef METRICS_TILE
#define METRICS_TILE 4096
#endif
constexpr int kTile = METRICS_TILE;  // elements per workgroup in every pass
constexpr int kItems = kTile / 256;

__device__ __forceinline__ uint64_t sortable_desc(double v) {
  // ascending order of the key == descending order of v (NaN keys sort first, as a "largest" value)
  uint64_t b = (uint64_t)__double_as_longlong(v);
  b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);  // ascending-sortable
  return ~b;
}

template <typename T>
__global__ __launch_bounds__(256) void range_check_kernel(const T* __restrict__ ind, int64_t n_ind,
                                                          const T* __restrict__ ood, int64_t n_ood,
                                                          unsigned* __restrict__ any_outside) {
  bool bad = false;
  const int64_t n = n_ind + n_ood;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const T v = (i < n_ind) ? ind[i] : ood[i - n_ind];
    bad = bad || !(v >= (T)0 && v <= (T)1);
  }
  // one word for the whole launch: atomics on it retire at ~88 per microsecond (31 000 waves of out-of-range scores took
  // 0.36 ms), so: one candidate per workgroup, a bounded grid, and only while the flag is still clear
  if (__syncthreads_or(bad) && threadIdx.x == 0 && __atomic_load_n(any_outside, __ATOMIC_RELAXED) == 0u)
    atomicOr(any_outside, 1u);
}

template <typename T>
__global__ __launch_bounds__(256) void make_keys_kernel(const T* __restrict__ ind, int64_t n_ind,
                                                        const T* __restrict__ ood, int64_t n_ood,
                                                        const unsigned* __restrict__ any_outside,
                                                        uint64_t* __restrict__ keys, uint8_t* __restrict__ labels) {
  const bool squash = *any_outside != 0u;
  const int64_t n = n_ind + n_ood;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T v = (i < n_ind) ? ind[i] : ood[i - n_ind];
    if (squash) v = (T)1 / ((T)1 + exp(-v));  // torch.sigmoid in the dtype of the scores
    // + 0.0: -0.0 becomes +0.0, so that the two zeros share a key and form ONE curve point, as torchmetrics' run
    // detection (preds[1:] - preds[:-1] != 0) has it
    keys[i] = sortable_desc((double)v + 0.0);
    labels[i] = (i < n_ind) ? 1 : 0;
  }
}

// ---- LSD radix sort, one 8-bit digit per pass ----------------------------------------------------------------
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                         unsigned* __restrict__ table, unsigned nblocks) {
  __shared__ unsigned hist[256];
  hist[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * kTile;
#pragma unroll
  for (int c = 0; c < kItems; ++c) {
    const int64_t i = t0 + c * 256 + threadIdx.x;
    if (i < n) atomicAdd(&hist[(unsigned)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  table[(size_t)threadIdx.x * nblocks + blockIdx.x] = hist[threadIdx.x];  // digit-major: scan order = (digit, block)
}

// Offsets of a radix pass in two levels, both parallel: workgroup d scans row d of the digit-major table (the counts of
// digit d in every tile) in place and leaves the row total in dtot[d]; every scatter workgroup then scans the 256 totals
// itself.  (One workgroup scanning all 256 x tiles counters took 139 us of a 2 M-key pass's 180 - 52 us once staged
// through LDS; the row scans take a few us.)
__global__ __launch_bounds__(256) void radix_row_scan_kernel(unsigned* __restrict__ table, unsigned nblocks,
                                                              unsigned* __restrict__ dtot) {
  __shared__ unsigned wsum[4];
  __shared__ unsigned carry_s;
  unsigned* row = table + (size_t)blockIdx.x * nblocks;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0u;
  __syncthreads();
  for (unsigned base = 0; base < nblocks; base += 256) {
    const unsigned i = base + tid;
    const unsigned v = (i < nblocks) ? row[i] : 0u;
    unsigned x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned woff = 0u;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const unsigned carry = carry_s;
    if (i < nblocks) row[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 255) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) dtot[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                            const uint8_t* __restrict__ lab_in,
                                                            uint64_t* __restrict__ keys_out, uint8_t* __restrict__ lab_out,
                                                            int64_t n, int shift, const unsigned* __restrict__ table,
                                                            unsigned nblocks, const unsigned* __restrict__ dtot) {
  __shared__ unsigned base[256];       // next free global slot of every digit for this workgroup
  __shared__ unsigned wcnt[4][256];    // per-wave digit counts of the current chunk
  __shared__ unsigned dsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    // first slot of digit `tid` = number of keys with a smaller digit (exclusive scan of the 256 row totals) ...
    const unsigned v = dtot[tid];
    unsigned x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) dsum[wave] = x;
    __syncthreads();
    unsigned woff = 0u;
    for (int w = 0; w < wave; ++w) woff += dsum[w];
    // ... plus the keys of this digit in earlier tiles
    base[tid] = woff + x - v + table[(size_t)tid * nblocks + blockIdx.x];
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * kTile;
  const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int c = 0; c < kItems; ++c) {  // chunks in order: the pass is stable
    const int64_t i = t0 + c * 256 + tid;
    const bool valid = i < n;
    const uint64_t key = valid ? keys_in[i] : 0ull;
    const unsigned d = (unsigned)(key >> shift) & 255u;
    uint64_t same = __ballot(valid);  // lanes of this wave holding the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned rank_in_wave = (unsigned)__popcll(same & lt_mask);
    if (valid && rank_in_wave == 0u) wcnt[wave][d] = (unsigned)__popcll(same);
    __syncthreads();
    if (valid) {
      unsigned off = base[d] + rank_in_wave;
      for (int w = 0; w < wave; ++w) off += wcnt[w][d];
      keys_out[off] = key;
      lab_out[off] = lab_in[i];
    }
    __syncthreads();
    base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
    __syncthreads();
  }
}

// ---- curve: cumulative positives and the previous run end of every run end ------------------------------------------------
// tile pass 1: (label sum, index of the last run end) of every tile
__device__ __forceinline__ void tile_summary_body(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                  int64_t n, unsigned* __restrict__ tile_sum, int* __restrict__ tile_end,
                                                  unsigned* __restrict__ tile_cnt, unsigned tile) {
  __shared__ unsigned ssum[4], scnt[4];
  __shared__ int send[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)tile * kTile;
  __syncthreads();  // (a caller that loops over tiles reuses the arrays above)
  unsigned s = 0u, cnt = 0u;
  int e = -1;
#pragma unroll
  for (int c = 0; c < kItems; ++c) {
    const int64_t i = t0 + c * 256 + tid;
    if (i < n) {
      s += lab[i];
      if (i == n - 1 || keys[i] != keys[i + 1]) { e = (int)i; ++cnt; }  // indices ascend with c
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    cnt += __shfl_xor(cnt, o, 64);
    e = max(e, __shfl_xor(e, o, 64));
  }
  if (lane == 0) { ssum[wave] = s; send[wave] = e; scnt[wave] = cnt; }
  __syncthreads();
  if (tid == 0) {
    tile_sum[tile] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    tile_end[tile] = max(max(send[0], send[1]), max(send[2], send[3]));
    tile_cnt[tile] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
  }
}
__global__ __launch_bounds__(256) void tile_summary_kernel(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                           int64_t n, unsigned* __restrict__ tile_sum,
                                                           int* __restrict__ tile_end, unsigned* __restrict__ tile_cnt) {
  tile_summary_body(keys, lab, n, tile_sum, tile_end, tile_cnt, blockIdx.x);
}

// exclusive scan of the tile summaries (sum: +, end: max) by one wave, 64 tiles per trip
__device__ __forceinline__ void tile_scan_body(unsigned* __restrict__ tile_sum, int* __restrict__ tile_end,
                                               unsigned* __restrict__ tile_cnt, int64_t ntiles) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  unsigned cs = 0u, cc = 0u;
  int ce = -1;
  for (int64_t base = 0; base < ntiles; base += 64) {
    const int64_t i = base + lane;
    const unsigned ts = (i < ntiles) ? tile_sum[i] : 0u;
    const unsigned tc = (i < ntiles) ? tile_cnt[i] : 0u;
    const int te = (i < ntiles) ? tile_end[i] : -1;
    unsigned x = ts, k = tc;
    int m = te;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      const unsigned q = __shfl_up(k, o, 64);
      const int z = __shfl_up(m, o, 64);
      if (lane >= o) { x += y; k += q; m = max(m, z); }
    }
    int em = __shfl_up(m, 1, 64);  // exclusive maximum
    if (lane == 0) em = -1;
    if (i < ntiles) {
      tile_sum[i] = cs + x - ts;
      tile_cnt[i] = cc + k - tc;
      tile_end[i] = max(ce, em);
    }
    cs += __shfl(x, 63, 64);
    cc += __shfl(k, 63, 64);
    ce = max(ce, __shfl(m, 63, 64));
  }
}
__global__ void tile_scan_kernel(unsigned* __restrict__ tile_sum, int* __restrict__ tile_end,
                                 unsigned* __restrict__ tile_cnt, int64_t ntiles) {
  if (blockIdx.x == 0) tile_scan_body(tile_sum, tile_end, tile_cnt, ntiles);
}

// tile pass 2: tps[i] (inclusive) for every element; prev_end[i] for every run end
template <bool RAW = false>  // RAW: the tile summaries as tile_summary_kernel left them - the carries are added up here
__device__ __forceinline__ void tile_prefix_body(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                 int64_t n, const unsigned* __restrict__ tile_sum,
                                                 const int* __restrict__ tile_end, const unsigned* __restrict__ tile_cnt,
                                                 unsigned* __restrict__ tps, int* __restrict__ prev_end,
                                                 unsigned* __restrict__ tps_out, unsigned* __restrict__ fps_out,
                                                 int64_t* __restrict__ n_points, unsigned tile) {
  // tps_out / fps_out (optional): torchmetrics' _binary_clf_curve - cumulative true / false positives at the end of every
  // run of equal scores, in descending score order (entry r belongs to the run end of rank r)
  __shared__ unsigned wsum[4], wcnt[4];
  __shared__ int wend[4];
  __shared__ unsigned carry_sum, carry_cnt;
  __shared__ int carry_end;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t0 = (int64_t)tile * kTile;
  __syncthreads();  // (a caller that loops over tiles reuses the carries)
  if constexpr (RAW) {
    // every workgroup adds up the summaries of the tiles before its own (<= 1 024 x 12 bytes from L2, ~1 us side by side)
    // instead of one wave scanning them in a launch of its own in front of this one (5.6 us + a launch boundary)
    unsigned ps = 0u, pc = 0u;
    int pe = -1;
    for (unsigned t = tid; t < tile; t += 256) { ps += tile_sum[t]; pc += tile_cnt[t]; pe = max(pe, tile_end[t]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ps += __shfl_xor(ps, o, 64);
      pc += __shfl_xor(pc, o, 64);
      pe = max(pe, __shfl_xor(pe, o, 64));
    }
    if (lane == 0) { wsum[wave] = ps; wcnt[wave] = pc; wend[wave] = pe; }
    __syncthreads();
    if (tid == 0) {
      carry_sum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      carry_cnt = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      carry_end = max(max(wend[0], wend[1]), max(wend[2], wend[3]));
    }
  } else {
    if (tid == 0) { carry_sum = tile_sum[tile]; carry_end = tile_end[tile]; carry_cnt = tile_cnt[tile]; }
  }
  __syncthreads();
  for (int c = 0; c < kItems; ++c) {
    const int64_t i = t0 + c * 256 + tid;
    const bool valid = i < n;
    const unsigned l = valid ? lab[i] : 0u;
    const bool is_end = valid && (i == n - 1 || keys[i] != keys[i + 1]);
    unsigned x = l;
    int m = is_end ? (int)i : -1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      const int q = __shfl_up(m, o, 64);
      if (lane >= o) { x += y; m = max(m, q); }
    }
    const uint64_t ends = __ballot(is_end);
    const unsigned ends_before = (unsigned)__popcll(ends & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
    if (lane == 63) { wsum[wave] = x; wend[wave] = m; wcnt[wave] = (unsigned)__popcll(ends); }
    __syncthreads();
    unsigned woff = 0u, coff = 0u;
    int wmax = -1;
    for (int w = 0; w < wave; ++w) { woff += wsum[w]; coff += wcnt[w]; wmax = max(wmax, wend[w]); }
    const unsigned cs = carry_sum, cc = carry_cnt;
    const int ce = carry_end;
    // exclusive maximum of the run-end indices before i
    int excl = __shfl_up(m, 1, 64);
    if (lane == 0) excl = -1;
    excl = max(max(excl, wmax), ce);
    if (valid) {
      tps[i] = cs + woff + x;
      if (is_end) {
        prev_end[i] = excl;
        if (tps_out) {
          const unsigned tp = cs + woff + x, rank = cc + coff + ends_before;
          tps_out[rank] = tp;
          fps_out[rank] = (unsigned)(i + 1) - tp;
          if (i == n - 1) *n_points = (int64_t)rank + 1;
        }
      }
    }
    __syncthreads();
    if (tid == 255) {
      carry_sum = cs + woff + x;
      carry_cnt = cc + coff + ends_before + (is_end ? 1u : 0u);
      carry_end = max(max(m, wmax), ce);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void tile_prefix_kernel(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                          int64_t n, const unsigned* __restrict__ tile_sum,
                                                          const int* __restrict__ tile_end,
                                                          const unsigned* __restrict__ tile_cnt, unsigned* __restrict__ tps,
                                                          int* __restrict__ prev_end, unsigned* __restrict__ tps_out,
                                                          unsigned* __restrict__ fps_out, int64_t* __restrict__ n_points) {
  tile_prefix_body(keys, lab, n, tile_sum, tile_end, tile_cnt, tps, prev_end, tps_out, fps_out, n_points, blockIdx.x);
}
__global__ __launch_bounds__(256) void tile_prefix_raw_kernel(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ lab,
                                                              int64_t n, const unsigned* __restrict__ tile_sum,
                                                              const int* __restrict__ tile_end,
                                                              const unsigned* __restrict__ tile_cnt, unsigned* __restrict__ tps,
                                                              int* __restrict__ prev_end, unsigned* __restrict__ tps_out,
                                                              unsigned* __restrict__ fps_out, int64_t* __restrict__ n_points) {
  tile_prefix_body<true>(keys, lab, n, tile_sum, tile_end, tile_cnt, tps, prev_end, tps_out, fps_out, n_points, blockIdx.x);
}

struct MetricsAccum {
  double roc_sum;            // sum of (fpr_j - fpr_{j-1}) * (tpr_j + tpr_{j-1})   (f32 terms)
  double pr_sum;             // sum of (recall_{j-1} - recall_j) * (precision_{j-1} + precision_j) and the end-point term
  unsigned long long fpr95_idx;  // smallest run-end index with tpr >= 0.95
};

__device__ __forceinline__ void curve_terms_body(const uint64_t* __restrict__ keys, int64_t n,
                                                 const unsigned* __restrict__ tps, const int* __restrict__ prev_end,
                                                 MetricsAccum* __restrict__ acc, unsigned block, unsigned blocks,
                                                 MetricsAccum* __restrict__ parts = nullptr) {
  __shared__ double sroc[4], spr[4];
  __shared__ unsigned long long sidx[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float P = (float)tps[n - 1];
  const float Nn = (float)((unsigned)n - tps[n - 1]);
  double roc = 0.0, pr = 0.0;
  unsigned long long first = ~0ull;
  // four consecutive elements per thread and trip, their loads requested together: a run end costs a chain of two dependent
  // loads (prev_end[i], then tps[prev_end[i]]), and with one element per trip a thread walked 15 such chains one after the
  // other at 2 M keys (54 us; the arithmetic is ~5 us)
  for (int64_t i0 = ((int64_t)block * 256 + tid) * 4; i0 < n; i0 += (int64_t)blocks * 1024) {
    uint64_t k[5];
    int pe[4];
    unsigned tq[4];
    if (i0 + 4 < n) {
      const ulonglong2 ka = *reinterpret_cast<const ulonglong2*>(keys + i0), kb = *reinterpret_cast<const ulonglong2*>(keys + i0 + 2);
      k[0] = ka.x; k[1] = ka.y; k[2] = kb.x; k[3] = kb.y; k[4] = keys[i0 + 4];
      const int4 pv = *reinterpret_cast<const int4*>(prev_end + i0);
      pe[0] = pv.x; pe[1] = pv.y; pe[2] = pv.z; pe[3] = pv.w;
      const uint4 tv = *reinterpret_cast<const uint4*>(tps + i0);
      tq[0] = tv.x; tq[1] = tv.y; tq[2] = tv.z; tq[3] = tv.w;
    } else {
#pragma unroll
      for (int j = 0; j < 5; ++j) k[j] = (i0 + j < n) ? keys[i0 + j] : 0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pe[j] = (i0 + j < n) ? prev_end[i0 + j] : -1;
        tq[j] = (i0 + j < n) ? tps[i0 + j] : 0u;
      }
    }
    bool end[4];
    unsigned tq0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      end[j] = (i0 + j < n) && (i0 + j == n - 1 || k[j] != k[j + 1]);
      tq0[j] = (end[j] && pe[j] >= 0) ? tps[pe[j]] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!end[j]) continue;
      const int64_t i = i0 + j;
      const int p = pe[j];
      const float tp = (float)tq[j], fp = (float)((unsigned)(i + 1) - tq[j]);
      const float tp0 = (p < 0) ? 0.f : (float)tq0[j], fp0 = (p < 0) ? 0.f : (float)((unsigned)(p + 1) - tq0[j]);
      const float tpr = tp / P, fpr = fp / Nn, tpr0 = tp0 / P, fpr0 = fp0 / Nn;
      roc += (double)((fpr - fpr0) * (tpr + tpr0));
      if (tpr >= 0.95f) first = min(first, (unsigned long long)i);
      // precision-recall points: this run end and the one before it (the first run end pairs with the (1, 0) end point)
      const float prec = tp / (tp + fp), rec = tpr;
      const float prec0 = (p < 0) ? 1.0f : tp0 / (tp0 + fp0), rec0 = (p < 0) ? 0.0f : tpr0;
      pr += (double)((rec0 - rec) * (prec0 + prec));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    roc += shfl_xor_f64(roc, o);
    pr += shfl_xor_f64(pr, o);
    const unsigned lo = __shfl_xor((unsigned)first, o, 64), hi = __shfl_xor((unsigned)(first >> 32), o, 64);
    first = min(first, ((unsigned long long)hi << 32) | lo);
  }
  if (lane == 0) { sroc[wave] = roc; spr[wave] = pr; sidx[wave] = first; }
  __syncthreads();
  if (tid == 0) {
    const double r = ((sroc[0] + sroc[1]) + sroc[2]) + sroc[3], q = ((spr[0] + spr[1]) + spr[2]) + spr[3];
    const unsigned long long f = min(min(sidx[0], sidx[1]), min(sidx[2], sidx[3]));
    if (parts) {  // the workgroup's own record: the last workgroup adds the records up in index order
      parts[block].roc_sum = r; parts[block].pr_sum = q; parts[block].fpr95_idx = f;
    } else {
      unsafeAtomicAdd(&acc->roc_sum, r);
      unsafeAtomicAdd(&acc->pr_sum, q);
      atomicMin(&acc->fpr95_idx, f);
    }
  }
}
__global__ __launch_bounds__(256) void curve_terms_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                          const unsigned* __restrict__ tps, const int* __restrict__ prev_end,
                                                          MetricsAccum* __restrict__ acc) {
  curve_terms_body(keys, n, tps, prev_end, acc, blockIdx.x, gridDim.x);
}

__device__ __forceinline__ void finalize_body(const MetricsAccum* __restrict__ acc, const unsigned* __restrict__ tps, int64_t n,
                                              double* __restrict__ out);
__global__ void finalize_kernel(const MetricsAccum* __restrict__ acc, const unsigned* __restrict__ tps, int64_t n,
                                double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  finalize_body(acc, tps, n, out);
}
__device__ __forceinline__ void finalize_body(const MetricsAccum* __restrict__ acc, const unsigned* __restrict__ tps, int64_t n,
                                              double* __restrict__ out) {
  const float Nn = (float)((unsigned)n - tps[n - 1]);
  out[0] = (double)(float)(acc->roc_sum * 0.5);                // trapz(tpr, fpr), reported as float32 like torchmetrics
  const unsigned long long i = acc->fpr95_idx;
  out[1] = (i == ~0ull) ? NAN : (double)((float)((unsigned)(i + 1) - tps[i]) / Nn);
  // pr_sum = sum over curve pairs of (recall_prev - recall_cur) * (precision_prev + precision_cur) <= 0, the first run end
  // paired with torchmetrics' (precision 1, recall 0) end point: twice trapz(precision, recall) along the decreasing
  // recall axis; sklearn.metrics.auc flips the sign of a decreasing axis
  out[2] = (double)(float)(-(acc->pr_sum * 0.5));
}

// ---- round 4: the sort as ONE most-significant-digit split + a sort per bucket (10 launches instead of ~31) ---------------
// The step is bound by its DEPENDENT LAUNCHES (~7.5 us each), not by bytes: 8 least-significant-digit passes x 3 launches
// moved every key 8 times to order 2 M keys.  Here: the probe also finds the range of the finite scores; a key's bucket (of
// 4 096) is linear in its (squashed) score over that range (msd_bucket); one launch counts the buckets while it makes the keys, one scans them, one scatters
// (per tile: bucket counts in LDS, ONE global atomic per non-empty (tile, bucket) reserves its range - a hot bucket costs a
// tile one atomic, not one per key), and one sorts every bucket where it lies: up to 2 048 keys by a bitonic network in
// LDS, a larger bucket (a heavily skewed score set) by a stable 8-bit radix sort that one workgroup runs over the bucket in
// global memory - slow but bounded, and the grid needs no second look.  The order inside a run of equal keys is irrelevant
// to the curve (a run is one point), so neither the scatter nor the network has to be stable.
#ifndef METRICS_MSD_BITS
#define METRICS_MSD_BITS 12
#endif
#ifndef METRICS_SCAT_ITEMS
#define METRICS_SCAT_ITEMS 32
#endif
constexpr int kMsdBits = METRICS_MSD_BITS, kMsdBuckets = 1 << kMsdBits;
constexpr int kScatTileHist = 2048;  // scores per workgroup of the sketch launch

constexpr unsigned kProbeBlocks = 256;
struct ProbeRec { unsigned long long nkmin, kmax; unsigned bad, pad; };
// device words of the split.  Nothing of it has to be cleared beforehand: the probe's workgroups leave records of their own
// (no atomics) and clear the counters the later launches add to; the key launch reduces the records (a memset in front of
// the step was two fill launches, ~10 us)
struct MsdState {
  unsigned long long nkmin;       // ~(smallest key of the raw, FINITE scores)   } written by workgroup 0 of the key launch
  unsigned long long kmax;        // their largest key                           }
  ProbeRec recs[kProbeBlocks];    // the probe's workgroups
  unsigned lin[kMsdBuckets];      // round 5: scores per bin of the raw range (msd_lin_hist_kernel), the sketch the buckets are equalised with
  unsigned hist[kMsdBuckets];
  unsigned start[kMsdBuckets + 1];
  unsigned done_blocks;           // curve_terms: the last block to finish writes the three scalars
  unsigned done_group[8 * 32];    // ... counted per residue of the workgroup id mod 8 first (one 128-byte line each)
  // the scatter's slot cursor of a bucket (low half) and - round 6, for the fused sort + curve launch - the bucket's positives
  // (in-distribution scores; high half): ONE returning atomic per non-empty (tile, bucket) serves both
  unsigned long long cursor64[kMsdBuckets];
  // ---- not cleared by the probe (written before read) ----
  unsigned long long split[kMsdBuckets];  // round 6: splitter keys of the equalised buckets (msd_split_kernel)
};

template <typename T>
__device__ __forceinline__ uint64_t score_key(T v, bool squash) {
  if (squash) v = (T)1 / ((T)1 + exp(-v));  // torch.sigmoid in the dtype of the scores
  return sortable_desc((double)v + 0.0);     // (-0.0 and +0.0 share a key: one tie group, as torchmetrics)
}

template <typename T>
__global__ __launch_bounds__(256) void msd_probe_kernel(const T* __restrict__ ind, int64_t n_ind, const T* __restrict__ ood,
                                                        int64_t n_ood, MsdState* st) {
  __shared__ unsigned long long smin[4], smax[4];
  {  // the words later launches add to: everything from `lin` up to the splitters
    constexpr unsigned kClearWords = (unsigned)((offsetof(MsdState, split) - offsetof(MsdState, lin)) / 4);
    unsigned* z = st->lin;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < kClearWords; i += gridDim.x * 256) z[i] = 0u;
  }
  bool bad = false;
  unsigned long long mn = ~0ull, mx = 0ull;  // key range of the raw, finite scores (no exp here: see MsdRange)
  const int64_t n = n_ind + n_ood;
  // (four loads in flight per thread and trip: one at a time the 32 trips of a thread at 2 M scores each waited for its load)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * stride) {
    T v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t ij = i + j * stride;
      v[j] = (ij < n) ? ((ij < n_ind) ? ind[ij] : ood[ij - n_ind]) : (T)0.5;  // (a filler inside [0, 1]; its key is skipped)
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bad = bad || !(v[j] >= (T)0 && v[j] <= (T)1);
      if (i + j * stride < n && v[j] - v[j] == (T)0) {  // finite
        const unsigned long long k = score_key<T>(v[j], false);
        mn = k < mn ? k : mn;
        mx = k > mx ? k : mx;
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = ((unsigned long long)__shfl_xor((unsigned)(mn >> 32), o, 64) << 32) | __shfl_xor((unsigned)mn, o, 64);
    const unsigned long long b = ((unsigned long long)__shfl_xor((unsigned)(mx >> 32), o, 64) << 32) | __shfl_xor((unsigned)mx, o, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if (lane == 0) { smin[wave] = mn; smax[wave] = mx; }
  const bool any_bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    unsigned long long a = smin[0], b = smax[0];
    for (int w = 1; w < 4; ++w) { a = smin[w] < a ? smin[w] : a; b = smax[w] > b ? smax[w] : b; }
    st->recs[blockIdx.x] = ProbeRec{~a, b, any_bad ? 1u : 0u, 0u};
  }
}

// the probe's records -> (any score outside [0, 1], key range); every workgroup of the key launch for itself, workgroup 0
// also for the launches behind it
struct ProbeResult { unsigned long long nkmin, kmax; bool squash; };
__device__ __forceinline__ ProbeResult msd_reduce_probe(MsdState* st, unsigned nrec, unsigned* __restrict__ any_outside) {
  __shared__ unsigned long long rmin[4], rmax[4];
  __shared__ unsigned rbad[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long a = 0ull, b = 0ull;
  unsigned bad = 0u;
  for (unsigned r = tid; r < nrec; r += 256) {
    const ProbeRec q = st->recs[r];
    a = q.nkmin > a ? q.nkmin : a;
    b = q.kmax > b ? q.kmax : b;
    bad |= q.bad;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a2 = ((unsigned long long)__shfl_xor((unsigned)(a >> 32), o, 64) << 32) | __shfl_xor((unsigned)a, o, 64);
    const unsigned long long b2 = ((unsigned long long)__shfl_xor((unsigned)(b >> 32), o, 64) << 32) | __shfl_xor((unsigned)b, o, 64);
    a = a2 > a ? a2 : a;
    b = b2 > b ? b2 : b;
    bad |= __shfl_xor(bad, o, 64);
  }
  if (lane == 0) { rmin[wave] = a; rmax[wave] = b; rbad[wave] = bad; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) { a = rmin[w] > a ? rmin[w] : a; b = rmax[w] > b ? rmax[w] : b; bad |= rbad[w]; }
    rmin[0] = a; rmax[0] = b; rbad[0] = bad;
    if (blockIdx.x == 0) { st->nkmin = a; st->kmax = b; *any_outside = bad; }
  }
  __syncthreads();
  return ProbeResult{rmin[0], rmax[0], rbad[0] != 0u};
}

// Bucket of a key: LINEAR in the (squashed) score over the range of the finite scores - the bit patterns of doubles are not
// spread evenly (the 12 bits below the keys' common prefix are mostly exponent: one bucket per binade, half of a normal
// score set in one of them).  b = (int)((largest - v) * scale) is monotone in v (a subtraction, a product with a positive
// factor and a truncation are), so the buckets follow the key order; the keys descend with the scores and the sigmoid is
// monotone, so the range of the squashed scores is [sigmoid(smallest), sigmoid(largest)] - two exps per workgroup instead of
// one per score in the probe.  +inf and NaN keys at the low end of the key space go to the first bucket, -inf and NaN keys
// at the high end to the last (the sort inside a bucket uses the whole key).
__device__ __forceinline__ double unkey(uint64_t k) {  // inverse of sortable_desc
  uint64_t b = ~k;
  b = (b >> 63) ? (b & 0x7fffffffffffffffull) : ~b;
  return __longlong_as_double((long long)b);
}
// Round 5: the buckets are linear in the RAW score, not in the squashed one.  The sigmoid turns the smooth score sets the
// postprocessors actually produce - LaREM's -chi2(256), LaRED's log-densities of -300 ... -2000, energies around 9 - into sets
// crowded against 0 or 1: value-linear buckets of sigmoid(score) then put nearly every key into ONE bucket, which takes the
// bounded-but-slow workgroup path (2 M LaREM scores: 81.7 ms against 0.22 ms for N(0, 1) scores; 20 000: 0.75 ms against
// 0.08).  The sigmoid is monotone, so a bucket index that rises as the raw score falls rises with the key as well: every key of
// bucket i still sorts before every key of bucket j > i (equal keys - saturated sigmoids - may straddle a boundary, which the
// order does not mind).  The key launch knows the raw score and leaves the bucket of every key in a 16-bit array for the
// scatter.  Non-finite scores take the end of the key space their KEY lies at.
struct MsdRange { double hi, scale; };
__device__ __forceinline__ MsdRange msd_range_of(unsigned long long nkmin, unsigned long long kmax) {
  MsdRange r{0.0, 0.0};
  const unsigned long long lo_k = ~nkmin, hi_k = kmax;
  if (lo_k > hi_k) return r;  // (no finite score)
  const double v_hi = unkey(lo_k), v_lo = unkey(hi_k);  // largest / smallest finite raw score
  r.hi = v_hi;
  const double span = v_hi - v_lo;
  r.scale = (span > 0.0 && span < __builtin_inf()) ? (double)kMsdBuckets / span : 0.0;
  return r;
}
// position of a finite raw score on the bin axis: [0, kMsdBuckets), rising as the score falls
__device__ __forceinline__ double msd_lin_pos(double raw, const MsdRange& r) {
  const double x = (r.hi - raw) * r.scale;
  constexpr double kTop = (double)kMsdBuckets * (1.0 - 0x1p-52);  // the largest position inside the last bin
  return (x >= kTop) ? kTop : (x > 0.0 ? x : 0.0);
}
// Equalised buckets (round 5; by splitter keys since round 6).  Raw-linear bins of a bell-shaped score set hold 0 ... 4 x the mean, and
// a bin beyond the wave sort's 1 024 keys takes the slow workgroup path (2 M N(0, 1) scores: 0.36 ms).  With the bin counts known
// (`lin`, one histogram launch over the scores), the cumulative count at a score - linear inside its bin - is its approximate RANK;
// bucket e starts at the raw score whose rank is e / 4 096 of the total (msd_split_kernel inverts that, msd_keys_split_kernel compares
// KEYS with the keys of those scores).
// the sketch: finite raw scores per linear bin - of every kSketchStride-th score once the set is large (a quarter of 2 M
// scores still puts ~120 into an average bin; the launch is a pass over the scores with LDS atomics, 19 us -> ~7 at 2 M)
constexpr int kSketchStride = 4;
constexpr int64_t kSketchAll = 1 << 17;  // up to here every score is counted
// Splitter e (1 .. nb - 1) of the bucket function: the key of the raw score at which the bucket index steps from e - 1 to e -
// squashed like every other key - made non-decreasing by a running maximum over e (so that "number of splitters <= key" is a
// monotone function of the key whatever the rounding of the lines below).  cum / cnt: the sketch (exclusive scan of the bin
// counts, bin counts) or null for raw-linear buckets; out[0] = 0.  One workgroup of 256 threads; nb a power of two <= kMsdBuckets.
template <typename T>
__device__ __forceinline__ void msd_make_splitters(const MsdRange& rg, bool squash, const unsigned* __restrict__ cum,
                                                   const unsigned* __restrict__ cnt, double buckets_per_key, int nb,
                                                   unsigned long long* __restrict__ out) {
  __shared__ unsigned long long wmax_s[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (nb + 255) / 256;  // consecutive entries per thread (<= 16)
  uint64_t run = 0ull, mine[kMsdBuckets / 256];
#pragma unroll
  for (int j0 = 0; j0 < kMsdBuckets / 256; j0 += 4) {
    // four entries at a time: their binary searches over the sketch are four independent chains of LDS reads (one chain per
    // entry left the sketch launch's last workgroup 22 us behind the others at 2 M scores)
    double xs[4];
    int bs[4];
    double ts[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid * per + j0 + q;
      xs[q] = (double)e;  // raw-linear bins: the boundary is the bin edge
      ts[q] = cum ? (double)e / buckets_per_key : 0.0;  // the rank at which bucket e starts
      bs[q] = 0;
    }
    if (cum) {  // (uniform)
#pragma unroll 1
      for (int step = kMsdBuckets / 2; step > 0; step >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if ((double)cum[bs[q] + step] <= ts[q]) bs[q] += step;  // largest bin with cum[b] <= t
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double c = (double)cnt[bs[q]];
        double frac = c > 0.0 ? (ts[q] - (double)cum[bs[q]]) / c : 0.0;
        frac = frac < 0.0 ? 0.0 : (frac > 1.0 ? 1.0 : frac);
        xs[q] = (double)bs[q] + frac;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + q, e = tid * per + j;
      uint64_t sk = 0ull;
      if (j < per && e >= 1 && e < nb)  // no finite range (rg.scale == 0): no key reaches a splitter - everything is bucket 0
        sk = (rg.scale > 0.0) ? score_key<T>((T)(rg.hi - xs[q] / rg.scale), squash) : ~0ull;
      run = sk > run ? sk : run;
      mine[j] = run;
    }
  }
  uint64_t x = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(x >> 32), o, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)x, o, 64);
    if (lane >= o) x = y > x ? y : x;
  }
  uint64_t before = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(x >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)x, 1, 64);
  if (lane == 0) before = 0ull;
  __syncthreads();  // (wmax_s may still be read by a previous caller's threads)
  if (lane == 63) wmax_s[wave] = x;
  __syncthreads();
  for (int w = 0; w < wave; ++w) before = wmax_s[w] > before ? wmax_s[w] : before;
#pragma unroll
  for (int j = 0; j < kMsdBuckets / 256; ++j) {
    const int e = tid * per + j;
    if (j < per && e < nb) out[e] = (e == 0) ? 0ull : (mine[j] > before ? mine[j] : before);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void msd_lin_hist_kernel(const T* __restrict__ ind, int64_t n_ind, const T* __restrict__ ood,
                                                           int64_t n_ood, unsigned* __restrict__ any_outside, MsdState* st,
                                                           unsigned n_probe_recs) {
  __shared__ unsigned lh[kMsdBuckets];
  const ProbeResult pr = msd_reduce_probe(st, n_probe_recs, any_outside);
  const MsdRange rg = msd_range_of(pr.nkmin, pr.kmax);
  for (int b = threadIdx.x; b < kMsdBuckets; b += 256) lh[b] = 0u;
  __syncthreads();
  const int64_t n = n_ind + n_ood;
  const int64_t step = (n > kSketchAll) ? kSketchStride : 1;  // (the grid is sized for the sampled count)
  const int64_t t0 = (int64_t)blockIdx.x * kScatTileHist;
#pragma unroll 4
  for (int c = 0; c < kScatTileHist / 256; ++c) {
    const int64_t j = t0 + c * 256 + threadIdx.x;
    const int64_t i = (j >> 8) * step * 256 + (j & 255);  // 256 consecutive scores of every `step`-th run of 256: whole lines, a quarter of them
    if (i < n) {
      const double v = (double)((i < n_ind) ? ind[i] : ood[i - n_ind]) + 0.0;
      if (v - v == 0.0) atomicAdd(&lh[(int)msd_lin_pos(v, rg)], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kMsdBuckets; b += 256) {
    const unsigned c = lh[b];
    if (c) atomicAdd(&st->lin[b], c);
  }
}

// round 6: the finished sketch -> the splitter keys of the equalised buckets, 256 per workgroup (16 workgroups), one per thread.
// (As the tail of the sketch launch - its last workgroup to arrive derived all 4 095 - the one workgroup worked 17 us while the
// chip waited; as a prologue of every key-launch workgroup, 60 us of that launch.)  The running maximum that makes them
// non-decreasing is taken by the key launch as it loads them.
template <typename T>
__global__ __launch_bounds__(256) void msd_split_kernel(unsigned* __restrict__ any_outside, MsdState* st, unsigned n_probe_recs) {
  __shared__ unsigned cum_s[kMsdBuckets], cnt_s[kMsdBuckets];
  __shared__ unsigned wsum_s[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const ProbeResult pr = msd_reduce_probe(st, n_probe_recs, any_outside);
  const MsdRange rg = msd_range_of(pr.nkmin, pr.kmax);
  constexpr int PER = kMsdBuckets / 256;
  unsigned v[PER], tot = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) { v[j] = st->lin[tid * PER + j]; tot += v[j]; }
  unsigned x = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum_s[wave] = x;
  __syncthreads();
  unsigned off = x - tot;
  for (int w = 0; w < wave; ++w) off += wsum_s[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    cum_s[tid * PER + j] = off;
    cnt_s[tid * PER + j] = v[j];
    off += v[j];
  }
  const unsigned finite_total = wsum_s[0] + wsum_s[1] + wsum_s[2] + wsum_s[3];
  __syncthreads();
  const int e = (int)blockIdx.x * 256 + tid;
  uint64_t sk = 0ull;
  if (e >= 1 && e < kMsdBuckets) {
    sk = ~0ull;  // no finite range: no key reaches a splitter - everything is bucket 0
    if (rg.scale > 0.0) {
      double xs = (double)e;
      if (finite_total) {
        const double t = (double)e * ((double)finite_total / (double)kMsdBuckets);  // the rank at which bucket e starts
        int b = 0;  // largest bin with cum[b] <= t
#pragma unroll 1
        for (int step = kMsdBuckets / 2; step > 0; step >>= 1)
          if ((double)cum_s[b + step] <= t) b += step;
        const double c = (double)cnt_s[b];
        double frac = c > 0.0 ? (t - (double)cum_s[b]) / c : 0.0;
        frac = frac < 0.0 ? 0.0 : (frac > 1.0 ? 1.0 : frac);
        xs = (double)b + frac;
      }
      sk = score_key<T>((T)(rg.hi - xs / rg.scale), pr.squash);
    }
  }
  if (e < kMsdBuckets) st->split[e] = sk;
}

constexpr int kScatItems = METRICS_SCAT_ITEMS, kScatTile = 256 * kScatItems;  // 8 192 keys per workgroup: ~2 per (tile, bucket)
template <typename T, int ITEMS = kScatItems>  // small sets: 8 keys per thread (20 000 scores: 10 workgroups instead of 3)
__global__ __launch_bounds__(256) void msd_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint8_t* __restrict__ lab_in,
                                                          uint64_t* __restrict__ keys_out, uint8_t* __restrict__ lab_out, int64_t n,
                                                          const uint16_t* __restrict__ bucket_of, MsdState* st) {
  __shared__ unsigned cnt[kMsdBuckets];   // keys of this tile per bucket, then the tile's first slot in the bucket
  __shared__ unsigned first[kMsdBuckets]; // keys in the buckets before b
  __shared__ unsigned lpos[kMsdBuckets];  // positives of this tile per bucket (round 6: the fused sort + curve launch starts from them)
  __shared__ unsigned wsum[4];
  for (int b = threadIdx.x; b < kMsdBuckets; b += 256) { cnt[b] = 0u; lpos[b] = 0u; }
  {
    // the exclusive scan of the bucket counts, by every workgroup for itself (16 KB of counts from L2) instead of by one
    // workgroup in a launch of its own in front of this one; workgroup 0 leaves it in st->start for the sort
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = kMsdBuckets / 256;
    unsigned v[PER], tot = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) { v[j] = st->hist[tid * PER + j]; tot += v[j]; }
    unsigned x = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned off = x - tot;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      first[tid * PER + j] = off;
      if (blockIdx.x == 0) st->start[tid * PER + j] = off;
      off += v[j];
    }
    if (blockIdx.x == 0 && tid == 255) st->start[kMsdBuckets] = off;
  }
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * (256 * ITEMS);
  uint64_t key[ITEMS];
  unsigned slot[ITEMS];  // bucket << 16 | rank of the key among the tile's keys of that bucket (< 16 384)
  unsigned labbits = 0u;      // label of item c in bit c
  static_assert(kScatTile <= 65536 && kMsdBits <= 16, "bucket and rank share a word; a bucket index fits the 16-bit array");
  static_assert(ITEMS <= 32, "one label bit per item");
#pragma unroll
  for (int c = 0; c < ITEMS; ++c) {
    const int64_t i = t0 + c * 256 + threadIdx.x;
    key[c] = (i < n) ? keys_in[i] : 0ull;
    const unsigned b = (i < n) ? (unsigned)bucket_of[i] : 0u;
    const unsigned lab = (i < n) ? (unsigned)lab_in[i] : 0u;
    labbits |= (lab & 1u) << c;
    slot[c] = (b << 16) | ((i < n) ? atomicAdd(&cnt[b], 1u) : 0u);
    if (lab) atomicAdd(&lpos[b], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kMsdBuckets; b += 256) {
    const unsigned c = cnt[b];
    // ONE global atomic per non-empty (tile, bucket): the tile's slots in the bucket (low half) + its positives (high half)
    cnt[b] = c ? first[b] + (unsigned)atomicAdd(&st->cursor64[b], (unsigned long long)c | ((unsigned long long)lpos[b] << 32)) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < ITEMS; ++c) {
    const int64_t i = t0 + c * 256 + threadIdx.x;
    if (i < n) {
      const unsigned pos = cnt[slot[c] >> 16] + (slot[c] & 0xffffu);
      keys_out[pos] = key[c];
      lab_out[pos] = (uint8_t)((labbits >> c) & 1u);
    }
  }
}

// Buckets 4 g .. 4 g + 3 belong to workgroup g.  First every wave sorts "its" bucket if it holds at most kMsdWaveCap keys: a
// bitonic network in the wave's REGISTERS (wave_sort_in_registers; 61 us per 2 M keys in 4 096 buckets of ~490 on average -
// value-linear buckets of a normal score set hold 0 ... 1 900).  Before it: the same network in the wave's LDS slice, wave-
// synchronous, 97 us; a workgroup per bucket with 45 barrier-separated rounds 173 us; a wave-level stable 8-bit radix sort over
// the bytes in which a bucket's keys differ (two LDS copies of the bucket: half the occupancy) 168 us.  Then the workgroup
// together takes each of its buckets that is larger.
constexpr int kMsdWaveCap = 1024;
#ifndef METRICS_REG_SORT
#define METRICS_REG_SORT 1
#endif
// A bucket of up to 64 * PER keys sorted by ONE wave in registers: element e = lane * PER + r, a bitonic network whose
// exchanges at distance j < PER are compare-and-selects between two of the lane's own registers and whose exchanges at
// distance j >= PER are two shuffles + a select per element with the lane at distance j / PER (21 of the 45 stages at
// PER = 8).  The LDS form - every stage two reads, a compare and up to two writes per pair, for keys and for labels, each
// stage waiting for the last - took 97 us per 2 M keys (4 096 buckets of ~490).  Equal keys never swap (their order does
// not matter to the curve: a run is one point), so both lanes of a pair decide alike.
template <int PER>
__device__ __forceinline__ void wave_sort_network(uint64_t (&k)[PER], unsigned (&l)[PER], int lane) {
  constexpr int LOG_PER = (PER == 1) ? 0 : (PER == 2) ? 1 : (PER == 4) ? 2 : (PER == 8) ? 3 : 4, LOG_M = LOG_PER + 6;
  static_assert((1 << LOG_PER) == PER, "PER is a power of two up to 16");
#pragma unroll
  for (int lk = 1; lk <= LOG_M; ++lk) {
#pragma unroll
    for (int lj = lk - 1; lj >= 0; --lj) {
      const int k2 = 1 << lk, j = 1 << lj;
      if (lj < LOG_PER) {  // both elements in this lane
#pragma unroll
        for (int r = 0; r < PER; ++r) {
          if ((r & j) != 0) continue;
          // direction: bit k2 of the element index lane * PER + r
          const bool up = (k2 < PER) ? ((r & k2) == 0) : ((((unsigned)lane * PER) & (unsigned)k2) == 0u);
          const uint64_t a = k[r], b = k[r | j];
          const unsigned la = l[r], lb = l[r | j];
          const bool swap = up ? (a > b) : (a < b);
          k[r] = swap ? b : a; k[r | j] = swap ? a : b;
          l[r] = swap ? lb : la; l[r | j] = swap ? la : lb;
        }
      } else {  // the partner element sits in the lane at distance j / PER, same register
        const int dl = j >> LOG_PER;
        const bool low = (lane & dl) == 0, up = (((unsigned)lane * PER) & (unsigned)k2) == 0u;
        const bool want_small = (low == up);
#pragma unroll
        for (int r = 0; r < PER; ++r) {
          const uint64_t a = k[r];
          const uint64_t o = ((uint64_t)(unsigned)__shfl_xor((int)(unsigned)(a >> 32), dl, 64) << 32) |
                             (uint64_t)(unsigned)__shfl_xor((int)(unsigned)a, dl, 64);
          const unsigned ol = (unsigned)__shfl_xor((int)l[r], dl, 64);
          const bool take = want_small ? (a > o) : (a < o);
          k[r] = take ? o : a;
          l[r] = take ? ol : l[r];
        }
      }
    }
  }
}
// the bucket's keys and labels into registers; (which unsorted key starts in which register does not matter: consecutive lanes
// read consecutive keys).  Padding: the largest key, behind every real one.
template <int PER>
__device__ __forceinline__ void wave_sort_load(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ labs, unsigned lo,
                                               unsigned nb, int lane, uint64_t (&k)[PER], unsigned (&l)[PER]) {
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const unsigned e = (unsigned)r * 64u + (unsigned)lane;
    k[r] = (e < nb) ? keys[lo + e] : ~0ull;
    l[r] = (e < nb) ? labs[lo + e] : 0u;
  }
}
template <int PER>
__device__ __forceinline__ void wave_sort_in_registers(uint64_t* __restrict__ keys, uint8_t* __restrict__ labs, unsigned lo,
                                                       unsigned nb, int lane) {
  uint64_t k[PER];
  unsigned l[PER];
  wave_sort_load<PER>(keys, labs, lo, nb, lane, k, l);
  wave_sort_network<PER>(k, l, lane);
  // (stores straight from the registers, PER consecutive keys per lane; through the wave's LDS slice for consecutive lanes to
  // write consecutive keys: 183 registers + 42 KB of LDS per workgroup, 211 -> 233 us per 2 M scores)
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const unsigned e = (unsigned)lane * PER + r;
    if (e < nb) { keys[lo + e] = k[r]; labs[lo + e] = (uint8_t)l[r]; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void msd_bucket_sort_kernel(uint64_t* __restrict__ keys, uint8_t* __restrict__ labs,
                                                              uint64_t* __restrict__ alt_keys, uint8_t* __restrict__ alt_labs,
                                                              const unsigned* __restrict__ any_outside, const MsdState* st) {
#if !METRICS_REG_SORT
  __shared__ uint64_t sk_all[4][kMsdWaveCap];
  __shared__ uint8_t sl_all[4][kMsdWaveCap];
#endif
  __shared__ unsigned base[256];
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned dsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  (void)any_outside;
  {
    const int b = 4 * blockIdx.x + wave;
    const unsigned lo = st->start[b], nb = st->start[b + 1] - lo;
#if METRICS_REG_SORT
    if (nb >= 2u && nb <= (unsigned)kMsdWaveCap) {  // (wave-uniform)
      if (nb <= 64u) wave_sort_in_registers<1>(keys, labs, lo, nb, lane);
      else if (nb <= 128u) wave_sort_in_registers<2>(keys, labs, lo, nb, lane);
      else if (nb <= 256u) wave_sort_in_registers<4>(keys, labs, lo, nb, lane);
      else if (nb <= 512u) wave_sort_in_registers<8>(keys, labs, lo, nb, lane);
      else wave_sort_in_registers<16>(keys, labs, lo, nb, lane);
    }
#else  // the LDS form of the network (round 4's first cut), kept for comparison
    if (nb >= 2u && nb <= (unsigned)kMsdWaveCap) {  // (wave-uniform)
      uint64_t* sk = sk_all[wave];
      uint8_t* sl = sl_all[wave];
      unsigned m = 2u;
      while (m < nb) m <<= 1;  // network size: the next power of two, padded with the largest key
      for (unsigned i = lane; i < m; i += 64) {
        sk[i] = (i < nb) ? keys[lo + i] : ~0ull;
        sl[i] = (i < nb) ? labs[lo + i] : 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (unsigned k2 = 2u; k2 <= m; k2 <<= 1) {
        for (unsigned j = k2 >> 1; j > 0u; j >>= 1) {
          for (unsigned t = lane; t < (m >> 1); t += 64) {  // pair t: i = the index with bit j clear
            const unsigned i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), p = i | j;
            const bool up = (i & k2) == 0u;
            const uint64_t a = sk[i], c = sk[p];
            if ((a > c) == up) {
              sk[i] = c; sk[p] = a;
              const uint8_t la = sl[i]; sl[i] = sl[p]; sl[p] = la;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
      for (unsigned i = lane; i < nb; i += 64) { keys[lo + i] = sk[i]; labs[lo + i] = sl[i]; }
    }
#endif
  }
  for (int b = 4 * blockIdx.x; b < 4 * (int)blockIdx.x + 4; ++b) {
    const unsigned lo = st->start[b], nb = st->start[b + 1] - lo;
    if (nb <= (unsigned)kMsdWaveCap) continue;  // (uniform)
    // a bucket that does not fit a wave's LDS slice: stable 8-bit radix passes over the bits below the bucket digit, by this
    // workgroup alone, between the bucket's range in `keys` and the same range of the other buffer
    // the bits in which the bucket's keys differ at all (a bucket of equal keys - saturated sigmoids, a constant score set -
    // needs no pass): OR of key ^ first key over the bucket
    __shared__ unsigned long long diff_s;
    __syncthreads();
    if (tid == 0) diff_s = 0ull;
    __syncthreads();
    {
      const uint64_t k0 = keys[lo];
      unsigned long long d = 0ull;
      for (unsigned i = tid; i < nb; i += 256) d |= keys[lo + i] ^ k0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        d |= ((unsigned long long)__shfl_xor((unsigned)(d >> 32), o, 64) << 32) | __shfl_xor((unsigned)d, o, 64);
      if (lane == 0 && d) atomicOr(&diff_s, d);
    }
    __syncthreads();
    const unsigned long long diff = diff_s;
    const int low_bits = diff ? 64 - __builtin_clzll(diff) : 0;
    const int first_bit = diff ? (__builtin_ctzll(diff) & ~7) : 0;
    uint64_t* src_k = keys + lo;
    uint8_t* src_l = labs + lo;
    uint64_t* dst_k = alt_keys + lo;
    uint8_t* dst_l = alt_labs + lo;
    int moved = 0;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int shift = first_bit; shift < low_bits; shift += 8) {
      __syncthreads();
      base[tid] = 0u;
      __syncthreads();
      for (unsigned i = tid; i < nb; i += 256) atomicAdd(&base[(unsigned)(src_k[i] >> shift) & 255u], 1u);
      __syncthreads();
      {
        const unsigned v = base[tid];
        unsigned x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned y = __shfl_up(x, o, 64);
          if (lane >= o) x += y;
        }
        if (lane == 63) dsum[wave] = x;
        __syncthreads();
        unsigned woff = 0u;
        for (int w = 0; w < wave; ++w) woff += dsum[w];
        base[tid] = woff + x - v;
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
      __syncthreads();
      for (unsigned i0 = 0; i0 < nb; i0 += 256) {  // chunks in order: the pass is stable
        const unsigned i = i0 + tid;
        const bool valid = i < nb;
        const uint64_t key = valid ? src_k[i] : 0ull;
        const unsigned d = (unsigned)(key >> shift) & 255u;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
          const uint64_t bal = __ballot((d >> bb) & 1u);
          same &= ((d >> bb) & 1u) ? bal : ~bal;
        }
        const unsigned rank_in_wave = (unsigned)__popcll(same & lt_mask);
        if (valid && rank_in_wave == 0u) wcnt[wave][d] = (unsigned)__popcll(same);
        __syncthreads();
        if (valid) {
          unsigned off = base[d] + rank_in_wave;
          for (int w = 0; w < wave; ++w) off += wcnt[w][d];
          dst_k[off] = key;
          dst_l[off] = src_l[i];
        }
        __syncthreads();
        base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
        __syncthreads();
      }
      __threadfence_block();
      uint64_t* tk = src_k; src_k = dst_k; dst_k = tk;
      uint8_t* tl = src_l; src_l = dst_l; dst_l = tl;
      moved ^= 1;
    }
    __syncthreads();
    if (moved)  // an odd number of passes left the bucket in the other buffer
      for (unsigned i = tid; i < nb; i += 256) { keys[lo + i] = src_k[i]; labs[lo + i] = src_l[i]; }
    __syncthreads();
  }
}

// curve terms + (the last block to finish) the three scalars: one launch instead of two.  Every workgroup leaves its sums in
// a record of its own and counts itself done with ONE atomic (three more on one record - 2 048 atomics on one line at ~15-45 ns
// each - were most of the launch); the last one adds the records in index order, so the three scalars are the same bits from
// run to run.
#ifndef METRICS_CURVE_BLOCKS
#define METRICS_CURVE_BLOCKS 512
#endif
constexpr unsigned kCurveBlocks = METRICS_CURVE_BLOCKS;
__global__ __launch_bounds__(256) void curve_terms_finalize_kernel(const uint64_t* __restrict__ keys, int64_t n,
                                                                   const unsigned* __restrict__ tps, const int* __restrict__ prev_end,
                                                                   MetricsAccum* __restrict__ parts, MsdState* st, double* __restrict__ out) {
  __shared__ int last;
  __shared__ double sroc[4], spr[4];
  __shared__ unsigned long long sidx[4];
  curve_terms_body(keys, n, tps, prev_end, nullptr, blockIdx.x, gridDim.x, parts);
  // (thread 0 wrote the record: it alone fences, below - a release fence is an L2 write-back, and four waves of every
  // workgroup issuing one was ~70 ns per workgroup of this launch)
  // "last one out": atomics on ONE word retire at ~60 ns each - 512 workgroups finishing together spent 30 us in that queue.
  // Eight counters on lines of their own (workgroup id mod 8) take 64 arrivals each side by side; whoever completes a
  // counter arrives at the common one.
  if (threadIdx.x == 0) {
    const unsigned grp = blockIdx.x & 7u, groups = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned in_grp = (gridDim.x + 7u - grp) / 8u;
    int l = 0;
    __threadfence();
    if (atomicAdd(&st->done_group[grp * 32], 1u) + 1u == in_grp) {
      __threadfence();
      l = (atomicAdd(&st->done_blocks, 1u) + 1u == groups) ? 1 : 0;
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double roc = 0.0, pr = 0.0;
  unsigned long long first = ~0ull;
  for (unsigned b = tid; b < gridDim.x; b += 256) {
    roc += __hip_atomic_load(&parts[b].roc_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pr += __hip_atomic_load(&parts[b].pr_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    first = min(first, __hip_atomic_load(&parts[b].fpr95_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    roc += shfl_xor_f64(roc, o);
    pr += shfl_xor_f64(pr, o);
    const unsigned lo = __shfl_xor((unsigned)first, o, 64), hi = __shfl_xor((unsigned)(first >> 32), o, 64);
    first = min(first, ((unsigned long long)hi << 32) | lo);
  }
  if (lane == 0) { sroc[wave] = roc; spr[wave] = pr; sidx[wave] = first; }
  __syncthreads();
  if (tid == 0) {
    MetricsAccum a;
    a.roc_sum = ((sroc[0] + sroc[1]) + sroc[2]) + sroc[3];
    a.pr_sum = ((spr[0] + spr[1]) + spr[2]) + spr[3];
    a.fpr95_idx = min(min(sidx[0], sidx[1]), min(sidx[2], sidx[3]));
    finalize_body(&a, tps, n, out);
  }
}

// ---- round 6: six launches instead of eight (four instead of seven for sets without a sketch, <= 262 144 scores) --------------------------------------------------------
// probe | [sketch | splitters] | keys | scatter | sort + curve + finalise.  Two changes make the back half ONE launch:
//   * the key launch assigns a key to its bucket by comparing the KEY with 4 095 splitter keys (the keys of the raw-score values
//     at which the equalised bucket function steps, squashed like every other key, made non-decreasing by a running maximum):
//     bucket = number of splitters <= key.  Equal keys can no longer lie in two buckets - squashing merges raw scores that the raw
//     bins keep apart: float32 energies, saturated LaREM scores - so the last key of a bucket always ends a run, and the
//     cumulative counts in front of a bucket are the sums of the counts / positives of the buckets before it.  (It also makes the
//     bucket order the key order BY CONSTRUCTION: the round-5 form derived the bucket from the raw score and relied on
//     1 / (1 + exp(-v)) being monotone - ADVICE r5.)  The arithmetic bucket of round 5 stays as the first guess (two LDS reads
//     confirm it); a binary search over the splitters only runs when the guess is off.
//   * the sort launch keeps a bucket's sorted keys in the wave's registers and forms the curve terms there - label prefix, run ends,
//     the previous run end of every run end as a running maximum of (index, positives) - instead of storing the keys for three
//     more passes (tile summary, tile prefix, curve terms: 55 us of kernels + the tps / prev_end arrays at 2 M scores).  Buckets
//     beyond a wave's 1 024 keys: a bucket of equal keys is one run end (its counts are known: nothing is read); anything else
//     is sorted by the workgroup as before and walked in chunks of 1 024 with carries.
// The three scalars come from per-workgroup records added in index order by the last workgroup (same bits from run to run).
struct FusedPart { double roc, pr; unsigned long long first_idx; unsigned first_fp, pad; };

template <typename T, int ITEMS = kItems>  // ITEMS keys per thread: small sets take 4 (20 000 scores: 20 workgroups instead of 5)
__global__ __launch_bounds__(256) void msd_keys_split_kernel(const T* __restrict__ ind, int64_t n_ind, const T* __restrict__ ood,
                                                             int64_t n_ood, unsigned* __restrict__ any_outside, MsdState* st,
                                                             unsigned n_probe_recs, uint64_t* __restrict__ keys,
                                                             uint8_t* __restrict__ labels, uint16_t* __restrict__ bucket_of, int equalise,
                                                             int nbk) {
  __shared__ unsigned long long split[kMsdBuckets];  // split[e], e = 1 .. nbk - 1: bucket of a key = number of splitters <= key
  __shared__ unsigned lh[kMsdBuckets];               // this tile's keys per bucket
  const int tid = threadIdx.x;
  const ProbeResult pr = msd_reduce_probe(st, n_probe_recs, any_outside);
  const bool squash = pr.squash;
  for (int b = tid; b < kMsdBuckets; b += 256) lh[b] = 0u;
  if (equalise) {  // (uniform) the splitters of the equalised buckets (msd_split_kernel), made non-decreasing by a running maximum
    __shared__ unsigned long long wmax_l[4];
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int PER = kMsdBuckets / 256;
    uint64_t mine[PER], run = 0ull;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint64_t v = st->split[tid * PER + j];
      run = v > run ? v : run;
      mine[j] = run;
    }
    uint64_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint64_t y = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(x >> 32), o, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)x, o, 64);
      if (lane >= o) x = y > x ? y : x;
    }
    uint64_t before = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(x >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)x, 1, 64);
    if (lane == 0) before = 0ull;
    if (lane == 63) wmax_l[wave] = x;
    __syncthreads();
    for (int w = 0; w < wave; ++w) before = wmax_l[w] > before ? wmax_l[w] : before;
#pragma unroll
    for (int j = 0; j < PER; ++j) split[tid * PER + j] = mine[j] > before ? mine[j] : before;
  } else {         // raw-linear buckets, nbk of them over the range of the finite scores
    MsdRange rg = msd_range_of(pr.nkmin, pr.kmax);
    rg.scale *= (double)nbk / (double)kMsdBuckets;
    msd_make_splitters<T>(rg, squash, nullptr, nullptr, 0.0, nbk, split);
  }
  __syncthreads();
  const int64_t n = n_ind + n_ood;
  const int64_t t0 = (int64_t)blockIdx.x * (256 * ITEMS);
  constexpr int W = (ITEMS % 8 == 0) ? 8 : 4;  // independent searches per trip: their LDS reads overlap
  static_assert(ITEMS % W == 0, "whole trips");
  for (int c0 = 0; c0 < ITEMS; c0 += W) {
    uint64_t key[W];
    unsigned b[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int64_t i = t0 + (c0 + j) * 256 + tid;
      const T v = (i < n) ? ((i < n_ind) ? ind[i] : ood[i - n_ind]) : (T)0;
      key[j] = score_key<T>(v, squash);
      b[j] = 0u;
    }
    for (unsigned step = (unsigned)nbk >> 1; step > 0u; step >>= 1) {
#pragma unroll
      for (int j = 0; j < W; ++j)
        if (split[b[j] + step] <= key[j]) b[j] += step;
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int64_t i = t0 + (c0 + j) * 256 + tid;
      if (i < n) {
        keys[i] = key[j];
        labels[i] = (i < n_ind) ? 1 : 0;
        bucket_of[i] = (uint16_t)b[j];
        atomicAdd(&lh[b[j]], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = tid; b < nbk; b += 256) {
    const unsigned c = lh[b];
    if (c) atomicAdd(&st->hist[b], c);
  }
}

// one curve term pair (ROC trapezoid, PR trapezoid) of the run end with cumulative (tp, fp), previous run end (tp0, fp0);
// has_prev = false: the first run end pairs with torchmetrics' (precision 1, recall 0) end point.  float32 arithmetic as in
// curve_terms_body.
struct CurveAcc { double roc, pr; unsigned long long first_idx; unsigned first_fp; };
__device__ __forceinline__ void curve_term(CurveAcc& a, unsigned long long idx, unsigned tpu, unsigned tp0u, unsigned fp0u, bool has_prev,
                                           float P, float Nn) {
  const float tp = (float)tpu, fp = (float)((unsigned)(idx + 1ull) - tpu);
  const float tp0 = has_prev ? (float)tp0u : 0.f, fp0 = has_prev ? (float)fp0u : 0.f;
  // x / P and x / Nn as x * (1 / den) corrected by its remainder: the correctly rounded quotient (counts and totals are integers
  // below 2^31, their float32 images finite and non-zero) in 3 instructions instead of the ~10 of an IEEE division, four per term
  const float rP = 1.0f / P, rN = 1.0f / Nn;
  auto quot = [](float a, float den, float r) { const float q = a * r; return fmaf(fmaf(-q, den, a), r, q); };
  const float tpr = quot(tp, P, rP), fpr = quot(fp, Nn, rN), tpr0 = quot(tp0, P, rP), fpr0 = quot(fp0, Nn, rN);
  a.roc += (double)((fpr - fpr0) * (tpr + tpr0));
  if (tpr >= 0.95f && idx < a.first_idx) { a.first_idx = idx; a.first_fp = (unsigned)(idx + 1ull) - tpu; }
  const float prec = tp / (tp + fp), rec = tpr;
  const float prec0 = has_prev ? tp0 / (tp0 + fp0) : 1.0f, rec0 = has_prev ? tpr0 : 0.0f;
  a.pr += (double)((rec0 - rec) * (prec0 + prec));
}

// a bucket of up to 64 * PER keys: sorted in the wave's registers, its curve terms formed there.  base_cnt / base_pos: keys /
// positives in the buckets before it (the bucket before it ended with a run end: see the head of this section).
template <int PER>
__device__ __forceinline__ void wave_bucket_curve(const uint64_t* __restrict__ keys, const uint8_t* __restrict__ labs, unsigned lo,
                                                  unsigned nb, unsigned base_pos, int lane, float P, float Nn, CurveAcc& acc) {
  uint64_t k[PER];
  unsigned l[PER];
  wave_sort_load<PER>(keys, labs, lo, nb, lane, k, l);
  // the network leaves element e = lane * PER + r in register r of lane `lane`; the load put element r * 64 + lane there - any
  // assignment of the unsorted keys to the slots will do
  wave_sort_network<PER>(k, l, lane);
  // inclusive label prefix
  unsigned tp[PER], s = 0u;
#pragma unroll
  for (int r = 0; r < PER; ++r) { s += l[r]; tp[r] = s; }
  unsigned x = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  const unsigned off = x - s;
  // run ends: the key differs from the next one; the bucket's last key always ends a run
  const uint64_t next0 = ((uint64_t)(unsigned)__shfl_down((int)(unsigned)(k[0] >> 32), 1, 64) << 32) | (unsigned)__shfl_down((int)(unsigned)k[0], 1, 64);
  bool end[PER];
  uint64_t pm[PER], run = 0ull;  // (element + 1) << 32 | positives at it, of the last run end before element r of this lane
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    const unsigned e = (unsigned)lane * PER + r;
    const uint64_t kn = (r + 1 < PER) ? k[(r + 1 < PER) ? r + 1 : r] : next0;
    end[r] = e < nb && (e + 1u == nb || k[r] != kn);
    pm[r] = run;
    const uint64_t packed = ((uint64_t)(e + 1u) << 32) | (uint64_t)(off + tp[r]);
    run = end[r] ? packed : run;  // indices rise with r: the latest run end is the maximum
  }
  uint64_t m = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t y = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(m >> 32), o, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)m, o, 64);
    if (lane >= o) m = y > m ? y : m;
  }
  uint64_t carry = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(m >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)m, 1, 64);
  if (lane == 0) carry = 0ull;
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    if (!end[r]) continue;
    const unsigned e = (unsigned)lane * PER + r;
    const uint64_t prev = pm[r] > carry ? pm[r] : carry;
    const unsigned tpu = base_pos + off + tp[r];
    if (prev != 0ull) {
      const unsigned pe = (unsigned)(prev >> 32) - 1u, ptp = base_pos + (unsigned)prev;
      curve_term(acc, (unsigned long long)lo + e, tpu, ptp, (lo + pe + 1u) - ptp, true, P, Nn);
    } else {  // the bucket's first run end: the one before it is the end of the keys in front of the bucket (none: the curve's start)
      curve_term(acc, (unsigned long long)lo + e, tpu, base_pos, lo - base_pos, lo != 0u, P, Nn);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void msd_sort_curve_kernel(uint64_t* __restrict__ keys, uint8_t* __restrict__ labs,
                                                             uint64_t* __restrict__ alt_keys, uint8_t* __restrict__ alt_labs,
                                                             MsdState* st, FusedPart* __restrict__ parts, float P, float Nn,
                                                             double* __restrict__ out) {
  __shared__ unsigned base[256];
  __shared__ unsigned wcnt[4][256];
  __shared__ unsigned dsum[4];
  __shared__ unsigned psum[4];
  __shared__ double sroc[4], spr[4];
  __shared__ unsigned long long sidx[4];
  __shared__ unsigned sfp[4];
  __shared__ unsigned long long diff_s;
  __shared__ unsigned carry_tp_s;
  __shared__ unsigned long long carry_prev_s;
  __shared__ unsigned long long wprev[4];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = 4 * blockIdx.x;
  // positives in the buckets before this workgroup's first one
  unsigned before_pos;
  {
    unsigned q = 0u;
    for (int c = tid; c < b0; c += 256) q += (unsigned)(st->cursor64[c] >> 32);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if (lane == 0) psum[wave] = q;
    __syncthreads();
    before_pos = psum[0] + psum[1] + psum[2] + psum[3];
  }
  CurveAcc acc{0.0, 0.0, ~0ull, 0u};
  {
    const int b = b0 + wave;
    const unsigned lo = st->start[b], nb = st->start[b + 1] - lo;
    unsigned base_pos = before_pos;
    for (int c = b0; c < b; ++c) base_pos += (unsigned)(st->cursor64[c] >> 32);
    if (nb >= 1u && nb <= (unsigned)kMsdWaveCap) {  // (wave-uniform)
      if (nb <= 64u) wave_bucket_curve<1>(keys, labs, lo, nb, base_pos, lane, P, Nn, acc);
      else if (nb <= 128u) wave_bucket_curve<2>(keys, labs, lo, nb, base_pos, lane, P, Nn, acc);
      else if (nb <= 256u) wave_bucket_curve<4>(keys, labs, lo, nb, base_pos, lane, P, Nn, acc);
      else if (nb <= 512u) wave_bucket_curve<8>(keys, labs, lo, nb, base_pos, lane, P, Nn, acc);
      else wave_bucket_curve<16>(keys, labs, lo, nb, base_pos, lane, P, Nn, acc);
    }
  }
  for (int b = b0; b < b0 + 4; ++b) {
    const unsigned lo = st->start[b], nb = st->start[b + 1] - lo;
    if (nb <= (unsigned)kMsdWaveCap) continue;  // (uniform)
    unsigned base_pos = before_pos;
    for (int c = b0; c < b; ++c) base_pos += (unsigned)(st->cursor64[c] >> 32);
    // the bits in which the bucket's keys differ at all: OR of key ^ first key over the bucket
    __syncthreads();
    if (tid == 0) diff_s = 0ull;
    __syncthreads();
    {
      const uint64_t k0 = keys[lo];
      unsigned long long d = 0ull;
      for (unsigned i = tid; i < nb; i += 256) d |= keys[lo + i] ^ k0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
        d |= ((unsigned long long)__shfl_xor((unsigned)(d >> 32), o, 64) << 32) | __shfl_xor((unsigned)d, o, 64);
      if (lane == 0 && d) atomicOr(&diff_s, d);
    }
    __syncthreads();
    const unsigned long long diff = diff_s;
    if (diff == 0ull) {  // one run: its end carries the bucket's counts (nothing else of it is read)
      if (tid == 0)
        curve_term(acc, (unsigned long long)lo + nb - 1u, base_pos + (unsigned)(st->cursor64[b] >> 32), base_pos, lo - base_pos, lo != 0u, P, Nn);
      continue;
    }
    const int low_bits = 64 - __builtin_clzll(diff);
    const int first_bit = __builtin_ctzll(diff) & ~7;
    uint64_t* src_k = keys + lo;
    uint8_t* src_l = labs + lo;
    uint64_t* dst_k = alt_keys + lo;
    uint8_t* dst_l = alt_labs + lo;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int shift = first_bit; shift < low_bits; shift += 8) {  // stable 8-bit radix passes, as msd_bucket_sort_kernel
      __syncthreads();
      base[tid] = 0u;
      __syncthreads();
      for (unsigned i = tid; i < nb; i += 256) atomicAdd(&base[(unsigned)(src_k[i] >> shift) & 255u], 1u);
      __syncthreads();
      {
        const unsigned v = base[tid];
        unsigned x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned y = __shfl_up(x, o, 64);
          if (lane >= o) x += y;
        }
        if (lane == 63) dsum[wave] = x;
        __syncthreads();
        unsigned woff = 0u;
        for (int w = 0; w < wave; ++w) woff += dsum[w];
        base[tid] = woff + x - v;
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
      __syncthreads();
      for (unsigned i0 = 0; i0 < nb; i0 += 256) {  // chunks in order: the pass is stable
        const unsigned i = i0 + tid;
        const bool valid = i < nb;
        const uint64_t key = valid ? src_k[i] : 0ull;
        const unsigned d = (unsigned)(key >> shift) & 255u;
        uint64_t same = __ballot(valid);
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) {
          const uint64_t bal = __ballot((d >> bb) & 1u);
          same &= ((d >> bb) & 1u) ? bal : ~bal;
        }
        const unsigned rank_in_wave = (unsigned)__popcll(same & lt_mask);
        if (valid && rank_in_wave == 0u) wcnt[wave][d] = (unsigned)__popcll(same);
        __syncthreads();
        if (valid) {
          unsigned o2 = base[d] + rank_in_wave;
          for (int w = 0; w < wave; ++w) o2 += wcnt[w][d];
          dst_k[o2] = key;
          dst_l[o2] = src_l[i];
        }
        __syncthreads();
        base[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
        __syncthreads();
      }
      __threadfence_block();
      uint64_t* tk = src_k; src_k = dst_k; dst_k = tk;
      uint8_t* tl = src_l; src_l = dst_l; dst_l = tl;
    }
    __syncthreads();
    // the sorted bucket lies at src_k / src_l: chunks of 1 024 (four consecutive keys per thread) with carries
    if (tid == 0) { carry_tp_s = 0u; carry_prev_s = 0ull; }
    __syncthreads();
    for (unsigned c0 = 0; c0 < nb; c0 += 1024u) {
      const unsigned e0 = c0 + 4u * (unsigned)tid;
      uint64_t kk[5];
      unsigned ll[4];
#pragma unroll
      for (int j = 0; j < 5; ++j) kk[j] = (e0 + j < nb) ? src_k[e0 + j] : ~0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) ll[j] = (e0 + j < nb) ? (unsigned)src_l[e0 + j] : 0u;
      unsigned tpj[4], ssum = 0u;
      bool endj[4];
      uint64_t pmj[4], run = 0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) { ssum += ll[j]; tpj[j] = ssum; }
      unsigned x = ssum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
      }
      if (lane == 63) dsum[wave] = x;
      __syncthreads();
      unsigned off = x - ssum + carry_tp_s;
      for (int w = 0; w < wave; ++w) off += dsum[w];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned e = e0 + j;
        endj[j] = e < nb && (e + 1u == nb || kk[j] != kk[j + 1]);
        pmj[j] = run;
        run = endj[j] ? (((uint64_t)(e + 1u) << 32) | (uint64_t)(off + tpj[j])) : run;
      }
      uint64_t m = run;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(m >> 32), o, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)m, o, 64);
        if (lane >= o) m = y > m ? y : m;
      }
      uint64_t carry = ((uint64_t)(unsigned)__shfl_up((int)(unsigned)(m >> 32), 1, 64) << 32) | (unsigned)__shfl_up((int)(unsigned)m, 1, 64);
      if (lane == 0) carry = 0ull;
      if (lane == 63) wprev[wave] = m;
      __syncthreads();
      carry = carry > carry_prev_s ? carry : carry_prev_s;
      for (int w = 0; w < wave; ++w) carry = wprev[w] > carry ? wprev[w] : carry;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!endj[j]) continue;
        const unsigned e = e0 + j;
        const uint64_t prev = pmj[j] > carry ? pmj[j] : carry;
        const unsigned tpu = base_pos + off + tpj[j];
        if (prev != 0ull) {
          const unsigned pe = (unsigned)(prev >> 32) - 1u, ptp = base_pos + (unsigned)prev;
          curve_term(acc, (unsigned long long)lo + e, tpu, ptp, (lo + pe + 1u) - ptp, true, P, Nn);
        } else {
          curve_term(acc, (unsigned long long)lo + e, tpu, base_pos, lo - base_pos, lo != 0u, P, Nn);
        }
      }
      __syncthreads();
      if (tid == 255) {
        carry_tp_s = off + ssum;
        const uint64_t mm = m > carry ? m : carry;
        carry_prev_s = mm;
      }
      __syncthreads();
    }
  }
  // the workgroup's record
  {
    double roc = acc.roc, prs = acc.pr;
    unsigned long long fi = acc.first_idx;
    unsigned ffp = acc.first_fp;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      roc += shfl_xor_f64(roc, o);
      prs += shfl_xor_f64(prs, o);
      const unsigned long long oi = ((unsigned long long)__shfl_xor((unsigned)(fi >> 32), o, 64) << 32) | __shfl_xor((unsigned)fi, o, 64);
      const unsigned of = __shfl_xor(ffp, o, 64);
      if (oi < fi) { fi = oi; ffp = of; }
    }
    __syncthreads();
    if (lane == 0) { sroc[wave] = roc; spr[wave] = prs; sidx[wave] = fi; sfp[wave] = ffp; }
    __syncthreads();
  }
  if (tid == 0) {
    FusedPart r;
    r.roc = ((sroc[0] + sroc[1]) + sroc[2]) + sroc[3];
    r.pr = ((spr[0] + spr[1]) + spr[2]) + spr[3];
    r.first_idx = sidx[0]; r.first_fp = sfp[0]; r.pad = 0u;
    for (int w = 1; w < 4; ++w)
      if (sidx[w] < r.first_idx) { r.first_idx = sidx[w]; r.first_fp = sfp[w]; }
    parts[blockIdx.x] = r;
    // "last one out", as curve_terms_finalize_kernel: eight counters on lines of their own, then the common one
    const unsigned grp = blockIdx.x & 7u, groups = gridDim.x < 8u ? gridDim.x : 8u;
    const unsigned in_grp = (gridDim.x + 7u - grp) / 8u;
    int l = 0;
    __threadfence();
    if (atomicAdd(&st->done_group[grp * 32], 1u) + 1u == in_grp) {
      __threadfence();
      l = (atomicAdd(&st->done_blocks, 1u) + 1u == groups) ? 1 : 0;
    }
    last = l;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  // the last workgroup: the records in index order (thread t: records t, t + 256, ... - then the fixed tree below)
  double roc = 0.0, prs = 0.0;
  unsigned long long fi = ~0ull;
  unsigned ffp = 0u;
  for (unsigned g = tid; g < gridDim.x; g += 256) {
    roc += __hip_atomic_load(&parts[g].roc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prs += __hip_atomic_load(&parts[g].pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long oi = __hip_atomic_load(&parts[g].first_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned of = __hip_atomic_load(&parts[g].first_fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (oi < fi) { fi = oi; ffp = of; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    roc += shfl_xor_f64(roc, o);
    prs += shfl_xor_f64(prs, o);
    const unsigned long long oi = ((unsigned long long)__shfl_xor((unsigned)(fi >> 32), o, 64) << 32) | __shfl_xor((unsigned)fi, o, 64);
    const unsigned of = __shfl_xor(ffp, o, 64);
    if (oi < fi) { fi = oi; ffp = of; }
  }
  if (lane == 0) { sroc[wave] = roc; spr[wave] = prs; sidx[wave] = fi; sfp[wave] = ffp; }
  __syncthreads();
  if (tid == 0) {
    const double r = ((sroc[0] + sroc[1]) + sroc[2]) + sroc[3], q = ((spr[0] + spr[1]) + spr[2]) + spr[3];
    unsigned long long f = sidx[0];
    unsigned fp = sfp[0];
    for (int w = 1; w < 4; ++w)
      if (sidx[w] < f) { f = sidx[w]; fp = sfp[w]; }
    out[0] = (double)(float)(r * 0.5);                 // trapz(tpr, fpr), reported as float32 like torchmetrics
    out[1] = (f == ~0ull) ? NAN : (double)((float)fp / Nn);
    out[2] = (double)(float)(-(q * 0.5));              // see finalize_body
  }
}

// ---- why the step stays a chain of launches (round 4) ------------------------------------------------------------------
// The whole step was written as ONE persistent launch - every workgroup resident, the phases above behind grid-wide
// barriers, a radix pass as one phase with a decoupled look-back over (aggregate | inclusive) words - and measured
// 7.6 ms per 2 M scores against 0.44 ms for the launches, 0.26 ms per 20 000 against 0.23: on this chip a kernel boundary IS
// the cheap grid barrier.  tools/microbench/grid_barrier.hip: arrive (one agent-scope atomic add) + spin costs 0.8 us for
// 5 workgroups, 11 us for 256, 21 us for 512 (atomics on one word retire at ~45 ns each), and with the two
// __threadfence() a barrier needs to publish plain stores across the eight L2s 2.5 / 40 / 90 us; the look-back of 256
// digit counters by 256 threads is serial over the predecessors, up to G - 1 of them when G tiles start together.  A
// dependent launch costs ~7.5 us.  Not kept; what did shorten the step is fewer passes over the keys (above).

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
  size_t keys_a, keys_b, lab_a, lab_b, tps, prev_end, table, dtot, tile_sum, tile_end, tile_cnt, accum, flag, msd, parts, bucket_of, parts2, total;
  unsigned nblocks;
};

Layout make_layout(int64_t n) {
  Layout L;
  L.nblocks = (unsigned)((n + kTile - 1) / kTile);
  size_t o = 0;
  L.keys_a = o; o += align256((size_t)n * 8);
  L.keys_b = o; o += align256((size_t)n * 8);
  L.lab_a = o; o += align256((size_t)n);
  L.lab_b = o; o += align256((size_t)n);
  L.tps = o; o += align256((size_t)n * 4);
  L.prev_end = o; o += align256((size_t)n * 4);
  L.table = o; o += align256((size_t)256 * L.nblocks * 4);
  L.dtot = o; o += align256(256 * 4);
  L.tile_sum = o; o += align256((size_t)L.nblocks * 4);
  L.tile_end = o; o += align256((size_t)L.nblocks * 4);
  L.tile_cnt = o; o += align256((size_t)L.nblocks * 4);
  L.accum = o; o += align256(sizeof(MetricsAccum));
  L.flag = o; o += 256;
  L.msd = o; o += align256(sizeof(MsdState));  // (accum, flag and msd are contiguous: one memset clears them)
  L.parts = o; o += align256(kCurveBlocks * sizeof(MetricsAccum));  // per-workgroup records of the curve-term launch (written before read)
  L.bucket_of = o; o += align256((size_t)n * 2);                     // bucket of every key (key launch -> scatter)
  L.parts2 = o; o += align256((size_t)(kMsdBuckets / 4) * sizeof(FusedPart));  // per-workgroup records of the fused sort + curve launch
  L.total = o;
  return L;
}

template <typename T>
int ood_metrics(const T* ind, int64_t n_ind, const T* ood, int64_t n_ood, double* out3, unsigned* tps_out, unsigned* fps_out,
                int64_t* n_points, void* workspace, size_t workspace_bytes, runia_stream_t stream) {
  if (n_ind < 1 || n_ood < 1 || n_ind + n_ood >= (1ll << 31)) return RUNIA_E_INVALID;
  if (!ind || !ood || !out3) return RUNIA_E_INVALID;
  if ((tps_out || fps_out || n_points) && !(tps_out && fps_out && n_points)) return RUNIA_E_INVALID;
  const int64_t n = n_ind + n_ood;
  const Layout L = make_layout(n);
  if (!workspace || workspace_bytes < L.total || (((uintptr_t)workspace) & 255) != 0) return RUNIA_E_WORKSPACE;
  char* w = reinterpret_cast<char*>(workspace);
  uint64_t* keys[2] = {reinterpret_cast<uint64_t*>(w + L.keys_a), reinterpret_cast<uint64_t*>(w + L.keys_b)};
  uint8_t* labs[2] = {reinterpret_cast<uint8_t*>(w + L.lab_a), reinterpret_cast<uint8_t*>(w + L.lab_b)};
  unsigned* tps = reinterpret_cast<unsigned*>(w + L.tps);
  int* prev_end = reinterpret_cast<int*>(w + L.prev_end);
  unsigned* table = reinterpret_cast<unsigned*>(w + L.table);
  unsigned* dtot = reinterpret_cast<unsigned*>(w + L.dtot);
  unsigned* tile_sum = reinterpret_cast<unsigned*>(w + L.tile_sum);
  int* tile_end = reinterpret_cast<int*>(w + L.tile_end);
  unsigned* tile_cnt = reinterpret_cast<unsigned*>(w + L.tile_cnt);
  MetricsAccum* acc = reinterpret_cast<MetricsAccum*>(w + L.accum);
  unsigned* flag = reinterpret_cast<unsigned*>(w + L.flag);
  hipStream_t s = as_stream(stream);
  // f32 scores widened to f64 have 29 zero mantissa bits at the bottom: the three lowest digits are the same for every
  // key, so those passes would move nothing
  const int first_pass = (sizeof(T) == 4) ? 3 : 0;
#ifndef METRICS_MSD
#define METRICS_MSD 1
#endif
#ifndef METRICS_FUSED
#define METRICS_FUSED 1
#endif
  if (METRICS_MSD) {
    MsdState* st = reinterpret_cast<MsdState*>(w + L.msd);
    // eight launches, nothing cleared beforehand (MsdState)
    const unsigned sgrid = runia_stream_grid(n, 256);
    const unsigned pgrid = sgrid < kProbeBlocks ? sgrid : kProbeBlocks;
    msd_probe_kernel<T><<<pgrid, 256, 0, s>>>(ind, n_ind, ood, n_ood, st);
    // Small sets (<= 64 keys per bucket on average) skip the sketch: a raw-linear bin of a bell-shaped set holds a few times
    // the mean, far below the wave sort's 1 024, and the launch would cost them 10 us of their 80
    const int equalise = n > (int64_t)64 * kMsdBuckets;
    if (equalise) {
      const int64_t sketch_n = (n > kSketchAll) ? (n + kSketchStride - 1) / kSketchStride : n;
      msd_lin_hist_kernel<T><<<(unsigned)((sketch_n + kScatTileHist - 1) / kScatTileHist), 256, 0, s>>>(ind, n_ind, ood, n_ood, flag, st, pgrid);
      msd_split_kernel<T><<<kMsdBuckets / 256, 256, 0, s>>>(flag, st, pgrid);
    }
    uint16_t* bucket_of = reinterpret_cast<uint16_t*>(w + L.bucket_of);
    // Front half of both forms (round 6): keys + buckets by splitter keys, scatter.  Buckets: 4 096 (equalised by the sketch) for
    // large sets; for small ones raw-linear bins, ~64 keys each - every (tile, bucket) pair costs the scatter an atomic and every
    // four buckets the sort launch a workgroup (20 000 scores in 4 096 buckets: 28 + 28 us for those two launches)
    int nbk = kMsdBuckets;
    if (!equalise) {
      nbk = 64;
      while (nbk < kMsdBuckets && (int64_t)nbk * 64 < n) nbk <<= 1;
    }
    if (n <= 65536) {  // small sets: smaller tiles, more workgroups (the two launches are latency-bound there)
      msd_keys_split_kernel<T, 4><<<(unsigned)((n + 1023) / 1024), 256, 0, s>>>(ind, n_ind, ood, n_ood, flag, st, pgrid, keys[0], labs[0], bucket_of, equalise, nbk);
      msd_scatter_kernel<T, 8><<<(unsigned)((n + 2047) / 2048), 256, 0, s>>>(keys[0], labs[0], keys[1], labs[1], n, bucket_of, st);
    } else if (n <= 524288) {  // mid-sized sets: 16 keys per thread in the scatter (200 000 scores: 85 -> 73 us; 2 M: 230 -> 240)
      msd_keys_split_kernel<T><<<L.nblocks, 256, 0, s>>>(ind, n_ind, ood, n_ood, flag, st, pgrid, keys[0], labs[0], bucket_of, equalise, nbk);
      msd_scatter_kernel<T, 16><<<(unsigned)((n + 4095) / 4096), 256, 0, s>>>(keys[0], labs[0], keys[1], labs[1], n, bucket_of, st);
    } else {
      msd_keys_split_kernel<T><<<L.nblocks, 256, 0, s>>>(ind, n_ind, ood, n_ood, flag, st, pgrid, keys[0], labs[0], bucket_of, equalise, nbk);
      msd_scatter_kernel<T><<<(unsigned)((n + kScatTile - 1) / kScatTile), 256, 0, s>>>(keys[0], labs[0], keys[1], labs[1], n, bucket_of, st);
    }
    if (METRICS_FUSED && !tps_out) {  // the three scalars alone: bucket sort + curve terms + finalise in ONE launch (six launches, four without the sketch)
      msd_sort_curve_kernel<T><<<(unsigned)(nbk / 4), 256, 0, s>>>(keys[1], labs[1], keys[0], labs[0], st,
                                                                   reinterpret_cast<FusedPart*>(w + L.parts2), (float)n_ind, (float)n_ood, out3);
      return runia_check_launch();
    }
    // the curve API: every run's cumulative counts have to land in memory - bucket sort, tile summary, tile prefix, terms
    msd_bucket_sort_kernel<T><<<kMsdBuckets / 4, 256, 0, s>>>(keys[1], labs[1], keys[0], labs[0], flag, st);  // (four buckets per workgroup)
    tile_summary_kernel<<<L.nblocks, 256, 0, s>>>(keys[1], labs[1], n, tile_sum, tile_end, tile_cnt);
    tile_prefix_raw_kernel<<<L.nblocks, 256, 0, s>>>(keys[1], labs[1], n, tile_sum, tile_end, tile_cnt, tps, prev_end, tps_out,
                                                     fps_out, n_points);
    curve_terms_finalize_kernel<<<(sgrid < kCurveBlocks ? sgrid : kCurveBlocks), 256, 0, s>>>(
        keys[1], n, tps, prev_end, reinterpret_cast<MetricsAccum*>(w + L.parts), st, out3);
    return runia_check_launch();
  }
  if (hipMemsetAsync(flag, 0, 4, s) != hipSuccess) return RUNIA_E_LAUNCH;
  if (hipMemsetAsync(acc, 0, 16, s) != hipSuccess) return RUNIA_E_LAUNCH;
  if (hipMemsetAsync(&acc->fpr95_idx, 0xFF, 8, s) != hipSuccess) return RUNIA_E_LAUNCH;
  const unsigned sgrid = runia_stream_grid(n, 256);
  range_check_kernel<T><<<(sgrid < 1024u ? sgrid : 1024u), 256, 0, s>>>(ind, n_ind, ood, n_ood, flag);
  make_keys_kernel<T><<<sgrid, 256, 0, s>>>(ind, n_ind, ood, n_ood, flag, keys[0], labs[0]);
  int cur = 0;
  for (int pass = first_pass; pass < 8; ++pass) {
    const int shift = 8 * pass;
    radix_hist_kernel<<<L.nblocks, 256, 0, s>>>(keys[cur], n, shift, table, L.nblocks);
    radix_row_scan_kernel<<<256, 256, 0, s>>>(table, L.nblocks, dtot);
    radix_scatter_kernel<<<L.nblocks, 256, 0, s>>>(keys[cur], labs[cur], keys[cur ^ 1], labs[cur ^ 1], n, shift, table,
                                                   L.nblocks, dtot);
    cur ^= 1;
  }
  tile_summary_kernel<<<L.nblocks, 256, 0, s>>>(keys[cur], labs[cur], n, tile_sum, tile_end, tile_cnt);
  tile_scan_kernel<<<1, 64, 0, s>>>(tile_sum, tile_end, tile_cnt, L.nblocks);
  tile_prefix_kernel<<<L.nblocks, 256, 0, s>>>(keys[cur], labs[cur], n, tile_sum, tile_end, tile_cnt, tps, prev_end, tps_out,
                                               fps_out, n_points);
  // bounded grid: every workgroup ends with three atomics on ONE record
  curve_terms_kernel<<<(sgrid < 512u ? sgrid : 512u), 256, 0, s>>>(keys[cur], n, tps, prev_end, acc);
  finalize_kernel<<<1, 64, 0, s>>>(acc, tps, n, out3);
  return runia_check_launch();
}

}  // namespace

extern "C" size_t runia_ood_metrics_workspace_bytes(int64_t n_total) {
  if (n_total <= 0) return 0;
  return make_layout(n_total).total;
}

extern "C" int runia_ood_metrics_f64(const double* ind_scores, int64_t n_ind, const double* ood_scores, int64_t n_ood,
                                     double* out3, void* workspace, size_t workspace_bytes, runia_stream_t stream) {
  return ood_metrics<double>(ind_scores, n_ind, ood_scores, n_ood, out3, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                             stream);
}

extern "C" int runia_ood_metrics_f32(const float* ind_scores, int64_t n_ind, const float* ood_scores, int64_t n_ood,
                                     double* out3, void* workspace, size_t workspace_bytes, runia_stream_t stream) {
  return ood_metrics<float>(ind_scores, n_ind, ood_scores, n_ood, out3, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                            stream);
}

// The same three scalars plus torchmetrics' _binary_clf_curve, from which get_auroc_results builds its ROC / PR curves
// (reference evaluation/metrics.py:70-81): tps / fps [n_ind + n_ood] u32 device buffers = cumulative true / false positives
// at the end of every run of equal scores in descending score order, *n_points (device) = number of runs.  Only that many
// entries need to leave the device; the O(N log N) part (sort, scans, compaction) stays on it.
extern "C" int runia_ood_clf_curve_f64(const double* ind_scores, int64_t n_ind, const double* ood_scores, int64_t n_ood,
                                       double* out3, unsigned* tps, unsigned* fps, int64_t* n_points, void* workspace,
                                       size_t workspace_bytes, runia_stream_t stream) {
  if (!tps || !fps || !n_points) return RUNIA_E_INVALID;
  return ood_metrics<double>(ind_scores, n_ind, ood_scores, n_ood, out3, tps, fps, n_points, workspace, workspace_bytes, stream);
}

extern "C" int runia_ood_clf_curve_f32(const float* ind_scores, int64_t n_ind, const float* ood_scores, int64_t n_ood,
                                       double* out3, unsigned* tps, unsigned* fps, int64_t* n_points, void* workspace,
                                       size_t workspace_bytes, runia_stream_t stream) {
  if (!tps || !fps || !n_points) return RUNIA_E_INVALID;
  return ood_metrics<float>(ind_scores, n_ind, ood_scores, n_ood, out3, tps, fps, n_points, workspace, workspace_bytes, stream);
}
