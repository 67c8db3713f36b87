// Row-streaming kernels (HBM-bound): Energy / MSP and the kNN normaliser.
//   a7: scipy.special.logsumexp / softmax-max (reference inference/postprocessors.py:549,606)
//   a8: normalizer (reference inference/funcs.py:105-115)
#include "common.hpp"
#include "entropy_core.hpp"  // sort_asc (register sorting network)

namespace {

// scipy.special.logsumexp: a_max = max(row) (0 when not finite); out = log(sum(exp(a - a_max))) + a_max
// scipy.special.softmax max: exp(a_max - a_max) / sum(exp(a - a_max)) = 1 / sum
// An infinite row maximum is reset to 0 by logsumexp only; softmax keeps it and
// therefore yields NaN (inf - inf), which is reproduced here.
__device__ __forceinline__ void finish_row(float m_raw, float m, float s, float* lse, float* msp,
                                           int64_t row) {
  if (lse) lse[row] = logf(s) + m;
  if (msp) msp[row] = (m_raw == INFINITY || m_raw == -INFINITY) ? NAN : 1.0f / s;
}

// ---- C <= 64: one row per lane, tile staged through LDS with coalesced loads ----
constexpr int kSmallRows = 256;  // rows per workgroup tile

__global__ __launch_bounds__(256) void lse_small_kernel(const float* __restrict__ x, float* lse,
                                                         float* msp, int64_t N, int C) {
  extern __shared__ float tile[];  // kSmallRows * C floats
  const int tid = threadIdx.x;
  for (int64_t r0 = (int64_t)blockIdx.x * kSmallRows; r0 < N; r0 += (int64_t)gridDim.x * kSmallRows) {
    const int rows = (int)((N - r0 < kSmallRows) ? (N - r0) : kSmallRows);
    const int total = rows * C;
    const float* src = x + r0 * C;
    __syncthreads();
    if ((((uintptr_t)src) & 15) == 0) {
      const int n4 = total >> 2;
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* t4 = reinterpret_cast<float4*>(tile);
      for (int i = tid; i < n4; i += 256) t4[i] = s4[i];
      for (int i = (n4 << 2) + tid; i < total; i += 256) tile[i] = src[i];
    } else {
      for (int i = tid; i < total; i += 256) tile[i] = src[i];
    }
    __syncthreads();
    if (tid < rows) {
      const float* row = tile + tid * C;
      // rotate the start column per lane so that lanes of one LDS access group hit
      // different banks when C shares a factor with the bank count
      int j0 = tid % C;
      float m = -INFINITY;
      int j = j0;
      for (int t = 0; t < C; ++t) {
        m = fmaxf(m, row[j]);
        j = (j + 1 == C) ? 0 : j + 1;
      }
      // fmaxf drops NaN, expf(NaN) then poisons the sum as NumPy's max/exp do
      const float m_raw = m;
      if (m == INFINITY || m == -INFINITY) m = 0.f;
      float s = 0.f;
      j = j0;
      for (int t = 0; t < C; ++t) {
        s += expf(row[j] - m);
        j = (j + 1 == C) ? 0 : j + 1;
      }
      finish_row(m_raw, m, s, lse, msp, r0 + tid);
    }
  }
}

// ---- C <= 16 (10-class logits: BASELINE config 1 / the x10 leg of config 3): one row per lane straight from global
// memory, the row in registers.  No LDS, no barrier: the lanes of a wave read 64 consecutive rows (64*C*4 contiguous
// bytes), every load instruction of the row touches the same cache lines, so HBM sees each byte once.
template <int CT>
__global__ __launch_bounds__(256) void lse_tiny_kernel(const float* __restrict__ x, float* lse, float* msp, int64_t N) {
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < N; row += (int64_t)gridDim.x * 256) {
    const float* p = x + row * CT;
    float v[CT];
    if constexpr (CT % 2 == 0) {
      if ((((uintptr_t)x) & 7) == 0) {
#pragma unroll
        for (int j = 0; j < CT / 2; ++j) {
          const float2 t = reinterpret_cast<const float2*>(p)[j];
          v[2 * j] = t.x; v[2 * j + 1] = t.y;
        }
      } else {
#pragma unroll
        for (int j = 0; j < CT; ++j) v[j] = p[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < CT; ++j) v[j] = p[j];
    }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < CT; ++j) m = fmaxf(m, v[j]);
    const float m_raw = m;
    if (m == INFINITY || m == -INFINITY) m = 0.f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CT; ++j) s += expf(v[j] - m);
    finish_row(m_raw, m, s, lse, msp, row);
  }
}

// ---- C > 64: one wave per row; the row stays in registers when it fits ----
template <int NCH>  // float4 chunks per lane; NCH == 0 -> re-read the row (any C)
__global__ __launch_bounds__(64 * kRowWaves) void lse_wave_kernel(const float* __restrict__ x, float* lse,
                                                                  float* msp, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * kRowWaves;
  for (int64_t row = (int64_t)blockIdx.x * kRowWaves + wave; row < N; row += wave_stride) {
    const float* p = x + row * C;
    float m = -INFINITY, s = 0.f, m_raw;
    if constexpr (NCH > 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      const int n4 = (int)(C >> 2);
      float4 v[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        v[c] = (i < n4) ? p4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) m = fmaxf(m, fmaxf(fmaxf(v[c].x, v[c].y), fmaxf(v[c].z, v[c].w)));
      m = wave_max_f32(m);
      m_raw = m;
      if (m == INFINITY || m == -INFINITY) m = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        s += (expf(v[c].x - m) + expf(v[c].y - m)) + (expf(v[c].z - m) + expf(v[c].w - m));
    } else {
      for (int64_t i = lane; i < C; i += 64) m = fmaxf(m, p[i]);
      m = wave_max_f32(m);
      m_raw = m;
      if (m == INFINITY || m == -INFINITY) m = 0.f;
      for (int64_t i = lane; i < C; i += 64) s += expf(p[i] - m);
    }
    s = wave_sum_f32(s);
    if (lane == 0) finish_row(m_raw, m, s, lse, msp, row);
  }
}

// ---- normalizer: y = x / (||x||_2 + 1e-10), f32, one wave per row; the row stays in registers when it fits ----
template <int NCH>  // float4 chunks per lane; NCH == 0 -> re-read the row (any D)
__global__ __launch_bounds__(64 * kRowWaves) void l2_normalize_kernel(const float* __restrict__ x,
                                                                      float* __restrict__ y, int64_t N, int64_t D) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * kRowWaves;
  for (int64_t row = (int64_t)blockIdx.x * kRowWaves + wave; row < N; row += wave_stride) {
    const float* p = x + row * D;
    float* q = y + row * D;
    float ss = 0.f;
    if constexpr (NCH > 0) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
      float4* q4 = reinterpret_cast<float4*>(q);
      const int n4 = (int)(D >> 2);
      float4 v[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        v[c] = (i < n4) ? p4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < n4) ss += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
      ss = wave_sum_f32(ss);
      const float den = sqrtf(ss) + 1e-10f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int i = lane + 64 * c;
        if (i < n4) q4[i] = make_float4(v[c].x / den, v[c].y / den, v[c].z / den, v[c].w / den);
      }
    } else {
      for (int64_t i = lane; i < D; i += 64) ss += p[i] * p[i];
      ss = wave_sum_f32(ss);
      const float den = sqrtf(ss) + 1e-10f;
      for (int64_t i = lane; i < D; i += 64) q[i] = p[i] / den;
    }
  }
}


// ---- per-row order statistics in registers (ASH-S pruning, GEN top-M) -------------------------------------
// One wave owns one row; lane l holds elements l, l+64, ...  The k-th largest value is found by a binary search, bit
// by bit, on the order-preserving integer image of the floats (count(key >= candidate)) - no sort.
__device__ __forceinline__ unsigned sort_key(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending in x
}

// count(key >= cand) on the scalar side: one v_cmp per register, population count and sum of the 64-bit masks in scalar
// registers - no per-lane counter, no cross-lane reduction (ASH-S 262 144 x 2048: 1.55 -> 1.46 ms, GEN 1 M x 1000: 9.8 -> 9.0 ms)
template <int R>
__device__ __forceinline__ int count_ge(const unsigned (&key)[R], unsigned cand) {
  int cnt = 0;
#pragma unroll
  for (int t = 0; t < R; ++t) cnt += __popcll(__ballot(key[t] >= cand));
  return cnt;
}

// The keys that agree with `prefix` above bit `bit` (the range the search has narrowed to), packed through the wave's LDS
// buffer into R2 registers per lane; empty slots hold 0, which no candidate (>= 1) counts.  The caller knows there are at
// most 64 * R2 of them.
template <int R, int R2>
__device__ __forceinline__ void compact_range(const unsigned (&key)[R], unsigned prefix, int bit, unsigned* buf,
                                              unsigned (&out)[R2], int lane) {
  const int sh = bit + 1;  // 1 .. 31
  int base = 0;
#pragma unroll
  for (int t = 0; t < R; ++t) {
    const bool in = ((key[t] ^ prefix) >> sh) == 0u;
    const unsigned long long mk = __ballot(in);
    if (in) buf[base + __popcll(mk & ((1ull << lane) - 1ull))] = key[t];
    base += __popcll(mk);
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < R2; ++r) out[r] = (lane + 64 * r < base) ? buf[lane + 64 * r] : 0u;
  __builtin_amdgcn_wave_barrier();  // buf is written again by the next compaction / the next row
}

constexpr int kth_slots(int NV) { return NV >= 8 ? NV / 4 : 0; }  // registers per lane after the first compaction (0: none)

// The step count (cnt_lo - cnt_hi = keys left in the range) is wave-uniform and known from the counts alone, so the
// search narrows its operand set as it goes: all NV registers until at most 64 * NV/4 keys are left, those - packed
// through `buf` (64 * NV/4 words of the wave's own LDS) - until at most 64 are left, then one register per lane.  With
// all 32 steps on all registers the scalar unit was the limit (two scalar instructions per register and step: GEN
// 1 M x 1000 3.3 ms, ASH-S 1 M x 2048 5.5 ms).
// `prefix` / `bit` / `lo0` (optional): bits above `bit` that every candidate shares, found by the caller (GEN: the wave's largest
// and smallest key agree in them) - the steps that would only confirm them are skipped; lo0 = number of keys >= prefix.
template <int NV>
__device__ __forceinline__ unsigned kth_largest_key(const unsigned (&key)[NV], int k, unsigned* buf, int lane,
                                                    unsigned prefix = 0u, int bit = 31, int lo0 = 64 * NV) {
  constexpr int R1 = kth_slots(NV);
  if constexpr (R1 == 0) {
#pragma unroll 1
    for (; bit >= 0; --bit) {
      const unsigned cand = prefix | (1u << bit);
      if (count_ge<NV>(key, cand) >= k) prefix = cand;  // wave-uniform
    }
    return prefix;
  } else {
    int lo = lo0, hi = 0;  // keys >= prefix, keys >= the range's upper end
#pragma unroll 1
    for (; bit >= 0 && lo - hi > 64 * R1; --bit) {
      const unsigned cand = prefix | (1u << bit);
      const int cnt = count_ge<NV>(key, cand);
      if (cnt >= k) { prefix = cand; lo = cnt; } else hi = cnt;
    }
    if (bit < 0) return prefix;
    unsigned k1[R1];
    compact_range<NV, R1>(key, prefix, bit, buf, k1, lane);
    int kk = k - hi;  // the rank among the keys that are left
    lo -= hi;
    hi = 0;
#pragma unroll 1
    for (; bit >= 0 && lo - hi > 64; --bit) {
      const unsigned cand = prefix | (1u << bit);
      const int cnt = count_ge<R1>(k1, cand);
      if (cnt >= kk) { prefix = cand; lo = cnt; } else hi = cnt;
    }
    if (bit < 0) return prefix;
    unsigned k2[1];
    compact_range<R1, 1>(k1, prefix, bit, buf, k2, lane);
    kk -= hi;
#pragma unroll 1
    for (; bit >= 0; --bit) {
      const unsigned cand = prefix | (1u << bit);
      if (count_ge<1>(k2, cand) >= kk) prefix = cand;
    }
    return prefix;  // key of the k-th largest element
  }
}

// ASH-S for 2-D activations (reference inference/funcs.py:234-261): keep the k = n - round(n*p/100) largest entries
// of each row, zero the rest, multiply by exp(sum(row) / sum(kept)).  Ties at the threshold are kept in index order.
template <int NV>
__global__ __launch_bounds__(64 * kRowWaves) void ash_s_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t N,
                                                     int D, int k) {
  __shared__ unsigned sel_buf[kRowWaves][64 * (kth_slots(NV) ? kth_slots(NV) : 1)];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * kRowWaves + wave; row < N; row += (int64_t)gridDim.x * kRowWaves) {
    const float* p = x + row * D;
    float v[NV];
    unsigned key[NV];
    float s1 = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const int j = lane + 64 * t;
      v[t] = (j < D) ? p[j] : 0.f;
      key[t] = (j < D) ? sort_key(v[t]) : 0u;  // padding sorts below every real value
      s1 += v[t];
    }
    s1 = wave_sum_f32(s1);
    float s2 = 0.f;
    if (k > 0) {
      const unsigned thr = kth_largest_key<NV>(key, k, sel_buf[wave], lane);
      int gt = 0;
#pragma unroll
      for (int t = 0; t < NV; ++t) gt += (key[t] > thr);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) gt += __shfl_xor(gt, o, 64);
      int ties_left = k - gt;  // how many elements equal to the threshold are kept (index order)
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const bool tie = (key[t] == thr) && (lane + 64 * t < D);
        const unsigned long long m = __ballot(tie);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        const bool keep = (key[t] > thr) || (tie && rank < ties_left);
        ties_left -= __popcll(m);
        v[t] = keep ? v[t] : 0.f;
        s2 += v[t];
      }
      s2 = wave_sum_f32(s2);
    } else {
#pragma unroll
      for (int t = 0; t < NV; ++t) v[t] = 0.f;
    }
    const float sc = expf(s1 / s2);
    float* q = y + row * D;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const int j = lane + 64 * t;
      if (j < D) q[j] = v[t] * sc;
    }
  }
}

constexpr int kGenCompact = 512;  // largest M whose selected probabilities are packed through LDS (2 KB per wave)

// GEN (reference inference/funcs.py:347-375 on softmax(logits)): -sum over the M largest probabilities of
// p^gamma * (1-p)^gamma, f32.
// C <= 16 (CIFAR-10-sized heads): one row per lane in registers, as lse_tiny_kernel; the M largest probabilities are
// the last M entries of the sorted row (16-input network, zero padding sorts to the front and contributes 0).
// 1 M x 10 rows: 0.61 ms with a wave per row (10 of 64 lanes busy) -> 0.14 ms.
template <int CT>
__global__ __launch_bounds__(256) void gen_tiny_kernel(const float* __restrict__ logits, float* __restrict__ score, int64_t N,
                                                        int M, float gamma, int from_probs) {
  for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < N; row += (int64_t)gridDim.x * 256) {
    const float* p = logits + row * CT;
    float v[16];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
      v[j] = p[j];
      m = fmaxf(m, v[j]);
    }
    float s = 1.f;
    if (!from_probs) {  // (uniform) the rows are logits: softmax first
#pragma unroll
      for (int j = 0; j < CT; ++j) v[j] = exp_nonpos(v[j] - m);
      // the row sum in NumPy's order (scipy.special.softmax: np.sum over the contiguous axis = pairwise_sum: below 8 terms one
      // chain; up to 128: eight interleaved partial sums r[j] += a[8 i + j], ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the remainder
      // behind).  On a confident row 1 - p cancels in float32 and (1 - p)^gamma moves 2.5 % per ulp of p: with the sum added in
      // another order a row could land 2e-5 from the reference's own float32 value (tools/fuzz_kernels.py, seed 99).
      if constexpr (CT < 8) {
        s = 0.f;
#pragma unroll
        for (int j = 0; j < CT; ++j) s += v[j];
      } else {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = v[j];
        constexpr int kWhole = CT - CT % 8;
#pragma unroll
        for (int i = 8; i < kWhole; i += 8)
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] += v[i + j];
        s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
        for (int i = kWhole; i < CT; ++i) s += v[i];
      }
    }
    const float rs = 1.0f / s;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (j < CT) ? (from_probs ? v[j] : div_by_rcp(v[j], s, rs)) : 0.f;
    if (M < CT) runia_entropy::sort_asc<16>(v);  // wave-uniform
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const bool take = (M >= CT) ? (j < CT) : (j >= 16 - M);
      if (take) acc += gen_term(v[j], gamma);
    }
    score[row] = -acc;
  }
}

// Round 6 (1 M x 1000, M = 100: 1.84 -> see profiles/README.md; the kernel is bound by vector-instruction issue - 834 per row):
//   * the M largest are selected on the softmax NUMERATORS e = exp(v - max) (the quotient e / s is monotone in e, and rows that tie
//     in p contribute the same term whichever of them is taken), so only the ~M selected values are divided, not all C;
//   * 16-byte loads when the rows allow it (VEC: lane holds 4 consecutive classes per 256-class stripe; the selection does not
//     care which lane holds what);
//   * the bit search starts below the bits the row's largest and smallest key share (numerators of a row span a few binades: the
//     sign and most exponent bits cost a full-width count each just to be confirmed).
template <int NV, bool VEC>
__global__ __launch_bounds__(64 * kRowWaves) void gen_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                   int64_t N, int C, int M, float gamma, int from_probs) {
  constexpr int kBufWords = (64 * kth_slots(NV) > kGenCompact) ? 64 * kth_slots(NV) : kGenCompact;
  __shared__ unsigned gen_sel[kRowWaves][kBufWords];  // the selection's packed keys, then the selected numerators (as bits)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n4 = C >> 2;
  for (int64_t row = (int64_t)blockIdx.x * kRowWaves + wave; row < N; row += (int64_t)gridDim.x * kRowWaves) {
    const float* p = logits + row * C;
    float v[NV];
    bool ok[NV];  // element t of this lane is a class of the row (compile-time pattern per chunk under VEC)
    if constexpr (VEC) {
      const float4* p4 = reinterpret_cast<const float4*>(p);
#pragma unroll
      for (int c = 0; c < NV / 4; ++c) {
        const bool in = lane + 64 * c < n4;
        const float4 t = in ? p4[lane + 64 * c] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        v[4 * c] = t.x; v[4 * c + 1] = t.y; v[4 * c + 2] = t.z; v[4 * c + 3] = t.w;
        ok[4 * c] = ok[4 * c + 1] = ok[4 * c + 2] = ok[4 * c + 3] = in;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        ok[t] = lane + 64 * t < C;
        v[t] = ok[t] ? p[lane + 64 * t] : -INFINITY;
      }
    }
    float s = 1.f;
    if (!from_probs) {  // (uniform) the rows are logits: softmax numerators first
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < NV; ++t) m = fmaxf(m, v[t]);
      m = wave_max_f32(m);
      s = 0.f;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        // (one v_exp_f32 of d * log2(e) with the product's low part - 3 instructions for exp_nonpos's 9 - measured 1.487 -> 1.416 ms
        // and lost the 1e-5 contract on narrow heads, tests/test_hip_kernels.py::test_gen_score_widths_and_m: not kept)
        v[t] = exp_nonpos(v[t] - m);  // padding -> 0
        s += v[t];
      }
      s = wave_sum_f32(s);
    }
    const float rs = 1.0f / s;
    auto prob = [&](float e) { return from_probs ? e : div_by_rcp(e, s, rs); };  // the (softmax) probability of a numerator
    float acc = 0.f;
    if (M >= C) {
#pragma unroll
      for (int t = 0; t < NV; ++t)
        if (ok[t]) acc += gen_term(prob(v[t]), gamma);
      acc = wave_sum_f32(acc);
    } else {
      unsigned key[NV];  // numerators / probabilities are >= 0: their bit patterns order them; padding = 0 sorts last
      unsigned kmax = 0u, kmin = 0xffffffffu;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        key[t] = ok[t] ? __float_as_uint(v[t]) : 0u;
        kmax = key[t] > kmax ? key[t] : kmax;
        const unsigned lowk = ok[t] ? key[t] : 0xffffffffu;
        kmin = lowk < kmin ? lowk : kmin;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)kmax, o, 64), b = (unsigned)__shfl_xor((int)kmin, o, 64);
        kmax = a > kmax ? a : kmax;
        kmin = b < kmin ? b : kmin;
      }
      const unsigned diff = kmax ^ kmin;  // wave-uniform
      unsigned thr;
      if (diff == 0u) {
        thr = kmax;  // every class has the same numerator
      } else {
        const int top = 31 - __builtin_clz(diff);
        const unsigned prefix0 = (top == 31) ? 0u : (kmax & ~((2u << top) - 1u));
        thr = kth_largest_key<NV>(key, M, gen_sel[wave], lane, prefix0, top, C);
      }
      int gt = 0;  // wave-uniform: elements above the threshold
      if (M <= kGenCompact) {
        // The selected numerators (< M of the row's C) are packed into LDS first, so that the division and the transcendentals of a
        // term are evaluated for M/64 elements per lane instead of all NV under a mask (C = 1000, M = 100: 2 instead of 16).
        unsigned* sel = gen_sel[wave];
#pragma unroll
        for (int t = 0; t < NV; ++t) {
          const bool take = key[t] > thr;
          const unsigned long long mk = __ballot(take);
          if (take) sel[gt + __popcll(mk & ((1ull << lane) - 1ull))] = key[t];
          gt += __popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < gt; i += 64) acc += gen_term(prob(__uint_as_float(sel[i])), gamma);
        __builtin_amdgcn_wave_barrier();  // sel is reused by this wave's next row
      } else {
#pragma unroll
        for (int t = 0; t < NV; ++t) {
          const bool take = key[t] > thr;
          if (take) acc += gen_term(prob(v[t]), gamma);
          gt += __popcll(__ballot(take));
        }
      }
      acc = wave_sum_f32(acc);
      acc += (float)(M - gt) * (gen_term(prob(__uint_as_float(thr)), gamma));
    }
    if (lane == 0) score[row] = -acc;
  }
}

}  // namespace

extern "C" int runia_row_lse_msp_f32(const float* logits, float* lse, float* msp, int64_t N, int64_t C,
                                     runia_stream_t stream) {
  if (N < 0 || C <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!logits || (!lse && !msp)) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (C <= 16) {
    const unsigned grid = runia_stream_grid(N, 256);
#define RUNIA_LSE_TINY(CT) case CT: lse_tiny_kernel<CT><<<grid, 256, 0, s>>>(logits, lse, msp, N); break;
    switch ((int)C) {
      RUNIA_LSE_TINY(1) RUNIA_LSE_TINY(2) RUNIA_LSE_TINY(3) RUNIA_LSE_TINY(4) RUNIA_LSE_TINY(5) RUNIA_LSE_TINY(6)
      RUNIA_LSE_TINY(7) RUNIA_LSE_TINY(8) RUNIA_LSE_TINY(9) RUNIA_LSE_TINY(10) RUNIA_LSE_TINY(11) RUNIA_LSE_TINY(12)
      RUNIA_LSE_TINY(13) RUNIA_LSE_TINY(14) RUNIA_LSE_TINY(15) RUNIA_LSE_TINY(16)
    }
#undef RUNIA_LSE_TINY
    return runia_check_launch();
  }
  if (C <= 64) {
    const size_t shmem = (size_t)kSmallRows * C * sizeof(float);
    lse_small_kernel<<<runia_stream_grid(N, kSmallRows), 256, shmem, s>>>(logits, lse, msp, N, (int)C);
    return runia_check_launch();
  }
  const unsigned grid = runia_rows_grid(N);
  constexpr int kT = 64 * kRowWaves;
  const bool vec = ((C & 3) == 0) && ((((uintptr_t)logits) & 15) == 0);
  const int64_t n4 = C >> 2;
  if (vec && n4 <= 64)
    lse_wave_kernel<1><<<grid, kT, 0, s>>>(logits, lse, msp, N, C);
  else if (vec && n4 <= 128)
    lse_wave_kernel<2><<<grid, kT, 0, s>>>(logits, lse, msp, N, C);
  else if (vec && n4 <= 256)
    lse_wave_kernel<4><<<grid, kT, 0, s>>>(logits, lse, msp, N, C);
  else if (vec && n4 <= 512)
    lse_wave_kernel<8><<<grid, kT, 0, s>>>(logits, lse, msp, N, C);
  else
    lse_wave_kernel<0><<<grid, kT, 0, s>>>(logits, lse, msp, N, C);
  return runia_check_launch();
}

extern "C" int runia_l2_normalize_f32(const float* x, float* y, int64_t N, int64_t D, runia_stream_t stream) {
  if (N < 0 || D <= 0 || (N > 0 && (!x || !y))) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  const unsigned grid = runia_rows_grid(N);
  constexpr int kT = 64 * kRowWaves;
  hipStream_t s = as_stream(stream);
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)y) & 15) == 0);
  const int64_t n4 = D >> 2;
  if (vec && n4 <= 64) l2_normalize_kernel<1><<<grid, kT, 0, s>>>(x, y, N, D);
  else if (vec && n4 <= 128) l2_normalize_kernel<2><<<grid, kT, 0, s>>>(x, y, N, D);
  else if (vec && n4 <= 256) l2_normalize_kernel<4><<<grid, kT, 0, s>>>(x, y, N, D);
  else if (vec && n4 <= 512) l2_normalize_kernel<8><<<grid, kT, 0, s>>>(x, y, N, D);
  else l2_normalize_kernel<0><<<grid, kT, 0, s>>>(x, y, N, D);
  return runia_check_launch();
}

extern "C" int runia_ash_s_f32(const float* x, float* y, int64_t N, int64_t D, int percentile,
                               runia_stream_t stream) {
  if (N < 0 || D <= 0 || D > 4096 || percentile < 0 || percentile > 100) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !y) return RUNIA_E_INVALID;
  // k = n - int(np.round(n * percentile / 100.0))   (round half to even, as NumPy)
  const double frac = (double)D * (double)percentile / 100.0;
  int k = (int)D - (int)nearbyint(frac);
  if (k == 0) k = (int)D;  // NumPy: x[:, -0:] is the whole row, i.e. nothing is pruned at percentile 100
  const unsigned grid = runia_rows_grid(N);
  constexpr int kT = 64 * kRowWaves;
  hipStream_t s = as_stream(stream);
  if (D <= 512) ash_s_kernel<8><<<grid, kT, 0, s>>>(x, y, N, (int)D, k);
  else if (D <= 1024) ash_s_kernel<16><<<grid, kT, 0, s>>>(x, y, N, (int)D, k);
  else if (D <= 2048) ash_s_kernel<32><<<grid, kT, 0, s>>>(x, y, N, (int)D, k);
  else ash_s_kernel<64><<<grid, kT, 0, s>>>(x, y, N, (int)D, k);
  return runia_check_launch();
}

int runia_gen_rows_wide(const float* logits, float* score, int64_t N, int64_t C, int M, double gamma, int from_probs,
                        hipStream_t s);  // funcs_rows.hip

static int gen_rows(const float* logits, float* score, int64_t N, int64_t C, int M, double gamma, int from_probs,
                    runia_stream_t stream) {
  if (N < 0 || C <= 0 || M < 1) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!logits || !score) return RUNIA_E_INVALID;
  if (C > 4096) return runia_gen_rows_wide(logits, score, N, C, M, gamma, from_probs, as_stream(stream));
  const unsigned grid = runia_rows_grid(N);
  constexpr int kT = 64 * kRowWaves;
  hipStream_t s = as_stream(stream);
  const float g = (float)gamma;
  if (C <= 16) {
    const unsigned tg = runia_stream_grid(N, 256);
#define RUNIA_GEN_TINY(CT) case CT: gen_tiny_kernel<CT><<<tg, 256, 0, s>>>(logits, score, N, M, g, from_probs); break;
    switch ((int)C) {
      RUNIA_GEN_TINY(1) RUNIA_GEN_TINY(2) RUNIA_GEN_TINY(3) RUNIA_GEN_TINY(4) RUNIA_GEN_TINY(5) RUNIA_GEN_TINY(6)
      RUNIA_GEN_TINY(7) RUNIA_GEN_TINY(8) RUNIA_GEN_TINY(9) RUNIA_GEN_TINY(10) RUNIA_GEN_TINY(11) RUNIA_GEN_TINY(12)
      RUNIA_GEN_TINY(13) RUNIA_GEN_TINY(14) RUNIA_GEN_TINY(15) RUNIA_GEN_TINY(16)
    }
#undef RUNIA_GEN_TINY
    return runia_check_launch();
  }
  const bool vec = (C & 3) == 0 && (((uintptr_t)logits) & 15) == 0;  // rows of whole, aligned 16-byte groups
  if (C <= 64) gen_kernel<1, false><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else if (C <= 256 && vec) gen_kernel<4, true><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else if (C <= 256) gen_kernel<4, false><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else if (C <= 1024 && vec) gen_kernel<16, true><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else if (C <= 1024) gen_kernel<16, false><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else if (vec) gen_kernel<64, true><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  else gen_kernel<64, false><<<grid, kT, 0, s>>>(logits, score, N, (int)C, M, g, from_probs);
  return runia_check_launch();
}

extern "C" int runia_gen_score_f32(const float* logits, float* score, int64_t N, int64_t C, int M, double gamma,
                                   runia_stream_t stream) {
  return gen_rows(logits, score, N, C, M, gamma, 0, stream);
}
// generalized_entropy on rows that already are probabilities (the reference's free function, inference/funcs.py:347-375)
extern "C" int runia_gen_entropy_f32(const float* probs, float* score, int64_t N, int64_t C, int M, double gamma,
                                     runia_stream_t stream) {
  return gen_rows(probs, score, N, C, M, gamma, 1, stream);
}
