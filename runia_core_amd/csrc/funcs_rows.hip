// Row-streaming kernels behind the free functions of the reference's inference/funcs.py that the postprocessor classes
// do not cover (round 4):
//   runia_mcd_uncertainty_f32   get_predictive_uncertainty_score / get_mcd_pred_uncertainty_score
//                               (reference inference/funcs.py:430-465, 378-427): softmax of every MC row, the mean
//                               distribution of an image, H[mean] (predictive entropy) and H[mean] - mean H (mutual
//                               information) - the pred_h / mi baselines the harness compares LaREx with
//                               (evaluation/latent_space.py:257-261).  ONE launch, the logits are read once.
//   runia_ash_s_rows_f32        ash_s_conv_layer (inference/funcs.py:194-227) and ash_s_linear_layer for rows longer than
//                               the register kernel of rowwise.hip holds: k-th largest by an 8-bit radix select over the
//                               row (workgroup per row), pruning (optionally in place, as the reference's view + scatter_
//                               does to its argument) and the exp(s1 / s2) sharpening.
// HBM-bound: n_mc * C * 4 bytes per image (+ the optional softmax output); 7 reads + 1-2 writes of a row for ASH-S (the row
// stays in L2 between the passes).
#include "common.hpp"

namespace {

__device__ __forceinline__ unsigned sort_key_f32(float x) {  // ascending in x
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---- pred_h / mi ---------------------------------------------------------------------------------------------------
// One wave per image; lane l holds classes l, l + 64, ...; the rows of an image are n_mc consecutive rows of `logits`
// (torch.split(samples, n_mc), image-major).  Arithmetic in f32 as torch (softmax = exp(x - max) / sum - exponentials and logarithms on the transcendental unit, ~1 ulp,
// denormals kept (common.hpp), the quotient correctly rounded -, p * log(p) with
// 0 * log(0) = NaN exactly as the reference's torch expression); the sums over classes run over the lanes in a fixed
// order.
template <int NV>
__global__ __launch_bounds__(64 * kRowWaves) void mcd_uncertainty_kernel(const float* __restrict__ logits,
                                                                         float* __restrict__ probs, float* __restrict__ pred_h,
                                                                         float* __restrict__ mi, int64_t N, int n_mc, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t img = (int64_t)blockIdx.x * kRowWaves + wave; img < N; img += (int64_t)gridDim.x * kRowWaves) {
    float mean[NV];
#pragma unroll
    for (int t = 0; t < NV; ++t) mean[t] = 0.f;
    float eh = 0.f;
    for (int s = 0; s < n_mc; ++s) {
      const float* p = logits + (img * n_mc + s) * (int64_t)C;
      float v[NV];
      float m = -INFINITY;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int j = lane + 64 * t;
        v[t] = (j < C) ? p[j] : -INFINITY;
        m = fmaxf(m, v[t]);
      }
      m = wave_max_f32(m);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        v[t] = (lane + 64 * t < C) ? exp_nonpos(v[t] - m) : 0.f;
        sum += v[t];
      }
      sum = wave_sum_f32(sum);
      const float rsum = 1.0f / sum;
      float h = 0.f;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int j = lane + 64 * t;
        if (j < C) {
          const float pr = div_by_rcp(v[t], sum, rsum);
          if (probs) probs[(img * n_mc + s) * (int64_t)C + j] = pr;
          mean[t] += pr;
          h += pr * log_nonneg(pr);
        }
      }
      eh -= wave_sum_f32(h);
    }
    float ph = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      if (lane + 64 * t < C) {
        const float e = mean[t] / (float)n_mc;
        ph += e * log_nonneg(e);
      }
    }
    ph = -wave_sum_f32(ph);
    if (lane == 0) {
      pred_h[img] = ph;
      mi[img] = ph - eh / (float)n_mc;
    }
  }
}

// C <= 16 (CIFAR-10-sized heads): one image per lane, its n_mc * C logits are one contiguous run.
__global__ __launch_bounds__(256) void mcd_uncertainty_tiny_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                                    float* __restrict__ pred_h, float* __restrict__ mi,
                                                                    int64_t N, int n_mc, int C) {
  for (int64_t img = (int64_t)blockIdx.x * 256 + threadIdx.x; img < N; img += (int64_t)gridDim.x * 256) {
    float mean[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) mean[j] = 0.f;
    float eh = 0.f;
    for (int s = 0; s < n_mc; ++s) {
      const float* p = logits + (img * n_mc + s) * (int64_t)C;
      float v[16];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = (j < C) ? p[j] : -INFINITY;
        m = fmaxf(m, v[j]);
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v[j] = (j < C) ? exp_nonpos(v[j] - m) : 0.f;
        sum += v[j];
      }
      const float rsum = 1.0f / sum;
      float h = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < C) {
          const float pr = div_by_rcp(v[j], sum, rsum);
          if (probs) probs[(img * n_mc + s) * (int64_t)C + j] = pr;
          mean[j] += pr;
          h += pr * log_nonneg(pr);
        }
      }
      eh -= h;
    }
    float ph = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < C) {
        const float e = mean[j] / (float)n_mc;
        ph += e * log_nonneg(e);
      }
    }
    pred_h[img] = -ph;
    mi[img] = -ph - eh / (float)n_mc;
  }
}

// ---- ASH-S, rows of any length ---------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_f32(float v, float* red, int tid) {  // 256 threads, fixed order
  v = wave_sum_f32(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup per row.  keep = the k largest entries (ties at the threshold in index order); y = keep * exp(s1 / s2);
// `pruned` (optional, may alias x): keep ? x : 0.
__global__ __launch_bounds__(256) void ash_s_rows_kernel(const float* x, float* __restrict__ y, float* pruned, int64_t N,
                                                          int64_t D, int64_t k) {
  __shared__ unsigned hist[256];
  __shared__ float red[4];
  __shared__ unsigned sel[2];
  __shared__ unsigned tie_base[5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    const float* p = x + row * D;
    float s1 = 0.f;
    for (int64_t j = tid; j < D; j += 256) s1 += p[j];
    s1 = block_sum_f32(s1, red, tid);
    unsigned thr = 0u, ties_kept = 0u;  // key of the k-th largest, and how many entries equal to it are kept
    if (k > 0) {
      unsigned prefix = 0u, want = (unsigned)k;  // k-th largest = want-th from the top among the keys matching the prefix
      for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int64_t j = tid; j < D; j += 256) {
          const unsigned key = sort_key_f32(p[j]);
          if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned r = want;
          int d8 = 255;
          for (; d8 > 0; --d8) {
            if (r <= hist[d8]) break;
            r -= hist[d8];
          }
          sel[0] = (unsigned)d8;
          sel[1] = r;
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        want = sel[1];
        __syncthreads();
      }
      thr = prefix;
      ties_kept = want;  // of the entries equal to the threshold, the first `want` in index order
    }
    // kept sum and output, in index order so that ties are resolved by position: chunks of 256 entries, the ties of a
    // chunk ranked by a ballot per wave + the four wave totals
    float s2 = 0.f;
    unsigned ties_seen = 0u;
    for (int pass = 0; pass < 2; ++pass) {
      float scale = 0.f;
      if (pass == 1) {
        s2 = block_sum_f32(s2, red, tid);
        scale = expf(s1 / s2);
        ties_seen = 0u;
      }
      for (int64_t j0 = 0; j0 < D; j0 += 256) {
        const int64_t j = j0 + tid;
        const float v = (j < D) ? p[j] : 0.f;
        const unsigned key = (j < D) ? sort_key_f32(v) : 0u;
        const bool tie = (k > 0) && (j < D) && key == thr;
        const unsigned long long mk = __ballot(tie);
        __syncthreads();
        if (lane == 0) tie_base[wave + 1] = (unsigned)__popcll(mk);
        __syncthreads();
        unsigned before = ties_seen + (unsigned)__popcll(mk & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; ++w) before += tie_base[w + 1];
        const bool keep = (k > 0) && (j < D) && (key > thr || (tie && before < ties_kept));
        ties_seen += tie_base[1] + tie_base[2] + tie_base[3] + tie_base[4];
        if (pass == 0) {
          if (keep) s2 += v;
        } else if (j < D) {
          y[row * D + j] = keep ? v * scale : 0.f * scale;
          if (pruned) pruned[row * D + j] = keep ? v : 0.f;
        }
      }
    }
    __syncthreads();
  }
}

// ---- heads wider than the register kernels hold (C > 4 096: ImageNet-21k, LLM vocabularies) ----------------------------------
// One workgroup per image / row, the row re-read from L2 between the passes (as ash_s_rows_kernel).  Same arithmetic per
// element as the wave-per-row kernels; the sums over classes run over 256 threads in a fixed order.
__device__ __forceinline__ float block_max_f32(float v, float* red, int tid) {
  v = wave_max_f32(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void mcd_uncertainty_wide_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                                    float* __restrict__ pred_h, float* __restrict__ mi,
                                                                    int64_t N, int n_mc, int64_t C) {
  extern __shared__ float row_stats[];  // [2 * n_mc]: max and sum of exp of every MC row of the image
  __shared__ float red[4];
  const int tid = threadIdx.x;
  for (int64_t img = blockIdx.x; img < N; img += gridDim.x) {
    float eh = 0.f;
    for (int s = 0; s < n_mc; ++s) {
      const float* p = logits + (img * n_mc + s) * C;
      float m = -INFINITY;
      for (int64_t j = tid; j < C; j += 256) m = fmaxf(m, p[j]);
      m = block_max_f32(m, red, tid);
      float sum = 0.f;
      for (int64_t j = tid; j < C; j += 256) sum += exp_nonpos(p[j] - m);
      sum = block_sum_f32(sum, red, tid);
      const float rsum = 1.0f / sum;
      float h = 0.f;
      for (int64_t j = tid; j < C; j += 256) {
        const float pr = div_by_rcp(exp_nonpos(p[j] - m), sum, rsum);
        if (probs) probs[(img * n_mc + s) * C + j] = pr;
        h += pr * log_nonneg(pr);
      }
      eh -= block_sum_f32(h, red, tid);
      if (tid == 0) {
        row_stats[2 * s] = m;
        row_stats[2 * s + 1] = sum;
      }
    }
    __syncthreads();
    float ph = 0.f;
    for (int64_t j = tid; j < C; j += 256) {
      float mean = 0.f;
      for (int s = 0; s < n_mc; ++s) {
        const float sum = row_stats[2 * s + 1];
        mean += div_by_rcp(exp_nonpos(logits[(img * n_mc + s) * C + j] - row_stats[2 * s]), sum, 1.0f / sum);
      }
      const float e = mean / (float)n_mc;
      ph += e * log_nonneg(e);
    }
    ph = -block_sum_f32(ph, red, tid);
    if (tid == 0) {
      pred_h[img] = ph;
      mi[img] = ph - eh / (float)n_mc;
    }
    __syncthreads();  // row_stats is rewritten by the next image
  }
}

// GEN on rows of any length: -sum over the M largest probabilities of p^gamma (1 - p)^gamma.  The M-th largest probability by
// an 8-bit radix select over the row's keys (4 passes), then one pass for the terms above it; ties at the threshold count
// (M - #above) times, as in gen_kernel.
__global__ __launch_bounds__(256) void gen_wide_kernel(const float* __restrict__ logits, float* __restrict__ score, int64_t N,
                                                        int64_t C, int64_t M, float gamma, int from_probs) {
  __shared__ unsigned hist[256];
  __shared__ float red[4];
  __shared__ unsigned sel[2];
  const int tid = threadIdx.x;
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    const float* p = logits + row * C;
    float m = 0.f, s = 1.f;
    if (!from_probs) {  // (uniform) the rows are logits: softmax first
      m = -INFINITY;
      for (int64_t j = tid; j < C; j += 256) m = fmaxf(m, p[j]);
      m = block_max_f32(m, red, tid);
      s = 0.f;
      for (int64_t j = tid; j < C; j += 256) s += exp_nonpos(p[j] - m);
      s = block_sum_f32(s, red, tid);
    }
    const float rs = 1.0f / s;
    auto prob = [&](int64_t j) { return from_probs ? p[j] : div_by_rcp(exp_nonpos(p[j] - m), s, rs); };
    float acc = 0.f;
    if (M >= C) {
      for (int64_t j = tid; j < C; j += 256) {
        const float pv = prob(j);
        acc += gen_term(pv, gamma);
      }
      acc = block_sum_f32(acc, red, tid);
    } else {
      unsigned prefix = 0u, want = (unsigned)M;
      for (int shift = 24; shift >= 0; shift -= 8) {
        hist[tid] = 0u;
        __syncthreads();
        const unsigned himask = (shift == 24) ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int64_t j = tid; j < C; j += 256) {
          const unsigned key = __float_as_uint(prob(j)) | 0x80000000u;  // p >= 0
          if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
          unsigned r = want;
          int d8 = 255;
          for (; d8 > 0; --d8) {
            if (r <= hist[d8]) break;
            r -= hist[d8];
          }
          sel[0] = (unsigned)d8;
          sel[1] = r;
        }
        __syncthreads();
        prefix |= sel[0] << shift;
        want = sel[1];
        __syncthreads();
      }
      // `want` of the entries equal to the threshold belong to the M largest
      for (int64_t j = tid; j < C; j += 256) {
        const float pv = prob(j);
        if ((__float_as_uint(pv) | 0x80000000u) > prefix) acc += gen_term(pv, gamma);
      }
      acc = block_sum_f32(acc, red, tid);
      const float pt = __uint_as_float(prefix & 0x7fffffffu);
      acc += (float)want * (gen_term(pt, gamma));
    }
    if (tid == 0) score[row] = -acc;
    __syncthreads();
  }
}

}  // namespace

// rowwise.hip: GEN of rows longer than its register kernels hold
int runia_gen_rows_wide(const float* logits, float* score, int64_t N, int64_t C, int M, double gamma, int from_probs,
                        hipStream_t s) {
  gen_wide_kernel<<<(unsigned)(N < 65536 ? N : 65536), 256, 0, s>>>(logits, score, N, C, (int64_t)M, (float)gamma, from_probs);
  return runia_check_launch();
}

extern "C" int runia_mcd_uncertainty_f32(const float* logits, float* probs, float* pred_h, float* mi, int64_t N, int n_mc,
                                         int64_t C, runia_stream_t stream) {
  if (N < 0 || n_mc < 1 || C <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!logits || !pred_h || !mi) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (C > 4096) {  // workgroup per image, the rows re-read from L2 (any width; n_mc bounded by the LDS table of row statistics)
    if (n_mc > 4096) return RUNIA_E_INVALID;
    mcd_uncertainty_wide_kernel<<<(unsigned)(N < 65536 ? N : 65536), 256, 2 * n_mc * sizeof(float), s>>>(logits, probs, pred_h, mi, N,
                                                                                                        n_mc, C);
    return runia_check_launch();
  }
  if (C <= 16) {
    mcd_uncertainty_tiny_kernel<<<runia_stream_grid(N, 256), 256, 0, s>>>(logits, probs, pred_h, mi, N, n_mc, (int)C);
    return runia_check_launch();
  }
  const unsigned grid = runia_rows_grid(N);
  constexpr int kT = 64 * kRowWaves;
  if (C <= 64) mcd_uncertainty_kernel<1><<<grid, kT, 0, s>>>(logits, probs, pred_h, mi, N, n_mc, (int)C);
  else if (C <= 256) mcd_uncertainty_kernel<4><<<grid, kT, 0, s>>>(logits, probs, pred_h, mi, N, n_mc, (int)C);
  else if (C <= 1024) mcd_uncertainty_kernel<16><<<grid, kT, 0, s>>>(logits, probs, pred_h, mi, N, n_mc, (int)C);
  else mcd_uncertainty_kernel<64><<<grid, kT, 0, s>>>(logits, probs, pred_h, mi, N, n_mc, (int)C);
  return runia_check_launch();
}

extern "C" int runia_ash_s_rows_f32(const float* x, float* y, float* pruned, int64_t N, int64_t D, int percentile,
                                    int keep_all_when_k_is_zero, runia_stream_t stream) {
  if (N < 0 || D <= 0 || percentile < 0 || percentile > 100) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !y) return RUNIA_E_INVALID;
  // k = n - int(np.round(n * percentile / 100.0))   (round half to even, as NumPy)
  int64_t k = D - (int64_t)nearbyint((double)D * (double)percentile / 100.0);
  if (k == 0 && keep_all_when_k_is_zero) k = D;  // NumPy's x[:, -0:] is the whole row; torch.topk(k = 0) keeps nothing
  ash_s_rows_kernel<<<(unsigned)(N < 65536 ? N : 65536), 256, 0, as_stream(stream)>>>(x, y, pruned, N, D, k);
  return runia_check_launch();
}
