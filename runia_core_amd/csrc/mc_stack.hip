// a1: MC-dropout latent stacking.
// Replaces MCSamplerModule.forward (reference feature_extraction/abstract_classes.py:81-101):
//   for each of n_mc DropBlock2D layers (dropblock==0.3.0):
//     mask = rand(1,H,W) < drop_prob / block_size^2
//     bm   = 1 - max_pool2d(mask, k=block_size, stride=1, pad=block_size//2)   (crop last row/col if even)
//     y    = x * bm * numel(bm) / sum(bm)
//     out  = fullmean(y) over W then H          (reference feature_extraction/utils.py:88-92)
// The block mask is shared by all channels, so per channel the work is a masked mean over H*W
// followed by one scalar rescale per drop layer.  HBM-bound: C*H*W*4 bytes in, n_mc*C*4 out per image.
//
// Grid: one workgroup per (image, 256-channel slab).  The n_mc block masks of the image are
// built once per workgroup in LDS from the caller's uniform draws; each thread then owns one
// channel, reads its H*W map with 16-byte loads and produces n_mc outputs.
#include "common.hpp"

namespace {

constexpr int kMaxHW = 1024;  // H*W limit (32x32 map)
constexpr int kMaxMC = 64;

__global__ __launch_bounds__(256) void mc_stack_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ rnd, int64_t rand_stride,
                                                        float* __restrict__ out, int C, int H, int W, int n_mc,
                                                        float gamma, int block_size, int identity) {
  extern __shared__ float lds[];       // bm [n_mc][HW] then scale [n_mc]
  const int HW = H * W;
  float* bm = lds;
  float* scale = lds + n_mc * HW;
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.y;
  const float* r = rnd ? rnd + img * rand_stride : nullptr;
  const int pad = block_size / 2;

  // 1) block masks
  for (int i = tid; i < n_mc * HW; i += 256) {
    float keep = 1.f;
    if (!identity) {
      const int s = i / HW, p = i - s * HW;
      const int y = p / W, xw = p - y * W;
      // output (y,xw) of the stride-1 max pool covers input rows y-pad .. y-pad+block_size-1
      bool dropped = false;
      for (int dy = 0; dy < block_size && !dropped; ++dy) {
        const int yy = y - pad + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < block_size; ++dx) {
          const int xx = xw - pad + dx;
          if (xx < 0 || xx >= W) continue;
          if (r[s * HW + yy * W + xx] < gamma) { dropped = true; break; }
        }
      }
      keep = dropped ? 0.f : 1.f;
    }
    bm[i] = keep;
  }
  __syncthreads();
  // 2) per-layer mask sum: one wave per layer
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int s = wave; s < n_mc; s += 4) {
      float acc = 0.f;
      for (int p = lane; p < HW; p += 64) acc += bm[s * HW + p];
      acc = wave_sum_f32(acc);  // exact: small integer counts
      if (lane == 0) scale[s] = acc;
    }
  }
  __syncthreads();
  // 3) masked means, one channel per thread; upstream order ((x*bm)*numel)/sum -> mean W -> mean H
  const int c = blockIdx.x * 256 + tid;
  if (c >= C) return;
  const float* xc = x + (img * C + c) * (int64_t)HW;
  float* o = out + img * n_mc * (int64_t)C + c;
  for (int s = 0; s < n_mc; ++s) {
    const float* b = bm + s * HW;
    const float den = scale[s];
    // mean over W then over H, each row added in torch's CPU order (common.hpp::torch_row_sum)
    const float col = torch_row_sum(
        [&](int y) {
          const float rowsum = torch_row_sum(
              [&](int xw) {
                const float q = (xc[y * W + xw] * (float)HW) / den;
                return (b[y * W + xw] != 0.f) ? q : 0.f;
              },
              W);
          return rowsum / (float)W;
        },
        H);
    o[(int64_t)s * C] = (den == 0.f) ? NAN : col / (float)H;
  }
}


// ---- register-resident form for small maps (H*W <= 64): the thread's whole map lives in VGPRs,
// the block masks are 64-bit words broadcast from LDS, and every element follows the upstream
// operation order  y = ((x * bm) * numel) / sum(bm)  -> mean over W -> mean over H  in f32.
// The per-element division by the (integer-valued) mask sum is a reciprocal product with one
// Newton correction (q + r*fma(-s, q, u)), 3 VALU ops instead of the ~10 of a full IEEE divide.
__device__ __forceinline__ float div_newton(float u, float den, float r) {
  const float q = u * r;
  const float e = fmaf(-den, q, u);
  return fmaf(e, r, q);
}

template <int HT, int WT>
__global__ __launch_bounds__(256) void mc_stack_small_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ rnd, int64_t rand_stride,
                                                              float* __restrict__ out, int C, int n_mc,
                                                              float gamma, int block_size, int identity) {
  constexpr int HW = HT * WT;
  __shared__ unsigned long long keep_bits[kMaxMC];
  __shared__ float msum[kMaxMC], mrcp[kMaxMC];
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.y;
  const int pad = block_size / 2;
  // 1) block masks: draws -> LDS with one coalesced pass, then one thread per (layer, position)
  __shared__ float draws[kMaxMC * HW];
  __shared__ unsigned keep_lo[kMaxMC], keep_hi[kMaxMC];
  if (tid < n_mc) { keep_lo[tid] = 0u; keep_hi[tid] = 0u; }
  if (!identity) {
    const float* r = rnd + img * rand_stride;
    for (int i = tid; i < n_mc * HW; i += 256) draws[i] = r[i];
  }
  __syncthreads();
  for (int i = tid; i < n_mc * HW; i += 256) {
    const int s = i / HW, p = i - s * HW;
    bool dropped = false;
    if (!identity) {
      const int y = p / WT, xw = p - y * WT;
      for (int dy = 0; dy < block_size; ++dy) {
        const int yy = y - pad + dy;
        if (yy < 0 || yy >= HT) continue;
        for (int dx = 0; dx < block_size; ++dx) {
          const int xx = xw - pad + dx;
          if (xx < 0 || xx >= WT) continue;
          dropped = dropped || (draws[s * HW + yy * WT + xx] < gamma);
        }
      }
    }
    if (!dropped) {
      if (p < 32) atomicOr(&keep_lo[s], 1u << p);
      else atomicOr(&keep_hi[s], 1u << (p - 32));
    }
  }
  __syncthreads();
  if (tid < n_mc) {
    const unsigned long long bits = ((unsigned long long)keep_hi[tid] << 32) | keep_lo[tid];
    keep_bits[tid] = bits;
    const float cnt = (float)__popcll(bits);
    msum[tid] = cnt;
    mrcp[tid] = 1.0f / cnt;
  }
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c >= C) return;
  const float* xc = x + (img * C + c) * (int64_t)HW;
  float u[HW];
  if constexpr (HW % 4 == 0) {
#pragma unroll
    for (int p = 0; p < HW / 4; ++p) {
      const float4 v = reinterpret_cast<const float4*>(xc)[p];
      u[4 * p] = v.x; u[4 * p + 1] = v.y; u[4 * p + 2] = v.z; u[4 * p + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int p = 0; p < HW; ++p) u[p] = xc[p];
  }
#pragma unroll
  for (int p = 0; p < HW; ++p) u[p] *= (float)HW;  // (x * bm) * numel; exact for power-of-two maps
  constexpr bool w_pow2 = (WT & (WT - 1)) == 0, h_pow2 = (HT & (HT - 1)) == 0;
  const float rW = 1.0f / (float)WT, rH = 1.0f / (float)HT;
  float* o = out + img * n_mc * (int64_t)C + c;
  for (int s = 0; s < n_mc; ++s) {
    const unsigned long long bits = keep_bits[s];
    const float den = msum[s], r = mrcp[s];
    float rm[HT];
#pragma unroll
    for (int y = 0; y < HT; ++y) {
      float rowsum = 0.f;
#pragma unroll
      for (int xi = 0; xi < WT; ++xi) {  // the row in torch's CPU summation order
        const int p = y * WT + torch_chain<WT>(xi);
        const float q = div_newton(u[p], den, r);
        rowsum += ((bits >> p) & 1ull) ? q : 0.f;
      }
      rm[y] = w_pow2 ? rowsum * rW : div_newton(rowsum, (float)WT, rW);
    }
    float col = 0.f;
#pragma unroll
    for (int yi = 0; yi < HT; ++yi) col += rm[torch_chain<HT>(yi)];
    float res = h_pow2 ? col * rH : div_newton(col, (float)HT, rH);
    if (den == 0.f) res = NAN;  // every position dropped: 0 * numel / 0 upstream
    o[(int64_t)s * C] = res;
  }
}


// ---- layer_type "FC" / "RPN" (feature_extraction/abstract_classes.py:95-99): no fullmean, every drop layer's
// output y = ((x * bm) * numel) / sum is flattened: out[img*n_mc + s][c*HW + p].  Write-bound (n_mc outputs per
// input element); one workgroup per (image, slab of 1024 elements), block masks built once per workgroup in LDS.
__global__ __launch_bounds__(256) void mc_drop_flat_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ rnd, int64_t rand_stride,
                                                            float* __restrict__ out, int C, int H, int W, int n_mc,
                                                            float gamma, int block_size, int identity) {
  extern __shared__ float lds[];  // bm [n_mc][HW] then mask sums [n_mc]
  const int HW = H * W;
  float* bm = lds;
  float* msum = lds + n_mc * HW;
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.y;
  const float* r = rnd ? rnd + img * rand_stride : nullptr;
  const int pad = block_size / 2;
  for (int i = tid; i < n_mc * HW; i += 256) {
    float keep = 1.f;
    if (!identity) {
      const int s = i / HW, p = i - s * HW;
      const int y = p / W, xw = p - y * W;
      bool dropped = false;
      for (int dy = 0; dy < block_size && !dropped; ++dy) {
        const int yy = y - pad + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = 0; dx < block_size; ++dx) {
          const int xx = xw - pad + dx;
          if (xx < 0 || xx >= W) continue;
          if (r[s * HW + yy * W + xx] < gamma) { dropped = true; break; }
        }
      }
      keep = dropped ? 0.f : 1.f;
    }
    bm[i] = keep;
  }
  __syncthreads();
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int s = wave; s < n_mc; s += 4) {
      float acc = 0.f;
      for (int p = lane; p < HW; p += 64) acc += bm[s * HW + p];
      acc = wave_sum_f32(acc);  // exact: small integer counts
      if (lane == 0) msum[s] = acc;
    }
  }
  __syncthreads();
  const int64_t E = (int64_t)C * HW;
  const float* xi = x + img * E;
  float* oi = out + img * n_mc * E;
  for (int64_t e = (int64_t)blockIdx.x * 1024 + tid; e < E && e < ((int64_t)blockIdx.x + 1) * 1024; e += 256) {
    const float v = xi[e];
    const int p = (int)(e % HW);
    for (int s = 0; s < n_mc; ++s) {
      // identity layers (eval mode / drop_prob 0) return x itself upstream, not x * numel / numel
      oi[(int64_t)s * E + e] = identity ? v : ((v * bm[s * HW + p]) * (float)HW) / msum[s];
    }
  }
}

}  // namespace

namespace {
// ---- large maps (H*W > 64: 14x14, 16x16, 28x28, 32x32 ...): the masked means of an image as ONE contraction -----------
// out[s][c] = sum_p x[c][p] * keep[s][p] / sum_p keep[s][p]   (what mean_H(mean_W((x * bm * numel) / sum)) adds up to)
// on the f64 matrix cores: A = 16 channels x K positions (f32 rows widened to f64, 16-byte loads), B = keep flags
// K x 16 drop layers from LDS, products exact (flags are 0 / 1), f64 accumulation, one division and one rounding to f32
// per sample.  The k order inside a group of 16 positions is permuted (step j of lane group g takes position
// 4 g + j): a sum does not care, and every lane then reads its four positions with one float4.
// The block masks are built per workgroup in LDS: seeds (draw < gamma), row dilation, column dilation (O(bs) each).
// The thread-per-channel kernel above re-reads its map n_mc times with lane-strided loads and divides per element:
// 2 000 x 256 x 14x14: 4.6 ms (0.1 TB/s); this one: 0.22 ms (2.1 TB/s; 16x16: 0.76 -> 0.08 ms).  Maps of up to 64 positions keep the kernels that
// add in torch's CPU order (bit-exact on the reference-run fixtures); for wider rows torch itself sums in vector lanes.
typedef double mc_d4 __attribute__((ext_vector_type(4)));

template <int NT>  // column tiles of 16 drop layers
__global__ __launch_bounds__(256) void mc_stack_mfma_kernel(const float* __restrict__ x, const float* __restrict__ rnd,
                                                             int64_t rand_stride, float* __restrict__ out, int C, int H,
                                                             int W, int n_mc, float gamma, int block_size, int identity) {
  extern __shared__ __attribute__((aligned(16))) unsigned char mc_smem[];
  constexpr int NSP = 16 * NT + 4;  // row pitch of the flag table: lane groups 4 positions apart land 16 banks apart
  const int HW = H * W, HWp = (HW + 15) & ~15;
  float* keepf = reinterpret_cast<float*>(mc_smem);                       // [HWp][NSP]
  unsigned char* seed = mc_smem + (size_t)HWp * NSP * sizeof(float);       // [n_mc][HW]
  unsigned char* hx = seed + (((size_t)n_mc * HW + 15) & ~(size_t)15);     // [n_mc][HW]
  int* cnt = reinterpret_cast<int*>(hx + (((size_t)n_mc * HW + 15) & ~(size_t)15));  // [16 * NT]
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.y;
  const int pad = block_size / 2;
  for (int i = tid; i < HWp * NSP; i += 256) keepf[i] = 0.f;
  if (tid < 16 * NT) cnt[tid] = 0;
  {
    const float* r = rnd ? rnd + img * rand_stride : nullptr;
    for (int i = tid; i < n_mc * HW; i += 256) seed[i] = (!identity && r[i] < gamma) ? 1 : 0;
  }
  __syncthreads();
  for (int i = tid; i < n_mc * HW; i += 256) {  // dropped-so-far(y, x) = OR over the window columns of the seeds
    const int s = i / HW, p = i - s * HW;
    const int y = p / W, xw = p - y * W;
    unsigned char v = 0;
    for (int dx = 0; dx < block_size; ++dx) {
      const int xx = xw - pad + dx;
      if (xx >= 0 && xx < W) v |= seed[s * HW + y * W + xx];
    }
    hx[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < n_mc * HW; i += 256) {  // ... then over the window rows
    const int s = i / HW, p = i - s * HW;
    const int y = p / W, xw = p - y * W;
    unsigned char v = 0;
    for (int dy = 0; dy < block_size; ++dy) {
      const int yy = y - pad + dy;
      if (yy >= 0 && yy < H) v |= hx[s * HW + yy * W + xw];
    }
    if (!v) {
      keepf[p * NSP + s] = 1.f;
      atomicAdd(&cnt[s], 1);
    }
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int c0 = blockIdx.x * 64 + wave * 16;
  if (c0 >= C) return;  // wave-uniform; no barrier follows
  const int ch = c0 + li;
  const bool valid = ch < C;
  const float* xr = x + ((int64_t)img * C + (valid ? ch : c0)) * (int64_t)HW;
  const bool vec = ((HW & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  mc_d4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (mc_d4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < HWp; k0 += 16) {
    const int kb = k0 + 4 * lg;  // this lane's four positions of the group
    float a4[4];
    if (vec) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && kb < HW) v = *reinterpret_cast<const float4*>(xr + kb);
      a4[0] = v.x; a4[1] = v.y; a4[2] = v.z; a4[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) a4[j] = (valid && kb + j < HW) ? xr[kb + j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double a = (double)a4[j];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const double b = (double)keepf[(kb + j) * NSP + 16 * t + li];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int s = 16 * t + li;
    if (s < n_mc) {
      const int den = cnt[s];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = c0 + lg + 4 * r;
        if (co < C)  // every position dropped: 0 * numel / 0 upstream
          out[((int64_t)img * n_mc + s) * C + co] = den ? (float)(acc[t][r] / (double)den) : NAN;
      }
    }
  }
}

static size_t mc_mfma_lds_bytes(int HW, int n_mc, int NT) {
  const size_t HWp = ((size_t)HW + 15) & ~(size_t)15, flags = (((size_t)n_mc * HW + 15) & ~(size_t)15);
  return HWp * (16 * NT + 4) * sizeof(float) + 2 * flags + 16 * NT * sizeof(int);
}

template <int NT>
int launch_mc_mfma(const float* x, const float* rnd, int64_t rand_image_stride, float* out, int64_t N, int C, int H, int W,
                   int n_mc, float gamma, int block_size, int identity, hipStream_t s) {
  const size_t lds = mc_mfma_lds_bytes(H * W, n_mc, NT);
  static std::atomic<uint64_t> lds_ok{0};
  if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(mc_stack_mfma_kernel<NT>), 160 * 1024, lds_ok) != RUNIA_OK)
    return RUNIA_E_LAUNCH;
  dim3 grid((C + 63) / 64, (unsigned)N);
  mc_stack_mfma_kernel<NT><<<grid, 256, lds, s>>>(x, rnd, rand_image_stride, out, C, H, W, n_mc, gamma, block_size,
                                                   identity);
  return runia_check_launch();
}
}  // namespace

extern "C" int runia_mc_stack_f32(const float* x, const float* rnd, int64_t rand_image_stride, float* out,
                                  int64_t N, int C, int H, int W, int n_mc, double drop_prob, int block_size,
                                  runia_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || n_mc < 1 || n_mc > kMaxMC || block_size < 1 ||
      (int64_t)H * W > kMaxHW || (N > 0 && (!x || !out)))
    return RUNIA_E_INVALID;
  const int identity = (drop_prob == 0.0);
  if (!identity && !rnd) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (N > 65535) return RUNIA_E_INVALID;  // grid.y limit; callers batch above this
  // the reference forms gamma in Python f64 and compares f32 draws against its f32 rounding
  const float gamma = (float)(drop_prob / (double)(block_size * block_size));
  if (H * W > 64) {  // large maps: one contraction per image on the matrix cores
    const int nt = (n_mc + 15) / 16;
    if (mc_mfma_lds_bytes(H * W, n_mc, nt) <= 160 * 1024) {
      hipStream_t s = as_stream(stream);
      if (nt == 1) return launch_mc_mfma<1>(x, rnd, rand_image_stride, out, N, C, H, W, n_mc, gamma, block_size, identity, s);
      if (nt == 2) return launch_mc_mfma<2>(x, rnd, rand_image_stride, out, N, C, H, W, n_mc, gamma, block_size, identity, s);
      if (nt == 3) return launch_mc_mfma<3>(x, rnd, rand_image_stride, out, N, C, H, W, n_mc, gamma, block_size, identity, s);
      return launch_mc_mfma<4>(x, rnd, rand_image_stride, out, N, C, H, W, n_mc, gamma, block_size, identity, s);
    }
  }
  const size_t shmem = ((size_t)n_mc * H * W + n_mc) * sizeof(float);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  dim3 grid((C + 255) / 256, (unsigned)N);
  const bool x16 = ((((uintptr_t)x) & 15) == 0);
#define RUNIA_MC_SMALL(HH, WW)                                                                              \
  if (H == HH && W == WW && (x16 || (HH * WW) % 4 != 0)) {                                                  \
    mc_stack_small_kernel<HH, WW><<<grid, 256, 0, as_stream(stream)>>>(x, rnd, rand_image_stride, out, C,  \
                                                                       n_mc, gamma, block_size, identity); \
    return runia_check_launch();                                                                            \
  }
  RUNIA_MC_SMALL(2, 2)
  RUNIA_MC_SMALL(3, 3)
  RUNIA_MC_SMALL(4, 4)
  RUNIA_MC_SMALL(5, 5)
  RUNIA_MC_SMALL(6, 6)
  RUNIA_MC_SMALL(7, 7)
  RUNIA_MC_SMALL(8, 8)
#undef RUNIA_MC_SMALL
  mc_stack_kernel<<<grid, 256, shmem, as_stream(stream)>>>(x, rnd, rand_image_stride, out, C, H, W, n_mc,
                                                          gamma, block_size, identity);
  return runia_check_launch();
}

namespace {
// Reductions of (already dropped) activation maps the extractor's other options ask for
// (feature_extraction/utils.py:70-92, 113-126): one thread per map.
//   mode 0 "mean"   : torch.mean(dim=3)               -> out [maps, H]
//   mode 1 "std"    : torch.std(torch.std(., 3), 2)   -> out [maps]   (unbiased, as torch's default; a map with one row or
//                                                         one column gives NaN there too)
// f64 accumulation (torch's CPU kernels accumulate f32 inputs in f64), one rounding to f32 at the end.
__global__ __launch_bounds__(256) void map_reduce_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int64_t maps, int H, int W, int mode) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < maps; m += (int64_t)gridDim.x * 256) {
    const float* p = x + m * (int64_t)(H * W);
    if (mode == 0) {
      for (int y = 0; y < H; ++y) {
        double s = 0.0;
        for (int xw = 0; xw < W; ++xw) s += (double)p[y * W + xw];
        out[m * H + y] = (float)(s / (double)W);
      }
    } else {
      // std over the row, rounded to f32 as torch materialises it, then std of those over the rows
      double mean2 = 0.0, m2 = 0.0;
      for (int y = 0; y < H; ++y) {
        double s = 0.0;
        for (int xw = 0; xw < W; ++xw) s += (double)p[y * W + xw];
        const double mu = s / (double)W;
        double q = 0.0;
        for (int xw = 0; xw < W; ++xw) {
          const double d = (double)p[y * W + xw] - mu;
          q += d * d;
        }
        const double sd = (double)(float)sqrt(q / (double)(W - 1));  // W == 1: 0/0 = NaN
        const double delta = sd - mean2;  // Welford over the H row values
        mean2 += delta / (double)(y + 1);
        m2 += delta * (sd - mean2);
      }
      out[m] = (float)sqrt(m2 / (double)(H - 1));
    }
  }
}
}  // namespace

extern "C" int runia_map_reduce_f32(const float* x, float* out, int64_t maps, int H, int W, int mode,
                                    runia_stream_t stream) {
  if (maps < 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return RUNIA_E_INVALID;
  if (maps == 0) return RUNIA_OK;
  if (!x || !out) return RUNIA_E_INVALID;
  map_reduce_kernel<<<runia_stream_grid(maps, 256), 256, 0, as_stream(stream)>>>(x, out, maps, H, W, mode);
  return runia_check_launch();
}

extern "C" int runia_mc_drop_flat_f32(const float* x, const float* rnd, int64_t rand_image_stride, float* out,
                                      int64_t N, int C, int H, int W, int n_mc, double drop_prob, int block_size,
                                      runia_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || n_mc < 1 || n_mc > kMaxMC || block_size < 1 ||
      (int64_t)H * W > kMaxHW || (N > 0 && (!x || !out)))
    return RUNIA_E_INVALID;
  const int identity = (drop_prob == 0.0);
  if (!identity && !rnd) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (N > 65535) return RUNIA_E_INVALID;  // grid.y limit; callers batch above this
  const float gamma = (float)(drop_prob / (double)(block_size * block_size));
  const size_t shmem = ((size_t)n_mc * H * W + n_mc) * sizeof(float);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  const int64_t E = (int64_t)C * H * W;
  dim3 grid((unsigned)((E + 1023) / 1024), (unsigned)N);
  mc_drop_flat_kernel<<<grid, 256, shmem, as_stream(stream)>>>(x, rnd, rand_image_stride, out, C, H, W, n_mc, gamma,
                                                              block_size, identity);
  return runia_check_launch();
}
