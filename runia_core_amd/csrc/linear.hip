// f4: the final linear layer that ReAct / ASH / DICE / ViM re-apply to (transformed) features
//     (reference inference/postprocessors.py:1193, 1441, 1466; RouteDICE.forward inference/funcs.py:180-189):
//     out [N, C] = min(x, clip) @ w.T + bias in f32.  Large heads take the 128 x 128 matrix-core tiles of nt_tile_f32.hpp; heads
//     of up to 16 classes a row-streaming kernel; a handful of rows / some hundred rows their own kernels (same bits as in a batch).
#include "nt_tile_f32.hpp"

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
