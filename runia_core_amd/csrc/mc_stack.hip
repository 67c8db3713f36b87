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
  // 2) per-layer rescale numel / sum(bm): one wave per layer
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int s = wave; s < n_mc; s += 4) {
      float acc = 0.f;
      for (int p = lane; p < HW; p += 64) acc += bm[s * HW + p];
      acc = wave_sum_f32(acc);  // exact: small integer counts
      if (lane == 0) scale[s] = (float)HW / acc;
    }
  }
  __syncthreads();
  // 3) masked means, one channel per thread
  const int c = blockIdx.x * 256 + tid;
  if (c >= C) return;
  const float* xc = x + (img * C + c) * (int64_t)HW;
  const float invW = 1.0f / (float)W, invH = 1.0f / (float)H;
  float* o = out + img * n_mc * (int64_t)C + c;
  for (int s = 0; s < n_mc; ++s) {
    const float* b = bm + s * HW;
    float col = 0.f;
    for (int y = 0; y < H; ++y) {
      float rowsum = 0.f;
      for (int xw = 0; xw < W; ++xw) rowsum += xc[y * W + xw] * b[y * W + xw];
      col += rowsum * invW;
    }
    o[(int64_t)s * C] = (col * invH) * scale[s];
  }
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
  const size_t shmem = ((size_t)n_mc * H * W + n_mc) * sizeof(float);
  if (shmem > 64 * 1024) return RUNIA_E_INVALID;
  dim3 grid((C + 255) / 256, (unsigned)N);
  mc_stack_kernel<<<grid, 256, shmem, as_stream(stream)>>>(x, rnd, rand_image_stride, out, C, H, W, n_mc,
                                                          gamma, block_size, identity);
  return runia_check_launch();
}
