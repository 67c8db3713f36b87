// f3: the step in front of the sampler for object-level inference (BASELINE config 4): torchvision.ops.roi_align as
// _dropblock_rois_get_entropy / _reduce_features_to_rois call it (reference feature_extraction/object_level.py:283-292,
// 340-349: output_size per hooked layer, spatial_scale = W_feat / W_img, sampling_ratio, aligned=True), written from the
// published algorithm (torchvision is absent from the image: roi_align kernel of torchvision/csrc/ops):
//   roi box scaled by spatial_scale, shifted by -0.5 when aligned; bin = roi / pooled; every bin averages a
//   grid_h x grid_w lattice of bilinear samples (grid = sampling_ratio, or ceil(roi / pooled) when <= 0);
//   samples further than one pixel outside the map contribute 0, coordinates are clamped to the map.
// Output [K, C, PH, PW] f32 = the (N, C, H, W) input of the sampler kernels (runia_mc_entropy_f32 for 2x2/4x4/7x7/8x8).
#include "common.hpp"

namespace {

__device__ __forceinline__ float bilinear(const float* __restrict__ in, int H, int W, float y, float x) {
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  const float v1 = in[y_low * W + x_low], v2 = in[y_low * W + x_high];
  const float v3 = in[y_high * W + x_low], v4 = in[y_high * W + x_high];
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ input, const float* __restrict__ boxes,
                                                        const int* __restrict__ batch_idx, float* __restrict__ out,
                                                        int64_t total, int64_t B, int C, int H, int W, int PH, int PW,
                                                        float spatial_scale, int sampling_ratio, int aligned) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pw = (int)(idx % PW), ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t k = idx / ((int64_t)PW * PH * C);
    const float* box = boxes + k * 4;
    const int b = batch_idx ? batch_idx[k] : 0;
    if (b < 0 || b >= B) {  // an image index outside the batch: a defined result (zeros), never a read outside the feature maps
      out[idx] = 0.f;
      continue;
    }
    const float offset = aligned ? 0.5f : 0.f;
    const float x1 = box[0] * spatial_scale - offset, y1 = box[1] * spatial_scale - offset;
    const float x2 = box[2] * spatial_scale - offset, y2 = box[3] * spatial_scale - offset;
    float roi_w = x2 - x1, roi_h = y2 - y1;
    if (!aligned) { roi_w = fmaxf(roi_w, 1.f); roi_h = fmaxf(roi_h, 1.f); }
    const float bin_h = roi_h / (float)PH, bin_w = roi_w / (float)PW;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)PH);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)PW);
    const float count = fmaxf((float)(grid_h * grid_w), 1.f);
    const float* in = input + ((int64_t)b * C + c) * (int64_t)H * W;
    float acc = 0.f;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)grid_w;
        acc += bilinear(in, H, W, y, x);
      }
    }
    out[idx] = acc / count;
  }
}

// ---- roi_align folded into the sampler's load (fused.hip, roi_load_map) ---------------------------------------------------
// The feature map in NHWC, so that the 64 channels of a wave read one bilinear tap as one contiguous run.
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                            int64_t HW) {
  __shared__ float tile[64][65];
  const int64_t b = blockIdx.z;
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = c0 + ty + 4 * r;
    const int64_t p = p0 + tx;
    tile[ty + 4 * r][tx] = (c < C && p < HW) ? in[(b * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t p = p0 + ty + 4 * r;
    const int c = c0 + tx;
    if (c < C && p < HW) out[(b * HW + p) * C + c] = tile[tx][ty + 4 * r];
  }
}

// mirrors RoiSource of fused.hip
struct RoiSourceHost {
  const float* quads;
  const unsigned* table;
  int64_t image_bytes;
  int C;
  int roi_dwords;
};
static_assert(sizeof(RoiSourceHost) <= 64, "the source description fits the head of the table");

// One thread per sample row or sample column of a ROI (PH * G rows, then PW * G columns).  The coordinates, the validity
// test, the clamping and the weights are those of `bilinear` / roi_align_kernel above, expression for expression (they are
// separable: y and x never meet before the four products).
__global__ __launch_bounds__(256) void roi_sample_table_kernel(RoiSourceHost src, const float* __restrict__ boxes,
                                                                const int* __restrict__ batch_idx, unsigned* __restrict__ tab,
                                                                int64_t K, int64_t B, int C, int H, int W, int PH, int PW,
                                                                float spatial_scale, int G, int aligned) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<RoiSourceHost*>(tab) = src;
  unsigned* roi_tab = tab + 16;  // 64 bytes of RoiSource in front
  const int nrows = PH * G, ncols = PW * G, per_roi = nrows + ncols, roi_dwords = 8 + 4 * per_roi;
  const int64_t total = K * per_roi;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t k = idx / per_roi;
    const int j = (int)(idx % per_roi);
    const bool is_row = j < nrows;
    const int sidx = is_row ? j : j - nrows;
    const int bin = sidx / G, sub = sidx % G;
    const float* box = boxes + k * 4;
    const float offset = aligned ? 0.5f : 0.f;
    const float lo = box[is_row ? 1 : 0] * spatial_scale - offset, hi = box[is_row ? 3 : 2] * spatial_scale - offset;
    float extent = hi - lo;
    if (!aligned) extent = fmaxf(extent, 1.f);
    const int P = is_row ? PH : PW, L = is_row ? H : W;
    const float bin_sz = extent / (float)P;
    float t = lo + (float)bin * bin_sz + ((float)sub + 0.5f) * bin_sz / (float)G;
    unsigned* hd = roi_tab + k * (int64_t)roi_dwords;
    unsigned* e = hd + 8 + 4 * j;
    // An image index outside [0, B) would move the loader's buffer descriptor off the feature maps (its own range check then
    // protects nothing): such a ROI is made of "outside" samples only - every tap beyond the buffer, weights 0 - on image 0,
    // i.e. the zeros runia_roi_align_f32 writes for it.
    const int image = batch_idx ? batch_idx[k] : 0;
    const bool bad_image = image < 0 || image >= B;
    if (j == 0) {
      hd[0] = bad_image ? 0u : (unsigned)image;
#pragma unroll
      for (int q = 1; q < 8; ++q) hd[q] = 0u;
    }
    const bool outside = bad_image || (t < -1.0f || t > (float)L);
    // a sample row / column more than a pixel outside the map contributes nothing (roi_align's `continue`): its byte offset is
    // the size of the image - every tap of it lies beyond the buffer the loader reads through, and such a load returns 0 -
    // and its weights are 0, so the sample is +0.0 without a test in the loader (the scalar offset is part of the range check
    // on gfx950: tools/microbench/buffer_soffset_range.hip)
    if (outside) {
      e[0] = e[1] = (unsigned)src.image_bytes;
      e[2] = e[3] = 0u;
      continue;
    }
    if (t <= 0.f) t = 0.f;
    int low = (int)t, high;
    if (low >= L - 1) { high = low = L - 1; t = (float)low; } else { high = low + 1; }
    const float l = t - (float)low, h = 1.f - l;
    const unsigned step = is_row ? (unsigned)W * (unsigned)C * 4u : (unsigned)C * 4u;
    e[0] = (unsigned)low * step;
    e[1] = (unsigned)high * step;
    e[2] = __float_as_uint(h);
    e[3] = __float_as_uint(l);
  }
}

}  // namespace

size_t runia_roi_sample_table_bytes(int64_t K, int PH, int PW, int G) {
  return 64 + (size_t)K * (size_t)(8 + 4 * (PH * G + PW * G)) * 4;
}

int runia_roi_sample_table(const float* feat_nhwc, const float* boxes, const int* batch_idx, void* table, size_t table_bytes,
                           int64_t K, int64_t B, int C, int H, int W, int PH, int PW, double spatial_scale, int G, int aligned,
                           hipStream_t s) {
  if (table_bytes < runia_roi_sample_table_bytes(K, PH, PW, G) || PH * G > 32 || PW * G > 32) return RUNIA_E_WORKSPACE;
  if ((int64_t)H * W * C * 4 >= (int64_t)1 << 30) return RUNIA_E_INVALID;  // (offsets of outside samples: row + column + channel stay below 2^32)
  RoiSourceHost src;
  src.quads = feat_nhwc;
  src.table = reinterpret_cast<const unsigned*>(table) + 16;
  src.image_bytes = (int64_t)H * W * C * 4;
  src.C = C;
  src.roi_dwords = 8 + 4 * (PH * G + PW * G);
  const int64_t total = K * (PH * G + PW * G);
  roi_sample_table_kernel<<<runia_stream_grid(total, 256), 256, 0, s>>>(src, boxes, batch_idx, reinterpret_cast<unsigned*>(table),
                                                                        K, B, C, H, W, PH, PW, (float)spatial_scale, G, aligned);
  return runia_check_launch();
}

extern "C" int runia_nchw_to_nhwc_f32(const float* in, float* out, int64_t B, int C, int64_t HW, runia_stream_t stream) {
  if (B < 0 || C <= 0 || HW <= 0 || B > 65535) return RUNIA_E_INVALID;
  if (B == 0) return RUNIA_OK;
  if (!in || !out) return RUNIA_E_INVALID;
  const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
  nchw_to_nhwc_kernel<<<grid, 256, 0, as_stream(stream)>>>(in, out, C, HW);
  return runia_check_launch();
}

extern "C" int runia_roi_align_f32(const float* input, const float* boxes, const int* batch_idx, float* out, int64_t K,
                                   int64_t B, int C, int H, int W, int PH, int PW, double spatial_scale,
                                   int sampling_ratio, int aligned, runia_stream_t stream) {
  if (K < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return RUNIA_E_INVALID;
  if (K == 0) return RUNIA_OK;
  if (!input || !boxes || !out) return RUNIA_E_INVALID;
  if (B > 1 && !batch_idx) return RUNIA_E_INVALID;
  const int64_t total = K * C * PH * PW;
  roi_align_kernel<<<runia_stream_grid(total, 256), 256, 0, as_stream(stream)>>>(
      input, boxes, batch_idx, out, total, B, C, H, W, PH, PW, (float)spatial_scale, sampling_ratio, aligned);
  return runia_check_launch();
}
