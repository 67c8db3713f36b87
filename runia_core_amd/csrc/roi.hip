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
                                                        int64_t total, int C, int H, int W, int PH, int PW,
                                                        float spatial_scale, int sampling_ratio, int aligned) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int pw = (int)(idx % PW), ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / ((int64_t)PW * PH)) % C);
    const int64_t k = idx / ((int64_t)PW * PH * C);
    const float* box = boxes + k * 4;
    const int b = batch_idx ? batch_idx[k] : 0;
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

}  // namespace

extern "C" int runia_roi_align_f32(const float* input, const float* boxes, const int* batch_idx, float* out, int64_t K,
                                   int64_t B, int C, int H, int W, int PH, int PW, double spatial_scale,
                                   int sampling_ratio, int aligned, runia_stream_t stream) {
  if (K < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return RUNIA_E_INVALID;
  if (K == 0) return RUNIA_OK;
  if (!input || !boxes || !out) return RUNIA_E_INVALID;
  if (B > 1 && !batch_idx) return RUNIA_E_INVALID;
  const int64_t total = K * C * PH * PW;
  roi_align_kernel<<<runia_stream_grid(total, 256), 256, 0, as_stream(stream)>>>(
      input, boxes, batch_idx, out, total, C, H, W, PH, PW, (float)spatial_scale, sampling_ratio, aligned);
  return runia_check_launch();
}
