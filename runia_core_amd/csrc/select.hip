// Setup-time order statistics of a flat float32 array on the device: the histogram pass of a radix select.
//   ReAct / DICE+ReAct take np.percentile(train_features.flatten(), p) as their clipping threshold (reference
//   inference/postprocessors.py:1441, 1466: 102 M activations at cfg3 size - 0.68 s of NumPy's introselect per fit, the largest
//   item of a `calculate_all_baselines` loop once the fits run on the device).  The percentile is an interpolation between two
//   neighbouring order statistics; an order statistic of 32-bit keys is three histogram passes (11 + 11 + 10 bits): each pass
//   counts, among the elements whose key starts with the prefix found so far, the next digit; the host picks the bin that holds
//   the rank and narrows the prefix (runia_core_amd/_hip.py: kth_smallest_flat).  One read of the array per pass at HBM rate.
// Keys: the usual order-preserving map of a float's bits (negative: all bits flipped, else the sign bit set): ascending in value,
// -0.0 directly below +0.0, NaNs above +inf (the caller rejects arrays with NaNs: NumPy's percentile returns NaN for them).
#include "common.hpp"

namespace {

constexpr int kSelectBins = 2048;

__device__ __forceinline__ unsigned select_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ x, int64_t n, unsigned prefix, unsigned prefix_mask,
                                                           int shift, unsigned* __restrict__ hist) {
  __shared__ unsigned bins[kSelectBins];
  for (int i = threadIdx.x; i < kSelectBins; i += 256) bins[i] = 0u;
  __syncthreads();
  // +0.0 is a seventh of a ReLU layer's output: its hits are counted once per wave (one LDS atomic instead of up to 64 on one word)
  const unsigned zero_key = 0x80000000u;
  const bool zero_in = (zero_key & prefix_mask) == prefix;
  const unsigned zero_bin = (zero_key >> shift) & (kSelectBins - 1);
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n; i0 += stride) {  // (uniform trip count per wave: the ballot sees whole waves)
    const int64_t i = i0 + threadIdx.x;
    const bool have = i < n;
    const unsigned key = have ? select_key(x[i]) : 0u;
    const bool is_zero = have && key == zero_key;
    const unsigned long long zm = __ballot(is_zero);
    if (zero_in && lane == 0 && zm) atomicAdd(&bins[zero_bin], (unsigned)__popcll(zm));
    if (have && !is_zero && (key & prefix_mask) == prefix) atomicAdd(&bins[(key >> shift) & (kSelectBins - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSelectBins; i += 256)
    if (bins[i]) atomicAdd(&hist[i], bins[i]);
}

}  // namespace

extern "C" int runia_select_hist_f32(const float* x, unsigned* hist, int64_t n, unsigned prefix, unsigned prefix_mask, int shift,
                                     runia_stream_t stream) {
  if (n < 0 || n >= ((int64_t)1 << 32) || shift < 0 || shift > 31) return RUNIA_E_INVALID;
  if (!hist || (n > 0 && !x)) return RUNIA_E_INVALID;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(hist, 0, kSelectBins * sizeof(unsigned), s) != hipSuccess) return RUNIA_E_LAUNCH;
  if (n == 0) return RUNIA_OK;
  int64_t grid = (n + 255) / 256;
  const int64_t cap = (int64_t)runia_cu_count() * 16;
  if (grid > cap) grid = cap;
  select_hist_kernel<<<(unsigned)grid, 256, 0, s>>>(x, n, prefix, prefix_mask, shift, hist);
  return runia_check_launch();
}
