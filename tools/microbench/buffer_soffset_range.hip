// Is the scalar offset of a raw buffer load part of the range check on gfx950?  (The fused ROI loader marks a sample outside
// the map by a scalar offset >= num_records and expects 0 back.)  A 4 KiB allocation of 1.0f, a descriptor over its first
// 1 KiB: lane l loads dword l with soffset 0 (expected 1), soffset 2048 (0 if the check includes it, 1 if not), and
// voffset 2048 (0 either way).
// hipcc --offload-arch=gfx950 -O2 buffer_soffset_range.hip -o /tmp/bsr && /tmp/bsr
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float* base, float* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 1024, 0x00020000);
  const int l = threadIdx.x;
  out[l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, 0, 0));
  out[64 + l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, 2048, 0));
  out[128 + l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4 + 2048, 0, 0));
  out[192 + l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, l * 4, 1024 - 128, 0));  // lanes 32.. cross the end
#endif
}
int main() {
  float *d, *o, h[1024], ho[256];
  for (int i = 0; i < 1024; ++i) h[i] = 1.0f;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o);
  hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
  printf("soffset 0: %g %g | soffset 2048 (beyond num_records): %g %g | voffset 2048: %g %g | soffset 896, lanes 31 / 32 (896 + 124 = 1020 in, 1024 out): %g %g\n",
         ho[0], ho[63], ho[64], ho[127], ho[128], ho[191], ho[192 + 31], ho[192 + 32]);
  return 0;
}
