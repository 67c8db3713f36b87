// Do vector-ALU instructions of one wave issue while another wave of the same SIMD streams v_mfma_f64_16x16x4_f64?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run the MFMA loop, waves 4-7 run an FMA loop.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void both(double* out, int mfma_iters, int valu_iters, double a0, float f0) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (mfma_iters == 0) return;
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = a0 - threadIdx.x;
    for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (valu_iters == 0) return;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f0 + i + threadIdx.x;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0000001f, 0.5f);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

static float run(double* out, int cus, int mi, int vi) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  both<<<cus, 512>>>(out, mi ? 10 : 0, vi ? 10 : 0, 1.0, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  both<<<cus, 512>>>(out, mi, vi, 1.0, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * 512 * cus);
  const int mi = 2000;            // 16 000 MFMAs per wave  (~1.1 M cycles at 68 cycles each)
  const int vi = 34000;           // 272 000 FMAs per wave (~1.1 M cycles at 4 cycles each)
  const float tm = run(out, cus, mi, 0), tv = run(out, cus, 0, vi), tb = run(out, cus, mi, vi);
  printf("MFMA wave alone %.3f ms, VALU wave alone %.3f ms, both on the same SIMD %.3f ms\n", tm, tv, tb);
  printf("-> %s\n", tb < 0.75f * (tm + tv) ? "the two streams overlap" : "the two streams serialise (shared issue)");
  return 0;
}
