// Sustained v_mfma_f64_16x16x4_f64 rate on gfx950: NACC independent accumulators per wave, WAVES waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void spin(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// same loop with 8 distinct A and 4 distinct B operands (the shape of a real tile: acc[a][c] += A[a] * B[c])
template <int NACC>
__global__ __launch_bounds__(256) void spin_ops(double* out, const double* in, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x + 256 * i];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x + 256 * (8 + i)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(i / 4 + it) & 7], b[i & 3], acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run_ops(int wgs_per_cu, int cus) {
  const int iters = 2000;
  double *out, *in;
  hipMalloc(&out, sizeof(double) * 256 * wgs_per_cu * cus);
  hipMalloc(&in, sizeof(double) * 256 * 12);
  hipMemset(in, 0, sizeof(double) * 256 * 12);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin_ops<NACC><<<wgs_per_cu * cus, 256>>>(out, in, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin_ops<NACC><<<wgs_per_cu * cus, 256>>>(out, in, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)wgs_per_cu * cus * 4 * iters * NACC;
  printf("distinct operands, NACC %2d waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.1f SIMD-cycles/MFMA at 2.4 GHz\n", NACC,
         wgs_per_cu, ms, mfmas * 2048 / ms * 1e-9, ms * 1e-3 * 2.4e9 * cus * 4 / mfmas);
  hipFree(out); hipFree(in);
}

template <int NACC>
void run(int wgs_per_cu, int cus) {
  const int iters = 2000;
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * wgs_per_cu * cus);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  spin<NACC><<<wgs_per_cu * cus, 256>>>(out, 10, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  spin<NACC><<<wgs_per_cu * cus, 256>>>(out, iters, 1.0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)wgs_per_cu * cus * 4 * iters * NACC;
  printf("NACC %2d waves/SIMD %d: %.3f ms, %.1f TFLOP/s, %.1f SIMD-cycles/MFMA at 2.4 GHz\n", NACC, wgs_per_cu, ms,
         mfmas * 2048 / ms * 1e-9, ms * 1e-3 * 2.4e9 * cus * 4 / mfmas);
  hipFree(out);
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, cus);
  run<1>(1, cus); run<4>(1, cus); run<8>(1, cus); run<16>(1, cus);
  run<4>(2, cus); run<8>(2, cus); run<4>(4, cus);
  run_ops<8>(1, cus); run_ops<16>(1, cus); run_ops<8>(2, cus); run_ops<16>(2, cus);
  return 0;
}
