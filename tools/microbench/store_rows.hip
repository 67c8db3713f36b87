// Store pattern of the stand-alone sampler (mc_stack): every thread owns one channel and produces n_mc = 16 samples
// that belong to 16 different rows of the [N*16, C] output.  What does the store shape cost on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o store_rows store_rows.hip && ./store_rows
//   A  one dword per lane per row (256 B per wave-instruction), blocks of 128 channels  [the shipped form]
//   B  same with 512-channel blocks (one image per workgroup)
//   C  transposed through LDS: 16-byte stores, a wave-instruction writes 1 KB of one row (512-channel blocks)
//   D  [N, C, 16] layout, 4 x 16-byte stores per lane (not the API layout; the ceiling of a thread-private store)
//   E  as A without the loads (stores only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int C = 512, HW = 16, NMC = 16;

__device__ __forceinline__ void make(const float* xc, float (&z)[NMC], bool load) {
  float u[HW];
  if (load) {
#pragma unroll
    for (int p = 0; p < HW / 4; ++p) {
      const float4 v = reinterpret_cast<const float4*>(xc)[p];
      u[4 * p] = v.x; u[4 * p + 1] = v.y; u[4 * p + 2] = v.z; u[4 * p + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int p = 0; p < HW; ++p) u[p] = (float)(threadIdx.x + p);
  }
#pragma unroll
  for (int s = 0; s < NMC; ++s) z[s] = u[s] + u[(s + 5) & 15] * 0.5f;
}

template <int BLOCK, bool LOAD>
__global__ __launch_bounds__(BLOCK) void k_a(const float* __restrict__ x, float* __restrict__ out, int64_t N) {
  constexpr unsigned chunks = C / BLOCK;
  const unsigned slot = blockIdx.x >> 3;
  const int64_t img = (int64_t)(slot / chunks) * 8 + (blockIdx.x & 7);
  if (img >= N) return;
  const int c = (int)(slot % chunks) * BLOCK + threadIdx.x;
  float z[NMC];
  make(x + (img * C + c) * (int64_t)HW, z, LOAD);
#pragma unroll
  for (int s = 0; s < NMC; ++s) out[(img * NMC + s) * (int64_t)C + c] = z[s];
}

__global__ __launch_bounds__(512) void k_c(const float* __restrict__ x, float* __restrict__ out, int64_t N) {
  __shared__ float tile[NMC][C + 4];
  const int64_t img = blockIdx.x;
  const int c = threadIdx.x;
  float z[NMC];
  make(x + (img * C + c) * (int64_t)HW, z, true);
#pragma unroll
  for (int s = 0; s < NMC; ++s) tile[s][c] = z[s];
  __syncthreads();
  float4* o = reinterpret_cast<float4*>(out + img * NMC * (int64_t)C);
#pragma unroll
  for (int t = 0; t < NMC * C / 4 / 512; ++t) {
    const int i = threadIdx.x + 512 * t;       // float4 index inside the image's [16, 512] block
    const int s = i / (C / 4), c4 = i % (C / 4);
    o[i] = *reinterpret_cast<const float4*>(&tile[s][4 * c4]);
  }
}

__global__ __launch_bounds__(128) void k_d(const float* __restrict__ x, float* __restrict__ out, int64_t N) {
  constexpr unsigned chunks = C / 128;
  const unsigned slot = blockIdx.x >> 3;
  const int64_t img = (int64_t)(slot / chunks) * 8 + (blockIdx.x & 7);
  if (img >= N) return;
  const int c = (int)(slot % chunks) * 128 + threadIdx.x;
  float z[NMC];
  make(x + (img * C + c) * (int64_t)HW, z, true);
  float4* o = reinterpret_cast<float4*>(out + (img * C + c) * (int64_t)NMC);
#pragma unroll
  for (int t = 0; t < 4; ++t) o[t] = make_float4(z[4 * t], z[4 * t + 1], z[4 * t + 2], z[4 * t + 3]);
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  const int64_t N = 10000;
  float *x, *out;
  CHECK(hipMalloc(&x, N * C * HW * 4));
  CHECK(hipMalloc(&out, N * NMC * C * 4));
  CHECK(hipMemset(x, 0, N * C * HW * 4));
  const double mb_rw = (double)N * C * (HW + NMC) * 4 * 1e-6, mb_w = (double)N * C * NMC * 4 * 1e-6;
  auto report = [&](const char* name, float ms, double mb) { printf("%-70s %8.4f ms  %7.1f GB/s\n", name, ms, mb / ms); fflush(stdout); };
  const unsigned g128 = (unsigned)(((N + 7) / 8) * 8 * (C / 128)), g512 = (unsigned)(((N + 7) / 8) * 8);
  report("A  dword per lane per row, 128-channel blocks", time_ms([&] { k_a<128, true><<<g128, 128>>>(x, out, N); }, 50), mb_rw);
  report("B  dword per lane per row, 512-channel blocks", time_ms([&] { k_a<512, true><<<g512, 512>>>(x, out, N); }, 50), mb_rw);
  report("C  LDS transpose, 16-byte stores, 1 KB of a row per wave-instruction", time_ms([&] { k_c<<<(unsigned)N, 512>>>(x, out, N); }, 50), mb_rw);
  report("D  [N, C, 16] layout, 16-byte stores (not the API layout)", time_ms([&] { k_d<<<g128, 128>>>(x, out, N); }, 50), mb_rw);
  report("E  A without the loads", time_ms([&] { k_a<128, false><<<g128, 128>>>(x, out, N); }, 50), mb_w);
  report("E  B without the loads", time_ms([&] { k_a<512, false><<<g512, 512>>>(x, out, N); }, 50), mb_w);
  return 0;
}
