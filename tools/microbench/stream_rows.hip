// How fast can "one wave per row" kernels read a row-major f32 matrix on gfx950?  (Energy / MSP, normalizer shape.)
//   hipcc --offload-arch=gfx950 -O3 -o stream_rows stream_rows.hip && ./stream_rows [N] [C]
// Variants of the log-sum-exp row kernel (rowwise.hip) that differ only in how the loads are issued; the
// arithmetic (row max, sum of expf(x - max)) is the same in all of them.  A read-only sweep of the same bytes with
// no arithmetic gives the ceiling of the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool NT>
__device__ __forceinline__ float4 ld4(const float4* p) {
  if constexpr (NT) {
    float4 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    v.z = __builtin_nontemporal_load(&p->z);
    v.w = __builtin_nontemporal_load(&p->w);
    return v;
  } else {
    return *p;
  }
}

template <int NCH>
__device__ __forceinline__ void load_row(const float* x, int64_t row, int64_t C, int lane, float4 (&v)[NCH]) {
  const float4* p4 = reinterpret_cast<const float4*>(x + row * C);
  const int n4 = (int)(C >> 2);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane + 64 * c;
    v[c] = (i < n4) ? p4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  }
}

template <int NCH>
__device__ __forceinline__ float row_lse(const float4 (&v)[NCH]) {
  float m = -INFINITY, s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) m = fmaxf(m, fmaxf(fmaxf(v[c].x, v[c].y), fmaxf(v[c].z, v[c].w)));
  m = wmax(m);
#pragma unroll
  for (int c = 0; c < NCH; ++c) s += (expf(v[c].x - m) + expf(v[c].y - m)) + (expf(v[c].z - m) + expf(v[c].w - m));
  s = wsum(s);
  return m + logf(s);
}

// A: the shipped form - grid-stride loop, one row in flight per wave
template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lse_a(const float* __restrict__ x, float* out, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < N; row += stride) {
    float4 v[NCH];
    load_row<NCH>(x, row, C, lane, v);
    const float r = row_lse<NCH>(v);
    if (lane == 0) out[row] = r;
  }
}

// B: next row's loads issued before this row's arithmetic (two rows of registers)
template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lse_b(const float* __restrict__ x, float* out, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t row = (int64_t)blockIdx.x * WAVES + wave;
  if (row >= N) return;
  float4 cur[NCH], nxt[NCH];
  load_row<NCH>(x, row, C, lane, cur);
  for (; row < N; row += stride) {
    const int64_t nr = row + stride;
    if (nr < N) load_row<NCH>(x, nr, C, lane, nxt);
    const float r = row_lse<NCH>(cur);
    if (lane == 0) out[row] = r;
#pragma unroll
    for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
  }
}

// C: a wave owns a CONTIGUOUS run of rows (rows_per_wave), prefetching as B; consecutive waves read consecutive runs
template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lse_c(const float* __restrict__ x, float* out, int64_t N, int64_t C,
                                                     int rows_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t row = ((int64_t)blockIdx.x * WAVES + wave) * rows_per_wave;
  const int64_t end = (row + rows_per_wave < N) ? row + rows_per_wave : N;
  if (row >= end) return;
  float4 cur[NCH], nxt[NCH];
  load_row<NCH>(x, row, C, lane, cur);
  for (; row < end; ++row) {
    if (row + 1 < end) load_row<NCH>(x, row + 1, C, lane, nxt);
    const float r = row_lse<NCH>(cur);
    if (lane == 0) out[row] = r;
#pragma unroll
    for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
  }
}

// D: one row per wave, no loop (N / WAVES workgroups)
template <int NCH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lse_d(const float* __restrict__ x, float* out, int64_t N, int64_t C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * WAVES + wave;
  if (row >= N) return;
  float4 v[NCH];
  load_row<NCH>(x, row, C, lane, v);
  const float r = row_lse<NCH>(v);
  if (lane == 0) out[row] = r;
}

// R: read-only ceiling: flat float4 sweep, 4 loads in flight per lane, no row structure
__global__ __launch_bounds__(256) void sweep(const float4* __restrict__ x, float* out, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
    float4 a = x[i], b = (i + 256 < n4) ? x[i + 256] : make_float4(0, 0, 0, 0);
    float4 c = (i + 512 < n4) ? x[i + 512] : make_float4(0, 0, 0, 0), d = (i + 768 < n4) ? x[i + 768] : make_float4(0, 0, 0, 0);
    acc += (a.x + b.y) + (c.z + d.w);
  }
  if (acc == 1234.5f) out[0] = acc;
}

template <typename F>
static float time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
  const int64_t C = argc > 2 ? atoll(argv[2]) : 1000;
  const size_t bytes = (size_t)N * C * 4;
  float *x, *out, *ref;
  CHECK(hipMalloc(&x, bytes));
  CHECK(hipMalloc(&out, N * 4));
  CHECK(hipMalloc(&ref, N * 4));
  {  // fill with a cheap pattern on the device
    float* h = (float*)malloc(bytes);
    unsigned s = 12345u;
    for (size_t i = 0; i < (size_t)N * C; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (8.f / 16777216.f) - 4.f; }
    CHECK(hipMemcpy(x, h, bytes, hipMemcpyHostToDevice));
    free(h);
  }
  const double gb = (double)bytes * 1e-9;
  constexpr int NCH = 4;
  if (C > 1024 || (C & 3)) { printf("C must be a multiple of 4, <= 1024\n"); return 1; }
  auto report = [&](const char* name, float ms) { printf("%-58s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); fflush(stdout); };
  const int reps = 20;
  report("R  flat float4 sweep, 2048 wgs", time_ms([&] { sweep<<<2048, 256>>>((const float4*)x, out, (int64_t)N * C / 4); }, reps));
  report("R  flat float4 sweep, 8192 wgs", time_ms([&] { sweep<<<8192, 256>>>((const float4*)x, out, (int64_t)N * C / 4); }, reps));
  lse_d<NCH, 4><<<(unsigned)((N + 3) / 4), 256>>>(x, ref, N, C);
  for (unsigned g : {2048u, 4096u, 8192u, 16384u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "A  grid-stride, 4 waves/wg, %u wgs", g);
    report(nm, time_ms([&] { lse_a<NCH, 4><<<g, 256>>>(x, out, N, C); }, reps));
  }
  for (unsigned g : {1024u, 2048u, 4096u, 8192u}) {
    char nm[96];
    snprintf(nm, sizeof nm, "B  grid-stride + next-row prefetch, 4 waves/wg, %u wgs", g);
    report(nm, time_ms([&] { lse_b<NCH, 4><<<g, 256>>>(x, out, N, C); }, reps));
  }
  for (int rpw : {4, 8, 16, 32}) {
    char nm[96];
    const unsigned g = (unsigned)((N + 4 * rpw - 1) / (4 * rpw));
    snprintf(nm, sizeof nm, "C  contiguous %d rows per wave + prefetch, %u wgs", rpw, g);
    report(nm, time_ms([&] { lse_c<NCH, 4><<<g, 256>>>(x, out, N, C, rpw); }, reps));
  }
  report("D  one row per wave, 4 waves/wg", time_ms([&] { lse_d<NCH, 4><<<(unsigned)((N + 3) / 4), 256>>>(x, out, N, C); }, reps));
  report("D  one row per wave, 8 waves/wg", time_ms([&] { lse_d<NCH, 8><<<(unsigned)((N + 7) / 8), 512>>>(x, out, N, C); }, reps));
  report("D  one row per wave, 1 wave/wg", time_ms([&] { lse_d<NCH, 1><<<(unsigned)N, 64>>>(x, out, N, C); }, reps));
  // agreement of the last variant with the first
  float *a = (float*)malloc(N * 4), *b = (float*)malloc(N * 4);
  CHECK(hipMemcpy(a, out, N * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(b, ref, N * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int64_t i = 0; i < N; ++i) bad += (a[i] != b[i]);
  printf("mismatching rows: %d\n", bad);
  return 0;
}
