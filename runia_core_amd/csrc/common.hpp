// Shared helpers for the gfx950 scoring kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <atomic>

#include "../../include/runia_hip.h"

#define RUNIA_WAVE 64

typedef double d4 __attribute__((ext_vector_type(4)));

static inline int runia_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RUNIA_OK : RUNIA_E_LAUNCH;
}

static inline hipStream_t as_stream(runia_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// runia_time_next_launch (lib.hip): a caller's event pair that the NEXT timed launch site of this thread attaches to its
// dispatch (hipExtLaunchKernelGGL: the events carry the kernel's own start / end timestamps - what rocprofv3's kernel trace
// reports - instead of bracketing the launch on the stream, which adds the dispatch gap behind the previous kernel, ~7 us).
struct RuniaTimedLaunch { hipEvent_t start, stop; };
RuniaTimedLaunch runia_take_timed_launch();
#define RUNIA_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...)                                                    \
  do {                                                                                                                 \
    const RuniaTimedLaunch tl_ = runia_take_timed_launch();                                                            \
    if (tl_.start) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream, tl_.start, tl_.stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__);                              \
  } while (0)

// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute belongs to the device that is current when it is set;
// a process that moves on to another GPU needs it again).  `done` is one static mask per call site: bit = device id.
// Two threads racing the first call both set the attribute, which is harmless.
static inline int runia_allow_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return RUNIA_E_LAUNCH;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return RUNIA_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return RUNIA_E_LAUNCH;
  done.fetch_or(bit, std::memory_order_release);
  return RUNIA_OK;
}

// Grid for thread-per-item streaming kernels: one trip per thread up to 2^20 workgroups (the kernels keep their
// grid-stride loop for more).  A 4 096-workgroup cap used to sit here; a grid just above it left a second trip to a
// fraction of the workgroups, and an address stream that follows the dispatch order reads faster (see below).
static inline unsigned runia_stream_grid(int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  const int64_t cap = 1 << 20;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// Grid for one-wave-per-row kernels: every wave takes ONE row and consecutive waves take consecutive rows, so the
// address stream follows the dispatch order.  Measured on the Energy shape (1 M x 1000 f32, tools/microbench/
// stream_rows.hip): 6.5 TB/s with 8 waves per workgroup and no loop, 5.5-5.6 TB/s as a 4 096-workgroup grid-stride
// loop (a flat float4 sweep of the same bytes reads 6.3 TB/s).  The kernels keep their loop for N beyond the grid limit.
constexpr int kRowWaves = 8;  // waves (rows) per workgroup of the wave-per-row kernels
static inline unsigned runia_rows_grid(int64_t rows, int waves = kRowWaves) {
  int64_t blocks = (rows + waves - 1) / waves;
  if (blocks > 0x7fffffffll) blocks = 0x7fffffffll;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

// Compute units of the current device (256 on MI355X); cached after the first call.
static inline int64_t runia_cu_count() {
  static int64_t cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// Summation order of torch.sum / torch.mean over a contiguous reduced dimension of a CPU f32 tensor (ATen
// SumKernel.cpp, cascade_sum) - the order in which the reference's fullmean (feature_extraction/utils.py:88-92) adds
// a row of the DropBlock output; restated in oracle/hotpath.py::torch_cpu_sum_lastdim and checked there against torch.
//   n < 8 : partial sums p[k] = a[k] (+ a[4+k]), remainder into p[0], then ((p0+p1)+p2)+p3: for n < 16 this is ONE
//           chain whose i-th term is torch_chain<N>(i):   n < 4: 0..n-1;  4 <= n < 8: 0, 4..n-1, 1, 2, 3;
//   n == 8: 0..7;   9 <= n < 16: 8..n-1, 0..7.
template <int N>
__host__ __device__ constexpr int torch_chain(int i) {
  static_assert(N >= 1 && N < 16, "single-chain range of the ATen row sum");
  if (N < 4 || N == 8) return i;
  if (N < 8) return i == 0 ? 0 : (i <= N - 4 ? i + 3 : i - (N - 4));
  return i < N - 8 ? 8 + i : i - (N - 8);
}

// The same order for a run-time length (rows of < 512 elements; ATen adds cascade levels beyond that): term(i) is
// the i-th element of the row.
template <class F>
__device__ __forceinline__ float torch_row_sum(F term, int n) {
  if (n < 8) {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    const int s = n >> 2;
    if (s) { p0 += term(0); p1 += term(1); p2 += term(2); p3 += term(3); }
    for (int i = 4 * s; i < n; ++i) p0 += term(i);
    p0 += p1; p0 += p2; p0 += p3;
    return p0;
  }
  const int vs = n >> 3, s = vs >> 2;
  float p[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int l = 0; l < 8; ++l) p[k][l] = 0.f;
  for (int j = 0; j < s; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int l = 0; l < 8; ++l) p[k][l] += term((j * 4 + k) * 8 + l);
  for (int i = 4 * s; i < vs; ++i)
#pragma unroll
    for (int l = 0; l < 8; ++l) p[0][l] += term(i * 8 + l);
#pragma unroll
  for (int k = 1; k < 4; ++k)
#pragma unroll
    for (int l = 0; l < 8; ++l) p[0][l] += p[k][l];
  float acc = 0.f;
  for (int k = vs * 8; k < n; ++k) acc += term(k);
#pragma unroll
  for (int l = 0; l < 8; ++l) acc += p[0][l];
  return acc;
}

__device__ __forceinline__ double kInfD() { return __builtin_inf(); }

__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double shfl_xor_f64(double v, int o) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return __hiloint2double(hi, lo);
}

// ---- GEN's transcendentals on the transcendental unit (v_exp_f32 / v_log_f32 are base-2 and ~1 ulp) -------------------
// The library expf / powf cost ~20 / ~170 vector instructions each; the wave-per-row GEN kernel spent half its issue slots
// in six powf per row (1 M x 1000, M = 100: 3.25 ms).  The raw instructions flush denormal inputs and results, which
// matters here: with gamma = 0.1 a probability of 1e-40 still contributes 1e-4, so both ends are rescaled.
__device__ __forceinline__ float exp2_nonpos(float a) {  // 2^a, a <= 0 (or NaN)
  const bool tiny = a < -126.0f;
  const float r = __builtin_amdgcn_exp2f(tiny ? a + 64.0f : a);
  return tiny ? r * 0x1p-64f : r;
}
// e^d, d <= 0 (or NaN): d log2(e) = n + r with the product formed once more inside an fma (r is exact to ~3e-8 whatever the
// size of d; rounding the product first put ~|d| 1e-7 into the exponent: 3e-6 relative at d = -30 - within the 1e-5 contract,
// but ten times the library's error in a sum over 1 000 classes), 2^r on the transcendental unit, 2^n by v_ldexp_f32 (which
// also forms denormal results, that v_exp_f32 alone flushes).  Below e^-104.5 the float32 result is 0.
__device__ __forceinline__ float exp_nonpos(float d) {
  const float n = __builtin_rintf(d * 1.44269504088896341f);
  float r = fmaf(d, 1.44269504088896341f, -n);
  r = fmaf(d, 1.92596299e-8f, r);  // log2(e) - float(log2(e))
  const float e = ldexpf(__builtin_amdgcn_exp2f(r), (int)n);
  return (d < -104.5f) ? 0.f : e;  // (also the -inf padding: n = -inf would make r NaN)
}
__device__ __forceinline__ float log2_nonneg(float x) {  // log2 x, x >= 0 (denormals included)
  const bool tiny = x < 1.17549435e-38f;
  const float r = __builtin_amdgcn_logf(tiny ? x * 0x1p+32f : x);
  return tiny ? r - 32.0f : r;
}
__device__ __forceinline__ float log_nonneg(float x) { return log2_nonneg(x) * 0.693147180559945309f; }  // ln x, x >= 0
// e / s from r ~ 1 / s: the product corrected by its remainder is the correctly rounded quotient (three instructions for
// the ten of an IEEE division).  GEN needs the quotient's bits: (1 - p)^gamma of a winner at p = 1 - 2.5e-7 moves by
// 2.5 % per ulp of p.
__device__ __forceinline__ float div_by_rcp(float e, float s, float r) {
  const float q = e * r;
  const float q1 = fmaf(fmaf(-q, s, e), r, q);
  return (q1 == q1) ? q1 : q;  // (inf / NaN operands: the plain product's result)
}
// p^gamma * (1 - p)^gamma as one exponential; pow(x, 0) is 1 for every x
__device__ __forceinline__ float gen_term(float p, float gamma) {
  if (gamma == 0.f) return 1.f;
  const float l = log2_nonneg(p) + __builtin_amdgcn_logf(1.0f - p);  // (1 - p is never denormal: >= 2^-24 or 0)
  const float a = gamma * l;
  // the argument can be positive (gamma < 0) or below the scaled range: the plain instruction covers both ends (inf / 0)
  return (a <= 0.f && a >= -180.f) ? exp2_nonpos(a) : __builtin_amdgcn_exp2f(a);
}
