// Shared helpers for the gfx950 scoring kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/runia_hip.h"

#define RUNIA_WAVE 64

typedef double d4 __attribute__((ext_vector_type(4)));

static inline int runia_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RUNIA_OK : RUNIA_E_LAUNCH;
}

static inline hipStream_t as_stream(runia_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Grid for grid-stride streaming kernels: enough workgroups to fill 256 CUs several
// times over, capped so that launch overhead stays flat (guide: Guideline 11).
static inline unsigned runia_stream_grid(int64_t work_items, int per_block) {
  int64_t blocks = (work_items + per_block - 1) / per_block;
  const int64_t cap = 256 * 16;
  if (blocks > cap) blocks = cap;
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
