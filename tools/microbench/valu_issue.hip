// Issue cost of the vector instructions K1 is made of, per SIMD, at 1 / 2 / 4 / 8 waves per SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
// Every wave runs ITER trips of 16 independent instructions of one kind; cycles per instruction per SIMD =
// elapsed * clock / (instructions issued on one SIMD).  The clock is taken as s_memtime ticks (shader clock)
// measured in-kernel, so DVFS does not enter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;

template <int KIND>
__global__ void bench(float* out, unsigned long long* cyc) {
  float a[16], b = out[1], c = out[2];
  double d[16], db = (double)out[3], dc = (double)out[4];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[16], pb = {b, c}, pc = {c, b};
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = out[5 + i] + threadIdx.x; d[i] = (double)a[i]; p[i] = (f2){a[i], a[i] + 1.f}; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (KIND == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if constexpr (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (KIND == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if constexpr (KIND == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if constexpr (KIND == 6) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if constexpr (KIND == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if constexpr (KIND == 8) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
      if constexpr (KIND == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
      if constexpr (KIND == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
      if constexpr (KIND == 11) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 12) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 13) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (KIND == 14) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if constexpr (KIND == 15) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      if constexpr (KIND == 16) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[i]));
      if constexpr (KIND == 17) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (KIND == 18) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc");
      if constexpr (KIND == 19) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
      if constexpr (KIND == 20) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 21) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 22) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 23) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 24) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 25) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (KIND == 26) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
      if constexpr (KIND == 27) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      // mixes in K1's proportions (predicted from the single-kind rows: (4.0 + 4.3 + 4.1 + 2.5) / 4 = 3.7)
      if constexpr (KIND == 30) {  // kinds alternate instruction by instruction
        if (i % 4 == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
        if (i % 4 == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        if (i % 4 == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (i % 4 == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
      // three different vector registers per instruction (the rows above reuse two operands throughout)
      if constexpr (KIND == 32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
      if constexpr (KIND == 33) asm volatile("v_add_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[(i + 5) & 15]), "v"(d[(i + 10) & 15]));
      if constexpr (KIND == 34) asm volatile("v_max_f64 %0, %1, %2" : "=v"(d[i]) : "v"(d[(i + 5) & 15]), "v"(d[(i + 10) & 15]));
      if constexpr (KIND == 35) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 5) & 15]), "s"(pb));
      if constexpr (KIND == 36) asm volatile("v_min_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[(i + 5) & 15]), "v"(a[(i + 10) & 15]));
      // dependent chains of 4 (K1's window scan and sort are chains, not 16 independent instructions)
      if constexpr (KIND == 37) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(d[4 + (i & 3)]));
      if constexpr (KIND == 38) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[0]) : "v"(d[4 + (i & 3)]));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y;
  if (s == 123.456f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
int run(const char* name, float* buf, unsigned long long* cyc) {
  printf("%-16s", name);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD; wps blocks per CU
    const int blocks = 256 * wps;
    bench<KIND><<<blocks, 256>>>(buf, cyc);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    bench<KIND><<<blocks, 256>>>(buf, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // in-kernel cycles of one wave for ITER*16 instructions; per SIMD `wps` waves ran concurrently
    // in-kernel shader cycles of one wave for its ITER*16 instructions; `wps` waves share the SIMD, so the SIMD retires one
    // instruction every c / (ITER*16*wps) cycles once it is saturated (wps >= 2)
    const double per_instr_per_simd = (double)c / (double)(ITER * 16) / wps;
    printf("  wps=%d: %5.2f (%.3f ms)", wps, per_instr_per_simd, ms);
  }
  printf("\n");
  return 0;
}

int main() {
  float* buf; unsigned long long* cyc;
  CHECK(hipMalloc(&buf, 4096)); CHECK(hipMemset(buf, 0, 4096)); CHECK(hipMalloc(&cyc, 8));
  run<0>("v_min_f32", buf, cyc); run<11>("v_add_f32", buf, cyc); run<1>("v_fma_f32", buf, cyc); run<10>("v_mov_b32", buf, cyc);
  run<12>("v_med3_f32", buf, cyc); run<17>("v_max3_f32", buf, cyc);
  run<2>("v_pk_fma_f32", buf, cyc); run<3>("v_pk_mul_f32", buf, cyc); run<13>("v_pk_add_f32", buf, cyc);
  run<4>("v_add_f64", buf, cyc); run<5>("v_max_f64", buf, cyc); run<6>("v_min_f64", buf, cyc); run<7>("v_mul_f64", buf, cyc);
  run<8>("v_fma_f64", buf, cyc); run<9>("v_cvt_f64_f32", buf, cyc); run<14>("v_log_f32", buf, cyc); run<15>("v_rcp_f64", buf, cyc);
  run<16>("v_frexp_mant_f64", buf, cyc); run<18>("v_cmp_lt_f64", buf, cyc); run<19>("v_cndmask_b32", buf, cyc);
  run<20>("v_min_u32", buf, cyc); run<21>("v_max_i32", buf, cyc); run<27>("v_min3_u32", buf, cyc); run<22>("v_or_b32", buf, cyc);
  run<23>("v_xor_b32", buf, cyc); run<24>("v_sub_f32", buf, cyc); run<25>("v_mul_f32", buf, cyc); run<26>("v_cvt_f32_f64", buf, cyc);
  run<30>("mix alternating", buf, cyc);
  run<32>("v_fma_f32 3reg", buf, cyc); run<36>("v_min_f32 3reg", buf, cyc); run<33>("v_add_f64 3reg", buf, cyc);
  run<34>("v_max_f64 3reg", buf, cyc); run<35>("v_pk_fma v,s,v", buf, cyc);
  run<37>("v_max_f64 4 chains", buf, cyc); run<38>("v_max_f64 1 chain", buf, cyc);
  return 0;
}
