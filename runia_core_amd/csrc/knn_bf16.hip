// a8 kNN, candidate distances of large problems on the bf16 matrix cores (gfx950: v_mfma_f32_32x32x16_bf16, 16x the
// per-instruction work of v_mfma_f32_32x32x2_f32 at the same issue cost).
//
// An f32 value is split into three bf16 pieces, x = h + m + l exactly up to 2^-24 |x| (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m): 8 significand bits each, f32's exponent range, so no scaling and no range restriction).  The dot
// product q.b is then the sum of the piece products; the six of order <= 2^-16 are kept,
//     q.b ~= l.h + h.l + m.m + m.h + h.m + h.h        (dropped: m.l + l.m + l.l <= 3 * 2^-24 |q||b|),
// each an exact bf16 product accumulated in f32 by the matrix cores - the accuracy of the f32 contraction this replaces
// (whose own accumulation error is ~ sqrt(D) 2^-24), at 6/16 of its matrix-pipe time; or only the three of order
// <= 2^-8 (m.h + h.m + h.h, error <= 3 * 2^-16 |q||b|, 3/16 of the time) - the default, KNN16_TERMS.  The result feeds
// the SAME selection + exact f32 re-measurement as the f32 kernel's distances (pairwise.hip, kth_select_range_kernel),
// whose window is widened to twice this kernel's error bound (runia_knn16_refine_rel): every bank row that could be the
// k-th neighbour is re-measured with exact f32 differences, so the caller gets the exactly re-measured k-th distance
// either way; this kernel only decides which bank rows are looked at.
//
// Kernel shape.  One workgroup = 256 queries x 256 bank rows, 4 waves of 128 x 128 (4 x 4 MFMA tiles, 256 accumulator
// registers, one wave per SIMD).  K runs over the piece pairs x D in stages of 64: a stage of a tile is 256 rows x 128
// bytes, staged by `buffer_load_dwordx4 ... lds` (16 per wave and stage, no staging registers, no ds_write) into one of
// two LDS buffers (2 x 64 KB).  LDS slot (16 bytes = 8 consecutive k of one row) of (row, kgroup g) is
// row * 8 + (g ^ ((row >> 1) & 7)): the eight lanes that fetch one row read its whole 128-byte line, and the
// ds_read_b128 of an MFMA operand (32 consecutive rows, one kgroup per half-wave) is conflict-free: the LDS serves a b128
// read in four groups of 16 lanes - {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 - and within a group the
// rows of one parity have row >> 1 in {0, 1, 6, 7, 10, 11, 12, 13} or {2, 3, 4, 5, 8, 9, 14, 15}, eight different values
// mod 8, so the 16 slots fall into the 16 different 16-byte bank groups.  The pipeline is described at the loop.
#include "common.hpp"

#include <cstdint>
#include <type_traits>

#ifndef KNN16_SAMECHUNK
#define KNN16_SAMECHUNK 0  // timing experiments only
#endif

namespace runia_knn16 {

constexpr int TQ = 256, TB = 256;            // tile
[[maybe_unused]] constexpr int KC = 64;      // k per LDS stage
[[maybe_unused]] constexpr int kStages = 2, kStageBytes = 2 * 256 * KC * 2;  // one stage = 64 k of both tiles = 64 KB
[[maybe_unused]] constexpr float kFltMax = 3.4028234663852886e38f;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float x) {  // bf16 bits of x, round to nearest even (NaN stays NaN)
  const unsigned b = __float_as_uint(x);
  if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// f32 [R, D] -> three bf16 planes [3][Rpad][Dp] (h, m, l), zero in the padding; 8 consecutive k per thread
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes,
                                                         int64_t R, int64_t D, int64_t Rpad, int64_t Dp) {
  const int64_t groups = Dp / 8, total = Rpad * groups;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / groups, k0 = (i % groups) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (row < R) {
      const float* p = x + row * D + k0;
      if (vec && k0 + 8 <= D) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < D) ? p[j] : 0.f;
      }
    }
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = bf16_rne(v[j]);
      const float hf = __uint_as_float(h[j] << 16);
      const bool fin = (__float_as_uint(hf) & 0x7f800000u) != 0x7f800000u;  // an infinite / NaN head has no tail
      const float r1 = fin ? v[j] - hf : 0.f;                               // exact: the low 16 bits of the significand
      m[j] = bf16_rne(r1);
      const float r2 = r1 - __uint_as_float(m[j] << 16);                    // exact
      l[j] = bf16_rne(r2);
    }
    const int64_t o = row * Dp + k0, plane = Rpad * Dp;
    *reinterpret_cast<uint4*>(planes + o) =
        make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    *reinterpret_cast<uint4*>(planes + plane + o) =
        make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    *reinterpret_cast<uint4*>(planes + 2 * plane + o) =
        make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
  }
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ i32x4 raw_buffer(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
// 64 lanes x 16 bytes from rsrc[voff(lane) + soff] straight into LDS at lds_bytes + 16 * lane.  Inline assembly on
// purpose: the compiler orders every later ds_read behind an LDS DMA builtin it cannot prove disjoint (one dynamic LDS
// array: s_waitcnt vmcnt(0) in front of each chunk's first operand read, the DMA of the NEXT chunk included).  There
// is no other vector-memory load in the chunk loop, so the one `s_waitcnt vmcnt(0)` at the head of a chunk is exact.
__device__ __forceinline__ void dma16(unsigned lds_bytes, unsigned voff, i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_bytes), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}
#endif

// piece pairs, smallest products first (h = 0, m = 1, l = 2): 6 terms  l.h h.l m.m m.h h.m h.h,  3 terms  m.h h.m h.h
// (error <= 3 * 2^-16 |q||b|: would need a wider refinement window in the caller).  One nibble per term - a table
// indexed at run time would live in constant memory, and its scalar load waits for every LDS read in flight.
template <int TERMS>
__device__ __forceinline__ void piece_pair(int t, int& pa, int& pb) {
  constexpr unsigned A = TERMS == 6 ? 0x001102u : 0x001u, B = TERMS == 6 ? 0x010120u : 0x010u;
  pa = (int)((A >> (4 * t)) & 15u);
  pb = (int)((B >> (4 * t)) & 15u);
}

template <int TERMS>
__global__ __launch_bounds__(256) void knn_dist_bf16_kernel(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ bp,
                                                             const float* __restrict__ qn, const float* __restrict__ bn,
                                                             float* __restrict__ dist, int64_t Q, int64_t M, int64_t Dp,
                                                             int64_t Qpad, int64_t Mpad) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];  // NS stages x (1024 A slots + 1024 B slots)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wb = wave & 1;
  // XCD-aware order: ids i, i + 8, ... (one XCD's share) walk a super-tile of 8 query tiles x 4 bank tiles; its 32
  // workgroups are resident on that XCD together and sweep K in step, so a tile's chunk is fetched into that L2 once
  // and read 4 (8) times.  Super-tiles are dealt round-robin to the XCDs.
  int64_t q0, m0;
  {
    const int64_t nqt = (Q + TQ - 1) / TQ, nbt = (M + TB - 1) / TB;
    const int64_t nqg = (nqt + 7) / 8;
    const int64_t l = blockIdx.x >> 3;
    const int64_t st = (l / 32) * 8 + (blockIdx.x & 7);
    const int r = (int)(l % 32);
    const int64_t qt = (st % nqg) * 8 + (r & 7), bt = (st / nqg) * 4 + (r >> 3);
    if (qt >= nqt || bt >= nbt) return;  // padding of the grid (uniform over the workgroup)
    q0 = qt * TQ;
    m0 = bt * TB;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  // DMA: a stage (64 k of both tiles) = 4096 slots = 64 instructions of 8 rows x 8 kgroups; wave w issues I = 16 w ..
  // 16 w + 15 (I < 32: query rows 8 I .. 8 I + 7, else bank rows).  Lane i of an instruction: row i / 8, slot position
  // i % 8, i.e. kgroup (i % 8) ^ ((row >> 1) & 7) - the eight lanes of a row fetch its whole 128-byte line.
  const bool mine_a = wave < 2;
  const i32x4 rsrc = mine_a ? raw_buffer(qp, (unsigned)(3 * Qpad * Dp * 2)) : raw_buffer(bp, (unsigned)(3 * Mpad * Dp * 2));
  const int64_t rows_pad = mine_a ? Qpad : Mpad;
  const int64_t tile_row0 = (mine_a ? q0 : m0) + 8 * 16 * (wave & 1);
  unsigned voff[2];  // by the parity of I: (row >> 1) & 7 = (4 I + (i >> 4)) & 7
#pragma unroll
  for (int par = 0; par < 2; ++par)
    voff[par] = (unsigned)((lane >> 3) * Dp * 2 + (((lane & 7) ^ ((lane >> 4) | (par << 2))) * 16));
  const int nk = (int)(Dp / KC), total = TERMS * nk;
  int next_t = 0, next_k = 0;  // piece pair and stage within it of the next DMA (scalar counters: no division per stage)
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // operand reads: lane l reads row (l & 31) of a 32-row tile, kgroup 2 ks + (l >> 5), ks = 0 .. 3 within a stage
  const int lrow = lane & 31, lhalf = lane >> 5;
  const uint4* a_lane = lds + (wq * 128 + lrow) * 8;
  const uint4* b_lane = lds + 2048 + (wb * 128 + lrow) * 8;
  int gk[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) gk[ks] = (2 * ks + lhalf) ^ ((lane >> 1) & 7);
  // operand n of a k-step in the order the matrix instructions first need them: a0, b0, b1, b2, b3, a1, a2, a3
  auto load_one = [&](uint4 (&fa)[4], uint4 (&fb)[4], int stage, int ks, int n) {
    const bool is_a = (n == 0 || n >= 5);
    const int t = (n == 0) ? 0 : (n >= 5 ? n - 4 : n - 1);
    const uint4* base = (is_a ? a_lane : b_lane) + stage * (kStageBytes / 16) + gk[ks] + t * 256;
#if KNN16_NOREAD  // timing experiment (wrong results): operands are not re-read
    if (ks == 99) { if (is_a) fa[t] = *base; else fb[t] = *base; }
#else
    if (is_a) fa[t] = *base;
    else fb[t] = *base;
#endif
  };
  auto mfma1 = [&](const uint4 (&fa)[4], const uint4 (&fb)[4], int n) {  // matrix instruction n = 4 i + j
    const int i = n >> 2, j = n & 3;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                        acc[i][j], 0, 0, 0);
  };
  // running source offset of the next stage: + 128 bytes per stage, re-based when the piece pair changes
  const unsigned plane_bytes = (unsigned)(rows_pad * Dp * 2), tile_bytes = (unsigned)(tile_row0 * Dp * 2);
  auto term_base = [&](int t) {
    int pa, pb;
    piece_pair<TERMS>(t, pa, pb);
    return (unsigned)(mine_a ? pa : pb) * plane_bytes + tile_bytes;
  };
  unsigned dma_next = term_base(0), dma_s0 = 0, dma_l0 = 0;
  auto dma_begin = [&](int stage) {
    dma_s0 = dma_next;
    dma_l0 = lds0 + (unsigned)stage * (unsigned)kStageBytes + (unsigned)wave * 16u * 1024u;
#if !KNN16_SAMECHUNK  // (timing experiment, wrong results: every stage re-reads the first one - all L2 hits)
    dma_next += KC * 2;
#endif
    if (++next_k == nk) {
      next_k = 0;
      if (++next_t == TERMS) { next_t = TERMS - 1; next_k = nk - 1; dma_next = dma_s0; }  // past the end: the last stage again (never read)
      else dma_next = term_base(KNN16_SAMECHUNK ? 0 : next_t);
    }
  };
#if KNN16_NODMA  // timing experiment (wrong results): no DMA inside the loop
  auto dma_one = [&](int j) { (void)j; };
#else
  auto dma_one = [&](int j) { dma16(dma_l0 + 1024u * j, voff[j & 1], rsrc, dma_s0 + (unsigned)(8 * j * Dp * 2)); };
#endif
  auto issue = [&](int stage) {
    dma_begin(stage);
#pragma unroll
    for (int j = 0; j < 16; ++j) dma_one(j);
  };
  // Pipeline: two LDS stages of 64 k (four k-steps of 16 MFMAs).  The DMA of stage d + 1 flies while stage d is
  // multiplied; the operand registers run one k-step ahead of the matrix instructions that consume them, across stage
  // boundaries too.  A wave issues matrix instructions back to back (32 cycles each); whatever else it has to do - 32
  // operand reads, 16 DMA instructions, one wait and one barrier per stage - is placed BETWEEN them, two matrix
  // instructions apart, so that it runs in their shadow (__builtin_amdgcn_sched_barrier pins the order: left alone, the
  // scheduler sinks every read to just in front of its first use and gathers the rest at the barrier).
  //   k-steps 0..2 of stage d:  16 MFMAs each  |  the 8 operand reads of the next k-step
  //   k-step 3 of stage d:      16 MFMAs       |  wait "DMA(d+1) landed" + barrier (everyone has read stage d to the
  //                                               end), DMA(d+2) into the buffer of stage d begins, the reads of (d+1, 0)
  //   The 16 DMA instructions of a stage go out ONE per pair of matrix instructions (7 in k-step 3, 8 in the next
  //   k-step 0, 1 in k-step 1): a vector-memory instruction holds the wave's issue for longer than a pair's shadow.
  // Every request of the DMA is a whole 128-byte line of a row (with 32-k stages - 64 bytes per row - the kernel ran at
  // the L2's half-line rate: 45 % matrix-pipe utilisation at 7.3 TB/s of L2 reads).  Past the end the last stage is
  // fetched again into a buffer nobody reads, so the wait is always the plain vmcnt(0).
#define RUNIA_PIN __builtin_amdgcn_sched_barrier(0)
  issue(0);
  dma_begin(1);  // stage 1 as the loop leaves a stage at its top: instructions 0 .. 6 issued, 7 .. 15 follow in k-steps 0 and 1
#pragma unroll
  for (int j = 0; j < 7; ++j) dma_one(j);
  asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  __syncthreads();
  uint4 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
  for (int n = 0; n < 8; ++n) load_one(fa0, fb0, 0, 0, n);
  int stage = 0;
  for (int d = 0; d < total; ++d) {
    RUNIA_PIN;
#pragma unroll
    for (int n = 0; n < 8; ++n) {  // k-step 0 (+ DMA instructions 7 .. 14 of the stage whose DMA began in the last k-step 3)
      load_one(fa1, fb1, stage, 1, n);
      dma_one(7 + n);
      RUNIA_PIN;
      mfma1(fa0, fb0, 2 * n);
      mfma1(fa0, fb0, 2 * n + 1);
      RUNIA_PIN;
    }
#pragma unroll
    for (int n = 0; n < 8; ++n) {  // k-step 1 (+ DMA instruction 15)
      load_one(fa0, fb0, stage, 2, n);
      if (n == 0) dma_one(15);
      RUNIA_PIN;
      mfma1(fa1, fb1, 2 * n);
      mfma1(fa1, fb1, 2 * n + 1);
      RUNIA_PIN;
    }
    dma_begin(stage);  // scalar bookkeeping of the DMA that starts in k-step 3, in the shadow of k-step 2
#pragma unroll
    for (int n = 0; n < 8; ++n) {  // k-step 2
      load_one(fa1, fb1, stage, 3, n);
      RUNIA_PIN;
      mfma1(fa0, fb0, 2 * n);
      mfma1(fa0, fb0, 2 * n + 1);
      RUNIA_PIN;
    }
    mfma1(fa1, fb1, 0);  // k-step 3
    mfma1(fa1, fb1, 1);
    RUNIA_PIN;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage ^= 1;
    RUNIA_PIN;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // one DMA instruction per pair of matrix instructions: a second or third one in the
      dma_one(g);                  // same gap outlasts the pair's 64 cycles and drains the matrix pipe
      RUNIA_PIN;
      mfma1(fa1, fb1, 2 + 2 * g);
      mfma1(fa1, fb1, 3 + 2 * g);
      RUNIA_PIN;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      dma_one(4 + g);
      load_one(fa0, fb0, stage, 0, 3 * g);
      load_one(fa0, fb0, stage, 0, 3 * g + 1);
      load_one(fa0, fb0, stage, 0, 3 * g + 2);
      RUNIA_PIN;
      mfma1(fa1, fb1, 10 + 2 * g);
      mfma1(fa1, fb1, 11 + 2 * g);
      RUNIA_PIN;
    }
    dma_one(6);
    load_one(fa0, fb0, stage, 0, 6);
    load_one(fa0, fb0, stage, 0, 7);
    RUNIA_PIN;
    mfma1(fa1, fb1, 14);
    mfma1(fa1, fb1, 15);
  }
#undef RUNIA_PIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail of the last (unused) DMA
  // epilogue: C[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31] of each 32 x 32 tile
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = m0 + wb * 128 + j * 32 + lrow;
      const float bnv = (col < M) ? bn[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = q0 + wq * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
        if (row < Q && col < M) {
          const float d = (qn[row] + bnv) - 2.0f * acc[i][j][r];
          // as the f32 kernel: a NaN or infinite distance counts as FLT_MAX (faiss never inserts it into its heap)
          dist[row * M + col] = (d == d) ? fminf(fmaxf(d, 0.f), kFltMax) : kFltMax;
        }
      }
    }
#endif
}

}  // namespace runia_knn16

// ---- host side (called from pairwise.hip) ----
int64_t runia_knn16_padded_rows(int64_t rows) { return (rows + 255) / 256 * 256; }
int64_t runia_knn16_padded_width(int64_t D) { return (D + 63) / 64 * 64; }
size_t runia_knn16_plane_bytes(int64_t rows, int64_t D) {
  return (size_t)(3 * runia_knn16_padded_rows(rows) * runia_knn16_padded_width(D)) * sizeof(uint16_t);
}
// the planes of a matrix must be addressable through one 32-bit buffer
bool runia_knn16_fits(int64_t rows, int64_t D) { return runia_knn16_plane_bytes(rows, D) < ((size_t)1 << 32) - 4096; }

int runia_knn16_split(const float* x, uint16_t* planes, int64_t R, int64_t D, hipStream_t s) {
  const int64_t Rpad = runia_knn16_padded_rows(R), Dp = runia_knn16_padded_width(D);
  runia_knn16::split_bf16_kernel<<<runia_stream_grid(Rpad * (Dp / 8), 256), 256, 0, s>>>(x, planes, R, D, Rpad, Dp);
  return runia_check_launch();
}

#ifndef KNN16_TERMS
#define KNN16_TERMS 3
#endif
// Half-width of the caller's refinement window, relative to the row's range bound R = (|q| + max|b|)^2: it has to be at
// least TWICE the largest error e of a candidate distance (then every bank row outside the window lies on the same side
// of the true k-th distance as of the approximate one).  Six terms: the dropped products are <= 3 * 2^-24 |q||b| - the
// f32 kernel's own allowance stands.  Three terms: dropped m.m + h.l + l.h <= 3 * 2^-16 sum|q_k||b_k| <= 3 * 2^-16 |q||b|
// (each piece is below 2^-8 of the one before; Cauchy-Schwarz), |q||b| <= R / 4, and the distance carries twice the
// product's error: e <= 2.3e-5 R, plus the accumulation allowance -> 5e-5 R.
int runia_knn16_terms() { return KNN16_TERMS; }
float runia_knn16_refine_rel() { return KNN16_TERMS == 6 ? 5e-6f : 5e-5f; }
int runia_knn16_dist(const uint16_t* qp, const uint16_t* bp, const float* qn, const float* bn, float* dist, int64_t Q,
                     int64_t M, int64_t D, hipStream_t s) {
  using namespace runia_knn16;
  constexpr size_t lds_bytes = (size_t)kStages * kStageBytes;
  static std::atomic<uint64_t> lds_ok{0};
  const int rc = runia_allow_dynamic_lds(reinterpret_cast<const void*>(knn_dist_bf16_kernel<KNN16_TERMS>), lds_bytes, lds_ok);
  if (rc != RUNIA_OK) return rc;
  const int64_t nqt = (Q + TQ - 1) / TQ, nbt = (M + TB - 1) / TB;
  const int64_t st = ((nqt + 7) / 8) * ((nbt + 3) / 4);
  const unsigned grid = (unsigned)(((st + 7) / 8) * 8 * 32);
  knn_dist_bf16_kernel<KNN16_TERMS><<<grid, 256, lds_bytes, s>>>(qp, bp, qn, bn, dist, Q, M, runia_knn16_padded_width(D),
                                                                  runia_knn16_padded_rows(Q), runia_knn16_padded_rows(M));
  return runia_check_launch();
}
