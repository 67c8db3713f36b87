// a8 kNN, candidate distances of large problems on the bf16 matrix cores (gfx950: v_mfma_f32_16x16x32_bf16, 16x the
// work per matrix-pipe cycle of the f32 forms).
//
// An f32 value is split into bf16 pieces, x = h + m + (rest <= 2^-16 |x|): h = bf16(x), m = bf16(x - h), 8 significand
// bits each, f32's exponent range, so no scaling and no range restriction.  The dot product q.b is taken as the three
// piece products of order <= 2^-8,
//     q.b ~= m.h + h.m + h.h        (dropped: m.m + h.l + l.h + ... <= 3 * 2^-16 |q||b|),
// each an exact bf16 product accumulated in f32 by the matrix cores, 3/16 of the f32 contraction's matrix-pipe time.  The
// result feeds the SAME selection + exact f32 re-measurement as the f32 kernel's distances (knn_f32.hip,
// kth_select_range_kernel), whose window is widened to twice this kernel's error bound (runia_knn16_refine_rel): every
// bank row that could be the k-th neighbour is re-measured with exact f32 differences, so the caller gets the exactly
// re-measured k-th distance either way; this kernel only decides which bank rows are looked at.  (All six products of
// order <= 2^-16 - f32 accuracy, window unchanged - were built and measured first: 9.3 ms per chunk against 5.3 for
// three; profiles/README.md.)
//
// Memory layout of the pieces ("planes"): per row, per block of 32 k, 64 bytes of h followed by 64 bytes of m - one
// 128-byte line holds both pieces of a block, so ONE fetched line feeds all three products of that block.
//
// Kernel shape.  One workgroup = 256 queries x 256 bank rows, 4 waves of 128 x 128 (8 x 8 MFMA tiles of 16 x 16, 256
// accumulator registers, one wave per SIMD).  K runs over D in stages of 32: a stage of a tile is 256 rows x 128 bytes
// (h | m), staged by `buffer_load_dwordx4 ... lds` (16 per wave and stage, no staging registers, no ds_write) into one of
// two LDS buffers (2 x 64 KB), and is multiplied three ways: 192 matrix instructions (16 cycles each) per 64 KB moved (the
// form with one product per staged chunk was bound by the bytes the DMA moves: profiles/README.md).  LDS slot (16 bytes =
// 8 consecutive k of one piece of one row) g = 0..3 (h) / 4..7 (m) of a row sits at row * 8 + (g ^ ((row >> 1) & 7)): the
// eight lanes that fetch one row read its whole 128-byte line, and the ds_read_b128 of an MFMA operand (lane l: row l & 15
// of a 16-row tile, slot l >> 4) is conflict-free: the LDS serves a b128 read in four groups of 16 lanes - {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 - i.e. rows {0-3, 12-15} of one slot with rows {4-11} of the next, and
// under the swizzle those 16 (row, slot) pairs fall into the 16 different 16-byte bank groups (SQ_LDS_BANK_CONFLICT = 0).
// The pipeline is described at the loop.
#include "common.hpp"
#include "knn_perm.hpp"

#include <cstdint>
#include <type_traits>

namespace runia_knn16 {

constexpr int TQ = 256, TB = 256;            // tile
[[maybe_unused]] constexpr int KC = 32;      // k per LDS stage
[[maybe_unused]] constexpr int kStages = 2, kStageBytes = 2 * 256 * KC * 4;  // one stage = 32 k of both tiles, h | m = 64 KB
[[maybe_unused]] constexpr float kFltMax = 3.4028234663852886e38f;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_rne(float x) {  // bf16 bits of x, round to nearest even (NaN stays NaN)
  const unsigned b = __float_as_uint(x);
  if ((b & 0x7fffffffu) > 0x7f800000u) return (b >> 16) | 0x40u;
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// f32 [R, D] -> pieces [Rpad][Dp / 32][h: 32 bf16 | m: 32 bf16], zero in the padding; 8 consecutive k per thread
// `pm`: piece row p holds matrix row knn_perm_row(p) (the bank; identity for queries); `norms` / `norms_p` (optional): the
// rows' squared norms in matrix order -> in piece order, NaN in the padding (a padding column never passes a threshold)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ planes,
                                                         int64_t R, int64_t D, int64_t Rpad, int64_t Dp, KnnPerm pm,
                                                         const float* __restrict__ norms, float* __restrict__ norms_p) {
  const int64_t groups = Dp / 8, total = Rpad * groups;
  const bool vec = ((D & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / groups, k0 = (i % groups) * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    const int64_t src = row < R ? knn_perm_row(row, pm) : row;
    if (norms_p && k0 == 0) norms_p[row] = row < R ? norms[src] : __builtin_nanf("");
    if (row < R) {
      const float* p = x + src * D + k0;
      if (vec && k0 + 8 <= D) {
        const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j < D) ? p[j] : 0.f;
      }
    }
    unsigned h[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h[j] = bf16_rne(v[j]);
      const float hf = __uint_as_float(h[j] << 16);
      const bool fin = (__float_as_uint(hf) & 0x7f800000u) != 0x7f800000u;  // an infinite / NaN head has no tail
      m[j] = bf16_rne(fin ? v[j] - hf : 0.f);                               // v - hf is exact (the low 16 significand bits)
    }
    uint16_t* o = planes + (row * Dp + (k0 / 32) * 32) * 2 + (k0 % 32);  // block of 32 k: 64 bytes h, then 64 bytes m
    *reinterpret_cast<uint4*>(o) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    *reinterpret_cast<uint4*>(o + 32) = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
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

// What leaves the kernel (FILTER):
//   false  the distances, dense: dist [Q, M].  `q_count` (optional, device): only the first *q_count - q_first query rows
//          exist (the dense fallback of the filter path works on however many rows overflowed; the grid is sized for Q)
//   true   per query row only the bank rows whose distance is <= thr[row], appended as (distance bits, col0 + column) to
//          lists[row * cap ..] with the running count in counts[row] (counted past cap, so an overflow shows): the
//          candidate filter of the k-th-neighbour search.  The Q x M matrix - 1.6 GB per 8 192 x 50 000 chunk, written
//          here and read again by the selection - is never formed; about k * M / (sample rows) entries per row are.
//          Per tile: every lane tests its 256 distances, the hits of a row are counted over the 16 lanes that hold it
//          (four rows' 8-bit counters per DPP scan), the two waves that share a row meet in LDS, ONE global atomic per
//          row and tile reserves the row's slots, and the hits are stored behind it.
struct KnnFilterOut {
  const float* thr;
  unsigned* counts;
  uint2* lists;
  int cap;
  int col0;
};
template <int CTRL>
__device__ __forceinline__ unsigned dpp_row_u32(unsigned v) {  // lanes shifted in from outside the row of 16 read 0
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

template <bool FILTER>
__global__ __launch_bounds__(256) void knn_dist_bf16_kernel(const uint16_t* __restrict__ qp, const uint16_t* __restrict__ bp,
                                                             const float* __restrict__ qn, const float* __restrict__ bn,
                                                             float* __restrict__ dist, int64_t Q, int64_t M, int64_t Dp,
                                                             int64_t Qpad, int64_t Mpad, const int* __restrict__ q_count,
                                                             int q_first, KnnFilterOut fo) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];  // 2 buffers x (2048 query slots + 2048 bank slots)
  const int tid = threadIdx.x, lane = tid & 63;
  if (!FILTER && q_count) {  // (uniform)
    const int64_t have = (int64_t)*q_count - q_first;
    if (have < Q) Q = have;
    if (Q <= 0) return;
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wq = wave >> 1, wb = wave & 1;
  // XCD-aware order: ids i, i + 8, ... (one XCD's share) walk a super-tile of 8 query tiles x 4 bank tiles; its 32
  // workgroups are resident on that XCD together and sweep K in step, so a tile's stage is fetched into that L2 once and
  // read 4 (8) times (measured: 79 % L2 hits of the ideal 81 %).  Super-tiles are dealt round-robin to the XCDs.  A batch
  // of fewer than 8 query tiles makes super-tiles of (all its query tiles) x 4 bank tiles, every workgroup with a tile:
  // workgroups go to compute units in a fixed rotation, and with a 256-row batch in 8 x 4 super-tiles the one valid
  // workgroup in eight landed on the same 4 compute units of each XCD - 6 rounds of them (0.55 ms instead of 0.15).
  int64_t q0, m0;
  {
    const int64_t nqt = (Q + TQ - 1) / TQ, nbt = (M + TB - 1) / TB;
    const int sq = nqt < 8 ? (int)nqt : 8, wps = 4 * sq;
    const int64_t nqg = (nqt + sq - 1) / sq;
    const int64_t l = blockIdx.x >> 3;
    const int64_t st = (l / wps) * 8 + (blockIdx.x & 7);
    const int r = (int)(l % wps);
    const int64_t qt = (st % nqg) * sq + (r >> 2), bt = (st / nqg) * 4 + (r & 3);
    if (qt >= nqt || bt >= nbt) return;  // padding of the grid (uniform over the workgroup)
    q0 = qt * TQ;
    m0 = bt * TB;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  // DMA: a stage (32 k of both tiles, h | m) = 4096 slots = 64 instructions of 8 rows x 8 slots; wave w issues I = 16 w ..
  // 16 w + 15 (I < 32: query rows 8 I .. 8 I + 7, else bank rows).  Lane i of an instruction: row i / 8, slot position
  // i % 8, i.e. slot (i % 8) ^ ((row >> 1) & 7) of the row - the eight lanes of a row fetch its whole 128-byte line.
  const bool mine_a = wave < 2;
  const unsigned row_bytes = (unsigned)(Dp * 4);  // h and m of one row
  const i32x4 rsrc = mine_a ? raw_buffer(qp, (unsigned)(Qpad * Dp * 4)) : raw_buffer(bp, (unsigned)(Mpad * Dp * 4));
  const int64_t tile_row0 = (mine_a ? q0 : m0) + 8 * 16 * (wave & 1);
  unsigned voff[2];  // by the parity of I: (row >> 1) & 7 = (4 I + (i >> 4)) & 7
#pragma unroll
  for (int par = 0; par < 2; ++par)
    voff[par] = (unsigned)(lane >> 3) * row_bytes + (unsigned)(((lane & 7) ^ ((lane >> 4) | (par << 2))) * 16);
  const int total = (int)(Dp / KC);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[8][8];  // 8 x 8 tiles of 16 x 16
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // operand reads (v_mfma_f32_16x16x32_bf16: lane l holds row l & 15, k 8 (l >> 4) .. + 7 of a 16-row tile): an operand
  // set = the h (or m) piece of the stage for the wave's 128 rows = 8 tiles; lane l reads slot (l >> 4) (+ 4 for m)
  const int lrow = lane & 15, lq = lane >> 4;
  const uint4* a_lane = lds + (wq * 128 + lrow) * 8;
  const uint4* b_lane = lds + 2048 + (wb * 128 + lrow) * 8;
  int gk[2];
#pragma unroll
  for (int piece = 0; piece < 2; ++piece) gk[piece] = (4 * piece + lq) ^ ((lane >> 1) & 7);
  typedef uint4 frag8[8];
  auto load_a = [&](frag8& f, int buf, int piece, int t) { f[t] = a_lane[buf * (kStageBytes / 16) + gk[piece] + t * 128]; };
  auto load_b = [&](frag8& f, int buf, int piece, int t) { f[t] = b_lane[buf * (kStageBytes / 16) + gk[piece] + t * 128]; };
  auto mfma1 = [&](const frag8& fa, const frag8& fb, int n) {  // matrix instruction n = 8 i + j
    const int i = n >> 3, j = n & 7;
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]),
                                                        acc[i][j], 0, 0, 0);
  };
  // running source offset of the next stage: + 128 bytes (one block of h | m) per stage
  unsigned dma_next = (unsigned)tile_row0 * row_bytes, dma_s0 = 0, dma_l0 = 0;
  int next_k = 0;
  auto dma_begin = [&](int buf) {
    dma_s0 = dma_next;
    dma_l0 = lds0 + (unsigned)buf * (unsigned)kStageBytes + (unsigned)wave * 16u * 1024u;
    if (++next_k < total) dma_next += 128u;  // past the end: the last stage again, into a buffer nobody reads
  };
  auto dma_one = [&](int j) { dma16(dma_l0 + 1024u * j, voff[j & 1], rsrc, dma_s0 + (unsigned)(8 * j) * row_bytes); };
  // Pipeline.  A stage (32 k) is multiplied three ways, 64 matrix instructions (16 x 16 x 32, 16 cycles each) per product:
  //     P1: Am x Bh   | the 8 reads of Ah
  //     P2: Ah x Bh   | the 8 reads of Bm
  //     P3: Ah x Bm   | wait "DMA(d+1) landed" + barrier (everyone has read stage d to the end), the 16 DMA
  //                     instructions of stage d + 2 into the buffer of stage d, the 16 reads of the next Am and Bh
  // In this order the product that overlaps the next stage's first reads (P3) uses neither of the sets those reads fill
  // (Am, Bh), so every operand set has ONE register home (R1 = Ah, R2 = Bh, R3 = Bm, R4 = Am) and the loop body is one
  // stage (a body of two stages with swapped roles made the compiler copy the 256 accumulators around the back edge).
  // A wave issues matrix instructions back to back; whatever else it has to do is placed BETWEEN them - at most one
  // vector-memory instruction and one read per four matrix instructions - so that it runs in their shadow
  // (__builtin_amdgcn_sched_barrier pins the order: left alone, the scheduler sinks every read to just in front of its
  // first use and gathers the rest at the barrier).  The 16 x 16 x 32 shape: the same cycles per FLOP as 32 x 32 x 16,
  // but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, 'DVFS give-back' (7): 1.12-1.15 x).
#define RUNIA_PIN __builtin_amdgcn_sched_barrier(0)
  frag8 R1, R2, R3, R4;
  dma_begin(0);
#pragma unroll
  for (int j = 0; j < 16; ++j) dma_one(j);
  dma_begin(1);
#pragma unroll
  for (int j = 0; j < 16; ++j) dma_one(j);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) { load_a(R4, 0, 1, t); load_b(R2, 0, 0, t); }
  int buf = 0;
  for (int d = 0; d < total; ++d) {
    RUNIA_PIN;
#pragma unroll
    for (int n = 0; n < 8; ++n) {  // P1: Am x Bh | read Ah -> R1
      load_a(R1, buf, 0, n);
      RUNIA_PIN;
#pragma unroll
      for (int u = 0; u < 8; ++u) mfma1(R4, R2, 8 * n + u);
      RUNIA_PIN;
    }
    dma_begin(buf);  // scalar bookkeeping of the DMA issued in P3, in the shadow of P2
#pragma unroll
    for (int n = 0; n < 8; ++n) {  // P2: Ah x Bh | read Bm -> R3
      load_b(R3, buf, 1, n);
      RUNIA_PIN;
#pragma unroll
      for (int u = 0; u < 8; ++u) mfma1(R1, R2, 8 * n + u);
      RUNIA_PIN;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
    RUNIA_PIN;
#pragma unroll
    for (int g = 0; g < 16; ++g) {  // P3: Ah x Bm | DMA of stage d + 2, reads of the next Am -> R4 and Bh -> R2
      dma_one(g);
      if (g == 0) load_a(R4, buf, 1, 0);
      else if (g <= 8) load_b(R2, buf, 0, g - 1);
      else load_a(R4, buf, 1, g - 8);
      RUNIA_PIN;
#pragma unroll
      for (int u = 0; u < 4; ++u) mfma1(R1, R3, 4 * g + u);
      RUNIA_PIN;
    }
  }
#undef RUNIA_PIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last (unused) DMA
  // epilogue: C[row = 4 * (lane >> 4) + reg][col = lane & 15] of each 16 x 16 tile
  if constexpr (FILTER) {
#if defined(KNN16_ABLATE) && KNN16_ABLATE == 1  // (timing experiments only: no epilogue at all)
    {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
      if (t == 123.456f) fo.counts[0] = 1u;
    }
    return;
#endif
    __syncthreads();  // the stage buffers are free: every wave has read its last operands
    float* qn_l = reinterpret_cast<float*>(lds);                 // [256] |q|^2 of the tile's rows
    float* th_l = qn_l + 256;                                    // [256] their thresholds (-inf: no such row)
    unsigned* c_l = reinterpret_cast<unsigned*>(th_l + 256);     // [256][2] hits of (row, column half)
    unsigned* b_l = c_l + 512;                                   // [256] first slot of the row's hits in its list
    {
      const int64_t row = q0 + tid;
      qn_l[tid] = row < Q ? qn[row] : 0.f;
      th_l[tid] = row < Q ? fo.thr[row] : -INFINITY;
    }
    float bnv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t col = m0 + wb * 128 + j * 16 + lrow;
      bnv[j] = (col < M) ? bn[col] : __builtin_nanf("");  // NaN: the comparison below is false
    }
    __syncthreads();
    unsigned maskp[8], exclp[8];  // per i: the hit masks of r = 0..3 (8 bits each) and the hits of the lanes before this one
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned cnt4 = 0u;
      maskp[i] = 0u;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowl = wq * 128 + i * 16 + 4 * lq + r;
        const float qnr = qn_l[rowl], tr = th_l[rowl];
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dd = (qnr + bnv[j]) - 2.0f * acc[i][j][r];
          m |= (dd <= tr) ? (1u << j) : 0u;
        }
        maskp[i] |= m << (8 * r);
        cnt4 |= (unsigned)__builtin_popcount(m) << (8 * r);
      }
      unsigned incl = cnt4;  // inclusive scan over the 16 lanes of a row (a row's total <= 128: no carry between the bytes)
      incl += dpp_row_u32<0x111>(incl);
      incl += dpp_row_u32<0x112>(incl);
      incl += dpp_row_u32<0x114>(incl);
      incl += dpp_row_u32<0x118>(incl);
      exclp[i] = incl - cnt4;
      if (lrow == 15) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c_l[(wq * 128 + i * 16 + 4 * lq + r) * 2 + wb] = (incl >> (8 * r)) & 255u;
      }
    }
    __syncthreads();
    {
      const unsigned n = c_l[2 * tid] + c_l[2 * tid + 1];
      b_l[tid] = (n != 0u) ? atomicAdd(&fo.counts[q0 + tid], n) : 0u;  // (rows beyond Q have no hits)
    }
    __syncthreads();
#if defined(KNN16_ABLATE) && KNN16_ABLATE == 2  // (timing experiments only: hits counted, not stored)
    return;
#endif
    // The hits, each behind the ones before it in its row's list; stored through a buffer descriptor (32-bit offsets: a
    // chunk's lists are < 4 GB); a row that would run over its list takes the checked form (it overflows anyway).
    // (Staging the hits in LDS and storing them with whole waves - 7 store instructions per wave instead of ~200 with two
    // or three lanes active - measured 3 us per tile SLOWER: what the stores cost, ~7 us of a 130 us tile, is not their
    // instruction count.)
    const __amdgpu_buffer_rsrc_t lrsrc =
        __builtin_amdgcn_make_buffer_rsrc(fo.lists, 0, (int)((size_t)Qpad * (size_t)fo.cap * 8u), 0x00020000);
    unsigned colj[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) colj[j] = (unsigned)(fo.col0 + (int)(m0 + wb * 128 + j * 16 + lrow));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (maskp[i] == 0u) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned m = (maskp[i] >> (8 * r)) & 255u;
        if (m == 0u) continue;
        const int rowl = wq * 128 + i * 16 + 4 * lq + r;
        const float qnr = qn_l[rowl];
        unsigned off = b_l[rowl] + (wb ? c_l[2 * rowl] : 0u) + ((exclp[i] >> (8 * r)) & 255u);
        const unsigned row_base = (unsigned)(q0 + rowl) * (unsigned)fo.cap;
        if (off + (unsigned)__builtin_popcount(m) <= (unsigned)fo.cap) {
          unsigned voff = (row_base + off) * 8u;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if ((m >> j) & 1u) {
              const float dd = fmaxf((qnr + bnv[j]) - 2.0f * acc[i][j][r], 0.f);
#if !(defined(KNN16_ABLATE) && KNN16_ABLATE == 3)  // (timing experiments only: no stores)
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_uint2(__float_as_uint(dd), colj[j])), lrsrc,
                                                    (int)voff, 0, 0);
#endif
              voff += 8u;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if ((m >> j) & 1u) {
              const float dd = fmaxf((qnr + bnv[j]) - 2.0f * acc[i][j][r], 0.f);
              if (off < (unsigned)fo.cap) fo.lists[row_base + off] = make_uint2(__float_as_uint(dd), colj[j]);
              ++off;
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t col = m0 + wb * 128 + j * 16 + lrow;
      const float bnv = (col < M) ? bn[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = q0 + wq * 128 + i * 16 + 4 * lq + r;
        if (row < Q && col < M) {
          const float dd = (qn[row] + bnv) - 2.0f * acc[i][j][r];
          // as the f32 kernel: a NaN or infinite distance counts as FLT_MAX (faiss never inserts it into its heap)
          dist[row * M + col] = (dd == dd) ? fminf(fmaxf(dd, 0.f), kFltMax) : kFltMax;
        }
      }
    }
#endif
}

}  // namespace runia_knn16

// ---- host side (called from knn_f32.hip) ----
int64_t runia_knn16_padded_rows(int64_t rows) { return (rows + 255) / 256 * 256; }
int64_t runia_knn16_padded_width(int64_t D) { return (D + 31) / 32 * 32; }
size_t runia_knn16_plane_bytes(int64_t rows, int64_t D) {
  return (size_t)(2 * runia_knn16_padded_rows(rows) * runia_knn16_padded_width(D)) * sizeof(uint16_t);
}
// the pieces of a matrix must be addressable through one 32-bit buffer
bool runia_knn16_fits(int64_t rows, int64_t D) { return runia_knn16_plane_bytes(rows, D) < ((size_t)1 << 32) - 4096; }

int runia_knn16_split(const float* x, uint16_t* planes, int64_t R, int64_t D, hipStream_t s) {
  const int64_t Rpad = runia_knn16_padded_rows(R), Dp = runia_knn16_padded_width(D);
  runia_knn16::split_bf16_kernel<<<runia_stream_grid(Rpad * (Dp / 8), 256), 256, 0, s>>>(x, planes, R, D, Rpad, Dp,
                                                                                        knn_perm_identity(), nullptr, nullptr);
  return runia_check_launch();
}
// the bank: rows in the sample order of knn_perm.hpp, with the squared norms `bn` (matrix order) copied into that order
// (`bn_p`, one per padded row, NaN in the padding)
int runia_knn16_split_bank(const float* x, uint16_t* planes, const float* bn, float* bn_p, int64_t R, int64_t D, hipStream_t s) {
  const int64_t Rpad = runia_knn16_padded_rows(R), Dp = runia_knn16_padded_width(D);
  runia_knn16::split_bf16_kernel<<<runia_stream_grid(Rpad * (Dp / 8), 256), 256, 0, s>>>(x, planes, R, D, Rpad, Dp,
                                                                                        knn_perm_for(R), bn, bn_p);
  return runia_check_launch();
}

// Half-width of the caller's refinement window, relative to the row's range bound R = (|q| + max|b|)^2: it has to be at
// least TWICE the largest error e of a candidate distance (then every bank row outside the window lies on the same side
// of the true k-th distance as of the approximate one).  Dropped products: m.m + h.l + l.h + ... <= 3 * 2^-16 sum|q_k||b_k|
// <= 3 * 2^-16 |q||b| (each piece is below 2^-8 of the one before; Cauchy-Schwarz), |q||b| <= R / 4, and the distance
// carries twice the product's error: e <= 2.3e-5 R.  Accumulation: the matrix cores add the 3 * D / 32 staged partial
// sums of a dot product in f32, each addition within 2^-24 of the running sum's magnitude <= sum|q_k||b_k| <= R / 4, and
// the 32 products inside an instruction are summed at no less than that accuracy: <= (3 D / 32 + 32) * 2^-24 * R / 4 per
// dot product, twice that in the distance - 6.7e-6 R at D = 2048 as a worst case (what is observed is the square root of
// the count: 1e-7 R).  Window = 2 e = 4.6e-5 R + (1.9e-6 + 5.6e-9 D) R, rounded up to (4.8e-5 + 5.6e-9 D) R; D is
// capped (runia_knn16_max_width; the f32 kernel takes wider rows) where the accumulation term would be twice the other:
// wider windows mean more exact re-measurements per row, not wrong answers.
int runia_knn16_terms() { return 3; }
int64_t runia_knn16_max_width() { return 16384; }
float runia_knn16_refine_rel(int64_t D) { return 4.8e-5f + 5.6e-9f * (float)D; }

static int knn16_launch_dims(int64_t Q, int64_t M, unsigned* grid) {
  using namespace runia_knn16;
  const int64_t nqt = (Q + TQ - 1) / TQ, nbt = (M + TB - 1) / TB;
  const int sq = nqt < 8 ? (int)nqt : 8;  // query tiles of a super-tile (see the kernel)
  const int64_t st = ((nqt + sq - 1) / sq) * ((nbt + 3) / 4);
  *grid = (unsigned)(((st + 7) / 8) * 8 * 4 * sq);
  return RUNIA_OK;
}

// dense distances [Q, M] of queries against piece rows 0 .. M - 1 of `bp` (`planes_rows`: padded rows behind bp);
// `q_count` / `q_first`: see the kernel
int runia_knn16_dist(const uint16_t* qp, const uint16_t* bp, const float* qn, const float* bn, float* dist, int64_t Q,
                     int64_t M, int64_t D, int64_t q_planes_rows, int64_t b_planes_rows, const int* q_count, int q_first,
                     hipStream_t s) {
  using namespace runia_knn16;
  constexpr size_t lds_bytes = (size_t)kStages * kStageBytes;
  static std::atomic<uint64_t> lds_ok{0};
  const int rc = runia_allow_dynamic_lds(reinterpret_cast<const void*>(knn_dist_bf16_kernel<false>), lds_bytes, lds_ok);
  if (rc != RUNIA_OK) return rc;
  unsigned grid;
  knn16_launch_dims(Q, M, &grid);
  knn_dist_bf16_kernel<false><<<grid, 256, lds_bytes, s>>>(qp, bp, qn, bn, dist, Q, M, runia_knn16_padded_width(D),
                                                            q_planes_rows, b_planes_rows, q_count, q_first,
                                                            KnnFilterOut{nullptr, nullptr, nullptr, 0, 0});
  return runia_check_launch();
}

// candidate lists (see the kernel): piece rows 0 .. M - 1 of `bp` are columns col0 .. col0 + M - 1
int runia_knn16_filter(const uint16_t* qp, const uint16_t* bp, const float* qn, const float* bn, const float* thr,
                       unsigned* counts, void* lists, int cap, int col0, int64_t Q, int64_t M, int64_t D,
                       int64_t q_planes_rows, int64_t b_planes_rows, hipStream_t s) {
  using namespace runia_knn16;
  constexpr size_t lds_bytes = (size_t)kStages * kStageBytes;
  static std::atomic<uint64_t> lds_ok{0};
  const int rc = runia_allow_dynamic_lds(reinterpret_cast<const void*>(knn_dist_bf16_kernel<true>), lds_bytes, lds_ok);
  if (rc != RUNIA_OK) return rc;
  unsigned grid;
  knn16_launch_dims(Q, M, &grid);
  knn_dist_bf16_kernel<true><<<grid, 256, lds_bytes, s>>>(qp, bp, qn, bn, nullptr, Q, M, runia_knn16_padded_width(D),
                                                           q_planes_rows, b_planes_rows, nullptr, 0,
                                                           KnnFilterOut{thr, counts, reinterpret_cast<uint2*>(lists), cap, col0});
  return runia_check_launch();
}
