// Shared pieces of the f64 matrix-core kernels (gemm_f64.hip, fused.hip).
//
// Packed weights.  A constant right-hand matrix B [K, n] is stored in MFMA fragment order:
//   packed[((s2*NT + ct)*64 + lane)*2 + h] = B[8*s2 + 4*h + (lane>>4)][16*ct + (lane&15)]
// for k-step pair s2 and 16-column tile ct (NT = n_pad/16).  K is zero padded to a multiple of 32 (= one staged
// A chunk, so the MFMA loops carry no tail conditions) plus FOUR extra all-zero k-step pairs (so the B prefetch of
// up to three pairs beyond the last chunk never needs a bounds check); n is zero padded to a multiple of 256.
//
// v_mfma_f64_16x16x4_f64 fragment maps: A lane l -> A[l&15][l>>4]; B lane l -> B[l>>4][l&15];
// C/D lane l, reg r -> C[(l>>4) + 4r][l&15].
#pragma once
#include "common.hpp"

namespace runia_mfma {

constexpr int KC = 32;      // k values per staged A chunk (4 k-step pairs)
constexpr int APITCH = 34;  // doubles; == 2 mod 32 -> the 16 rows x 2 k of a ds_read_b64 group hit 32 distinct bank pairs
constexpr int BN = 256;     // columns per workgroup pass: 4 waves x 4 tiles x 16

__host__ __device__ inline int64_t n_padded(int64_t n) { return (n + BN - 1) / BN * BN; }
__host__ __device__ inline int64_t k_padded(int64_t K) { return (K + KC - 1) / KC * KC; }        // multiple of 32
__host__ __device__ inline int64_t packed_pairs(int64_t K) { return k_padded(K) / 8 + 4; }       // + 4 zero pairs
__host__ __device__ inline int64_t packed_elems(int64_t K, int64_t n) { return packed_pairs(K) * 8 * n_padded(n); }

// Multiply-accumulate one 32-deep chunk: acc[a][c] += A[16a.., chunk] * B[chunk, this wave's 4 column tiles].
//   a_tile : LDS, row-major [>=16*RT rows][pitch doubles], already offset to the chunk's first k
//   bp     : this wave's packed-B pointer at the chunk's first k-step pair, already offset by (ctbase*64 + lane);
//            consecutive pairs are pair_stride double2 apart, consecutive column tiles 64 double2 apart
//   b0     : B fragments of the chunk's first pair (loaded by the previous call / the prologue); on return it holds
//            the first pair of the NEXT chunk, so B loads stay one pair (8 MFMAs = 512 matrix-pipe cycles) ahead.
// Branch-free on purpose: a conditional around an MFMA makes hipcc shuttle the accumulators between VGPRs and
// AGPRs (64 v_accvgpr moves per 8 MFMAs were measured before this form).
// NCT = column tiles per wave (4: a workgroup pass covers 256 columns; 2: 128 columns).
template <int RT, int NCT = 4>
__device__ __forceinline__ void mfma_chunk(d4 (&acc)[RT][NCT], const double* a_tile, int pitch, int li, int lg,
                                           const double2* __restrict__ bp, int64_t pair_stride,
                                           double2 (&b0)[NCT]) {
  double av[RT][8];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int t = 0; t < 8; ++t) av[a][t] = a_tile[(16 * a + li) * pitch + 4 * t + lg];
  double2 b1[NCT];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
    if ((s2 & 1) == 0) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) b1[c] = bp[(s2 + 1) * pair_stride + c * 64];
    } else {
#pragma unroll
      for (int c = 0; c < NCT; ++c) b0[c] = bp[(s2 + 1) * pair_stride + c * 64];
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const double2 bb = (s2 & 1) ? b1[c] : b0[c];
        const double bv = hh ? bb.y : bb.x;
#pragma unroll
        for (int a = 0; a < RT; ++a)
          acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][2 * s2 + hh], bv, acc[a][c], 0, 0, 0);
      }
    }
  }
}

// The same chunk product with the B fragments fetched THREE k-step pairs ahead (ring of four pair buffers, one per pair
// of a chunk, indexed statically): with few MFMAs per pair (RT * NCT * 2 = 4 for the 16 x 128 tiles of K2') the one-pair
// look-ahead of mfma_chunk is 256 matrix-pipe cycles, well under an L2 hit under load; three pairs are 768.
//   b[j] : on entry pairs 0..2 of THIS chunk are in b[0..2]; on return b[0..2] hold pairs 0..2 of the next chunk.
template <int RT, int NCT>
__device__ __forceinline__ void mfma_chunk_ring(d4 (&acc)[RT][NCT], const double* a_tile, int pitch, int li, int lg,
                                                const double2* __restrict__ bp, int64_t pair_stride,
                                                double2 (&b)[4][NCT]) {
  double av[RT][8];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int t = 0; t < 8; ++t) av[a][t] = a_tile[(16 * a + li) * pitch + 4 * t + lg];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) b[(s2 + 3) & 3][c] = bp[(s2 + 3) * pair_stride + c * 64];  // pair 3, then the next chunk's 0..2
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const double bv = hh ? b[s2][c].y : b[s2][c].x;
#pragma unroll
        for (int a = 0; a < RT; ++a)
          acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][2 * s2 + hh], bv, acc[a][c], 0, 0, 0);
      }
    }
  }
}

#if defined(__HIP_DEVICE_COMPILE__)  // buffer resources and their builtins exist in the device pass only
// ---- buffer-addressed form (K2'): no vector instruction steps an address ----
// gfx950 buffer loads take a 128-bit resource (scalar registers), a per-lane byte offset (one loop-invariant vector
// register), a scalar byte offset (stepped on the scalar unit) and an immediate; `buffer_load_dwordx4 ... lds` writes
// the 64 x 16 bytes of a wave straight into LDS at M0 + 16 * lane.  Out-of-range bytes read as zero.
using u4 = unsigned __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// LDS layout of a staged 32-deep chunk for the DMA form: 16-byte slots (two consecutive k of one row), slot index
// = kpair * BM + row (kpair = 0..15).  The A fragment of k-step s for lane (li, lg) is the double (lg & 1) of slot
// (2s + (lg >> 1)) * BM + row: the 64 lanes of a ds_read_b64 cover two contiguous 256-byte runs (no bank conflict), and
// a wave's DMA instruction j fills slots 64j .. 64j+63 from 16-byte global reads.
template <int RT, int NCT, int BM>
__device__ __forceinline__ void mfma_chunk_dma(d4 (&acc)[RT][NCT], const double* a_slots, int li, int lg,
                                               __amdgpu_buffer_rsrc_t brsrc, unsigned lane_bytes, unsigned pair0_bytes,
                                               unsigned pair_stride_bytes, double2 (&b)[4][NCT]) {
  double av[RT][8];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int t = 0; t < 8; ++t) av[a][t] = a_slots[(((2 * t + (lg >> 1)) * BM + 16 * a + li) << 1) + (lg & 1)];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(brsrc, lane_bytes + c * 1024u,
                                                         pair0_bytes + (unsigned)(s2 + 3) * pair_stride_bytes, 0);
      b[(s2 + 3) & 3][c] = __builtin_bit_cast(double2, v);
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const double bv = hh ? b[s2][c].y : b[s2][c].x;
#pragma unroll
        for (int a = 0; a < RT; ++a)
          acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][2 * s2 + hh], bv, acc[a][c], 0, 0, 0);
      }
    }
  }
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace runia_mfma
