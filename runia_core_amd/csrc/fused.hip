// Fused LaREM row pipeline (a11): LaRExInference.get_score after the backbone
// (reference inference/image_level.py:115-119) as three launches per batch instead of
// ~100 tiny ones per image:
//
//   K0  mc_mask      DropBlock draws (N,n_mc,H,W) -> per-image keep-flag table (the DropBlock2D masks of
//                    feature_extraction/abstract_classes.py:91-96 as 0/1 floats, drop layers sorted by mask sum).
//                    Bit arithmetic on one 64-bit word per drop layer; ~9 us per 10 000 images.
//   K1  mc_entropy   hooked latent maps (N,C,H,W) + table -> per-dimension entropies H (N,C) f64
//                    = MCSamplerModule.forward (feature_extraction/abstract_classes.py:81-101) fused with
//                      the per-dimension loop of get_dl_h_z (evaluation/entropy.py:77-82).  One thread owns
//                      one (image, channel): its H*W map stays in VGPRs, the n_mc MC samples are produced,
//                      sorted and reduced to one entropy without ever leaving registers.  VALU-issue-bound.
//   K2' proj_sq      H (N,D) f64 -> LaREM score (N) = apply_pca_transform (dimensionality_reduction.py:86)
//                    + MDLatentSpace.postprocess (inference/postprocessors.py:241-242) folded into ONE contraction
//                    on the f64 matrix cores (K2 pca_md is the two-contraction form, kept for non-PSD precisions).
//
// Sample order inside an image is irrelevant to the entropy (order statistics), so K1 visits the drop
// layers sorted by their mask sum and recomputes the per-element quotients (x*numel)/sum only when the sum
// changes (wave-uniform branch): same bits as the upstream op order.
// K1 (vector ALU) and K2' (f64 matrix instructions) share the SIMD's issue port on gfx950 and do not overlap
// (tools/microbench/mfma_valu_overlap.hip); they run back to back on one stream.
#include "common.hpp"
#include "entropy_core.hpp"
#include "mfma_f64_tile.hpp"
#include "philox.hpp"
#include "div_consts.hpp"

namespace {

using namespace runia_entropy;
using namespace runia_mfma;

constexpr int kMaxMC = 64;
typedef float f2 __attribute__((ext_vector_type(2)));

// Keep-flag table of one image (K0 -> K1), n_mc*(HW+2) floats:
//   n_mc records of HW floats: 0.0 where the drop layer removes the position, 2^-a where it keeps it, with the layer's
//     mask sum written as sum = 2^a * m (m odd).  The DropBlock rescale x * numel / sum = (x * numel / m) * 2^-a and a
//     power-of-two factor commutes with every rounding that follows (products, row sums, means), so it rides on the
//     flags for free;  positions in K1's operand order (mask_slot);
//   n_mc floats zh = RN(1/m) and n_mc floats zl = RN(1/m - zh): the layer's quotients are q = fma(u, zh, u * zl),
//     the correctly rounded u / m in two instructions (div_consts.hpp, proven by tools/verify_div_constants.py).
//     A layer that drops the whole map has zh = NaN (0 * numel / 0 upstream).
// Drop layers are sorted by m (layers that share m share their quotients; m = 1 - sums 1, 2, 4, 8, 16, ... - needs none).
__device__ __forceinline__ void layer_consts(int cnt, int& m, float& flag, float& zh, float& zl) {
  const int a = cnt ? __ffs(cnt) - 1 : 0;
  m = cnt >> a;
  flag = __uint_as_float((unsigned)(127 - a) << 23);  // 2^-a
  zh = cnt ? __uint_as_float(runia_div::kDivHi[(m - 1) >> 1]) : NAN;
  zl = cnt ? __uint_as_float(runia_div::kDivLo[(m - 1) >> 1]) : 0.f;
}

__device__ __forceinline__ float add_f32(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float div_newton(float u, float den, float r) {
  const float q = u * r;
  const float e = fmaf(-den, q, u);
  return fmaf(e, r, q);
}

// position p of the map -> slot of its keep flag in a table record (K1's operand order)
template <int HT, int WT>
__host__ __device__ constexpr int mask_slot(int p) {
  // even H: rows 2k and 2k+1 are interleaved so that one 64-bit scalar operand feeds a packed FMA
  if constexpr (HT % 2 == 0) {
    const int y = p / WT, xw = p - y * WT;
    return ((y >> 1) * WT + xw) * 2 + (y & 1);
  } else {
    return p;
  }
}

// K0 for maps of at most 64 positions (2x2, 4x4, 7x7, 8x8 ...): a drop layer is one 64-bit word and
// the whole derivation is bit arithmetic in registers - no LDS, no barrier, one HBM round trip.
//   seeds   : lane = (layer, position); `draw < gamma` -> ballot -> 64/HW layers per word
//   dilation: lane = layer; separable OR of the seed word shifted over the block window (columns, then rows)
//   sort    : rank by popcount through readlane broadcasts; each lane stores its own table record
// The LDS version above spends ~110 instructions per (layer, position) and 17-18 us per 10 000 images whatever its
// workgroup shape (an empty launch of the same grid takes 5 us; profiles/README.md).
template <int HT, int WT>
__host__ __device__ constexpr int slot_position(int slot) {  // inverse of mask_slot
  if constexpr (HT % 2 == 0) {
    const int e = slot & 1, t = slot >> 1;
    const int k = t / WT, xw = t - k * WT;
    return (2 * k + e) * WT + xw;
  } else {
    return slot;
  }
}

template <int HT, int WT>
__device__ __forceinline__ unsigned long long columns_below(int k) {  // positions (y, x) with x < k
  unsigned long long row = (k >= 64) ? ~0ull : ((1ull << k) - 1ull), m = 0ull;
#pragma unroll
  for (int y = 0; y < HT; ++y) m |= row << (y * WT);
  return m;
}

constexpr int kMaskBitsWaves = 4;  // images per workgroup (independent waves)

template <int HT, int WT, int NP, bool REDRAW = false>
__global__ __launch_bounds__(64 * kMaskBitsWaves) void mc_mask_bits_kernel(const float* __restrict__ rnd,
                                                                           int64_t rand_stride,
                                                                           float* __restrict__ table, int64_t N,
                                                                           int n_mc, float gamma, int block_size,
                                                                           int identity, int sort_layers,
                                                                           uint64_t rng_seed, int64_t first_image) {
  constexpr int HW = HT * WT;
  static_assert(HW <= 64 && NP <= 64, "one drop layer per 64-bit word");
  // draw i = layer * HW + position sits in bit i % 64 of ballot word i / 64.  Map sizes that divide 64 (2x2, 4x4, 8x8)
  // keep whole layers inside a word; any other size (7x7: 49 bits) lets a layer straddle two words, and the lane that
  // owns the layer stitches its bits together from them.
  constexpr bool DIV = (64 % HW == 0);
  constexpr int WORDS = (NP * HW + 63) / 64;
  constexpr unsigned long long FULL = (HW == 64) ? ~0ull : ((1ull << HW) - 1ull);
  const int lane = threadIdx.x & 63;
  const int64_t img = (int64_t)blockIdx.x * kMaskBitsWaves + (threadIdx.x >> 6);
  if (img >= N) return;  // wave-uniform
  const int bit0 = lane * HW, word0 = bit0 >> 6, off = bit0 & 63;  // where this lane's layer (lane = layer) starts
  const int pad = block_size / 2;

  // seed word of this lane's layer from attempt `attempt` of the draws (attempt > 0: counter mode only)
  auto seeds_of = [&](unsigned attempt) -> unsigned long long {
    float d[WORDS];
    if (rnd) {
      const float* r = rnd + img * rand_stride;
#pragma unroll
      for (int t = 0; t < WORDS; ++t) {
        const int i = lane + 64 * t;
        d[t] = (i < n_mc * HW) ? r[i] : 1.0f;
      }
    } else {  // counter mode (philox.hpp): this lane's words 4q .. 4q+3 are the four components of one Philox block
#pragma unroll
      for (int q = 0; q < (WORDS + 3) / 4; ++q) {
        const runia_philox::u4 b = runia_philox::lane_block(rng_seed, (uint64_t)(first_image + img), lane, q, attempt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = 4 * q + j;
          if (t < WORDS) d[t] = (lane + 64 * t < n_mc * HW) ? runia_philox::component(b, j) : 1.0f;
        }
      }
    }
    unsigned long long lo = 0ull, hi = 0ull;
#pragma unroll
    for (int t = 0; t < WORDS; ++t) {
      const unsigned long long w = __ballot(d[t] < gamma);
      if (t == word0) lo = w;
      if (!DIV && t == word0 + 1) hi = w;
    }
    unsigned long long sd = lo >> off;
    if (!DIV && off + HW > 64) sd |= hi << (64 - off);
    return sd & FULL;
  };
  auto keep_of = [&](unsigned long long seed) -> unsigned long long {
    unsigned long long hx = 0ull;
    for (int dx = 0; dx < block_size; ++dx) {  // dropped(y, x) |= seed(y, x + ox)
      const int ox = dx - pad;
      if (ox >= WT || -ox >= WT) continue;
      hx |= (ox >= 0) ? ((seed >> ox) & columns_below<HT, WT>(WT - ox))
                      : ((seed << -ox) & ~columns_below<HT, WT>(-ox) & FULL);
    }
    unsigned long long dropped = 0ull;
    for (int dy = 0; dy < block_size; ++dy) {  // ... |= hx(y + oy, x)
      const int oy = dy - pad;
      if (oy >= HT || -oy >= HT) continue;
      dropped |= (oy >= 0) ? (hx >> (oy * WT)) : ((hx << (-oy * WT)) & FULL);
    }
    return ~dropped & FULL;
  };

  unsigned long long keep = identity ? FULL : keep_of(seeds_of(0u));
  if (REDRAW && !identity && !rnd) {
    // throughput mode, opt-in: a drop layer that removed the whole map (0 * numel / 0 = NaN upstream) draws again from
    // the next counter block of the same image - a bounded, wave-uniform loop that only images with such a layer enter
    for (unsigned attempt = 1; attempt <= 16u; ++attempt) {
      if (__ballot(lane < n_mc && keep == 0ull) == 0ull) break;
      const unsigned long long again = keep_of(seeds_of(attempt));
      if (keep == 0ull) keep = again;
    }
  }
  const int cnt = __popcll(keep);
  int m;
  float flag, zh, zl;
  layer_consts(cnt, m, flag, zh, zl);
  const int key = m ? m : 127;  // fully dropped maps last
  int rank = 0;  // stable counting sort of the drop layers by the odd part of their mask sum
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int o = __shfl(key, j, 64);
    rank += (j < n_mc) && ((o < key) || (o == key && j < lane));
  }
  if (!sort_layers) rank = lane;  // table in the order of the draws (runia_mc_stack_table_f32)
  if (lane >= n_mc) return;
  float* out = table + img * (int64_t)(n_mc * (HW + 2));
  float* rec = out + rank * HW;
  if constexpr (HW % 4 == 0) {
#pragma unroll
    for (int c = 0; c < HW / 4; ++c) {
      float4 v;
      v.x = ((keep >> slot_position<HT, WT>(4 * c)) & 1ull) ? flag : 0.f;
      v.y = ((keep >> slot_position<HT, WT>(4 * c + 1)) & 1ull) ? flag : 0.f;
      v.z = ((keep >> slot_position<HT, WT>(4 * c + 2)) & 1ull) ? flag : 0.f;
      v.w = ((keep >> slot_position<HT, WT>(4 * c + 3)) & 1ull) ? flag : 0.f;
      reinterpret_cast<float4*>(rec)[c] = v;
    }
  } else {
#pragma unroll
    for (int q = 0; q < HW; ++q) rec[q] = ((keep >> slot_position<HT, WT>(q)) & 1ull) ? flag : 0.f;
  }
  out[n_mc * HW + rank] = zh;
  out[n_mc * (HW + 1) + rank] = zl;
}

// ------------------------------------------------------------------------------------------
// K1: latent map -> MC samples -> entropy.   one workgroup = (image, block of kK1Block channels); no LDS, no barrier.
// The mask table is wave-uniform: it arrives through the scalar cache (s_load) and enters the arithmetic as
// SGPR operands.  A drop layer is acc = fma(q, keep, acc) over the map in the upstream summation order
// (keep = 1: the add of the reference; keep = 0: acc unchanged), two rows per v_pk_fma_f32.
// ------------------------------------------------------------------------------------------
#ifndef K1_BLOCK
#define K1_BLOCK 128
#endif
constexpr int kK1Block = K1_BLOCK;  // channels (threads) per workgroup: 185 us vs 199 us with 256 at N = 10 000

// ---- ROI source of K1 (BASELINE config 4, reference feature_extraction/object_level.py:340-349 + 312-367): instead of a
// (K, C, PH, PW) tensor that roi_align wrote, a thread builds its PH x PW map in registers from the feature map itself.
// `x` of the kernel then points at a RoiSource (written by roi_sample_table_kernel, roi.hip): the feature map in NHWC
// (channel = lane: every bilinear tap of a wave is one contiguous 4 * 64-byte run) and, per ROI, a SEPARABLE table of its
// bilinear samples - per sample row (ph, iy): byte offsets of the two map rows and the weights hy, ly; per sample column
// (pw, ix): byte offsets of the two map columns and hx, lx; two bit masks of the rows / columns
// more than a pixel outside the map (their samples contribute 0) - 480 bytes for 7x7 bins of 2x2 samples,
// wave-uniform, read through the scalar cache.  (A table of all 196 samples with their four offsets and four products,
// 6.3 KB per ROI, kept every vector instruction out of the weights but streamed 376 MB per 60 000 ROIs through the scalar
// cache: 7 of the 11 ms the launch took.)  A bin is the sum of its G x G samples in (iy, ix) order, each
// (hy hx) v1 + (hy lx) v2 + (ly hx) v3 + (ly lx) v4, divided by G * G: the arithmetic of roi_align_kernel, same bits
// (tests/test_api_gpu.py).  The taps are buffer loads with the sample's offset as the scalar operand: no vector instruction
// goes into addressing.  How it got to 3.6 ms per 60 000 ROIs x 256 channels x 7x7 bins (roi_align 9.4 + K1 1.1 as two
// launches): a table of all 196 samples (offsets and products, 6.3 KB per ROI) streamed 376 MB through the scalar cache -
// 11.2 ms, 7 of them scalar-cache misses; the separable table 8.7 ms, every sample's four loads followed by a wait (four
// loads in flight per wave, ~380 ns each); all 56 taps of a sample row requested before the first is used: 3.6 ms, the
// vector L1 path now at ~18 TB/s of taps.  (The four taps of a sample as ONE 16-byte load from a map of quads
// (f[y][x], f[y][x+1], f[y+1][x], f[y+1][x+1]): the same bytes, a quarter of the instructions, no reuse left for the L1 -
// 11.1 ms in the unpipelined form, not kept.)
struct RoiSource {
  const float* nhwc;      // [B, H, W, C]
  const unsigned* table;  // per ROI: 8 dwords (image, row mask, column mask, 0 ...) + (PH * G + PW * G) x 4 dwords
  int64_t image_bytes;    // H * W * C * 4
  int C;
  int roi_dwords;         // 8 + 4 * (PH * G + PW * G)
};

template <int HT, int WT, int G>
__device__ __forceinline__ void roi_load_map(float (&u)[HT * WT], const RoiSource* __restrict__ src, int64_t roi, int c) {
  // the table is read through the constant address space: loads the compiler may keep on the scalar unit (through a
  // plain pointer loaded from memory it reads the wave-uniform entries with vector loads and wraps every tap in a
  // readfirstlane loop)
  typedef const __attribute__((address_space(4))) unsigned* cptr;
  cptr tab = (cptr)(src->table + roi * (int64_t)src->roi_dwords);  // wave-uniform
  const unsigned image = tab[0];
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(src->nhwc) + (int64_t)image * (src->image_bytes / 4), 0, (int)src->image_bytes, 0x00020000);
  const int voff = c * 4;
  cptr rows = tab + 8, cols = tab + 8 + 4 * (HT * G);
  constexpr float count = (float)(G * G);
  // The column entries first (offsets stay in scalar registers, the weights move to vector registers: a vector multiply
  // takes one scalar operand, the row's), then per sample row ALL its taps are requested before the first is used: left
  // to the compiler every sample's four loads were followed by a wait - four loads in flight per wave, 380 ns each,
  // 8.7 ms per 60 000 ROIs x 256 channels with the load path a sixth busy.
  constexpr int SX = WT * G, SY = HT * G;
  unsigned ox_lo[SX], ox_hi[SX];
  float hx[SX], lx[SX];
#pragma unroll
  for (int sx = 0; sx < SX; ++sx) {
    ox_lo[sx] = cols[4 * sx];
    ox_hi[sx] = cols[4 * sx + 1];
    hx[sx] = __uint_as_float(cols[4 * sx + 2]);
    lx[sx] = __uint_as_float(cols[4 * sx + 3]);
  }
  float acc[HT * WT];
#pragma unroll
  for (int p = 0; p < HT * WT; ++p) acc[p] = 0.f;
  // The two pixel rows of a sample row stay in registers: sample rows are 1/G of a bin apart, so on a box of fewer than
  // 2 * SY feature rows the next sample row uses the same two pixel rows again (no load) or the upper one becomes the lower
  // (SX * 2 loads instead of SX * 4).  Which of the three it is depends on the table alone (wave-uniform), and a pixel read
  // once is the pixel read twice: the same bits as one load per tap.
  float tl[SX][2], th[SX][2];
  unsigned cur_lo = 0xffffffffu, cur_hi = 0xffffffffu;  // (no pixel row starts at this byte offset)
#pragma unroll
  for (int sy = 0; sy < SY; ++sy) {
    const unsigned oy_lo = rows[4 * sy], oy_hi = rows[4 * sy + 1];
    const float hy = __uint_as_float(rows[4 * sy + 2]), ly = __uint_as_float(rows[4 * sy + 3]);
    if (oy_lo != cur_lo || oy_hi != cur_hi) {
      if (oy_lo == cur_hi) {
#pragma unroll
        for (int sx = 0; sx < SX; ++sx) { tl[sx][0] = th[sx][0]; tl[sx][1] = th[sx][1]; }
      } else {
#pragma unroll
        for (int sx = 0; sx < SX; ++sx) {
          tl[sx][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (int)(oy_lo + ox_lo[sx]), 0));
          tl[sx][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (int)(oy_lo + ox_hi[sx]), 0));
        }
      }
#pragma unroll
      for (int sx = 0; sx < SX; ++sx) {
        th[sx][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (int)(oy_hi + ox_lo[sx]), 0));
        th[sx][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (int)(oy_hi + ox_hi[sx]), 0));
      }
      cur_lo = oy_lo;
      cur_hi = oy_hi;
    }
    __builtin_amdgcn_sched_barrier(0);  // (the loads above stay above)
#pragma unroll
    for (int sx = 0; sx < SX; ++sx) {
      // (a sample outside the map: zero weights and taps beyond the buffer, which read as 0 - the table sees to both)
      const float w1 = hy * hx[sx], w2 = hy * lx[sx], w3 = ly * hx[sx], w4 = ly * lx[sx];
      const float val = w1 * tl[sx][0] + w2 * tl[sx][1] + w3 * th[sx][0] + w4 * th[sx][1];
      float& bin = acc[(sy / G) * WT + (sx / G)];
      bin += val;  // a bin receives its samples in (iy, ix) order
      // added NOW: left alone the compiler keeps all SY * SX sample values in registers to add them in pairs at the end
      // (247 vector registers, two waves per SIMD; the loader needs the bins, two pixel rows and little else)
      asm volatile("" : "+v"(bin));
    }
  }
#pragma unroll
  for (int p = 0; p < HT * WT; ++p) u[p] = acc[p] / count;
}

#ifndef K1_IMGS
#define K1_IMGS 1  // images per workgroup, one after the other (timing experiments: fewer, longer waves)
#endif
template <int HT, int WT, int NP, int K, bool FULL, bool ENTROPY, int ROI_G>
__device__ __forceinline__ void mc_entropy_item(const float* __restrict__ x, const float* __restrict__ table,
                                                double* __restrict__ h, float* __restrict__ z_out,
                                                double* __restrict__ zero_fill, int64_t N, int C, int n_mc_rt,
                                                double min_dist, double const_term, double inv_n, int64_t img, int c) {
  constexpr int HW = HT * WT;
  constexpr bool PAIRS = (HT % 2 == 0);
  const int n_mc = FULL ? NP : n_mc_rt;
  if (img >= N) return;
  if (zero_fill && c == 0) zero_fill[img] = 0.0;  // the accumulator of the score launch that follows (optional)
  if (c >= C) return;
  const float* mk = table + img * (int64_t)(n_mc * (HW + 2));  // wave-uniform
  const float* zhs = mk + n_mc * HW;
  const float* zls = zhs + n_mc;
  float u[HW];
  if constexpr (ROI_G > 0) {
    roi_load_map<HT, WT, ROI_G>(u, reinterpret_cast<const RoiSource*>(x), img, c);
  } else {
    // 16-byte loads at a 4*HW-byte lane stride: measured faster than staging the block's contiguous run
    // through LDS (214 vs 271 us at N = 10 000, profiles/README.md) - the kernel is VALU-bound, not HBM-bound
    const float* xc = x + (img * C + c) * (int64_t)HW;
    if constexpr (HW % 4 == 0) {
#pragma unroll
      for (int p = 0; p < HW / 4; ++p) {
        const float4 v = reinterpret_cast<const float4*>(xc)[p];
        u[4 * p] = v.x; u[4 * p + 1] = v.y; u[4 * p + 2] = v.z; u[4 * p + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int p = 0; p < HW; ++p) u[p] = xc[p];
    }
  }
  constexpr bool w_pow2 = (WT & (WT - 1)) == 0, h_pow2 = (HT & (HT - 1)) == 0;
  // Power-of-two scalings are exact and commute with rounding (no overflow / underflow at feature-map magnitudes),
  // so they are moved to where they cost nothing: with sum(bm) = 2^a * m, (x*numel)/sum = (x*numel/m) * 2^-a, the
  // factor 2^-a sits on the keep flags (K0), and when numel = H*W is a power of two it cancels against the 1/(H*W) of
  // mean_H(mean_W(.)): the sample is the plain sum of the quotients x/m over the kept positions, in upstream order.
  constexpr bool hw_pow2 = w_pow2 && h_pow2;
  if constexpr (!hw_pow2) {
#pragma unroll
    for (int p = 0; p < HW; ++p) u[p] *= (float)HW;
  }
  const float rW = 1.0f / (float)WT, rH = 1.0f / (float)HT;
  float z[NP];
  float q[PAIRS ? 1 : HW];   // quotients (x*numel)/sum ...
  f2 q2[PAIRS ? HW / 2 : 1];  // ... as row pairs in mask_slot order when H is even
  // quotients start out as x / 1 (the m = 1 group comes first in the sorted table).  The divisor in use is tracked by its
  // BIT PATTERN: both values sit in scalar registers, and gfx950 compares scalar integers on the scalar unit but floats only
  // on the vector unit (v_mov + v_cmp per drop layer: 32 of the kernel's 772 vector instructions)
  unsigned cur_zh_bits = __float_as_uint(1.0f);
#pragma unroll
  for (int p = 0; p < HW; ++p) {
    if constexpr (PAIRS) q2[mask_slot<HT, WT>(p) >> 1][mask_slot<HT, WT>(p) & 1] = u[p];
    else q[p] = u[p];
  }
  // G drop layers per trip of a rolled loop: G*HW keep flags live in SGPRs at a time (a fully unrolled loop
  // lets the compiler hoist all n_mc*HW scalar loads and spill them).  z is a shift register: constant indices
  // only, and the sample order is irrelevant to the sort that follows.
#ifndef K1_GCAP
#define K1_GCAP 128
#endif
  // (a run-time n_mc adds the clamped indices to the scalar side: with 8 layers per trip that instantiation spilled
  // scalar registers to vector lanes - 254 v_readlane / v_writelane; 4 per trip do not)
  constexpr int GCAP = FULL ? K1_GCAP : K1_GCAP / 2;
  constexpr int G = (GCAP / HW < 1) ? 1 : ((GCAP / HW > NP) ? NP : GCAP / HW);
  // the +inf fill only matters when fewer than NP samples are produced (FULL writes every slot: NP / G whole trips)
  if constexpr (!(FULL && NP % G == 0)) {
#pragma unroll
    for (int s = 0; s < NP; ++s) z[s] = INFINITY;
  }
#ifndef K1_TRIP_UNROLL
#define K1_TRIP_UNROLL 1
#endif
#pragma unroll K1_TRIP_UNROLL
  for (int s0 = 0; s0 < NP; s0 += G) {
    float znew[G];
    // all scalar operands of this trip are requested up front (one wait instead of one per drop layer)
    float dg[G], rg[G], mg[G][HW];
    // FULL: one base pointer per trip and constant offsets from it (the scalar loads then carry the layer's offset as an
    // immediate; with sc = s0 + g inside the index the compiler kept one 64-bit address per drop layer in scalar registers)
    const float* mk_t = mk + (FULL ? s0 * HW : 0);
    const float* zhs_t = zhs + (FULL ? s0 : 0);
    const float* zls_t = zls + (FULL ? s0 : 0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int sc = FULL ? g : ((s0 + g < n_mc) ? s0 + g : n_mc - 1);
      dg[g] = zhs_t[sc];
      rg[g] = zls_t[sc];
#pragma unroll
      for (int p = 0; p < HW; ++p) mg[g][p] = mk_t[sc * HW + p];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int s = s0 + g;
      znew[g] = INFINITY;
      if (FULL || s < n_mc) {
        const float zh = dg[g], zl = rg[g];
        if (__float_as_uint(zh) != cur_zh_bits) {  // wave-uniform: every thread of the block works on the same image
#pragma unroll
          for (int p = 0; p < HW; ++p) {
            const float qv = fmaf(u[p], zh, u[p] * zl);  // u / m, correctly rounded (div_consts.hpp)
            if constexpr (PAIRS) q2[mask_slot<HT, WT>(p) >> 1][mask_slot<HT, WT>(p) & 1] = qv;
            else q[p] = qv;
          }
          cur_zh_bits = __float_as_uint(zh);
        }
        const float* m = mg[g];
        float col;
        if constexpr (PAIRS) {
          float rows[HT];
#pragma unroll
          for (int k2 = 0; k2 < HT / 2; ++k2) {
            const f2* qq = q2 + k2 * WT;
            const float* mm = m + 2 * k2 * WT;
            f2 acc = qq[0] * (f2){mm[0], mm[1]};  // torch_chain<WT>(0) == 0
#pragma unroll
            for (int xi = 1; xi < WT; ++xi) {  // the row in torch's CPU summation order (common.hpp)
              const int xw = torch_chain<WT>(xi);
              acc = __builtin_elementwise_fma(qq[xw], (f2){mm[2 * xw], mm[2 * xw + 1]}, acc);
            }
            if constexpr (hw_pow2) {
              // scaled once below
            } else if constexpr (w_pow2) {
              acc = acc * (f2){rW, rW};
            } else {
              acc = (f2){div_newton(acc.x, (float)WT, rW), div_newton(acc.y, (float)WT, rW)};
            }
            rows[2 * k2] = acc.x;
            rows[2 * k2 + 1] = acc.y;
          }
          // plain v_add_f32: left to itself the compiler adds the halves of the row-pair registers with v_pk_add_f32 and
          // op_sel, one useful add per packed instruction (4.2 issue cycles instead of 2.5)
          col = rows[0];
#pragma unroll
          for (int yi = 1; yi < HT; ++yi) col = add_f32(col, rows[torch_chain<HT>(yi)]);
        } else {
          float rm[HT];
#pragma unroll
          for (int y = 0; y < HT; ++y) {
            float rowsum = q[y * WT] * m[y * WT];
#pragma unroll
            for (int xi = 1; xi < WT; ++xi) {
              const int xw = torch_chain<WT>(xi);
              rowsum = fmaf(q[y * WT + xw], m[y * WT + xw], rowsum);
            }
            rm[y] = w_pow2 ? rowsum * rW : div_newton(rowsum, (float)WT, rW);
          }
          col = rm[0];
#pragma unroll
          for (int yi = 1; yi < HT; ++yi) col += rm[torch_chain<HT>(yi)];
        }
        if constexpr (hw_pow2 && PAIRS) znew[g] = col;
        else znew[g] = h_pow2 ? col * rH : div_newton(col, (float)HT, rH);
        if (z_out)  // optional copy of the MC samples (drop-layer order is the table's; tests only)
          z_out[(img * n_mc + s) * (int64_t)C + c] = znew[g];
      }
    }
#pragma unroll
    for (int i = NP - 1; i >= G; --i) z[i] = z[i - G];
#pragma unroll
    for (int g = 0; g < G; ++g) z[g] = znew[g];
  }
  if constexpr (ENTROPY) {
    // A NaN sample - a fully dropped map (0*numel/0 upstream), a NaN or infinite activation - makes the entropy NaN
    // upstream; the min/max sort below would silently drop it, so it is caught here: a sum is NaN iff a term is
    // (+inf pads of a short column cannot cancel: the samples are finite otherwise).
    float nan_probe = z[0];
#pragma unroll
    for (int s = 1; s < NP; ++s) nan_probe += z[s];
    const bool bad = nan_probe != nan_probe;
    sort_asc<NP>(z);
    double res = const_term + inv_n * column_log_sum<NP, K, FULL>(z, n_mc, min_dist);
    if (bad) res = NAN;
    h[img * C + c] = res;
  }
}

template <int HT, int WT, int NP, int K, bool FULL, bool ENTROPY = true, int ROI_G = 0>
#ifdef K1_WAVES
__attribute__((amdgpu_waves_per_eu(K1_WAVES, 8)))
#endif
__global__ __launch_bounds__(kK1Block) void mc_entropy_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ table,
                                                          double* __restrict__ h, float* __restrict__ z_out,
                                                          double* __restrict__ zero_fill, int64_t N, int C,
                                                          int n_mc_rt, double min_dist, double const_term,
                                                          double inv_n) {
  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the
  // channel blocks of one image are given ids that are congruent mod 8 - the image's keep-flag table is then
  // fetched into ONE L2 (with the plain (block, image) grid up to 4 XCDs fetched it: +35 MB per 10 000 images).
  const unsigned chunks = (unsigned)(C + kK1Block - 1) / kK1Block;
  const unsigned slot = blockIdx.x >> 3;
  const int c = (int)(slot % chunks) * kK1Block + threadIdx.x;
#pragma unroll 1
  for (int rep = 0; rep < K1_IMGS; ++rep) {
    const int64_t img = ((int64_t)(slot / chunks) * K1_IMGS + rep) * 8 + (blockIdx.x & 7);
    mc_entropy_item<HT, WT, NP, K, FULL, ENTROPY, ROI_G>(x, table, h, z_out, zero_fill, N, C, n_mc_rt, min_dist, const_term,
                                                        inv_n, img, c);
  }
}

// ------------------------------------------------------------------------------------------
// K2: H -> PCA (optional) -> LaREM score.  One workgroup = BM rows; both contractions on f64 MFMA.
// Packed weights layout: see gemm_f64.hip.
// ------------------------------------------------------------------------------------------
struct PcaMdArgs {
  const double* h;         // [N, D]
  const double* packed_ct; // pack(C.T [D, n]) or null (no PCA: n == D)
  const double* bias;      // [n]
  const double* scale;     // [n] or null
  const double* md_mean;   // [n]
  const double* packed_p;  // pack(P [n, n])
  double* score;           // [N]
  double* y_out;           // optional [N, n]: the projected rows (tests)
  int64_t N, D, n;
};

template <int RT>  // row tiles of 16 per workgroup (BM = 16*RT)
__global__ __launch_bounds__(256) void pca_md_kernel(PcaMdArgs g) {
  constexpr int BM = 16 * RT;
  extern __shared__ double lds[];
  // layout: lds_a [2][BM][APITCH] | lds_y [BM][ypitch] | part [4][BM]
  const int64_t n_pad = n_padded(g.n);
  const int ypitch = (int)n_pad + 2;  // == 2 mod 32 -> conflict-free A-fragment reads
  double* lds_a = lds;
  double* lds_y = lds + 2 * BM * APITCH;
  double* part = lds_y + (size_t)BM * ypitch;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t NT = n_pad / 16;
  const int64_t r0 = (int64_t)blockIdx.x * BM;

  // ---------------- phase A: d = (H C^T - bias) / scale - md_mean  -> lds_y ----------------
  if (g.packed_ct) {
    const int64_t nchunks = k_padded(g.D) / KC;
    constexpr int PER_T = BM * KC / 256;  // doubles staged per thread per chunk (4 or 2)
    for (int64_t cb = 0; cb < n_pad / BN; ++cb) {
      const int64_t ctbase = cb * 16 + wave * 4;
      d4 acc[RT][4];
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
      const double2* bp = reinterpret_cast<const double2*>(g.packed_ct) + ctbase * 64 + lane;
      double2 b0[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) b0[c] = bp[c * 64];
      double areg[PER_T];
      const int srow = (tid * PER_T) / KC, skk = (tid * PER_T) % KC;
      auto load_a = [&](int64_t kc) {
        const int64_t gr = r0 + srow;
#pragma unroll
        for (int q = 0; q < PER_T; ++q) {
          const int64_t gk = kc + skk + q;
          areg[q] = (gr < g.N && gk < g.D) ? g.h[gr * g.D + gk] : 0.0;
        }
      };
      load_a(0);
      int buf = 0;
      for (int64_t ch = 0; ch < nchunks; ++ch) {
#pragma unroll
        for (int q = 0; q < PER_T; ++q) lds_a[(buf * BM + srow) * APITCH + skk + q] = areg[q];
        __syncthreads();
        if (ch + 1 < nchunks) load_a((ch + 1) * KC);
        mfma_chunk<RT>(acc, lds_a + buf * BM * APITCH, APITCH, li, lg, bp + ch * 4 * NT * 64, NT * 64, b0);
        buf ^= 1;
      }
      // epilogue: sklearn transform then the LaREM centring, kept in LDS
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t col = (ctbase + c) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * a + lg + 4 * r;
            double d = 0.0;
            if (col < g.n) {
              double y = acc[a][c][r] - g.bias[col];
              if (g.scale) y = y / g.scale[col];
              if (g.y_out && r0 + row < g.N) g.y_out[(r0 + row) * g.n + col] = y;
              d = y - g.md_mean[col];
            }
            lds_y[row * ypitch + col] = d;  // zero padding beyond n keeps phase B exact
          }
        }
      __syncthreads();
    }
  } else {
    // no PCA: d = h - md_mean straight into lds_y (n == D)
    for (int i = tid; i < BM * (int)n_pad; i += 256) {
      const int row = i / (int)n_pad, col = i - row * (int)n_pad;
      double d = 0.0;
      if (r0 + row < g.N && col < g.n) d = g.h[(r0 + row) * g.D + col] - g.md_mean[col];
      lds_y[row * ypitch + col] = d;
    }
    __syncthreads();
  }

  // ---------------- phase B: score = -sum_j (d P)_j d_j, A fragments straight from lds_y ------
  double rowdot[RT][4];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) rowdot[a][r] = 0.0;
  const int64_t kchunks = k_padded(g.n) / KC;
  for (int64_t cb = 0; cb < n_pad / BN; ++cb) {
    const int64_t ctbase = cb * 16 + wave * 4;
    d4 acc[RT][4];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
    const double2* bp = reinterpret_cast<const double2*>(g.packed_p) + ctbase * 64 + lane;
    double2 b0[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) b0[c] = bp[c * 64];
    for (int64_t ch = 0; ch < kchunks; ++ch)
      mfma_chunk<RT>(acc, lds_y + ch * KC, ypitch, li, lg, bp + ch * 4 * NT * 64, NT * 64, b0);
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = (int)((ctbase + c) * 16) + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) rowdot[a][r] += acc[a][c][r] * lds_y[(16 * a + lg + 4 * r) * ypitch + col];
      }
  }
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = rowdot[a][r];
      v += shfl_xor_f64(v, 1);
      v += shfl_xor_f64(v, 2);
      v += shfl_xor_f64(v, 4);
      v += shfl_xor_f64(v, 8);
      if (li == 0) part[wave * BM + 16 * a + lg + 4 * r] = v;
    }
  __syncthreads();
  if (tid < BM) {
    const int64_t row = r0 + tid;
    if (row < g.N) g.score[row] = -(((part[tid] + part[BM + tid]) + part[2 * BM + tid]) + part[3 * BM + tid]);
  }
}

// ------------------------------------------------------------------------------------------
// K2': the same score as K2 from ONE contraction.  With d = A h + b (A = diag(1/scale) C, b = -bias/scale - mu) and
// P = W^T W (W = diag(1/sqrt(s)) U^T over the eigenpairs pinvh kept),  -d^T P d = -|| (W A) h + W b ||^2, so the
// 512->256 projection, the whitening, the centring and the 256x256 quadratic form collapse into M = W A (r x D) and
// c = W b, folded once at setup.  2*D*r FLOP per row instead of 2*D*n + 2*n^2; no LDS round trip for the projection.
// ------------------------------------------------------------------------------------------
struct ProjSqArgs {
  const double* h;        // [N, D]
  const double* packed_m; // pack(M^T [D, r])
  const double* c;        // [r]
  double* score;          // [N]            (column split 1)
  double* partial;        // [2, N] row sums (column split 2; proj_sq_combine_kernel finishes)
  int64_t N, D, r;
};

// NCT column tiles per wave: 4 = the workgroup covers every 256-column block whole (grid.y = 1); 2 = two
// workgroups share a row tile, each taking one 128-column half of every block (grid.y = 2) and leaving its row
// sums of squares in `partial`.  The split halves the unit of work: at N = 10 000 the 625 row tiles are 2.44 per
// CU (3 on some, 2 on most: 19 % of the matrix pipes idle at the end); 1250 half tiles finish within 3 %.
// Few components (the reference's default is nro_components = 16): SPLIT = false with NCT = 1 covers r <= 64 (each wave
// one 16-column tile), with NCT = 2 r <= 128 - the zero-padded column tiles beyond r are simply not computed (the
// 256-column forms spend the same 45-50 us on r = 16 as on r = 256).
// 16-row forms: 96 vector registers = 5 waves per SIMD, so the 1 250 workgroups of 10 000 rows are all resident at once
// (98 registers = 4 waves left 226 workgroups for a second, thinly occupied round: 45.5 -> 44.5 us)
#ifndef K2_WAVES
#define K2_WAVES 5
#endif
#ifndef K2_DMA
#define K2_DMA 1
#endif
#ifndef K2_WAVES_DMA
#define K2_WAVES_DMA 5  // (the DMA form needs 74 vector registers, but 6 waves per SIMD measured 2 % slower than 5)
#endif
template <int RT, int NCT, bool ACCUMULATE, bool SPLIT, bool DMA>
__global__ __launch_bounds__(256, (RT == 1 && NCT <= 2) ? (DMA ? K2_WAVES_DMA : K2_WAVES) : 1) void proj_sq_kernel(ProjSqArgs g) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the launch stub; the body uses device-only buffer builtins)
  constexpr int BM = 16 * RT;
  __shared__ double lds_a[DMA ? 1 : 2 * BM * APITCH];
  // the DMA form's two chunk buffers are separate objects: the compiler orders a ds_read behind every LDS DMA it cannot
  // prove disjoint (one array: `s_waitcnt vmcnt(0)` in front of each chunk's first read, the DMA of the NEXT chunk included)
  __shared__ __attribute__((aligned(16))) double lds_d0[DMA ? BM * KC : 2];
  __shared__ __attribute__((aligned(16))) double lds_d1[DMA ? BM * KC : 2];
  constexpr int NG = (NCT + 1) / 2;  // 32-column groups per wave: the unit of the (launch-independent) summation order
  __shared__ double part[4 * NG * BM];
  const int64_t n_pad = n_padded(g.r);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t NT = n_pad / 16;
  // Column split (NCT == 2): a 1-D grid whose ids i and i + 8 are the two column halves of one row tile.  Consecutive
  // workgroup ids go round-robin over the 8 XCDs, so the two halves run on the SAME XCD one dispatch round apart and the
  // second one finds the row tile in that XCD's L2 (with a (tile, half) 2-D grid they sat 625 ids apart: both read the
  // tile from HBM, 90.9 MB fetched for 42 MB of rows).
  int64_t tile = blockIdx.x;
  int half = 0;
  if constexpr (SPLIT) {
    tile = (int64_t)(blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    half = (blockIdx.x >> 3) & 1;
    if (tile * BM >= g.N) return;  // padding of the last group of 8 tiles (uniform over the workgroup)
  }
  const int64_t r0 = tile * BM;
  const int nchunks = (int)(k_padded(g.D) / KC);
  constexpr int PER_T = BM * KC / 256;
  double rowsq[NG][RT][4];
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) rowsq[q][a][r] = 0.0;
  for (int64_t cb = 0; cb < n_pad / BN; ++cb) {
    const int64_t ctbase = cb * 16 + (int64_t)half * (4 * NCT) + wave * NCT;
    d4 acc[RT][NCT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
    double2 bring[4][NCT];  // B fragments three k-step pairs ahead
    if constexpr (DMA) {
      // DMA form (whole chunks of a matrix below 4 GiB, the usual case; proj_sq_dma_ok): per chunk and wave ONE buffer_load ... lds for
      // the rows (no staging registers, no ds_write), B fragments through a scalar offset, the LDS buffer a compile-time
      // constant (chunks are taken two at a time): the vector ALU issues the matrix instructions and nothing else.
      const __amdgpu_buffer_rsrc_t brsrc = buffer_of(g.packed_m, (unsigned)(packed_elems(g.D, g.r) * 8));
      const int64_t rows_here = (g.N - r0 < BM) ? g.N - r0 : BM;  // rows past N read as zero (range check)
      const __amdgpu_buffer_rsrc_t arsrc = buffer_of(g.h + r0 * g.D, (unsigned)(rows_here * g.D * 8));
      const unsigned lane_bytes = (unsigned)lane * 16u;
      const unsigned pair_stride_bytes = (unsigned)NT * 1024u;
      const unsigned ct_bytes = (unsigned)ctbase * 1024u;
      // DMA instruction j of a chunk fills slots 64j .. 64j+63: slot = kpair * BM + row
      constexpr int NDMA = BM / 4 / 4;  // per wave (BM / 4 instructions per chunk over 4 waves)
      unsigned a_lane_bytes[NDMA];
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int slot = 64 * (wave + 4 * i) + lane;
        a_lane_bytes[i] = (unsigned)(((slot % BM) * g.D + 2 * (slot / BM)) * 8);
      }
      auto dma_rows = [&](int bufc, int kc) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              arsrc, (__attribute__((address_space(3))) void*)((bufc ? lds_d1 : lds_d0) + 128 * (wave + 4 * i)), 16,
              a_lane_bytes[i], (unsigned)kc * 8u, 0, 0);
        asm volatile("" ::: "memory");  // the B loads of the chunk stay behind the DMA in program order (the wait below counts them)
      };
      auto body = [&](auto buf_tag, int ch) {
        constexpr int bufc = decltype(buf_tag)::value;
        // this wave's DMA of chunk ch is older than at least 3 * NCT B loads (four pairs in the loop, three in the prologue):
        // once at most that many loads are outstanding it has landed (loads retire in order); the barrier then publishes
        // all four waves' parts.  The oldest of those B loads is needed by the first matrix instruction anyway.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NCT) : "memory");
        __syncthreads();
        // (no branch: the last chunk fetches itself again into the idle buffer - with the DMA under a condition the loop
        // is several basic blocks and the compiler's own wait at their join is vmcnt(0))
        dma_rows(bufc ^ 1, ((ch + 1 < nchunks) ? ch + 1 : ch) * KC);
        mfma_chunk_dma<RT, NCT, BM>(acc, bufc ? lds_d1 : lds_d0, li, lg, brsrc, lane_bytes,
                                    ct_bytes + (unsigned)ch * 4u * pair_stride_bytes, pair_stride_bytes, bring);
      };
      // prologue in the loop's own order - DMA first, then the three ring pairs - so the count in body() holds for chunk 0
      // too (and the compiler's own wait for its DMA -> ds_read dependence merges to the same vmcnt(3 * NCT) at the loop
      // header instead of vmcnt(0))
      dma_rows(0, 0);
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          bring[j][c] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(
                                                        brsrc, lane_bytes + c * 1024u, ct_bytes + j * pair_stride_bytes, 0));
      int ch = 0;
      for (; ch + 1 < nchunks; ch += 2) {
        body(std::integral_constant<int, 0>{}, ch);
        body(std::integral_constant<int, 1>{}, ch + 1);
      }
      if (ch < nchunks) body(std::integral_constant<int, 0>{}, ch);
    } else {
      const double2* bp = reinterpret_cast<const double2*>(g.packed_m) + ctbase * 64 + lane;
    #pragma unroll
      for (int j = 0; j < 3; ++j)
  #pragma unroll
        for (int c = 0; c < NCT; ++c) bring[j][c] = bp[j * NT * 64 + c * 64];
      // staging: element q*256 + tid of the BM x KC chunk (a wave covers two 256-byte row segments per pass)
      double areg[PER_T];
      const int srow = tid / KC, skk = tid % KC;  // + 256/KC rows per q
      // interior row tile of a matrix whose width is a whole number of chunks (uniform over the workgroup): plain loads off
      // one running pointer - predicated loads and 64-bit index arithmetic are vector instructions the matrix pipe waits for
      const bool interior = (r0 + BM <= g.N) && (g.D % KC == 0);
      const double* pa = g.h + (r0 + srow) * g.D + skk;
      auto load_a = [&](int64_t kc) {
        if (interior) {
  #pragma unroll
          for (int q = 0; q < PER_T; ++q) areg[q] = pa[(int64_t)q * (256 / KC) * g.D + kc];
          return;
        }
        const int64_t gk = kc + skk;
  #pragma unroll
        for (int q = 0; q < PER_T; ++q) {
          const int64_t gr = r0 + srow + q * (256 / KC);
          areg[q] = (gr < g.N && gk < g.D) ? g.h[gr * g.D + gk] : 0.0;
        }
      };
      load_a(0);
      int buf = 0;
      for (int64_t ch = 0; ch < nchunks; ++ch) {
  #pragma unroll
        for (int q = 0; q < PER_T; ++q) lds_a[(buf * BM + srow + q * (256 / KC)) * APITCH + skk] = areg[q];
        __syncthreads();
        if (ch + 1 < nchunks) load_a((ch + 1) * KC);
        mfma_chunk_ring<RT, NCT>(acc, lds_a + buf * BM * APITCH, APITCH, li, lg, bp + ch * 4 * NT * 64, NT * 64, bring);
        buf ^= 1;
      }
    }
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int64_t col = (ctbase + c) * 16 + li;
        const double cc = (col < g.r) ? g.c[col] : 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = acc[a][c][r] + cc;  // zero-padded columns contribute 0
          rowsq[c / 2][a][r] = fma(v, v, rowsq[c / 2][a][r]);
        }
      }
    __syncthreads();
  }
  // A row's sum of squares is added up in ONE order whatever the launch shape (so a batch scores the same bits
  // sharded, chunked or whole): 32-column groups g = 0..7 of every 256-column block, each accumulated over the
  // blocks and its 16 lanes, then ((g0+g1)+g2)+g3 and ((g4+g5)+g6)+g7, then the sum of the two halves.
#pragma unroll
  for (int q = 0; q < NG; ++q)
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = rowsq[q][a][r];
        v += shfl_xor_f64(v, 1);
        v += shfl_xor_f64(v, 2);
        v += shfl_xor_f64(v, 4);
        v += shfl_xor_f64(v, 8);
        if (li == 0) part[(wave * NG + q) * BM + 16 * a + lg + 4 * r] = v;
      }
  __syncthreads();
  if (tid < BM) {
    const int64_t row = r0 + tid;
    if (row < g.N) {
      const double lo = ((part[tid] + part[BM + tid]) + part[2 * BM + tid]) + part[3 * BM + tid];
      if constexpr (NCT == 4) {
        const double hi = ((part[4 * BM + tid] + part[5 * BM + tid]) + part[6 * BM + tid]) + part[7 * BM + tid];
        g.score[row] = -(lo + hi);
      } else if constexpr (!SPLIT) {  // this workgroup holds the whole row (r <= 64 * NCT)
        if constexpr (ACCUMULATE) unsafeAtomicAdd(&g.score[row], -lo);
        else g.score[row] = -lo;
      } else if constexpr (ACCUMULATE) {
        // score was zeroed earlier in the stream: two addends per row, and 0 + a + b = 0 + b + a bit for bit
        unsafeAtomicAdd(&g.score[row], -lo);
      } else {
        g.partial[(int64_t)half * g.N + row] = lo;
      }
    }
  }
#endif
}

__global__ void proj_sq_combine_kernel(const double* __restrict__ partial, double* __restrict__ score, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) score[i] = -(partial[i] + partial[N + i]);
}

double digamma_diff(int n, int k) {
  double s = 0.0;
  for (int j = n - 1; j >= k; --j) s += 1.0 / (double)j;
  return s;
}

template <int RT>
int launch_pca_md(const PcaMdArgs& g, hipStream_t s) {
  constexpr int BM = 16 * RT;
  const int64_t n_pad = n_padded(g.n);
  const size_t shmem = ((size_t)2 * BM * APITCH + (size_t)BM * (n_pad + 2) + 4 * BM) * sizeof(double);
  if (shmem > 160 * 1024) return RUNIA_E_INVALID;
  static std::atomic<uint64_t> lds_ok{0};
  if (runia_allow_dynamic_lds(reinterpret_cast<const void*>(pca_md_kernel<RT>), 160 * 1024, lds_ok) != RUNIA_OK)
    return RUNIA_E_LAUNCH;
  const int64_t tiles = (g.N + BM - 1) / BM;
  pca_md_kernel<RT><<<(unsigned)tiles, 256, shmem, s>>>(g);
  return runia_check_launch();
}

}  // namespace

extern "C" int runia_pca_md_score_f64(const double* h, const double* packed_ct, const double* bias,
                                      const double* scale, const double* md_mean, const double* packed_p,
                                      double* score, double* y_out, int64_t N, int64_t D, int64_t n,
                                      runia_stream_t stream) {
  if (N < 0 || D <= 0 || n <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!h || !md_mean || !packed_p || !score) return RUNIA_E_INVALID;
  if (packed_ct ? !bias : (n != D)) return RUNIA_E_INVALID;
  PcaMdArgs g{h, packed_ct, bias, scale, md_mean, packed_p, score, y_out, N, D, n};
  // 32-row tiles halve the L2 traffic of the packed weights but need >= ~4 tiles per CU to balance
  const int64_t tiles32 = (N + 31) / 32;
  const int64_t n_pad = n_padded(n);
  const bool fits32 = ((size_t)2 * 32 * APITCH + (size_t)32 * (n_pad + 2) + 128) * 8 <= 160 * 1024;
  if (tiles32 >= 1024 && fits32) return launch_pca_md<2>(g, as_stream(stream));
  return launch_pca_md<1>(g, as_stream(stream));
}

namespace {
template <int HH, int WW, int NPP>
void launch_mask(const float* rnd, int64_t rand_image_stride, float* table, int64_t N, int n_mc, float gamma,
                 int block_size, int identity, int sort_layers, uint64_t seed, int64_t first_image, int redraw,
                 hipStream_t s) {
  static_assert(HH * WW <= 64, "mc_mask_bits_kernel holds a drop layer in one 64-bit word");
  const unsigned grid = (unsigned)((N + kMaskBitsWaves - 1) / kMaskBitsWaves);
  if (redraw)  // throughput mode, opt-in: its own instantiation, so that the common launch carries neither the check nor the code
    mc_mask_bits_kernel<HH, WW, NPP, true><<<grid, 64 * kMaskBitsWaves, 0, s>>>(rnd, rand_image_stride, table, N, n_mc, gamma,
                                                                              block_size, identity, sort_layers, seed, first_image);
  else
    mc_mask_bits_kernel<HH, WW, NPP, false><<<grid, 64 * kMaskBitsWaves, 0, s>>>(rnd, rand_image_stride, table, N, n_mc, gamma,
                                                                               block_size, identity, sort_layers, seed, first_image);
}

// explicit draws of the counter generator, [N, n_mc, H, W] (tests; callers that want the values themselves)
__global__ void mc_draws_kernel(float* __restrict__ out, int64_t N, int per_image, uint64_t seed, int64_t first_image) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * per_image) return;
  const int64_t img = i / per_image;
  out[i] = runia_philox::draw(seed, (uint64_t)(first_image + img), (int)(i - img * per_image));
}
}  // namespace

extern "C" int runia_mc_entropy_supported(int H, int W, int n_mc, int k);

extern "C" size_t runia_mc_entropy_workspace_bytes(int64_t N, int H, int W, int n_mc) {
  if (N <= 0 || H <= 0 || W <= 0 || n_mc <= 0) return 0;
  return (size_t)N * (size_t)n_mc * (size_t)(H * W + 2) * sizeof(float);
}

static int mc_args_ok(int64_t N, int H, int W, int n_mc, const void* workspace, size_t workspace_bytes) {
  if (N < 0 || N > 65535 || H <= 0 || W <= 0 || n_mc < 2 || n_mc > kMaxMC) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!workspace || workspace_bytes < runia_mc_entropy_workspace_bytes(N, H, W, n_mc) ||
      (((uintptr_t)workspace) & 15) != 0)
    return RUNIA_E_WORKSPACE;
  return RUNIA_OK;
}

#define RUNIA_MCE_SHAPES(F) \
  F(4, 4, 16, 5) F(4, 4, 32, 5) F(4, 4, 8, 5) F(2, 2, 16, 5) F(2, 2, 32, 5) F(7, 7, 16, 5) F(7, 7, 32, 5) F(8, 8, 16, 5) F(8, 8, 32, 5)

static int mc_mask_table(const float* rnd, int64_t rand_image_stride, void* workspace, size_t workspace_bytes,
                         int64_t N, int H, int W, int n_mc, double drop_prob, int block_size, int sort_layers,
                         runia_stream_t stream, bool counter = false, uint64_t seed = 0, int64_t first_image = 0,
                         int redraw = 0) {
  if (block_size < 1) return RUNIA_E_INVALID;
  if (int rc = mc_args_ok(N, H, W, n_mc, workspace, workspace_bytes)) return rc;
  if (N == 0) return RUNIA_OK;
  const int identity = (drop_prob == 0.0);
  if (!identity && !rnd && !counter) return RUNIA_E_INVALID;
  if (counter) rnd = nullptr;
  float* table = reinterpret_cast<float*>(workspace);
  const float gamma = (float)(drop_prob / (double)(block_size * block_size));
  hipStream_t s = as_stream(stream);
#define RUNIA_MCE(HH, WW, NPP, KK)                                                                          \
  if (H == HH && W == WW && n_mc <= NPP && n_mc > NPP / 2) {                                                \
    launch_mask<HH, WW, NPP>(rnd, rand_image_stride, table, N, n_mc, gamma, block_size, identity,           \
                             sort_layers, seed, first_image, redraw, s);                                    \
    return runia_check_launch();                                                                            \
  }
  RUNIA_MCE_SHAPES(RUNIA_MCE)
#undef RUNIA_MCE
  return RUNIA_E_INVALID;
}

extern "C" int runia_mc_mask_table_f32(const float* rnd, int64_t rand_image_stride, void* workspace,
                                       size_t workspace_bytes, int64_t N, int H, int W, int n_mc, double drop_prob,
                                       int block_size, runia_stream_t stream) {
  return mc_mask_table(rnd, rand_image_stride, workspace, workspace_bytes, N, H, W, n_mc, drop_prob, block_size, 1,
                       stream);
}

// MCSamplerModule.forward alone (the samples, in the order of the draws) on the table path: the keep-flag table is
// left unsorted and K1 stops after the sampler.  Same bits as runia_mc_stack_f32.
extern "C" int runia_mc_stack_table_f32(const float* x, const float* rnd, int64_t rand_image_stride, float* z,
                                        void* workspace, size_t workspace_bytes, int64_t N, int C, int H, int W,
                                        int n_mc, double drop_prob, int block_size, runia_stream_t stream) {
  if (C <= 0) return RUNIA_E_INVALID;
  if (int rc = mc_args_ok(N, H, W, n_mc, workspace, workspace_bytes)) return rc;
  if (N == 0) return RUNIA_OK;
  if (!x || !z || !runia_mc_entropy_supported(H, W, n_mc, 5)) return RUNIA_E_INVALID;
  if (!(((((uintptr_t)x) & 15) == 0) || (H * W) % 4 != 0)) return RUNIA_E_INVALID;
  if (int rc = mc_mask_table(rnd, rand_image_stride, workspace, workspace_bytes, N, H, W, n_mc, drop_prob,
                             block_size, 0, stream))
    return rc;
  const float* table = reinterpret_cast<const float*>(workspace);
  const unsigned grid = (unsigned)((((N + 7) / 8 + K1_IMGS - 1) / K1_IMGS) * 8 * ((C + kK1Block - 1) / kK1Block));
  hipStream_t s = as_stream(stream);
#define RUNIA_MCE(HH, WW, NPP, KK)                                                                          \
  if (H == HH && W == WW && n_mc <= NPP && n_mc > NPP / 2) {                                                \
    if (n_mc == NPP)                                                                                        \
      mc_entropy_kernel<HH, WW, NPP, KK, true, false><<<grid, kK1Block, 0, s>>>(x, table, nullptr, z,       \
                                                                                nullptr, N, C, n_mc, 0.0,   \
                                                                                0.0, 0.0);                  \
    else                                                                                                    \
      mc_entropy_kernel<HH, WW, NPP, KK, false, false><<<grid, kK1Block, 0, s>>>(x, table, nullptr, z,      \
                                                                                 nullptr, N, C, n_mc, 0.0,  \
                                                                                 0.0, 0.0);                 \
    return runia_check_launch();                                                                            \
  }
  RUNIA_MCE_SHAPES(RUNIA_MCE)
#undef RUNIA_MCE
  return RUNIA_E_INVALID;
}

extern "C" int runia_mc_entropy_from_table_f32(const float* x, const void* workspace, size_t workspace_bytes,
                                               double* h, float* z_out, double* zero_fill, int64_t N, int C,
                                               int H, int W, int n_mc, int k, double min_dist,
                                               runia_stream_t stream) {
  if (C <= 0 || k < 1 || k >= n_mc) return RUNIA_E_INVALID;
  if (int rc = mc_args_ok(N, H, W, n_mc, workspace, workspace_bytes)) return rc;
  if (N == 0) return RUNIA_OK;
  if (!x || !h) return RUNIA_E_INVALID;
  if (!runia_mc_entropy_supported(H, W, n_mc, k)) return RUNIA_E_INVALID;  // callers: mc_stack + kl_entropy_per_dim
  const float* table = reinterpret_cast<const float*>(workspace);
  const double ct = digamma_diff(n_mc, k), inv_n = 1.0 / (double)n_mc;
  const unsigned grid = (unsigned)((((N + 7) / 8 + K1_IMGS - 1) / K1_IMGS) * 8 * ((C + kK1Block - 1) / kK1Block));
  hipStream_t s = as_stream(stream);
  const bool x16 = ((((uintptr_t)x) & 15) == 0);
#define RUNIA_MCE(HH, WW, NPP, KK)                                                                          \
  if (H == HH && W == WW && n_mc <= NPP && n_mc > NPP / 2 && k == KK && (x16 || (HH * WW) % 4 != 0)) {      \
    if (n_mc == NPP)                                                                                        \
      RUNIA_LAUNCH_TIMED((mc_entropy_kernel<HH, WW, NPP, KK, true>), grid, kK1Block, 0, s, x, table, h, z_out, \
                         zero_fill, N, C, n_mc, min_dist, ct, inv_n);                                       \
    else                                                                                                    \
      RUNIA_LAUNCH_TIMED((mc_entropy_kernel<HH, WW, NPP, KK, false>), grid, kK1Block, 0, s, x, table, h, z_out, \
                         zero_fill, N, C, n_mc, min_dist, ct, inv_n);                                       \
    return runia_check_launch();                                                                            \
  }
  RUNIA_MCE_SHAPES(RUNIA_MCE)
#undef RUNIA_MCE
  return RUNIA_E_INVALID;  // e.g. misaligned x: callers use runia_mc_stack_f32 + runia_kl_entropy_per_dim_f32
}

extern "C" int runia_mc_entropy_f32(const float* x, const float* rnd, int64_t rand_image_stride, double* h,
                                    float* z_out, double* zero_fill, void* workspace, size_t workspace_bytes,
                                    int64_t N, int C, int H, int W, int n_mc, double drop_prob, int block_size,
                                    int k, double min_dist, runia_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || n_mc < 2 || n_mc > kMaxMC || block_size < 1 || k < 1 || k >= n_mc)
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !h || N > 65535) return RUNIA_E_INVALID;
  if (!runia_mc_entropy_supported(H, W, n_mc, k)) return RUNIA_E_INVALID;
  if (drop_prob != 0.0 && !rnd) return RUNIA_E_INVALID;
  if (!(((((uintptr_t)x) & 15) == 0) || (H * W) % 4 != 0)) return RUNIA_E_INVALID;  // before anything is launched
  if (int rc = runia_mc_mask_table_f32(rnd, rand_image_stride, workspace, workspace_bytes, N, H, W, n_mc, drop_prob,
                                       block_size, stream))
    return rc;
  return runia_mc_entropy_from_table_f32(x, workspace, workspace_bytes, h, z_out, zero_fill, N, C, H, W, n_mc, k,
                                         min_dist, stream);
}

// ---- ROI source (config 4): roi_align folded into K1's load ------------------------------------------------------------
int runia_roi_sample_table(const float* feat_nhwc, const float* boxes, const int* batch_idx, void* table, size_t table_bytes,
                           int64_t K, int64_t B, int C, int H, int W, int PH, int PW, double spatial_scale, int G, int aligned,
                           hipStream_t s);  // roi.hip
size_t runia_roi_sample_table_bytes(int64_t K, int PH, int PW, int G);

// The fused ROI launch exists for exactly these (PH, PW, register samples NP, k, samples per bin side G): one list feeds both
// the shape query and the dispatch, so `supported` can never promise a shape the entry point has no kernel for.  n_mc in
// (NP/2, NP] takes the NP instantiation.  Everything else (adaptive sampling - ratio <= 0 - has a per-ROI sample count; 4x4 /
// 8x8 bins with one sample; n_mc <= 8) goes through runia_roi_align_f32 + runia_mc_entropy_f32, same bits.
#define RUNIA_ROI_MCE_SHAPES(X)                                                                                    \
  X(7, 7, 16, 5, 2) X(7, 7, 32, 5, 2) X(7, 7, 16, 5, 1) X(7, 7, 32, 5, 1)                                          \
  X(4, 4, 16, 5, 2) X(4, 4, 32, 5, 2) X(8, 8, 16, 5, 2) X(8, 8, 32, 5, 2)

extern "C" int runia_roi_mc_entropy_supported(int PH, int PW, int n_mc, int k, int sampling_ratio) {
  if (!runia_mc_entropy_supported(PH, PW, n_mc, k)) return 0;
#define RUNIA_ROI_HAS(HH, WW, NPP, KK, GG) \
  if (PH == HH && PW == WW && n_mc <= NPP && n_mc > NPP / 2 && k == KK && sampling_ratio == GG) return 1;
  RUNIA_ROI_MCE_SHAPES(RUNIA_ROI_HAS)
#undef RUNIA_ROI_HAS
  return 0;
}
extern "C" size_t runia_roi_mc_entropy_workspace_bytes(int64_t K, int PH, int PW, int n_mc, int sampling_ratio) {
  if (K <= 0 || sampling_ratio < 1) return 0;
  const size_t masks = (runia_mc_entropy_workspace_bytes(K, PH, PW, n_mc) + 255) / 256 * 256;
  return masks + runia_roi_sample_table_bytes(K, PH, PW, sampling_ratio);
}
extern "C" int runia_roi_mc_entropy_f32(const float* feat_nhwc, const float* boxes, const int* batch_idx, const float* rnd,
                                        int64_t rand_image_stride, double* h, float* z_out, void* workspace,
                                        size_t workspace_bytes, int64_t K, int64_t B, int C, int H, int W, int PH, int PW,
                                        double spatial_scale, int sampling_ratio, int aligned, int n_mc, double drop_prob,
                                        int block_size, int k, double min_dist, runia_stream_t stream) {
  if (K < 0 || B <= 0 || C <= 0 || H <= 0 || W <= 0 || n_mc < 2 || n_mc > kMaxMC || block_size < 1 || k < 1 || k >= n_mc)
    return RUNIA_E_INVALID;
  if (K == 0) return RUNIA_OK;
  if (!feat_nhwc || !boxes || !h || K > 65535 || (B > 1 && !batch_idx)) return RUNIA_E_INVALID;
  if (!runia_roi_mc_entropy_supported(PH, PW, n_mc, k, sampling_ratio)) return RUNIA_E_INVALID;
  if ((int64_t)H * W * C * 4 >= ((int64_t)1 << 30)) return RUNIA_E_INVALID;  // one image's map behind a 32-bit buffer (RUNIA_ROI_FUSED_MAX_IMAGE_BYTES)
  if (drop_prob != 0.0 && !rnd) return RUNIA_E_INVALID;
  if (!workspace || (((uintptr_t)workspace) & 15) != 0 ||
      workspace_bytes < runia_roi_mc_entropy_workspace_bytes(K, PH, PW, n_mc, sampling_ratio))
    return RUNIA_E_WORKSPACE;
  const size_t masks = (runia_mc_entropy_workspace_bytes(K, PH, PW, n_mc) + 255) / 256 * 256;
  if (int rc = runia_mc_mask_table_f32(rnd, rand_image_stride, workspace, masks, K, PH, PW, n_mc, drop_prob, block_size, stream))
    return rc;
  char* tab = reinterpret_cast<char*>(workspace) + masks;
  hipStream_t s = as_stream(stream);
  if (int rc = runia_roi_sample_table(feat_nhwc, boxes, batch_idx, tab, workspace_bytes - masks, K, B, C, H, W, PH, PW,
                                      spatial_scale, sampling_ratio, aligned, s))
    return rc;
  const float* table = reinterpret_cast<const float*>(workspace);
  const float* src = reinterpret_cast<const float*>(tab);  // the RoiSource at the head of the sample table
  const double ct = digamma_diff(n_mc, k), inv_n = 1.0 / (double)n_mc;
  const unsigned grid = (unsigned)((((K + 7) / 8 + K1_IMGS - 1) / K1_IMGS) * 8 * ((C + kK1Block - 1) / kK1Block));
#define RUNIA_ROI_MCE(HH, WW, NPP, KK, GG)                                                                         \
  if (PH == HH && PW == WW && n_mc <= NPP && n_mc > NPP / 2 && k == KK && sampling_ratio == GG) {                  \
    if (n_mc == NPP)                                                                                               \
      mc_entropy_kernel<HH, WW, NPP, KK, true, true, GG><<<grid, kK1Block, 0, s>>>(src, table, h, z_out, nullptr,  \
                                                                                   K, C, n_mc, min_dist, ct, inv_n); \
    else                                                                                                           \
      mc_entropy_kernel<HH, WW, NPP, KK, false, true, GG><<<grid, kK1Block, 0, s>>>(src, table, h, z_out, nullptr, \
                                                                                    K, C, n_mc, min_dist, ct, inv_n); \
    return runia_check_launch();                                                                                   \
  }
  RUNIA_ROI_MCE_SHAPES(RUNIA_ROI_MCE)
#undef RUNIA_ROI_MCE
  return RUNIA_E_INVALID;
}

// ---- throughput mode: the DropBlock draws come from the counter generator inside K0 (philox.hpp) ----------------
extern "C" int runia_mc_draws_f32(float* out, int64_t N, int n_mc, int H, int W, uint64_t seed, int64_t first_image,
                                  runia_stream_t stream) {
  if (N < 0 || n_mc < 1 || H <= 0 || W <= 0 || first_image < 0 || (int64_t)n_mc * H * W > (1 << 20))
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!out) return RUNIA_E_INVALID;
  const int64_t total = N * n_mc * H * W;
  if ((total + 255) / 256 > 0x7fffffffLL) return RUNIA_E_INVALID;
  mc_draws_kernel<<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(out, N, n_mc * H * W, seed,
                                                                                first_image);
  return runia_check_launch();
}

extern "C" int runia_mc_mask_table_counter_f32(uint64_t seed, int64_t first_image, void* workspace,
                                               size_t workspace_bytes, int64_t N, int H, int W, int n_mc,
                                               double drop_prob, int block_size, int redraw_dead_layers,
                                               runia_stream_t stream) {
  if (first_image < 0) return RUNIA_E_INVALID;
  return mc_mask_table(nullptr, 0, workspace, workspace_bytes, N, H, W, n_mc, drop_prob, block_size, 1, stream, true,
                       seed, first_image, redraw_dead_layers ? 1 : 0);
}

extern "C" int runia_mc_entropy_counter_f32(const float* x, uint64_t seed, int64_t first_image, double* h,
                                            float* z_out, double* zero_fill, void* workspace, size_t workspace_bytes,
                                            int64_t N, int C, int H, int W, int n_mc, double drop_prob,
                                            int block_size, int k, double min_dist, int redraw_dead_layers,
                                            runia_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || n_mc < 2 || n_mc > kMaxMC || block_size < 1 || k < 1 || k >= n_mc ||
      first_image < 0)
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !h || N > 65535) return RUNIA_E_INVALID;
  if (!runia_mc_entropy_supported(H, W, n_mc, k)) return RUNIA_E_INVALID;
  if (!(((((uintptr_t)x) & 15) == 0) || (H * W) % 4 != 0)) return RUNIA_E_INVALID;  // before anything is launched
  if (int rc = runia_mc_mask_table_counter_f32(seed, first_image, workspace, workspace_bytes, N, H, W, n_mc,
                                               drop_prob, block_size, redraw_dead_layers, stream))
    return rc;
  return runia_mc_entropy_from_table_f32(x, workspace, workspace_bytes, h, z_out, zero_fill, N, C, H, W, n_mc, k,
                                         min_dist, stream);
}

extern "C" int runia_mc_entropy_supported(int H, int W, int n_mc, int k) {
  if (k != 5 || k >= n_mc) return 0;
  const bool hw = (H == 4 && W == 4);
  if (hw && n_mc > 4 && n_mc <= 32) return 1;
  // 9 ... 32 samples (32 = the reference's default mcd_samples_nro, evaluation/entropy.py:41) on the maps RoI-align
  // produces (7x7), 8x8 and 2x2
  if (n_mc > 8 && n_mc <= 32 && ((H == 2 && W == 2) || (H == 7 && W == 7) || (H == 8 && W == 8))) return 1;
  return 0;
}

// The DMA form addresses the rows and the packed matrix through 32-bit buffer offsets and stages whole 32-deep chunks.
// Measured against the register-staged form (tools/ablate/run_proj_acc.py, us per launch, D = 512, r = 256):
//   N = 2 000 / 5 000 / 8 000 / 10 000: 20.9 / 26.1 / 33.9 / 41.4 vs 23.0 / 27.9 / 35.8 / 43.5 (every workgroup resident
//   at once: 5 per CU); 12 000 / 15 000 / 20 000: 56.8 / 69.7 / 86.2 vs 52.0 / 67.5 / 83.7 (a second, partial round of
//   16-row workgroups: slower); 32 x 256 tiles at 40 000 / 100 000: 158.8 / 410.3 vs 167.3 / 426.3.
static bool proj_sq_dma_ok(int64_t D, int64_t r) {
  return K2_DMA && D % KC == 0 && D < (1 << 23) && packed_elems(D, r) < ((int64_t)1 << 29);
}
static bool proj_sq_one_round(unsigned grid) { return (int64_t)grid <= (int64_t)K2_WAVES_DMA * runia_cu_count(); }
template <int RT, int NCT, bool ACCUMULATE, bool SPLIT>
static void launch_proj_sq(unsigned grid, hipStream_t s, const ProjSqArgs& g, bool dma) {
  if (dma && (RT > 1 || proj_sq_one_round(grid))) proj_sq_kernel<RT, NCT, ACCUMULATE, SPLIT, true><<<grid, 256, 0, s>>>(g);
  else proj_sq_kernel<RT, NCT, ACCUMULATE, SPLIT, false><<<grid, 256, 0, s>>>(g);
}

// 32 x 256 tiles (fewer re-reads of M, 65 TFLOP/s when they fill the chip evenly) or 16 x 128 half tiles (62 TFLOP/s,
// whatever the batch)?  The large tile only pays when its last round of workgroups is nearly full: N = 20 000 is 625
// large tiles = 2.44 rounds on 256 CUs and ran at 51 TFLOP/s; 16 384 / 32 768 / 65 536 rows (whole rounds) at 63-65.
static bool proj_sq_large_tiles(int64_t N, int64_t cus) {
  const int64_t tiles32 = (N + 31) / 32;
  const int64_t rounds = (tiles32 + cus - 1) / cus;
  return tiles32 >= 2 * cus && 100 * tiles32 >= 93 * rounds * cus;
}

extern "C" int runia_proj_sq_accumulate_f64(const double* h, const double* packed_m, const double* c, double* score,
                                            int64_t N, int64_t D, int64_t r, runia_stream_t stream) {
  if (N < 0 || D <= 0 || r <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!h || !packed_m || !c || !score) return RUNIA_E_INVALID;
  ProjSqArgs g{h, packed_m, c, score, nullptr, N, D, r};
  hipStream_t s = as_stream(stream);
  const int64_t tiles16 = (N + 15) / 16, cus = runia_cu_count();
  const bool dma = proj_sq_dma_ok(D, r);
  if (r <= 64) launch_proj_sq<1, 1, true, false>((unsigned)tiles16, s, g, dma);
  else if (r <= 128) launch_proj_sq<1, 2, true, false>((unsigned)tiles16, s, g, dma);
  else if (proj_sq_large_tiles(N, cus)) launch_proj_sq<2, 4, false, false>((unsigned)((N + 31) / 32), s, g, dma);
  else if (tiles16 > cus / 2) launch_proj_sq<1, 2, true, true>((unsigned)((tiles16 + 7) / 8 * 16), s, g, dma);
  else launch_proj_sq<1, 4, false, false>((unsigned)tiles16, s, g, dma);
  return runia_check_launch();
}

extern "C" size_t runia_proj_sq_workspace_bytes(int64_t N) { return N > 0 ? (size_t)N * 2 * sizeof(double) : 0; }

extern "C" int runia_proj_sq_score_f64(const double* h, const double* packed_m, const double* c, double* score,
                                       void* workspace, size_t workspace_bytes, int64_t N, int64_t D, int64_t r,
                                       runia_stream_t stream) {
  if (N < 0 || D <= 0 || r <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!h || !packed_m || !c || !score) return RUNIA_E_INVALID;
  ProjSqArgs g{h, packed_m, c, score, reinterpret_cast<double*>(workspace), N, D, r};
  hipStream_t s = as_stream(stream);
  const int64_t tiles16 = (N + 15) / 16, cus = runia_cu_count();
  const bool dma = proj_sq_dma_ok(D, r);
  if (r <= 64) {
    launch_proj_sq<1, 1, false, false>((unsigned)tiles16, s, g, dma);
  } else if (r <= 128) {
    launch_proj_sq<1, 2, false, false>((unsigned)tiles16, s, g, dma);
  } else if (proj_sq_large_tiles(N, cus)) {
    launch_proj_sq<2, 4, false, false>((unsigned)((N + 31) / 32), s, g, dma);
  } else if (tiles16 > cus / 2 && workspace && workspace_bytes >= runia_proj_sq_workspace_bytes(N)) {
    // (32-, 48- and 64-row tiles with the same column split measured 64, 64 and 78 us against 59 us)
    launch_proj_sq<1, 2, false, true>((unsigned)((tiles16 + 7) / 8 * 16), s, g, dma);
    proj_sq_combine_kernel<<<(unsigned)((N + 255) / 256), 256, 0, s>>>(g.partial, score, N);
  } else {
    launch_proj_sq<1, 4, false, false>((unsigned)tiles16, s, g, dma);
  }
  return runia_check_launch();
}
