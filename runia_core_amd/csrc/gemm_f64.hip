// f64 dense contractions of the scoring path on the CDNA4 matrix cores
// (v_mfma_f64_16x16x4_f64):
//   a4  PCA transform     Y = X @ C.T - mean @ C.T ; Y /= scale   (reference dimensionality_reduction.py:86)
//   a5  LaREM / MD score  -diag(diff @ P @ diff.T)                 (reference inference/postprocessors.py:241-242)
//   a6  Mahalanobis       max_c -(x-mu_c) P (x-mu_c)^T             (reference inference/funcs.py:88-100)
//
// Layout.  The right-hand matrix B [K, n] is constant after setup(), so it is repacked
// once into MFMA fragment order: for k-step pair s2 (8 k values) and 16-column tile ct,
//   packed[((s2*NT + ct)*64 + lane)*2 + h] = B[8*s2 + 4*h + (lane>>4)][16*ct + (lane&15)]
// (zero padded to K%32==0 plus four zero k-step pairs, n%256==0; see mfma_f64_tile.hpp).  One wave-wide 16-byte load then yields the B operand
// of two MFMA k-steps, fully coalesced (1 KiB per instruction, 4 KiB per wave per k-step pair)
// and served by the XCD L2 (the whole matrix is <= a few MiB).
// The left-hand rows stream from HBM exactly once: a 32-row x 32-k chunk is staged into LDS
// (pitch 34 doubles => conflict-free ds_read_b64 of the A fragment) while the previous chunk
// is being multiplied.  A workgroup (4 waves) owns 32 rows x 256 columns; wave w owns the
// 64-column slice w, i.e. 2 x 4 accumulator tiles (64 VGPRs).
//
// f64 MFMA fragment maps (guide section 3): A lane l -> A[l&15][l>>4]; B lane l -> B[l>>4][l&15];
// C/D lane l, reg r -> C[(l>>4) + 4r][l&15].
#include "common.hpp"
#include "mfma_f64_tile.hpp"

namespace {

using namespace runia_mfma;

constexpr int BM = 32;  // rows per workgroup (2 row tiles of 16)

enum Epilogue { EPI_PCA = 0, EPI_ROWDOT = 1, EPI_STORE = 2, EPI_ROWNORM = 3, EPI_KDE = 4, EPI_MAHA = 5 };
constexpr int kMahaMaxClasses = 16;  // classes the fused Mahalanobis epilogue keeps in LDS

struct GemmArgs {
  const void* x;        // [N, K] rows (TA), ld = ldx
  int64_t ldx;
  const double* packed; // packed B
  int64_t N, K, n;
  // prologue
  const void* sub;      // optional [K] (TS): a = x - sub[k]; subtraction in f32 when TA = TS = float (NumPy dtype rules)
  // EPI_PCA
  const double* bias;   // [n]
  const double* scale;  // [n] or null
  // EPI_KDE: out[row] = logsumexp_col alpha * (rown[row] + coln[col] - 2 (x B)[row][col]) + addc
  const double* rown;   // [N] squared norms of the rows of x
  const double* coln;   // [n] squared norms of the columns of B
  double alpha, addc;
  // EPI_MAHA: out[row] = max_c -(t P t^T), t = fl(x - mu_c) in the input dtype (see maha_class_kernel for the algebra)
  const void* class_mean;  // [C, K] (TA)
  const double* mu_p;      // [C, K] f64 = class_mean @ P
  int n_classes;
  // EPI_KDE, few row tiles: one workgroup per (row tile, 256-column block) stores its alpha * (...) values to kde_vals
  // [tile][block][thread][RT * 4][NCT]; kde_replay_kernel then runs the online logsumexp over the blocks in order
  double* kde_vals;
  // EPI_MAHA, few row tiles: one workgroup per (row tile, 256-column block); the (block, wave, row, class) partial sums go
  // to maha_part [n_blocks][4][N][C] and maha_split_finish_kernel adds them up in the unsplit kernel's order
  double* maha_part;
  // EPI_ROWDOT (MD), few row tiles: one workgroup per (row tile, 256-column block) stores its products (d P)_j d_j to md_vals
  // [tile][block][thread][RT * 4][NCT]; md_replay_kernel adds them up per lane in the unsplit kernel's order
  double* md_vals;
  // outputs
  double* out;          // EPI_PCA / EPI_STORE: [N, n] (ld = n); EPI_ROWDOT / EPI_ROWNORM / EPI_KDE: [N]
  // EPI_ROWNORM with a TRIANGULAR right-hand matrix (round 6, runia_md_score_tril_*): B = W^T with W lower triangular, i.e.
  // B[k][col] = 0 for k > col - the 256-column block cb only multiplies its first (cb + 1) * 256 k values (the chunk loop ends at
  // the diagonal block: about half the products of a wide matrix), and neg_sq writes -sum of squares instead of the norm
  int tri, neg_sq;      // neg_sq: 0 = sqrt(sum), 1 = -sum, 2 = the sum itself (a first launch whose columns a second one completes)
  // EPI_ROWNORM over a column RANGE of the packed matrix (round 6, ViM: 1 048 columns = four full 256-column blocks + 24): the
  // blocks [cb0, cb1) (cb1 = 0: all of them) with the launch's first column tile at ct0; acc_in (optional) = sums of squares of
  // the columns another launch took, added before the final sqrt / sign
  int64_t cb0, cb1, ct0;
  const double* acc_in;
};

__global__ __launch_bounds__(256) void pack_weights_kernel(const double* __restrict__ B, int64_t ldb,
                                                            int64_t K, int64_t n, double* __restrict__ packed,
                                                            int64_t NT, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i & 1);
    const int lane = (int)((i >> 1) & 63);
    const int64_t t = i >> 7;  // s2*NT + ct
    const int64_t ct = t % NT, s2 = t / NT;
    const int64_t k = 8 * s2 + 4 * h + (lane >> 4);
    const int64_t c = 16 * ct + (lane & 15);
    packed[i] = (k < K && c < n) ? B[k * ldb + c] : 0.0;
  }
}

// x - sub with NumPy's promotion: f32 - f32 stays f32 (then promoted), anything else is f64
template <typename TA, typename TS>
__device__ __forceinline__ double sub_promote(TA x, TS m) {
  if constexpr (sizeof(TA) == 4 && sizeof(TS) == 4) return (double)(x - m);
  else return (double)x - (double)m;
}

// This thread's part of a staged (16 RT) x 32 chunk of rows between its loads and its LDS store.  The loads are issued in
// front of a chunk's matrix instructions and their values first touched behind them (row_chunk_values): widened (f32 rows)
// or centred at the load, every wave waited out its loads' latency before it multiplied (Mahalanobis on f32 rows: a
// s_waitcnt + four v_cvt_f64_f32 in front of each chunk's 64 matrix instructions).
//   mode 1  rows as loaded (interior chunk of an aligned matrix, the common case; uniform over the workgroup)
//   mode 2  rows and the slice of `sub` as loaded (the same with a mean to subtract)
//   mode 0  values computed at the load (edges, unaligned matrices): predicated scalar loads
// (SUB: the epilogues that centre their rows - MD, ViM; the others carry no slot for `sub`.  f64 rows keep the values of
// mode 0 in x itself.)
template <typename TA, typename TS, int RT, bool SUB>
struct RowChunk {
  static constexpr int PER = 2 * RT;  // elements per thread: 4 (32-row tiles) or 2 (16 rows); 8 for the 64-row experiments
  TA x[PER];
  TS m[SUB ? PER : 1];
  double v[sizeof(TA) == 8 ? 1 : PER];
  int mode;
};

template <typename T, int PER>
__device__ __forceinline__ void load_run(const T* __restrict__ p, T (&r)[PER]) {  // PER consecutive elements, 16-byte loads
  if constexpr (sizeof(T) == 4 && PER == 2) {
    const float2 a = *reinterpret_cast<const float2*>(p);
    r[0] = a.x; r[1] = a.y;
  } else if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int q = 0; q < PER; q += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + q);
      r[q] = a.x; r[q + 1] = a.y; r[q + 2] = a.z; r[q + 3] = a.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < PER; q += 2) {
      const double2 a = *reinterpret_cast<const double2*>(p + q);
      r[q] = a.x; r[q + 1] = a.y;
    }
  }
}

template <typename TA, typename TS, int RT, bool SUB>
__device__ __forceinline__ void load_a_regs(const GemmArgs& g, int64_t r0, int64_t kc, int tid, RowChunk<TA, TS, RT, SUB>& c) {
  constexpr int PER = 2 * RT;
  constexpr int TPR = KC / PER;            // threads per row: 8 or 16
  const int row = tid / TPR;               // 0 .. 16*RT-1
  const int kk = (tid % TPR) * PER;
  const int64_t gr = r0 + row;
  const int64_t gk = kc + kk;
  const TA* x = reinterpret_cast<const TA*>(g.x);
  const TS* sub = reinterpret_cast<const TS*>(g.sub);
  const bool centred = SUB && g.sub != nullptr;
  // (vector instructions do not overlap the matrix pipe: one 16-byte load - two for f64 rows of a 32-row tile - instead of
  // predicated scalar loads)
  if (r0 + 16 * RT <= g.N && kc + KC <= g.K && (g.ldx & 3) == 0 && (((uintptr_t)g.x) & 15) == 0 &&
      (!centred || (((uintptr_t)g.sub) & 15) == 0)) {
    load_run<TA, PER>(x + gr * g.ldx + gk, c.x);
    if constexpr (SUB) {
      if (centred) load_run<TS, PER>(sub + gk, c.m);
    }
    c.mode = centred ? 2 : 1;
    return;
  }
  c.mode = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    double v = 0.0;
    if (gr < g.N && gk + q < g.K) {
      const TA xv = x[gr * g.ldx + gk + q];
      v = centred ? sub_promote<TA, TS>(xv, sub[gk + q]) : (double)xv;
    }
    if constexpr (sizeof(TA) == 8) c.x[q] = v;
    else c.v[q] = v;
  }
}

template <typename TA, typename TS, int RT, bool SUB>
__device__ __forceinline__ double row_chunk_value(const RowChunk<TA, TS, RT, SUB>& c, int q) {
  if constexpr (SUB) {
    if (c.mode == 2) return sub_promote<TA, TS>(c.x[q], c.m[q]);
  }
  if constexpr (sizeof(TA) == 8) return (double)c.x[q];
  else return (c.mode == 1) ? (double)c.x[q] : c.v[q];
}

// One workgroup = 32 rows x all n columns (256 at a time); wave w owns the 64-column slice w of each pass.
// The rows stream from HBM once per 256-column pass through a double-buffered 32x32 LDS chunk; the packed
// weights come from L2 one k-step pair ahead (mfma_chunk).
// NCT column tiles per wave (4: the workgroup covers 256 columns per pass; 2 / 1: 128 / 64 columns, for outputs that are
// no wider - PCA-16 ... PCA-128 and the MD that follows, the reference's default being 16 components: the padded tiles of
// the 256-column form cost the same 57 us at n = 16 as at n = 256)
// EPI_KDE, shared by the fused kernel and by the replay of the column-split launch (the same arithmetic in the same
// order: a row scores the same bits on both):
// one block's NCT values of a lane's accumulator row into the lane's running (max, sum of exp(. - max))
template <int NCT>
__device__ __forceinline__ void kde_online_update(const double (&val)[NCT], double& rowmax, double& rowdot) {
#if defined(KDE_ABLATE) && KDE_ABLATE >= 1  // (timing experiments only: what the epilogue's exponentials cost)
#pragma unroll
  for (int c = 0; c < NCT; ++c) rowdot += val[c];
  rowmax = 0.0;
  return;
#endif
  double gmax = -kInfD();
#pragma unroll
  for (int c = 0; c < NCT; ++c) gmax = fmax(gmax, val[c]);
  if (gmax > -kInfD()) {
    const double mnew = fmax(rowmax, gmax);
    double part = 0.0;
#pragma unroll
    for (int c = 0; c < NCT; ++c) part += exp(val[c] - mnew);
    rowdot = rowdot * exp(rowmax - mnew) + part;
    rowmax = mnew;
  }
}
// the (max, sum) pairs of the 16 lanes that share a row, then of the four waves -> out[row]
template <int RT>
__device__ __forceinline__ void kde_merge_store(double (&rowdot)[RT][4], double (&rowmax)[RT][4], double (*lds_m)[16 * RT],
                                                double (*lds_s)[16 * RT], int wave, int li, int lg, int tid, int64_t r0,
                                                int64_t N, double addc, double* __restrict__ out) {
  constexpr int BM = 16 * RT;
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double m = rowmax[a][r], sm = rowdot[a][r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const double m2 = shfl_xor_f64(m, o), s2 = shfl_xor_f64(sm, o);
        const double mn = fmax(m, m2);
        sm = (mn > -kInfD()) ? sm * exp(m - mn) + s2 * exp(m2 - mn) : 0.0;
        m = mn;
      }
      if (li == 0) {
        lds_m[wave][16 * a + lg + 4 * r] = m;
        lds_s[wave][16 * a + lg + 4 * r] = sm;
      }
    }
  __syncthreads();
  if (tid < BM) {
    const int64_t row = r0 + tid;
    if (row < N) {
      const double gm = fmax(fmax(lds_m[0][tid], lds_m[1][tid]), fmax(lds_m[2][tid], lds_m[3][tid]));
      double gs = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (lds_s[w][tid] > 0.0) gs += lds_s[w][tid] * exp(lds_m[w][tid] - gm);
      out[row] = log(gs) + gm + addc;
    }
  }
}

template <typename TA, typename TS, int EPI, int RT = 2, int NCT = 4>
__global__ __launch_bounds__(256) void gemm_rows_kernel(GemmArgs g) {
  constexpr int BM = 16 * RT;  // rows per workgroup (shadows the file-level default of 32)
  __shared__ double lds_a[2][BM][APITCH];
  __shared__ double lds_part[4][BM];
  __shared__ double lds_part2[(EPI == EPI_KDE) ? 4 : 1][BM];
  // EPI_MAHA: per (wave, row, class) partial of sum_j (G_j - (mu_c P)_j)(2 t_j - a_j); every slot is owned by one lane
  __shared__ double lds_cls[(EPI == EPI_MAHA) ? 4 * BM * kMahaMaxClasses : 1];
  if constexpr (EPI == EPI_MAHA) {
    for (int i = threadIdx.x; i < 4 * BM * kMahaMaxClasses; i += 256) lds_cls[i] = 0.0;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t n_pad = n_padded(g.n);
  const int64_t NT = n_pad / 16;
  const int64_t nchunks = k_padded(g.K) / KC;
  int64_t tile_id = blockIdx.x, cb_begin = g.cb0, cb_end = g.cb1 ? g.cb1 : n_pad / BN;
  if constexpr (EPI == EPI_MAHA || EPI == EPI_KDE || EPI == EPI_ROWDOT || EPI == EPI_ROWNORM) {
    if ((EPI == EPI_MAHA) ? (g.maha_part != nullptr) : (EPI == EPI_KDE) ? (g.kde_vals != nullptr) : (g.md_vals != nullptr)) {  // column-split launch (uniform)
      const int64_t nb = n_pad / BN;
      if constexpr (EPI == EPI_MAHA) {
        // block-major: the workgroups resident at any moment (consecutive ids) are consecutive row tiles of the SAME
        // 256-column block of P, so the block (K x 256 f64: 4 MB at K = 2048, the size of an XCD's L2) is fetched once per
        // XCD and launch instead of once per row tile
        const int64_t tiles = (g.N + BM - 1) / BM;
        tile_id = blockIdx.x % tiles;
        cb_begin = blockIdx.x / tiles;
      } else {
        tile_id = blockIdx.x / nb;
        cb_begin = blockIdx.x % nb;
      }
      cb_end = cb_begin + 1;
    }
  }
  const int64_t r0 = tile_id * BM;

  double rowdot[RT][4];   // EPI_KDE: running sum of exp(. - rowmax)
  double rowmax[RT][4];   // EPI_KDE only
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { rowdot[a][r] = 0.0; rowmax[a][r] = -kInfD(); }

  // EPI_KDE: the tile's row norms once (they were fetched in front of every block's epilogue - a global load's latency per
  // block and lane); the block's column norms are requested at the head of the block and arrive under its products
  __shared__ double lds_rown[(EPI == EPI_KDE) ? BM : 1];
  if constexpr (EPI == EPI_KDE) {
    if (tid < BM) lds_rown[tid] = (r0 + tid < g.N) ? g.rown[r0 + tid] : 0.0;  // (published by the chunk loop's barriers)
  }
  constexpr bool SUB = (EPI == EPI_ROWDOT || EPI == EPI_ROWNORM);
  RowChunk<TA, TS, RT, SUB> areg;  // this thread's part of the next staged chunk of rows
  // weights three k-step pairs ahead (mfma_chunk_ring) wherever the registers allow it: with one pair of look-ahead - 1 024
  // matrix-pipe cycles at 32 x 256, 512 at 16 x 256 - a fetch that misses the L2 stalls the products (PCA 100 000 x 1024 ->
  // 256: 0.93 -> 0.83 ms; Mahalanobis 262 144 x 2048: 35.84 -> 35.66 ms, its weights come from the Infinity Cache either way)
#ifndef GEMM_RING
#define GEMM_RING 1
#endif
  constexpr bool RING = GEMM_RING != 0;
  double2 b0[NCT];
  double2 bring[RING ? 4 : 1][NCT];
  auto load_b_head = [&](const double2* p) {  // the first k-step pair(s) of a block's weights
    if constexpr (RING) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < NCT; ++c) bring[j][c] = p[j * NT * 64 + c * 64];
    } else {
#pragma unroll
      for (int c = 0; c < NCT; ++c) b0[c] = p[c * 64];
    }
  };
  for (int64_t cb = cb_begin; cb < cb_end; ++cb) {
    const int64_t ctbase = g.ct0 + cb * 16 + wave * NCT;
    double cn[(EPI == EPI_KDE) ? NCT : 1];
    if constexpr (EPI == EPI_KDE) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int64_t col = (ctbase + c) * 16 + li;
        cn[c] = (col < g.n) ? g.coln[col] : 0.0;
      }
    }
    d4 acc[RT][NCT];
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int c = 0; c < NCT; ++c) acc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
    const double2* bp = reinterpret_cast<const double2*>(g.packed) + ctbase * 64 + lane;
    if (cb == cb_begin) load_b_head(bp);  // (later blocks: requested behind the previous block's last chunk)

    if (cb == cb_begin) load_a_regs<TA, TS, RT, SUB>(g, r0, 0, tid, areg);
    int buf = 0;
    int64_t nch = nchunks;  // chunks of this column block
    if constexpr (EPI == EPI_ROWNORM) {
      if (g.tri && (cb + 1) * (BN / KC) < nchunks) nch = (cb + 1) * (BN / KC);  // (uniform) nothing below the diagonal block
    }
    for (int64_t ch = 0; ch < nch; ++ch) {
      {
        constexpr int PER = 2 * RT, TPR = KC / PER;
        const int row = tid / TPR, kk = (tid % TPR) * PER;
#pragma unroll
        for (int q = 0; q < PER; ++q) lds_a[buf][row][kk + q] = row_chunk_value<TA, TS, RT, SUB>(areg, q);
      }
      __syncthreads();
      // the next chunk's rows are requested before this chunk's products; behind the last chunk that is the FIRST chunk of
      // the next 256-column block (the same rows again), so its latency passes under the products and the epilogue instead
      // of in front of every block (KDE at K = 256: 8 chunks per block)
      if (ch + 1 < nch) load_a_regs<TA, TS, RT, SUB>(g, r0, (ch + 1) * KC, tid, areg);
      else if (cb + 1 < cb_end) load_a_regs<TA, TS, RT, SUB>(g, r0, 0, tid, areg);
      if constexpr (RING) mfma_chunk_ring<RT, NCT>(acc, &lds_a[buf][0][0], APITCH, li, lg, bp + ch * 4 * NT * 64, NT * 64, bring);
      else mfma_chunk<RT, NCT>(acc, &lds_a[buf][0][0], APITCH, li, lg, bp + ch * 4 * NT * 64, NT * 64, b0);
      buf ^= 1;
    }
    // the next block's first weights travel under this block's epilogue (the look-ahead of the last chunk fetched the zero
    // padding behind K instead)
    if (cb + 1 < cb_end) load_b_head(bp + 16 * 64);

    // ---- epilogue for this 256-column block ----
#if defined(MAHA_ABLATE) && MAHA_ABLATE >= 1  // (timing experiments only: what the class-term epilogue costs)
    if constexpr (EPI == EPI_MAHA) {
      double keep = 0.0;
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) keep += acc[a][c][0] + acc[a][c][1] + acc[a][c][2] + acc[a][c][3];
      if (li == 0 && lg == 0) lds_cls[wave * BM * kMahaMaxClasses] += keep;
    } else
#endif
    if constexpr (EPI == EPI_MAHA) {
      // class terms of this 256-column block straight from the accumulators (G = X P never goes to memory).
      // Where the kernel's time goes (round 5, 262 144 x 2048 rows, 10 classes, tools/ablate/run_maha_variants.sh): 35.3 ms as
      // it is (0.79 of the f64 matrix peak), 30.7 ms with this epilogue compiled out (0.91) - ~37 vector instructions per
      // (row, class, 4 columns), and on gfx950 vector instructions do not overlap v_mfma_f64 (tools/microbench/
      // mfma_valu_overlap.hip), so the class terms cost their issue time 1:1.  Measured and not kept: 64-row tiles at one wave
      // per SIMD 40.1 ms (-DMAHA_RT=4), one pair of weight look-ahead instead of three 35.6 ms, the column-split launches in
      // block-major order 144.5 vs 136.1 ms per 1 M rows, and ranking the classes first so that only the classes that can still
      // be the maximum get the f32-difference term: 33.2 ms with the ranking for free, 35.3 ms with a 32 x D x C ranking prologue
      // in this kernel (its loads are latency-bound: 2.1 ms) - profiles/README.md, round 5.
      // for every class c, row partial += (G_j - (mu_c P)_j) * (2 t_j - a_j) over the lane's 4 columns, reduced over the
      // 16 lanes that share a row, added to the (wave, row, class) slot this lane group owns.  ~5 % of the block's MFMA time.
      const TA* xg = reinterpret_cast<const TA*>(g.x);
      const TA* mug = reinterpret_cast<const TA*>(g.class_mean);
      TA xv[RT][NCT][4];
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int64_t col = (ctbase + c) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + 16 * a + lg + 4 * r;
            xv[a][c][r] = (row < g.N && col < g.n) ? xg[row * g.ldx + col] : (TA)0;
          }
        }
      for (int cls = 0; cls < g.n_classes; ++cls) {
        TA mv[NCT];
        double qv[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int64_t col = (ctbase + c) * 16 + li;
          mv[c] = (col < g.n) ? mug[(int64_t)cls * g.K + col] : (TA)0;
          qv[c] = (col < g.n) ? g.mu_p[(int64_t)cls * g.K + col] : 0.0;
        }
#pragma unroll
        for (int a = 0; a < RT; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double part = 0.0;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
              const double ad = (double)xv[a][c][r] - (double)mv[c];
              const double td = (double)(TA)(xv[a][c][r] - mv[c]);
              part = fma(acc[a][c][r] - qv[c], 2.0 * td - ad, part);  // zero beyond n: G = 0, q = 0
            }
            part += shfl_xor_f64(part, 1);
            part += shfl_xor_f64(part, 2);
            part += shfl_xor_f64(part, 4);
            part += shfl_xor_f64(part, 8);
            if (li == 0) lds_cls[(wave * BM + 16 * a + lg + 4 * r) * kMahaMaxClasses + cls] += part;
          }
      }
    } else if constexpr (EPI == EPI_KDE) {
      // online logsumexp over this lane's 4 columns of the block, per accumulator row
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double rn = lds_rown[16 * a + lg + 4 * r];
          double val[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) {
            const int64_t col = (ctbase + c) * 16 + li;
            val[c] = (col < g.n) ? g.alpha * (rn + cn[c] - 2.0 * acc[a][c][r]) : -kInfD();
          }
          if (g.kde_vals) {  // column-split launch: the values go to memory, kde_replay_kernel does the rest
            double* dst = g.kde_vals + ((((tile_id * (n_pad / BN) + cb) * 256 + tid) * (RT * 4) + (a * 4 + r)) * NCT);
#pragma unroll
            for (int c = 0; c < NCT; ++c) dst[c] = val[c];
          } else {
            kde_online_update<NCT>(val, rowmax[a][r], rowdot[a][r]);
          }
        }
    } else if ((EPI == EPI_ROWDOT || EPI == EPI_ROWNORM) && g.md_vals) {  // column-split launch: the products go to memory, md_replay_kernel adds them
      const TA* x = reinterpret_cast<const TA*>(g.x);
#pragma unroll
      for (int a = 0; a < RT; ++a)
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const int64_t col = (ctbase + c) * 16 + li;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + 16 * a + lg + 4 * r;
            double prod = 0.0;  // (a term the unsplit kernel skips: adding +0.0 leaves its running sum as it is)
            if (row < g.N && col < g.n) {
              if constexpr (EPI == EPI_ROWNORM) {
                prod = acc[a][c][r] * acc[a][c][r];
              } else {
                const TA xv = x[row * g.ldx + col];
                const double d = g.sub ? sub_promote<TA, TS>(xv, reinterpret_cast<const TS*>(g.sub)[col]) : (double)xv;
                prod = acc[a][c][r] * d;
              }
            }
            g.md_vals[(((tile_id * (n_pad / BN) + cb) * 256 + tid) * (RT * 4) + (a * 4 + r)) * NCT + c] = prod;
          }
        }
    } else
#pragma unroll
    for (int a = 0; a < RT; ++a) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int64_t col = (ctbase + c) * 16 + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = r0 + 16 * a + lg + 4 * r;
          if (row < g.N && col < g.n) {
            const double v = acc[a][c][r];
            if constexpr (EPI == EPI_PCA) {
              double y = v - g.bias[col];
              if (g.scale) y = y / g.scale[col];
              g.out[row * g.n + col] = y;
            } else if constexpr (EPI == EPI_STORE) {
              g.out[row * g.n + col] = v;
            } else if constexpr (EPI == EPI_ROWNORM) {  // || (x - sub) B ||_2: sum of squares here, sqrt at the end
              rowdot[a][r] += v * v;
            } else {  // EPI_ROWDOT: sum_j (d P)_j d_j with d = x - sub
              const TA* x = reinterpret_cast<const TA*>(g.x);
              const TA xv = x[row * g.ldx + col];
              const double d = g.sub ? sub_promote<TA, TS>(xv, reinterpret_cast<const TS*>(g.sub)[col]) : (double)xv;
              rowdot[a][r] += v * d;
            }
          }
        }
      }
    }
    __syncthreads();  // all waves done with the last chunk before the next block restages LDS
  }

  if constexpr (EPI == EPI_MAHA) {
    __syncthreads();
    if (g.maha_part) {  // this block's partial sums, as they stand in LDS
      const int C = g.n_classes;
      for (int i = tid; i < 4 * BM * C; i += 256) {
        const int w = i / (BM * C), rr = (i / C) % BM, cls = i % C;
        const int64_t row = r0 + rr;
        if (row < g.N)
          g.maha_part[((cb_begin * 4 + w) * g.N + row) * C + cls] = lds_cls[(w * BM + rr) * kMahaMaxClasses + cls];
      }
      return;
    }
    if (tid < BM) {
      const int64_t row = r0 + tid;
      if (row < g.N) {
        double best = -kInfD();
        for (int cls = 0; cls < g.n_classes; ++cls) {
          const double t = ((lds_cls[(0 * BM + tid) * kMahaMaxClasses + cls] + lds_cls[(1 * BM + tid) * kMahaMaxClasses + cls]) +
                            lds_cls[(2 * BM + tid) * kMahaMaxClasses + cls]) + lds_cls[(3 * BM + tid) * kMahaMaxClasses + cls];
          double sc = -t;
          if (sc != sc) sc = -kInfD();  // NaN (class without training samples) -> -inf, as the reference
          best = fmax(best, sc);
        }
        g.out[row] = best;
      }
    }
  }
  if constexpr (EPI == EPI_KDE) {
    if (g.kde_vals) return;
    kde_merge_store<RT>(rowdot, rowmax, lds_part, lds_part2, wave, li, lg, tid, r0, g.N, g.addc, g.out);
  }
  if constexpr (EPI == EPI_ROWDOT || EPI == EPI_ROWNORM) {
    if ((EPI == EPI_ROWDOT || EPI == EPI_ROWNORM) && g.md_vals) return;
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = rowdot[a][r];
        v += shfl_xor_f64(v, 1);
        v += shfl_xor_f64(v, 2);
        v += shfl_xor_f64(v, 4);
        v += shfl_xor_f64(v, 8);
        if (li == 0) lds_part[wave][16 * a + lg + 4 * r] = v;
      }
    __syncthreads();
    if (tid < BM) {
      const int64_t row = r0 + tid;
      if (row < g.N) {
        double t = ((lds_part[0][tid] + lds_part[1][tid]) + lds_part[2][tid]) + lds_part[3][tid];
        if constexpr (EPI == EPI_ROWNORM) {
          if (g.acc_in) t += g.acc_in[row];
          g.out[row] = (g.neg_sq == 0) ? sqrt(t) : (g.neg_sq == 1 ? -t : t);
        } else {
          g.out[row] = -t;
        }
      }
    }
  }
}

// squared L2 norm of every row, f64: one wave per row, fixed summation order
__global__ __launch_bounds__(256) void row_sqnorm_f64_kernel(const double* __restrict__ x, double* __restrict__ out,
                                                              int64_t N, int64_t D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  double s = 0.0;
  for (int64_t i = lane; i < D; i += 64) {
    const double v = x[row * D + i];
    s = fma(v, v, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += shfl_xor_f64(s, o);
  if (lane == 0) out[row] = s;
}

// 32-row tiles halve the L2 traffic of the packed weights but need several tiles per CU to balance: below 4 per CU the
// 16-row instantiation runs (N = 10 000: 313 tiles of 32 rows are 1.2 rounds on 256 CUs - PCA transform 75 us, MD
// 51 us; 625 tiles of 16 rows: 59 and 33 us; LaRED 8 192 x 10 000 x 256: 0.94 -> 0.88 ms).  A row's bits do not depend on
// the tile height (tests/test_full_size_gpu.py).
//
// Last round (round 4).  All workgroups of a launch take the same time, so a grid of 6.1 x (resident workgroups) runs 7
// rounds, the last one on a tenth of the chip (100 000 x 1024 -> 256: matrix pipe busy 0.72 against the 0.81 of the same
// kernel on Mahalanobis' 61 rounds).  The whole rounds therefore go first as 32-row tiles and the remaining rows follow
// as a second launch of units a fraction of that size: 16-row tiles (half a unit; taken when they fit the chip at once),
// or - KDE / MD with a workspace, see their entry points - one workgroup per (16-row tile, 256-column block).
template <auto KERNEL>
static int64_t resident_workgroups() {  // of one instantiation on the current device; cached
  static int64_t slots = 0;
  if (slots == 0) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    slots = (int64_t)per_cu * runia_cu_count();
  }
  return slots;
}

// the rows from `first` on, as a problem of its own (every per-row pointer moved; column-split workspaces are not set here)
template <typename TA>
static GemmArgs gemm_rows_from(const GemmArgs& g, int64_t first, int64_t out_ld) {
  GemmArgs t = g;
  t.x = reinterpret_cast<const TA*>(g.x) + first * g.ldx;
  t.N = g.N - first;
  if (g.rown) t.rown = g.rown + first;
  if (g.acc_in) t.acc_in = g.acc_in + first;
  t.out = g.out + first * out_ld;
  return t;
}

// rows [0, return value) are whole rounds of 32-row tiles; 0 = no split (few tiles, or the last round is nearly full)
template <typename TA, int EPI, typename TS, int NCT>
static int64_t gemm_whole_round_rows(const GemmArgs& g) {
  const int64_t tiles = (g.N + BM - 1) / BM;
  const int64_t slots = resident_workgroups<gemm_rows_kernel<TA, TS, EPI, 2, NCT>>();
  const int64_t whole = tiles / slots * slots, rest = tiles - whole;
  if (whole == 0 || rest == 0 || rest * 8 >= slots * 7) return 0;
  return whole * BM;
}

template <typename TA, int EPI, typename TS, int NCT>
int launch_gemm_nct(const GemmArgs& g, hipStream_t s) {
  const int64_t tiles = (g.N + BM - 1) / BM;
  if (tiles > 0x7fffffff) return RUNIA_E_INVALID;
  if (EPI != EPI_MAHA && tiles < 4 * runia_cu_count()) {
    gemm_rows_kernel<TA, TS, EPI, 1, NCT><<<(unsigned)((g.N + 15) / 16), 256, 0, s>>>(g);
    return runia_check_launch();
  }
  if constexpr (EPI != EPI_MAHA) {
    const int64_t head = gemm_whole_round_rows<TA, EPI, TS, NCT>(g);
    const int64_t rest16 = (g.N - head + 15) / 16;
    if (head > 0 && rest16 <= resident_workgroups<gemm_rows_kernel<TA, TS, EPI, 1, NCT>>()) {
      GemmArgs h = g;
      h.N = head;
      gemm_rows_kernel<TA, TS, EPI, 2, NCT><<<(unsigned)(head / BM), 256, 0, s>>>(h);
      const GemmArgs t = gemm_rows_from<TA>(g, head, (EPI == EPI_PCA || EPI == EPI_STORE) ? g.n : 1);
      gemm_rows_kernel<TA, TS, EPI, 1, NCT><<<(unsigned)rest16, 256, 0, s>>>(t);
      return runia_check_launch();
    }
  }
#if defined(MAHA_RT) && MAHA_RT == 4  // (experiments: 64-row tiles, one workgroup per compute unit)
  if constexpr (EPI == EPI_MAHA && NCT == 4) {
    gemm_rows_kernel<TA, TS, EPI, 4, NCT><<<(unsigned)((g.N + 63) / 64), 256, 0, s>>>(g);
    return runia_check_launch();
  }
#endif
  gemm_rows_kernel<TA, TS, EPI, 2, NCT><<<(unsigned)tiles, 256, 0, s>>>(g);
  return runia_check_launch();
}

template <typename TA, int EPI, typename TS = double>
int launch_gemm(const GemmArgs& g, hipStream_t s) {
  // narrow outputs (n <= 64 / 128 columns): one / two column tiles per wave instead of four - the choice depends on n
  // alone, so a row's bits do not depend on the batch
  if constexpr (EPI == EPI_PCA || EPI == EPI_ROWDOT || EPI == EPI_STORE || EPI == EPI_ROWNORM) {
    if (g.n <= 64) return launch_gemm_nct<TA, EPI, TS, 1>(g, s);
    if (g.n <= 128) return launch_gemm_nct<TA, EPI, TS, 2>(g, s);
  }
  return launch_gemm_nct<TA, EPI, TS, 4>(g, s);
}

// ---- Mahalanobis class terms: one wave per row over G = X P ---------------------------
// s_c = -(t P t^T), t = fl(x - mu_c) in the input dtype.  With a = x - mu_c exact in f64 and
// t = a + e:  t P t^T = sum_j (G_j - (mu_c P)_j) * (2 t_j - a_j) + O(e^2)   (P symmetric).
template <typename TX>
__global__ __launch_bounds__(256) void maha_class_kernel(const TX* __restrict__ x, const TX* __restrict__ mu,
                                                          const double* __restrict__ G,
                                                          const double* __restrict__ muP,
                                                          double* __restrict__ score, int64_t rows, int64_t D,
                                                          int C) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const TX* xr = x + row * D;
    const double* gr = G + row * D;
    double best = -kInfD();
    for (int c = 0; c < C; ++c) {
      const TX* m = mu + (int64_t)c * D;
      const double* q = muP + (int64_t)c * D;
      double acc = 0.0;
      for (int64_t j = lane; j < D; j += 64) {
        const TX xv = xr[j], mv = m[j];
        const double a = (double)xv - (double)mv;
        const double t = (double)(TX)(xv - mv);
        acc += (gr[j] - q[j]) * (2.0 * t - a);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
      double sc = -acc;
      if (sc != sc) sc = -kInfD();  // NaN (class without training samples) -> -inf
      best = fmax(best, sc);
    }
    if (lane == 0) score[row] = best;
  }
}

// ---- many classes (C > 16): the class terms as a second contraction --------------------------------------------------
// a P a^T = x P x^T - 2 (x P) mu_c^T + mu_c P mu_c^T = qx - 2 S_c + qc with G = x P, S = G M^T (M = class means, D x C
// on the matrix cores), qc fixed per class.  That is the exact quadratic form of a = x - mu_c; the reference's
// t = fl32(a) differs from it by ~1e-7 relative, so S only RANKS the classes: every class whose base value comes within
// 1e-3 of the best one is then re-evaluated with the f32-diff formula of maha_class_kernel (typically one class per row).
// Work per row: 2 D^2 + 2 D C on MFMA + O(D) per candidate, instead of ~6 D C vector operations + C re-reads of G
// (C = 1000, D = 2048: 77 ms per 16 384 rows with the class loop).
template <typename TX>  // packed layout of pack_weights_kernel with B[k][c] = mean[c][k]
__global__ __launch_bounds__(256) void pack_class_means_kernel(const TX* __restrict__ mean, int64_t D, int C,
                                                                double* __restrict__ packed, int64_t NT, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int h = (int)(i & 1);
    const int lane = (int)((i >> 1) & 63);
    const int64_t t = i >> 7;
    const int64_t ct = t % NT, s2 = t / NT;
    const int64_t k = 8 * s2 + 4 * h + (lane >> 4);
    const int64_t c = 16 * ct + (lane & 15);
    packed[i] = (k < D && c < C) ? (double)mean[c * D + k] : 0.0;
  }
}

template <typename TX>  // qc[c] = mu_c P mu_c^T = mu_c . (mu_c P): one wave per class
__global__ __launch_bounds__(256) void class_quad_kernel(const TX* __restrict__ mean, const double* __restrict__ muP,
                                                          double* __restrict__ qc, int64_t D, int C) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double acc = 0.0;
  for (int64_t j = lane; j < D; j += 64) acc = fma((double)mean[(int64_t)c * D + j], muP[(int64_t)c * D + j], acc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
  if (lane == 0) qc[c] = acc;
}

template <typename TX>
__global__ __launch_bounds__(256) void maha_refine_kernel(const TX* __restrict__ x, const TX* __restrict__ mu,
                                                           const double* __restrict__ G, const double* __restrict__ muP,
                                                           const double* __restrict__ S, const double* __restrict__ qc,
                                                           double* __restrict__ score, int64_t rows, int64_t D, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // wave-uniform
  const TX* xr = x + row * D;
  const double* gr = G + row * D;
  const double* sr = S + row * (int64_t)C;
  double qx = 0.0;
  for (int64_t j = lane; j < D; j += 64) qx = fma(gr[j], (double)xr[j], qx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qx += shfl_xor_f64(qx, o);
  // v_c = 2 S_c - qc = qx - (a P a^T): the larger, the closer the class (NaN = class without samples: never a candidate)
  double m = -kInfD();
  for (int c = lane; c < C; c += 64) {
    const double v = 2.0 * sr[c] - qc[c];
    if (v == v) m = fmax(m, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, shfl_xor_f64(m, o));
  double best = -kInfD();
  if (m > -kInfD()) {
    const double thr = m - (1e-3 * fabs(qx - m) + 1e-12 * (fabs(qx) + fabs(m)));
    for (int c0 = 0; c0 < C; c0 += 64) {
      const int c = c0 + lane;
      bool cand = false;
      if (c < C) {
        const double v = 2.0 * sr[c] - qc[c];
        cand = (v >= thr);
      }
      unsigned long long mask = __ballot(cand);
      while (mask) {  // wave-uniform
        const int cc = c0 + __builtin_ctzll(mask);
        mask &= mask - 1;
        const TX* mm = mu + (int64_t)cc * D;
        const double* q = muP + (int64_t)cc * D;
        double acc = 0.0;
        for (int64_t j = lane; j < D; j += 64) {
          const TX xv = xr[j], mv = mm[j];
          const double a = (double)xv - (double)mv;
          const double t = (double)(TX)(xv - mv);
          acc += (gr[j] - q[j]) * (2.0 * t - a);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += shfl_xor_f64(acc, o);
        double sc = -acc;
        if (sc != sc) sc = -kInfD();
        best = fmax(best, sc);
      }
    }
  }
  if (lane == 0) score[row] = best;
}

}  // namespace

extern "C" size_t runia_packed_weights_bytes(int64_t K, int64_t n) {
  if (K <= 0 || n <= 0) return 0;
  return (size_t)packed_elems(K, n) * sizeof(double);
}

extern "C" int runia_pack_weights_f64(const double* B, int64_t ldb, int64_t K, int64_t n, double* packed,
                                      runia_stream_t stream) {
  if (!B || !packed || K <= 0 || n <= 0 || ldb < n) return RUNIA_E_INVALID;
  const int64_t total = packed_elems(K, n);
  pack_weights_kernel<<<runia_stream_grid(total, 256), 256, 0, as_stream(stream)>>>(B, ldb, K, n, packed,
                                                                                     n_padded(n) / 16, total);
  return runia_check_launch();
}

static int pca_impl(const void* x, bool f32in, const double* packed_ct, const double* bias, const double* scale,
                    double* y, int64_t N, int64_t D, int64_t n, int whiten, runia_stream_t stream) {
  if (N < 0 || D <= 0 || n <= 0 || (N > 0 && (!x || !y)) || !packed_ct || !bias || (whiten && !scale))
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  GemmArgs g{};
  g.x = x; g.ldx = D; g.packed = packed_ct; g.N = N; g.K = D; g.n = n;
  g.sub = nullptr; g.bias = bias; g.scale = whiten ? scale : nullptr; g.out = y;
  return f32in ? launch_gemm<float, EPI_PCA>(g, as_stream(stream))
               : launch_gemm<double, EPI_PCA>(g, as_stream(stream));
}

extern "C" int runia_pca_transform_f64(const double* x, const double* packed_ct, const double* bias,
                                       const double* scale, double* y, int64_t N, int64_t D, int64_t n,
                                       int whiten, runia_stream_t stream) {
  return pca_impl(x, false, packed_ct, bias, scale, y, N, D, n, whiten, stream);
}
extern "C" int runia_pca_transform_f32in(const float* x, const double* packed_ct, const double* bias,
                                         const double* scale, double* y, int64_t N, int64_t D, int64_t n,
                                         int whiten, runia_stream_t stream) {
  return pca_impl(x, true, packed_ct, bias, scale, y, N, D, n, whiten, stream);
}

template <typename TA, typename TS>
static int md_impl(const TA* x, const TS* mean, const double* packed_p, double* score, int64_t N, int64_t n,
                   runia_stream_t stream) {
  if (N < 0 || n <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !score || !mean || !packed_p) return RUNIA_E_INVALID;
  GemmArgs g{};
  g.x = x; g.ldx = n; g.packed = packed_p; g.N = N; g.K = n; g.n = n;
  g.sub = mean; g.bias = nullptr; g.scale = nullptr; g.out = score;
  return launch_gemm<TA, EPI_ROWDOT, TS>(g, as_stream(stream));
}

extern "C" int runia_md_score_f64(const double* x, const double* mean, const double* packed_p, double* score,
                                  int64_t N, int64_t n, runia_stream_t stream) {
  return md_impl<double, double>(x, mean, packed_p, score, N, n, stream);
}
extern "C" int runia_md_score_f32(const float* x, const float* mean, const double* packed_p, double* score,
                                  int64_t N, int64_t n, runia_stream_t stream) {
  return md_impl<float, float>(x, mean, packed_p, score, N, n, stream);
}
extern "C" int runia_md_score_f32x_f64mean(const float* x, const double* mean, const double* packed_p,
                                           double* score, int64_t N, int64_t n, runia_stream_t stream) {
  return md_impl<float, double>(x, mean, packed_p, score, N, n, stream);
}

// Few rows of wide features (MD on un-reduced 2048-d features, one image at a time): a 16-row tile walks the whole 33.5 MB
// precision matrix on ONE compute unit (0.9 ms at any batch <= 512 rows).  With a workspace the 256-column blocks of a tile go
// to separate workgroups, which store their products (d P)_j d_j, and a second launch adds them per lane in the unsplit kernel's
// order (blocks in order, the lane's four column tiles in order, then the same 16-lane and 4-wave sums): the same bits.
__global__ __launch_bounds__(256) void md_replay_kernel(const double* __restrict__ vals, double* __restrict__ out, int64_t N,
                                                         int64_t nb) {
  constexpr int NCT = 4;
  __shared__ double lds_part[4][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  double rowdot[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t cb = 0; cb < nb; ++cb) {
    const double* src = vals + (((int64_t)blockIdx.x * nb + cb) * 256 + tid) * (4 * NCT);
#pragma unroll
    for (int c = 0; c < NCT; ++c)  // the unsplit epilogue visits (column tile, register) in this order
#pragma unroll
      for (int r = 0; r < 4; ++r) rowdot[r] += src[r * NCT + c];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = rowdot[r];
    v += shfl_xor_f64(v, 1);
    v += shfl_xor_f64(v, 2);
    v += shfl_xor_f64(v, 4);
    v += shfl_xor_f64(v, 8);
    if (li == 0) lds_part[wave][lg + 4 * r] = v;
  }
  __syncthreads();
  if (tid < 16) {
    const int64_t row = r0 + tid;
    if (row < N) out[row] = -(((lds_part[0][tid] + lds_part[1][tid]) + lds_part[2][tid]) + lds_part[3][tid]);
  }
}

static bool md_split_wanted(int64_t N, int64_t n) { return (N + 15) / 16 < runia_cu_count() && n_padded(n) / BN > 1; }
extern "C" size_t runia_md_score_workspace_bytes(int64_t N, int64_t n) {
  if (N <= 0 || n <= 0 || !md_split_wanted(N, n)) return 0;
  return (size_t)(((N + 15) / 16) * (n_padded(n) / BN) * 256 * 16) * sizeof(double);
}
template <typename TA, typename TS>
static int md_ws_impl(const TA* x, const TS* mean, const double* packed_p, double* score, void* workspace, size_t workspace_bytes,
                      int64_t N, int64_t n, runia_stream_t stream) {
  const size_t need = runia_md_score_workspace_bytes(N, n);
  if (need == 0 || !workspace || workspace_bytes < need || N <= 0 || !x || !score || !mean || !packed_p)
    return md_impl<TA, TS>(x, mean, packed_p, score, N, n, stream);  // (large batches, narrow features, no workspace: the one launch)
  GemmArgs g{};
  g.x = x; g.ldx = n; g.packed = packed_p; g.N = N; g.K = n; g.n = n;
  g.sub = mean; g.out = score;
  g.md_vals = reinterpret_cast<double*>(workspace);
  hipStream_t s = as_stream(stream);
  const int64_t tiles = (N + 15) / 16, nb = n_padded(n) / BN;
  gemm_rows_kernel<TA, TS, EPI_ROWDOT, 1, 4><<<(unsigned)(tiles * nb), 256, 0, s>>>(g);
  md_replay_kernel<<<(unsigned)tiles, 256, 0, s>>>(g.md_vals, score, N, nb);
  return runia_check_launch();
}
extern "C" int runia_md_score_ws_f64(const double* x, const double* mean, const double* packed_p, double* score,
                                     void* workspace, size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream) {
  return md_ws_impl<double, double>(x, mean, packed_p, score, workspace, workspace_bytes, N, n, stream);
}
extern "C" int runia_md_score_ws_f32(const float* x, const float* mean, const double* packed_p, double* score, void* workspace,
                                     size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream) {
  return md_ws_impl<float, float>(x, mean, packed_p, score, workspace, workspace_bytes, N, n, stream);
}
extern "C" int runia_md_score_ws_f32x_f64mean(const float* x, const double* mean, const double* packed_p, double* score,
                                              void* workspace, size_t workspace_bytes, int64_t N, int64_t n,
                                              runia_stream_t stream) {
  return md_ws_impl<float, double>(x, mean, packed_p, score, workspace, workspace_bytes, N, n, stream);
}

// MD / LaREM score with the triangular factor of the precision (round 6): P = W^T W, W lower triangular ->
// -(x - mean) P (x - mean)^T = -|| W (x - mean) ||^2 with packed_wt = pack(W^T): the zero half of W is not multiplied.  Few rows
// of wide features take the column-split launch + replay of runia_md_score_ws_* (same workspace size, same bits as the one launch).
template <typename TA, typename TS>
static int md_tril_impl(const TA* x, const TS* mean, const double* packed_wt, double* score, void* workspace,
                        size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream) {
  if (N < 0 || n <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !packed_wt || !score) return RUNIA_E_INVALID;
  GemmArgs g{};
  g.x = x; g.ldx = n; g.packed = packed_wt; g.N = N; g.K = n; g.n = n;
  g.sub = mean; g.out = score; g.tri = 1; g.neg_sq = 1;
  hipStream_t s = as_stream(stream);
  const size_t need = runia_md_score_workspace_bytes(N, n);
  if (need == 0 || !workspace || workspace_bytes < need) return launch_gemm<TA, EPI_ROWNORM, TS>(g, s);
  g.md_vals = reinterpret_cast<double*>(workspace);
  const int64_t tiles = (N + 15) / 16, nb = n_padded(n) / BN;
  gemm_rows_kernel<TA, TS, EPI_ROWNORM, 1, 4><<<(unsigned)(tiles * nb), 256, 0, s>>>(g);
  md_replay_kernel<<<(unsigned)tiles, 256, 0, s>>>(g.md_vals, score, N, nb);
  return runia_check_launch();
}
extern "C" int runia_md_score_tril_f64(const double* x, const double* mean, const double* packed_wt, double* score,
                                       void* workspace, size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream) {
  return md_tril_impl<double, double>(x, mean, packed_wt, score, workspace, workspace_bytes, N, n, stream);
}
extern "C" int runia_md_score_tril_f32(const float* x, const float* mean, const double* packed_wt, double* score, void* workspace,
                                       size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream) {
  return md_tril_impl<float, float>(x, mean, packed_wt, score, workspace, workspace_bytes, N, n, stream);
}
extern "C" int runia_md_score_tril_f32x_f64mean(const float* x, const double* mean, const double* packed_wt, double* score,
                                                void* workspace, size_t workspace_bytes, int64_t N, int64_t n,
                                                runia_stream_t stream) {
  return md_tril_impl<float, double>(x, mean, packed_wt, score, workspace, workspace_bytes, N, n, stream);
}

// ViM residual: || (x - u) @ NS ||_2 per row (reference inference/postprocessors.py:1106): x - u follows NumPy's
// dtype rules (f32 - f32 in f32), the projection and the norm are f64.
template <typename TA, typename TS>
static int proj_norm_impl(const TA* x, const TS* u, const double* packed_ns, double* norm, int64_t N, int64_t D,
                          int64_t n, runia_stream_t stream) {
  if (N < 0 || D <= 0 || n <= 0) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!x || !u || !packed_ns || !norm) return RUNIA_E_INVALID;
  GemmArgs g{};
  g.x = x; g.ldx = D; g.packed = packed_ns; g.N = N; g.K = D; g.n = n;
  g.sub = u; g.bias = nullptr; g.scale = nullptr; g.out = norm;
  // A ragged last block: n = 1 048 (ViM at D = 2048, DIM = 1000) pads to 1 280 columns and the fifth 256-column block multiplies
  // 24 real ones (92 ms per 1 M rows, 0.59 of the matrix peak per useful product).  Two launches instead: the whole blocks leave
  // their sums of squares in `norm`, then the tail - one 64- or 128-column pass of the narrow-tile kernel over the same packed
  // matrix, started at its column tile - adds its own and takes the root.
  const int64_t whole = n / BN, tail = n - whole * BN;
  if (whole >= 1 && tail > 0 && tail <= 128) {
    hipStream_t s = as_stream(stream);
    GemmArgs a = g;
    a.cb0 = 0; a.cb1 = whole; a.neg_sq = 2;
    int rc = launch_gemm_nct<TA, EPI_ROWNORM, TS, 4>(a, s);
    if (rc != RUNIA_OK) return rc;
    GemmArgs b = g;
    b.cb0 = 0; b.cb1 = 1; b.ct0 = whole * (BN / 16); b.acc_in = norm; b.neg_sq = 0;
    return tail <= 64 ? launch_gemm_nct<TA, EPI_ROWNORM, TS, 1>(b, s) : launch_gemm_nct<TA, EPI_ROWNORM, TS, 2>(b, s);
  }
  return launch_gemm<TA, EPI_ROWNORM, TS>(g, as_stream(stream));
}
extern "C" int runia_proj_norm_f32(const float* x, const float* u, const double* packed_ns, double* norm, int64_t N,
                                   int64_t D, int64_t n, runia_stream_t stream) {
  return proj_norm_impl<float, float>(x, u, packed_ns, norm, N, D, n, stream);
}
extern "C" int runia_proj_norm_f64(const double* x, const double* u, const double* packed_ns, double* norm, int64_t N,
                                   int64_t D, int64_t n, runia_stream_t stream) {
  return proj_norm_impl<double, double>(x, u, packed_ns, norm, N, D, n, stream);
}

extern "C" int runia_row_sqnorm_f64(const double* x, double* out, int64_t N, int64_t D, runia_stream_t stream) {
  if (N < 0 || D <= 0 || (N > 0 && (!x || !out))) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  row_sqnorm_f64_kernel<<<(unsigned)((N + 3) / 4), 256, 0, as_stream(stream)>>>(x, out, N, D);
  return runia_check_launch();
}

// Second half of the column-split KDE launch: per 16-row tile, the blocks' values in block order through the same
// update and the same merges as the fused kernel (thread t of the tile's workgroup is thread t of the fused kernel).
__global__ __launch_bounds__(256) void kde_replay_kernel(const double* __restrict__ vals, double* __restrict__ out,
                                                          int64_t N, int64_t nb, double addc) {
  constexpr int RT = 1, NCT = 4;
  __shared__ double lds_m[4][16 * RT], lds_s[4][16 * RT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 16 * RT;
  double rowdot[RT][4], rowmax[RT][4];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { rowdot[a][r] = 0.0; rowmax[a][r] = -kInfD(); }
  for (int64_t cb = 0; cb < nb; ++cb) {
    const double* src = vals + (((int64_t)blockIdx.x * nb + cb) * 256 + tid) * (RT * 4 * NCT);
#pragma unroll
    for (int a = 0; a < RT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double val[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) val[c] = src[(a * 4 + r) * NCT + c];
        kde_online_update<NCT>(val, rowmax[a][r], rowdot[a][r]);
      }
  }
  kde_merge_store<RT>(rowdot, rowmax, lds_m, lds_s, wave, li, lg, tid, r0, N, addc, out);
}

// query norms, + the values of the column-split launch when the batch has fewer 16-row tiles than the chip has compute units
static bool kde_split_wanted(int64_t N, int64_t M) { return (N + 15) / 16 < runia_cu_count() && n_padded(M) / BN > 1; }
static size_t kde_split_bytes(int64_t rows, int64_t M) {
  return (size_t)(((rows + 15) / 16) * (n_padded(M) / BN) * 256 * 16) * sizeof(double);
}
// Large batches: the rows behind the whole rounds of 32-row tiles (launch_gemm_nct) go through the same column split - units
// of 1/(2 * blocks) of a 32-row tile instead of a last round on part of the chip (100 000 x 256 against 4 000 rows: 3 125
// tiles on 512 resident workgroups = 6.1 rounds; 4.40 -> 3.97 ms).  0 = no such rows.
static int64_t kde_last_round_rows(int64_t N, int64_t M) {
  if (n_padded(M) / BN < 2 || (N + BM - 1) / BM < 4 * runia_cu_count()) return 0;
  GemmArgs g{};
  g.N = N;
  const int64_t head = gemm_whole_round_rows<double, EPI_KDE, double, 4>(g);
  return head > 0 ? N - head : 0;
}
extern "C" size_t runia_kde_workspace_bytes(int64_t N, int64_t M) {
  if (N <= 0 || M <= 0) return 0;
  size_t bytes = (((size_t)N * sizeof(double)) + 255) / 256 * 256;
  if (kde_split_wanted(N, M)) bytes += kde_split_bytes(N, M);
  else bytes += kde_split_bytes(kde_last_round_rows(N, M), M);
  return bytes;
}

extern "C" int runia_kde_score_packed_f64(const double* packed_train_t, const double* train_sqnorm, const double* x,
                                          double* score, void* workspace, size_t workspace_bytes, int64_t M,
                                          int64_t N, int64_t D, double bandwidth, runia_stream_t stream) {
  if (M <= 0 || N < 0 || D <= 0 || !(bandwidth > 0.0)) return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  if (!packed_train_t || !train_sqnorm || !x || !score) return RUNIA_E_INVALID;
  if (!workspace || workspace_bytes < (size_t)N * sizeof(double)) return RUNIA_E_WORKSPACE;
  double* qn = reinterpret_cast<double*>(workspace);
  if (int rc = runia_row_sqnorm_f64(x, qn, N, D, stream)) return rc;
  GemmArgs g{};
  g.x = x; g.ldx = D; g.packed = packed_train_t; g.N = N; g.K = D; g.n = M;
  g.rown = qn; g.coln = train_sqnorm;
  g.alpha = -0.5 / (bandwidth * bandwidth);
  g.addc = -log((double)M) - (double)D * log(bandwidth) - 0.5 * (double)D * log(2.0 * M_PI);
  g.out = score;
  // Few rows (LaRED on the ~100 proposals of one image): one workgroup per 16-row tile walks the whole training set on ONE
  // compute unit (8 rows against 10 000 x 256: 0.6 ms).  With the workspace runia_kde_workspace_bytes asks for, the
  // 256-column blocks of a tile go to separate workgroups, which store their values, and a second launch replays the
  // online logsumexp over them in block order with the fused kernel's own update and merges: the same bits.
  if (kde_split_wanted(N, M) && workspace_bytes >= runia_kde_workspace_bytes(N, M)) {
    hipStream_t s = as_stream(stream);
    const int64_t tiles = (N + 15) / 16, nb = n_padded(M) / BN;
    g.kde_vals = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + (((size_t)N * sizeof(double)) + 255) / 256 * 256);
    gemm_rows_kernel<double, double, EPI_KDE, 1, 4><<<(unsigned)(tiles * nb), 256, 0, s>>>(g);
    kde_replay_kernel<<<(unsigned)tiles, 256, 0, s>>>(g.kde_vals, score, N, nb, g.addc);
    return runia_check_launch();
  }
  if (const int64_t rest = kde_last_round_rows(N, M); rest > 0 && workspace_bytes >= runia_kde_workspace_bytes(N, M)) {
    hipStream_t s = as_stream(stream);
    const int64_t head = N - rest, tiles = (rest + 15) / 16, nb = n_padded(M) / BN;
    if (tiles * nb <= 0x7fffffff) {
      GemmArgs h = g;
      h.N = head;
      gemm_rows_kernel<double, double, EPI_KDE, 2, 4><<<(unsigned)(head / BM), 256, 0, s>>>(h);
      GemmArgs t = gemm_rows_from<double>(g, head, 1);
      t.kde_vals = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + (((size_t)N * sizeof(double)) + 255) / 256 * 256);
      gemm_rows_kernel<double, double, EPI_KDE, 1, 4><<<(unsigned)(tiles * nb), 256, 0, s>>>(t);
      kde_replay_kernel<<<(unsigned)tiles, 256, 0, s>>>(t.kde_vals, t.out, rest, nb, g.addc);
      return runia_check_launch();
    }
  }
  return launch_gemm<double, EPI_KDE>(g, as_stream(stream));
}

extern "C" size_t runia_mahalanobis_workspace_bytes(int64_t N, int64_t D) {
  if (N <= 0 || D <= 0) return 0;
  // G = X P for a chunk of rows; chunk capped at 64 Ki rows so that the buffer stays L2/MALL friendly
  const int64_t rows = N < 65536 ? N : 65536;
  return (size_t)(rows * D) * sizeof(double);
}

// Workspace that also lets C > 16 classes take the matrix-core form of the class terms (below): the packed class means,
// qc, and G and S = G M^T for a chunk of rows.  (With only runia_mahalanobis_workspace_bytes the class loop runs.)
static size_t maha_class_carve_bytes(int64_t D, int C) {
  return (size_t)packed_elems(D, C) * sizeof(double) + (((size_t)C * sizeof(double) + 255) / 256) * 256;
}
extern "C" size_t runia_mahalanobis_workspace_bytes_classes(int64_t N, int64_t D, int C) {
  if (N <= 0 || D <= 0 || C <= 0) return 0;
  if (C <= kMahaMaxClasses) return runia_mahalanobis_workspace_bytes(N, D);
  const int64_t rows = N < 65536 ? N : 65536;
  return maha_class_carve_bytes(D, C) + (size_t)rows * (size_t)(D + C) * sizeof(double);
}

// Finish of the column-split Mahalanobis launch: per (row, class) the blocks' partial sums of each wave in block order
// from 0.0 - what the unsplit kernel's LDS slot accumulates - then the four waves as there: the same bits.
__global__ __launch_bounds__(256) void maha_split_finish_kernel(const double* __restrict__ part, double* __restrict__ score,
                                                                 int64_t N, int C, int64_t nb) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= N) return;
  double best = -kInfD();
  for (int cls = 0; cls < C; ++cls) {
    double sw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      double acc = 0.0;
      for (int64_t cb = 0; cb < nb; ++cb) acc += part[((cb * 4 + w) * N + row) * C + cls];
      sw[w] = acc;
    }
    double sc = -(((sw[0] + sw[1]) + sw[2]) + sw[3]);
    if (sc != sc) sc = -kInfD();  // NaN (class without training samples) -> -inf, as the reference
    best = fmax(best, sc);
  }
  score[row] = best;
}

// RUNIA_MAHA_SPLIT=1 sends large batches through the column-split launches as well (measurements only; same bits either way)
static bool maha_split_large() {
  static const bool on = [] { const char* e = getenv("RUNIA_MAHA_SPLIT"); return e && e[0] == '1'; }();
  return on;
}

template <typename TX>
static int maha_impl(const TX* x, const TX* class_mean, const double* packed_p, const double* mu_p,
                     double* score, void* workspace, size_t workspace_bytes, int64_t N, int64_t D, int C,
                     runia_stream_t stream) {
  if (N < 0 || D <= 0 || C <= 0 || (N > 0 && (!x || !score)) || !class_mean || !packed_p || !mu_p)
    return RUNIA_E_INVALID;
  if (N == 0) return RUNIA_OK;
  hipStream_t s = as_stream(stream);
  if (C <= kMahaMaxClasses) {
    // fused: the class terms are taken from the GEMM accumulators of every 256-column block; G = X P is never written
    // (the two-launch form below spends 20 % of its time re-reading it per class).  The workspace is not touched.
    GemmArgs g{};
    g.x = x; g.ldx = D; g.packed = packed_p; g.N = N; g.K = D; g.n = D;
    g.class_mean = class_mean; g.mu_p = mu_p; g.n_classes = C; g.out = score;
    // Few rows (serving, small batches): one workgroup per row tile walks all of P on ONE compute unit - a single row
    // against a 2048 x 2048 precision took 1.6 ms, 512 rows 1.4 ms.  With the caller's workspace the 256-column blocks
    // of a tile go to separate workgroups (8 x the parallelism at D = 2048) and a finishing launch adds their partial
    // sums in the unsplit kernel's order: a row scores the same bits in a batch of any size.
    //
    // Large batches stay on the one launch.  Round 5 measured them on the same split in block-major order (slices of 65 536
    // rows; RUNIA_MAHA_SPLIT=1 still selects it): the resident workgroups then share ONE 4 MB block of P instead of walking
    // all 33.5 MB out of phase (687 GB of L2 misses per 1 M rows in the one-launch form, profiles/r4_cfg3_pmc_summary.json) -
    // and 1 M x 2048 rows took 144.5 ms against 136.1 ms (profiles/r5_maha_split.txt): the rows are read once per block
    // (8 x 8.2 GB) and every (tile, block) workgroup refills its pipeline, which costs more than the misses did - they are
    // served by the Infinity Cache under the three-pair look-ahead of the weights.  Same bits either way.
    const int64_t nb = n_padded(D) / BN;
    const size_t per_row = (size_t)(nb * 4 * C) * sizeof(double);
    if (nb > 1 && workspace && (((uintptr_t)workspace) & 7) == 0) {
      int64_t cap = (int64_t)(workspace_bytes / per_row);
      if (cap > 65536) cap = 65536;
      if (cap >= N) cap = N; else cap = cap / BM * BM;  // slices end on tile boundaries
      const bool few = (N + BM - 1) / BM < runia_cu_count();
      if ((few && cap >= N) || (!few && maha_split_large() && cap >= 8192)) {
        for (int64_t r0 = 0; r0 < N; r0 += cap) {
          GemmArgs t = gemm_rows_from<TX>(g, r0, 1);
          if (t.N > cap) t.N = cap;
          const int64_t tiles = (t.N + BM - 1) / BM;
          t.maha_part = reinterpret_cast<double*>(workspace);
          gemm_rows_kernel<TX, double, EPI_MAHA, 2, 4><<<(unsigned)(tiles * nb), 256, 0, s>>>(t);
          maha_split_finish_kernel<<<(unsigned)((t.N + 255) / 256), 256, 0, s>>>(t.maha_part, t.out, t.N, C, nb);
        }
        return runia_check_launch();
      }
    }
    return launch_gemm<TX, EPI_MAHA>(g, s);
  }
  if (!workspace) return RUNIA_E_WORKSPACE;
  const size_t carve = maha_class_carve_bytes(D, C);
  if ((((uintptr_t)workspace) & 15) == 0 && workspace_bytes >= carve + (size_t)(D + C) * sizeof(double)) {
    // class terms on the matrix cores: S = G M^T ranks the classes, the f32-diff formula finishes the candidates
    double* packed_mt = reinterpret_cast<double*>(workspace);
    double* qc = packed_mt + packed_elems(D, C);
    double* G = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + carve);
    const int64_t cap_rows = (int64_t)((workspace_bytes - carve) / ((size_t)(D + C) * sizeof(double)));
    const int64_t total = packed_elems(D, C);
    pack_class_means_kernel<TX><<<runia_stream_grid(total, 256), 256, 0, s>>>(class_mean, D, C, packed_mt,
                                                                               n_padded(C) / 16, total);
    class_quad_kernel<TX><<<(unsigned)((C + 3) / 4), 256, 0, s>>>(class_mean, mu_p, qc, D, C);
    int rc = runia_check_launch();
    if (rc != RUNIA_OK) return rc;
    for (int64_t r0 = 0; r0 < N; r0 += cap_rows) {
      const int64_t rows = (N - r0 < cap_rows) ? (N - r0) : cap_rows;
      double* S = G + rows * D;
      GemmArgs g{};
      g.x = x + r0 * D; g.ldx = D; g.packed = packed_p; g.N = rows; g.K = D; g.n = D; g.out = G;
      rc = launch_gemm<TX, EPI_STORE>(g, s);
      if (rc != RUNIA_OK) return rc;
      GemmArgs h{};
      h.x = G; h.ldx = D; h.packed = packed_mt; h.N = rows; h.K = D; h.n = C; h.out = S;
      rc = launch_gemm<double, EPI_STORE>(h, s);
      if (rc != RUNIA_OK) return rc;
      maha_refine_kernel<TX><<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(x + r0 * D, class_mean, G, mu_p, S, qc,
                                                                          score + r0, rows, D, C);
      rc = runia_check_launch();
      if (rc != RUNIA_OK) return rc;
    }
    return RUNIA_OK;
  }
  const int64_t cap_rows = (int64_t)(workspace_bytes / ((size_t)D * sizeof(double)));
  if (cap_rows < 1) return RUNIA_E_WORKSPACE;
  double* G = reinterpret_cast<double*>(workspace);
  for (int64_t r0 = 0; r0 < N; r0 += cap_rows) {
    const int64_t rows = (N - r0 < cap_rows) ? (N - r0) : cap_rows;
    GemmArgs g{};
    g.x = x + r0 * D; g.ldx = D; g.packed = packed_p; g.N = rows; g.K = D; g.n = D;
    g.sub = nullptr; g.bias = nullptr; g.scale = nullptr; g.out = G;
    int rc = launch_gemm<TX, EPI_STORE>(g, s);
    if (rc != RUNIA_OK) return rc;
    maha_class_kernel<TX><<<runia_stream_grid(rows, 4), 256, 0, s>>>(
        x + r0 * D, class_mean, G, mu_p, score + r0, rows, D, C);
    rc = runia_check_launch();
    if (rc != RUNIA_OK) return rc;
  }
  return RUNIA_OK;
}

extern "C" int runia_mahalanobis_score_f32(const float* x, const float* class_mean, const double* packed_p,
                                           const double* mu_p, double* score, void* workspace,
                                           size_t workspace_bytes, int64_t N, int64_t D, int C,
                                           runia_stream_t stream) {
  return maha_impl<float>(x, class_mean, packed_p, mu_p, score, workspace, workspace_bytes, N, D, C, stream);
}
extern "C" int runia_mahalanobis_score_f64(const double* x, const double* class_mean, const double* packed_p,
                                           const double* mu_p, double* score, void* workspace,
                                           size_t workspace_bytes, int64_t N, int64_t D, int C,
                                           runia_stream_t stream) {
  return maha_impl<double>(x, class_mean, packed_p, mu_p, score, workspace, workspace_bytes, N, D, C, stream);
}
