/*
 * runia_hip.h — C ABI of the MI355X (gfx950) scoring library `librunia_hip.so`.
 *
 * This is the drop-in boundary for the post-hoc OOD scoring hot path of
 * CEA-LIST/runia_core (reference paths below are relative to
 * /root/reference/runia_core/).  The reference is pure Python: each entry point
 * replaces one host numerical call (NumPy / SciPy / scikit-learn / faiss /
 * entropy_estimators) that the reference makes on this path, and is what a
 * `ctypes` stub in the reference would bind.
 *
 * Conventions (last row):
 *   - every pointer is a DEVICE pointer (HBM) unless the name says `host`;
 *   - matrices are row-major and dense unless a leading dimension is given;
 *   - `stream` is a hipStream_t passed as void*; calls are stream-ordered,
 *     never synchronise, never allocate, and are re-entrant;
 *   - return value: 0 on success, negative RUNIA_E_* code otherwise
 *     (`runia_error_string` gives the text).
 */
#ifndef RUNIA_HIP_H
#define RUNIA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RUNIA_OK 0
#define RUNIA_E_INVALID (-1)   /* bad shape / null pointer / unsupported size */
#define RUNIA_E_LAUNCH (-2)    /* hipGetLastError() after the launch was not hipSuccess */
#define RUNIA_E_NODEVICE (-3)  /* no HIP device visible */
#define RUNIA_E_WORKSPACE (-4) /* caller-provided workspace too small */

typedef void* runia_stream_t;

/* ---- library ------------------------------------------------------------ */
int runia_abi_version(void);
const char* runia_error_string(int code);
/* number of visible HIP devices (0 when none; never fails) */
int runia_device_count(void);
/* Clock reading for measurement records (no counterpart upstream: bench.py brackets its timed regions with it).  One wave
 * runs `chain` (16 .. 2^24, rounded up to 16) dependent v_fma_f32 between two readings of the shader-clock counter
 * (s_memtime) and of the constant 100 MHz counter (s_memrealtime):
 *   out4[0] = shader-clock ticks, out4[1] = 100 MHz ticks, out4[2] = FMAs executed, out4[3] unused.
 * Clock held = out4[0] / (out4[1] * 10 ns). */
int runia_clock_probe(uint64_t* out4, int chain, runia_stream_t stream);
/* Kernel-only timing of ONE launch (measurement records): the next call of this thread to an entry point with a timed launch
 * site (runia_mc_entropy_from_table_f32: the dominant kernel of the headline step) attaches the two hipEvent_t to its dispatch
 * - they then carry the kernel's own start and end timestamps, as rocprofv3's kernel trace does, without the dispatch gap an
 * event pair recorded around the launch includes.  Both events must exist (have been recorded once); (NULL, NULL) clears. */
int runia_time_next_launch(void* start_event, void* stop_event);

/* ---- a1  MC-dropout latent stacking ------------------------------------- *
 * Replaces MCSamplerModule.forward (feature_extraction/abstract_classes.py:81-101)
 * = n_mc x DropBlock2D (dropblock==0.3.0) + get_mean_or_fullmean_ls_sample("fullmean")
 * (feature_extraction/utils.py:88-92), for a batch of N latent maps.
 *   x     [N, C, H, W] f32 (NCHW)
 *   rand  uniform draws of the drop layers, [n_mc, H, W] per image; image i reads
 *         rand + i*rand_image_stride (stride 0 = the same draws for every image)
 *   out   [N*n_mc, C] f32, image-major (the n_mc rows of one image are contiguous)
 * drop_prob == 0 -> plain full mean replicated n_mc times (DropBlock identity). */
int runia_mc_stack_f32(const float* x, const float* rand, int64_t rand_image_stride, float* out,
                       int64_t N, int C, int H, int W, int n_mc, double drop_prob, int block_size,
                       runia_stream_t stream);

/* layer_type "FC" / "RPN" of the same module (feature_extraction/abstract_classes.py:95-99: no fullmean, each drop
 * layer's output is flattened):  out [N*n_mc, C*H*W] f32 = ((x * bm) * numel) / sum(bm), image-major. */
int runia_mc_drop_flat_f32(const float* x, const float* rand, int64_t rand_image_stride, float* out,
                           int64_t N, int C, int H, int W, int n_mc, double drop_prob, int block_size,
                           runia_stream_t stream);

/* Reductions of dropped activation maps for the other options of FastMCDSamplesExtractor
 * (feature_extraction/image_level.py:205-236 through feature_extraction/utils.py:70-92, 113-126):
 *   x [maps, H, W] f32 (e.g. the output of runia_mc_drop_flat_f32 seen as N*n_mc*C maps)
 *   mode 0: reduction_method="mean" = torch.mean(dim=3)            -> out [maps, H]
 *   mode 1: return_stds = torch.std(torch.std(., dim=3), dim=2)    -> out [maps]  (unbiased; NaN when H or W is 1) */
int runia_map_reduce_f32(const float* x, float* out, int64_t maps, int H, int W, int mode, runia_stream_t stream);

/* ---- a2  Kozachenko-Leonenko kNN entropy --------------------------------- *
 * Replaces the loops of get_dl_h_z / single_image_entropy_calculation
 * (evaluation/entropy.py:20-93) over entropy_estimators.continuous.get_h(col, k,
 * norm="max", min_dist).
 *   z  [N*n_mc, D] f32 image-major;  h  [N, D] f64;  h_mvn [N] f64
 * 2 <= n_mc <= 64, 1 <= k < n_mc. */
int runia_kl_entropy_per_dim_f32(const float* z, double* h, int64_t N, int n_mc, int64_t D, int k,
                                 double min_dist, runia_stream_t stream);
int runia_kl_entropy_joint_f32(const float* z, double* h_mvn, int64_t N, int n_mc, int64_t D, int k,
                               double min_dist, runia_stream_t stream);
/* Both outputs of one get_dl_h_z call (evaluation/entropy.py:67-84: the joint entropy AND the per-dimension entropies of every
 * image) from ONE pass over z: h_mvn [N] f64 and h [N, D] f64 carry the bits of the two entry points above.  The single-read
 * kernel covers 5 <= n_mc <= 32 with k = 5 (n_mc <= 8 also k = 4) on rows of whole 16-byte (n_mc <= 16) / 8-byte vectors;
 * runia_kl_entropy_both_fused tells (1 / 0); other shapes run the two kernels one after the other inside the same call. */
int runia_kl_entropy_both_fused(int n_mc, int64_t D, int k);
int runia_kl_entropy_both_f32(const float* z, double* h_mvn, double* h, int64_t N, int n_mc, int64_t D, int k,
                              double min_dist, runia_stream_t stream);

/* ---- dense f64 weights, packed once at setup ------------------------------ *
 * The f64 contractions (PCA projection, quadratic forms) read their constant
 * right-hand matrix B [K, n] (row-major, ld = ldb) from a fragment-ordered copy
 * so that every MFMA operand load is one coalesced 512-byte wave access.
 * `runia_packed_weights_bytes` gives the size of that copy. */
size_t runia_packed_weights_bytes(int64_t K, int64_t n);
int runia_pack_weights_f64(const double* B, int64_t ldb, int64_t K, int64_t n, double* packed,
                           runia_stream_t stream);

/* ---- a4  PCA transform ---------------------------------------------------- *
 * Replaces apply_pca_transform (dimensionality_reduction.py:75-87) = sklearn
 * PCA.transform:  Y = X @ C.T - mean @ C.T;  Y /= scale  (when whiten != 0).
 *   x [N, D] (f64 or f32), packed_ct = pack(C.T [D, n]), bias [n] = mean @ C.T,
 *   scale [n] = max(sqrt(explained_variance_), eps), y [N, n] f64. */
int runia_pca_transform_f64(const double* x, const double* packed_ct, const double* bias,
                            const double* scale, double* y, int64_t N, int64_t D, int64_t n,
                            int whiten, runia_stream_t stream);
int runia_pca_transform_f32in(const float* x, const double* packed_ct, const double* bias,
                              const double* scale, double* y, int64_t N, int64_t D, int64_t n,
                              int whiten, runia_stream_t stream);

/* ---- a5  LaREM = MDLatentSpace.postprocess -------------------------------- *
 * Replaces -np.diag(diff @ P @ diff.T) (inference/postprocessors.py:241-242).
 *   x [N, n], mean [n], packed_p = pack(P [n, n]), score [N] f64.
 * `diff` follows NumPy's dtype rules: f32 - f32 is rounded to f32 before the f64
 * quadratic form (the reference's own unit test feeds f32 features); every other
 * combination subtracts in f64 (an f32 mean against f64 rows is widened by the caller). */
int runia_md_score_f64(const double* x, const double* mean, const double* packed_p, double* score,
                       int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_f32(const float* x, const float* mean, const double* packed_p, double* score,
                       int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_f32x_f64mean(const float* x, const double* mean, const double* packed_p,
                                double* score, int64_t N, int64_t n, runia_stream_t stream);
/* runia_md_score_tril_*: the same score from the TRIANGULAR factor of the precision (round 6): precision = W^T W with W lower
 *   triangular (e.g. the reversed Cholesky factor, see runia_cholesky_f64), packed_wt = runia_pack_weights_f64 of W^T [n, n]:
 *   score = -|| W (x - mean) ||^2 = -(x - mean) precision (x - mean)^T (inference/postprocessors.py:241-242) with only the
 *   k <= column part of every 256-column block multiplied - n^2 + 256 n multiply-adds per row instead of 2 n^2.  Same dtype
 *   combinations and centring rules as runia_md_score_*.  The caller keeps runia_md_score_* for a precision that has no such
 *   factor (rank-deficient pinvh).  workspace (optional): runia_md_score_workspace_bytes(N, n) bytes, as runia_md_score_ws_*
 *   (few rows of wide features: column blocks on separate workgroups + a replay launch, same bits). */
int runia_md_score_tril_f64(const double* x, const double* mean, const double* packed_wt, double* score, void* workspace,
                            size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_tril_f32(const float* x, const float* mean, const double* packed_wt, double* score, void* workspace,
                            size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_tril_f32x_f64mean(const float* x, const double* mean, const double* packed_wt, double* score,
                                     void* workspace, size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);
/* The same scores (bit for bit) with a workspace (round 4): for few rows of wide features - MD on un-reduced 2048-d features,
 * one image at a time - the 256-column blocks of a 16-row tile go to separate workgroups and a second launch adds their
 * products in the one-launch kernel's order (0.9 -> 0.1 ms at <= 512 rows x 2048).  runia_md_score_workspace_bytes returns
 * 0 where the one launch is taken anyway (then NULL / 0 may be passed). */
size_t runia_md_score_workspace_bytes(int64_t N, int64_t n);
int runia_md_score_ws_f64(const double* x, const double* mean, const double* packed_p, double* score, void* workspace,
                          size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_ws_f32(const float* x, const float* mean, const double* packed_p, double* score, void* workspace,
                          size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);
int runia_md_score_ws_f32x_f64mean(const float* x, const double* mean, const double* packed_p, double* score,
                                   void* workspace, size_t workspace_bytes, int64_t N, int64_t n, runia_stream_t stream);

/* ---- a6  class-conditional Mahalanobis ------------------------------------ *
 * Replaces mahalanobis_postprocess (inference/funcs.py:69-102): for each row and
 * class c, t = x - mu_c (in the dtype of the inputs), s_c = -t P t^T in f64,
 * NaN -> -inf, max over classes.
 *   x [N, D] f32 (…_f32) or f64 (…_f64); class_mean [C, D] same dtype as x;
 *   packed_p = pack(P [D, D]); mu_p [C, D] f64 = class_mean @ P; score [N] f64.
 *   workspace: C <= 16: not touched (the class terms come out of the GEMM accumulators).  C > 16: holds G = X P for a
 *   chunk of rows (runia_mahalanobis_workspace_bytes; any size >= one row works, rows are processed in chunks that
 *   fit); with runia_mahalanobis_workspace_bytes_classes(N, D, C) bytes (16-byte aligned) the class terms become a
 *   second contraction on the matrix cores - S = G M^T ranks the classes, the f32-difference formula is evaluated for the
 *   classes within 1e-3 of the best - instead of a loop over all classes per row (same scores; C = 1000: 20x faster). */
size_t runia_mahalanobis_workspace_bytes(int64_t N, int64_t D);
size_t runia_mahalanobis_workspace_bytes_classes(int64_t N, int64_t D, int C);
int runia_mahalanobis_score_f32(const float* x, const float* class_mean, const double* packed_p,
                                const double* mu_p, double* score, void* workspace,
                                size_t workspace_bytes, int64_t N, int64_t D, int C,
                                runia_stream_t stream);
int runia_mahalanobis_score_f64(const double* x, const double* class_mean, const double* packed_p,
                                const double* mu_p, double* score, void* workspace,
                                size_t workspace_bytes, int64_t N, int64_t D, int C,
                                runia_stream_t stream);

/* ---- a7  Energy / MSP ------------------------------------------------------ *
 * Replaces scipy.special.logsumexp(x, axis=1) and np.max(softmax(x, axis=1), axis=1)
 * (inference/postprocessors.py:549, 606).  logits [N, C] f32; outputs [N] f32;
 * either output pointer may be NULL. */
int runia_row_lse_msp_f32(const float* logits, float* lse, float* msp, int64_t N, int64_t C,
                          runia_stream_t stream);

/* ---- a8  kNN ---------------------------------------------------------------- *
 * normalizer (inference/funcs.py:105-115): y = x / (||x||_2 + 1e-10), f32. */
int runia_l2_normalize_f32(const float* x, float* y, int64_t N, int64_t D, runia_stream_t stream);
/* faiss.IndexFlatL2(D).add(bank); search(q, k) -> -D[:, -1]
 * (inference/postprocessors.py:396-397,419,850-851,878).  q [N, D] and bank [M, D]
 * are already normalised f32; score [N] f32 = -(k-th smallest squared L2), or
 * -FLT_MAX when k > M.  `workspace` holds the distance tiles
 * (runia_knn_workspace_bytes) and, for large problems, the bf16 planes of the
 * candidate-distance kernel: with a workspace of the size asked for, the
 * candidates of a large problem come from bf16 piece products
 * (runia_knn_piece_products of them, 0 = the f32 matrix-core kernel); with a
 * smaller workspace always from the f32 kernel.  The score is the exactly
 * re-measured f32 distance on either path (identical bits). */
size_t runia_knn_workspace_bytes(int64_t N, int64_t M, int64_t D, int k);
int runia_knn_piece_products(int64_t N, int64_t M, int64_t D);
/* The same search against a bank prepared once (the index of a deployed postprocessor; faiss does its `add` once, too):
 * `state` (runia_knn_bank_state_bytes, 16-byte aligned) receives |b|^2, their maximum and - for banks the bf16 kernel can
 * take - the bf16 pieces; a call then needs runia_knn_prepared_workspace_bytes of workspace (one chunk of distances,
 * |q|^2, the chunk's pieces) and skips the bank passes.  Same scores, bit for bit, as runia_knn_kth_f32; a state of only
 * (M + 1) floats (rounded up to 256 bytes) or a smaller workspace keeps the f32 kernel. */
size_t runia_knn_bank_state_bytes(int64_t M, int64_t D);
int runia_knn_prepare_bank_f32(const float* bank, void* state, size_t state_bytes, int64_t M, int64_t D,
                               runia_stream_t stream);
size_t runia_knn_prepared_workspace_bytes(int64_t N, int64_t M, int64_t D, int k);
int runia_knn_kth_prepared_f32(const float* q, const float* bank, const void* state, size_t state_bytes, float* score,
                               void* workspace, size_t workspace_bytes, int64_t N, int64_t M, int64_t D, int k,
                               runia_stream_t stream);
int runia_knn_kth_f32(const float* q, const float* bank, float* score, void* workspace,
                      size_t workspace_bytes, int64_t N, int64_t M, int64_t D, int k,
                      runia_stream_t stream);

/* ---- a9  LaRED = KernelDensity.score_samples (gaussian) --------------------- *
 * Replaces DetectorKDE.get_density_scores (inference/postprocessors.py:118-128):
 * logsumexp_i(-|x - t_i|^2 / (2 h^2)) - log(M) - D log(h) - (D/2) log(2 pi).
 *   train [M, D] f64, x [N, D] f64, score [N] f64 */
int runia_kde_score_f64(const double* train, const double* x, double* score, int64_t M, int64_t N,
                        int64_t D, double bandwidth, runia_stream_t stream);
/* runia_kde_score_kernel_f64: the same log-density for every kernel sklearn's KernelDensity offers (DetectorKDE(kernel=...)
 * forwards any, inference/postprocessors.py:78-128): kind 0 gaussian (= runia_kde_score_f64), 1 tophat, 2 epanechnikov,
 * 3 exponential, 4 linear, 5 cosine, with sklearn's normalisation; a query with no training row in range scores -inf. */
int runia_kde_score_kernel_f64(const double* train, const double* x, double* score, int64_t M, int64_t N, int64_t D,
                               double bandwidth, int kind, runia_stream_t stream);
/* The same log-density with the pair distances on the f64 matrix cores, |x - t|^2 = |x|^2 + |t|^2 - 2 x.t (for
 * wide embeddings, D > 64): packed_train_t = runia_pack_weights_f64 of train^T [D, M] and train_sqnorm [M] =
 * runia_row_sqnorm_f64(train), both made once at setup; workspace: N doubles (the query norms).  The logsumexp over
 * the M training rows is kept online in the accumulator lanes; no [N, M] matrix is written. */
int runia_row_sqnorm_f64(const double* x, double* out, int64_t N, int64_t D, runia_stream_t stream);
/* workspace of runia_kde_score_packed_f64: the N query norms, plus room for the column-split units (same bits) that take
 * either a batch of fewer 16-row tiles than the chip has compute units (a fraction of the latency) or, since ABI 4, the rows
 * of a large batch behind its last whole round of 32-row tiles (no last round on part of the chip); N doubles are the minimum */
size_t runia_kde_workspace_bytes(int64_t N, int64_t M);
int runia_kde_score_packed_f64(const double* packed_train_t, const double* train_sqnorm, const double* x,
                               double* score, void* workspace, size_t workspace_bytes, int64_t M, int64_t N,
                               int64_t D, double bandwidth, runia_stream_t stream);

/* ---- a11 fused LaREM row pipeline ------------------------------------------- *
 * LaRExInference.get_score after the backbone (inference/image_level.py:115-119) as two
 * launches per batch.
 * (1) runia_mc_entropy_f32 = MCSamplerModule.forward fused with the per-dimension loop of
 *     get_dl_h_z: latent maps x [N, C, H, W] f32 + DropBlock draws (layout as
 *     runia_mc_stack_f32) -> entropies h [N, C] f64; the MC samples never leave registers.
 *     z_out (optional, may be NULL): [N*n_mc, C] f32 copy of the samples, drop layers in
 *     mask-sum order (entropy is invariant to their order).  Shapes outside
 *     runia_mc_entropy_supported() return RUNIA_E_INVALID: use the two unfused calls.
 *     zero_fill (optional, may be NULL): [N] f64 set to 0.0 by the launch - the accumulator that
 *     runia_proj_sq_accumulate_f64 adds into, cleared here for free instead of by a launch of its own.
 *     workspace: runia_mc_entropy_workspace_bytes(N, H, W, n_mc) bytes of device memory, 16-byte
 *     aligned (the per-image keep-flag table a first small launch derives from the draws;
 *     RUNIA_E_WORKSPACE if missing or short).
 * (2) runia_pca_md_score_f64 = apply_pca_transform + MDLatentSpace.postprocess:
 *     h [N, D] f64 -> score [N] f64; packed_ct/bias/scale as runia_pca_transform_f64
 *     (packed_ct NULL = no PCA, n == D), md_mean [n], packed_p = pack(P [n, n]);
 *     y_out (optional) receives the projected rows [N, n]. */
int runia_mc_entropy_supported(int H, int W, int n_mc, int k);
size_t runia_mc_entropy_workspace_bytes(int64_t N, int H, int W, int n_mc);
/* The two launches of runia_mc_entropy_f32 on their own (same arguments, same workspace): the keep-flag table
 * of the draws (the DropBlock2D mask of feature_extraction/abstract_classes.py:91-96, drop layers sorted by mask
 * sum), then sampler + entropy from that table.  A table may be reused for other latents of the same batch
 * geometry (upstream draws the masks once per forward pass and applies them to every hooked layer). */
int runia_mc_mask_table_f32(const float* rand, int64_t rand_image_stride, void* workspace,
                            size_t workspace_bytes, int64_t N, int H, int W, int n_mc, double drop_prob,
                            int block_size, runia_stream_t stream);
/* runia_mc_stack_table_f32 = runia_mc_stack_f32 (same samples, same order, same bits) on the table path, for the
 * map shapes of runia_mc_entropy_supported(H, W, n_mc, 5); workspace as for runia_mc_entropy_f32. */
int runia_mc_stack_table_f32(const float* x, const float* rand, int64_t rand_image_stride, float* z,
                             void* workspace, size_t workspace_bytes, int64_t N, int C, int H, int W, int n_mc,
                             double drop_prob, int block_size, runia_stream_t stream);
int runia_mc_entropy_from_table_f32(const float* x, const void* workspace, size_t workspace_bytes, double* h,
                                    float* z_out, double* zero_fill, int64_t N, int C, int H, int W, int n_mc,
                                    int k, double min_dist, runia_stream_t stream);
int runia_mc_entropy_f32(const float* x, const float* rand, int64_t rand_image_stride, double* h,
                         float* z_out, double* zero_fill, void* workspace, size_t workspace_bytes, int64_t N,
                         int C, int H, int W, int n_mc, double drop_prob, int block_size, int k, double min_dist,
                         runia_stream_t stream);
/* Throughput mode of the sampler: the DropBlock draws are not read from memory but made
 * inside the keep-flag launch by a counter generator - Philox4x32-10 keyed by `seed`; draw i (= layer*H*W + position)
 * of image g is component (i/64)&3 of philox(counter = (g.lo, g.hi, i%64 + 64*(i/256), 0)), u = (bits >> 8) * 2^-24.
 * Image ids are first_image .. first_image+N-1, so a batch scored whole, in chunks or sharded draws the same masks.
 * runia_mc_draws_f32 writes the same draws out, [N, n_mc, H, W]: feeding them to the `rand` argument of
 * runia_mc_entropy_f32 / runia_mc_stack_f32 gives the same bits as the counter entry points. */
int runia_mc_draws_f32(float* out, int64_t N, int n_mc, int H, int W, uint64_t seed, int64_t first_image,
                       runia_stream_t stream);
/* redraw_dead_layers != 0 (opt-in, ABI 3): a drop layer whose block mask removes the WHOLE map - 0 * numel / 0 = NaN in
 * the reference as well (dropblock==0.3.0 does not guard it) - draws again from the same image's next counter block
 * (fourth Philox counter word = attempt 1, 2, ... up to 16), so that a batched caller gets no NaN score.  Not the
 * reference's semantics (it has no redraw); counter mode is not the reference's random stream to begin with. */
int runia_mc_mask_table_counter_f32(uint64_t seed, int64_t first_image, void* workspace, size_t workspace_bytes,
                                    int64_t N, int H, int W, int n_mc, double drop_prob, int block_size,
                                    int redraw_dead_layers, runia_stream_t stream);
int runia_mc_entropy_counter_f32(const float* x, uint64_t seed, int64_t first_image, double* h, float* z_out,
                                 double* zero_fill, void* workspace, size_t workspace_bytes, int64_t N, int C, int H,
                                 int W, int n_mc, double drop_prob, int block_size, int k, double min_dist,
                                 int redraw_dead_layers, runia_stream_t stream);
int runia_pca_md_score_f64(const double* h, const double* packed_ct, const double* bias,
                           const double* scale, const double* md_mean, const double* packed_p,
                           double* score, double* y_out, int64_t N, int64_t D, int64_t n,
                           runia_stream_t stream);
/* (2') runia_proj_sq_score_f64: the same LaREM score from ONE contraction, score = -|| M h + c ||^2, where the
 *     caller folded PCA transform, centring and the factor W of precision = W^T W into M [r, D] = W diag(1/scale) C
 *     and c [r] = W (-bias/scale - md_mean) at setup (exact algebra, f64): packed_m = pack(M.T [D, r]). */
/*     workspace (optional: NULL / 0 is accepted): runia_proj_sq_workspace_bytes(N) bytes let batches of a few
 *     row tiles per compute unit be cut in column halves (two partial row sums + one small combine launch). */
size_t runia_proj_sq_workspace_bytes(int64_t N);
/*     runia_proj_sq_accumulate_f64: the same score ADDED into `score`, which the caller zeroed earlier in the
 *     stream (runia_mc_entropy_f32's zero_fill): the column halves of a row tile each add their row sums with one
 *     f64 atomic - two addends per row, so the result does not depend on their order - and no combine launch or
 *     workspace is needed.  Bit-identical to runia_proj_sq_score_f64. */
int runia_proj_sq_accumulate_f64(const double* h, const double* packed_m, const double* c, double* score,
                                 int64_t N, int64_t D, int64_t r, runia_stream_t stream);
int runia_proj_sq_score_f64(const double* h, const double* packed_m, const double* c, double* score,
                            void* workspace, size_t workspace_bytes, int64_t N, int64_t D, int64_t r,
                            runia_stream_t stream);

/* ---- f1  setup-time covariance on the device (SURVEY 8f "next #1") ----------- *
 * Replaces np.cov(X.T, bias=1) inside sklearn EmpiricalCovariance.fit
 * (inference/postprocessors.py:217-220, inference/funcs.py:62-66):
 * mean [D] = column means, cov [D, D] = (X - mean)^T (X - mean) / N, f64 (f32 rows are promoted).
 * x [N, D]; workspace from runia_covariance_workspace_bytes. */
size_t runia_covariance_workspace_bytes(int64_t N, int64_t D);
int runia_covariance_f64(const double* x, double* mean, double* cov, void* workspace,
                         size_t workspace_bytes, int64_t N, int64_t D, runia_stream_t stream);
int runia_covariance_f32in(const float* x, double* mean, double* cov, void* workspace,
                           size_t workspace_bytes, int64_t N, int64_t D, runia_stream_t stream);

/* ---- f3  the step in front of the sampler for object-level inference ------------------------ *
 * torchvision.ops.roi_align as the reference calls it (feature_extraction/object_level.py:283-292, 340-349):
 *   input [B, C, H, W] f32, boxes [K, 4] f32 (x1, y1, x2, y2 in image pixels), batch_idx [K] int32 (NULL when B == 1),
 *   out [K, C, PH, PW] f32; bins average sampling_ratio^2 bilinear samples (ceil(roi / pooled) per axis when
 *   sampling_ratio <= 0); aligned != 0 shifts the box by -0.5 pixel.  The output is the (N, C, H, W) input of
 *   runia_mc_entropy_f32 / runia_mc_stack_f32: roi_align -> per-ROI MC DropBlock -> entropy with no host round trip. */
int runia_roi_align_f32(const float* input, const float* boxes, const int* batch_idx, float* out, int64_t K, int64_t B,
                        int C, int H, int W, int PH, int PW, double spatial_scale, int sampling_ratio, int aligned,
                        runia_stream_t stream);
/* roi_align folded into the sampler + entropy launch (round 4; _dropblock_rois_get_entropy, feature_extraction/
 * object_level.py:312-367, as ONE pass from the hooked feature map to the per-ROI entropies - the (K, C, PH, PW) tensor is
 * never written).  runia_nchw_to_nhwc_f32: the hooked map [B, C, H*W] -> [B, H*W, C] once per image batch (channel = lane:
 * every bilinear tap of a wave is one contiguous run).  runia_roi_mc_entropy_f32: feat_nhwc [B, H, W, C], boxes [K, 4] xyxy,
 * batch_idx [K] or NULL (B == 1), draws / workspace as runia_mc_entropy_f32 (workspace:
 * runia_roi_mc_entropy_workspace_bytes: the keep-flag table + 16 bytes per sample row and sample column of every ROI) -> h [K, C] f64,
 * the same bits as runia_roi_align_f32 followed by runia_mc_entropy_f32.  Supported (runia_roi_mc_entropy_supported): the
 * fused sampler's map shapes with sampling_ratio 1 or 2; others take the two calls. */
int runia_nchw_to_nhwc_f32(const float* in, float* out, int64_t B, int C, int64_t HW, runia_stream_t stream);
int runia_roi_mc_entropy_supported(int PH, int PW, int n_mc, int k, int sampling_ratio);
size_t runia_roi_mc_entropy_workspace_bytes(int64_t K, int PH, int PW, int n_mc, int sampling_ratio);
int runia_roi_mc_entropy_f32(const float* feat_nhwc, const float* boxes, const int* batch_idx, const float* rand,
                             int64_t rand_image_stride, double* h, float* z_out, void* workspace, size_t workspace_bytes,
                             int64_t K, int64_t B, int C, int H, int W, int PH, int PW, double spatial_scale,
                             int sampling_ratio, int aligned, int n_mc, double drop_prob, int block_size, int k,
                             double min_dist, runia_stream_t stream);

/* Symmetric eigen-decomposition without a vendor solver: two-sided cyclic Jacobi, f64, parallel ordering.  What
 * scipy.linalg.pinvh (EmpiricalCovariance.fit, inference/postprocessors.py:213-220, inference/funcs.py:52-66), the
 * "covariance_eigh" / "full" PCA fit (dimensionality_reduction.py:70-71) and eigen_score (llm_uncertainty/scores.py:49-66)
 * need.  runia_eigh_init_f64: V = I, matrix norm.  runia_eigh_sweep_f64: ONE sweep in place (A -> J^T A J, V -> V J);
 * `rotations` (device, unsigned) grows by the number of rotations applied - the caller repeats the call until a sweep adds
 * none (typically 8-10 sweeps), then diag(A) holds the eigenvalues and the columns of V the eigenvectors.
 * A, V [n, n] f64 row-major (A symmetric, overwritten); workspace: runia_eigh_workspace_bytes(n), 16-byte aligned. */
size_t runia_eigh_workspace_bytes(int64_t n);
int runia_eigh_init_f64(const double* A, double* V, int64_t n, void* workspace, size_t workspace_bytes,
                        runia_stream_t stream);
int runia_eigh_sweep_f64(double* A, double* V, int64_t n, void* workspace, size_t workspace_bytes,
                         unsigned* rotations, runia_stream_t stream);
/* Blocked form of the same cyclic Jacobi method (csrc/eigh_block.hip): a sweep is (n/32 - 1) steps of two launches -
 * one 64 x 64 sub-problem per pair of 32-column blocks solved in LDS, then A <- R^T A R, V <- V R as 64^3 products on the
 * f64 matrix cores - instead of 2 (n - 1) launches of scalar rotations.  Works on PADDED matrices: N =
 * runia_eigh_block_padded(n) (an even number of whole blocks), A zero outside its n x n corner;
 * runia_eigh_block_init_f64(A, V, N, ...) first (V = I, norm in a fixed summation order), with a workspace of
 * runia_eigh_block_workspace_bytes(n) bytes, 16-byte aligned. */
int64_t runia_eigh_block_padded(int64_t n);
size_t runia_eigh_block_workspace_bytes(int64_t n);
int runia_eigh_block_init_f64(const double* A, double* V, int64_t N, void* workspace, size_t workspace_bytes,
                              runia_stream_t stream);
int runia_eigh_block_sweep_f64(double* A, double* V, int64_t N, void* workspace, size_t workspace_bytes,
                               unsigned* rotations, runia_stream_t stream);
/* C [M, N] = A [M, K] * B  (B [K, N], or B [N, K] read transposed when transpose_b != 0); f64, setup-time sizes. */
int runia_matmul_f64(const double* A, const double* B, double* C, int64_t M, int64_t N, int64_t K, int transpose_b,
                     runia_stream_t stream);
/* f4: Gram form of eigen_score's covariance: G [n, n] = Ec Ec^T / denom with Ec = E - column means, E [n, H] f32
 * (torch.cov(E.T) is Ec^T Ec / (n-1), [H, H] of rank < n: its non-zero spectrum is that of G). */
int runia_centred_gram_f32(const float* E, double* G, int64_t n, int64_t H, double denom, runia_stream_t stream);

/* ---- f2  metrics after the path (SURVEY 8f "next #2") -------------------------------------- *
 * AUROC, FPR@95 and AUPR of in-distribution (positive) against out-of-distribution scores as get_auroc_results
 * computes them (evaluation/metrics.py:37-100: torchmetrics binary auroc / roc / precision_recall_curve +
 * sklearn.metrics.auc): sigmoid in the dtype of the scores when any score is outside [0, 1], descending sort, one
 * curve point per run of equal scores, float32 curve points and trapezoid terms.  Device-resident: radix sort + scans.
 *   ind_scores [n_ind], ood_scores [n_ood] (device), out3 [3] f64 (device) = {auroc, fpr@95, aupr} (float32 values);
 *   workspace: runia_ood_metrics_workspace_bytes(n_ind + n_ood) bytes, 256-byte aligned. */
size_t runia_ood_metrics_workspace_bytes(int64_t n_total);
int runia_ood_metrics_f64(const double* ind_scores, int64_t n_ind, const double* ood_scores, int64_t n_ood,
                          double* out3, void* workspace, size_t workspace_bytes, runia_stream_t stream);
int runia_ood_metrics_f32(const float* ind_scores, int64_t n_ind, const float* ood_scores, int64_t n_ood,
                          double* out3, void* workspace, size_t workspace_bytes, runia_stream_t stream);
/* The same launch sequence, additionally leaving torchmetrics' _binary_clf_curve on the device (what get_auroc_results'
 * roc / precision_recall_curve calls are built from, evaluation/metrics.py:70-81): tps / fps [n_ind + n_ood] u32 =
 * cumulative true / false positives at the end of every run of equal scores, descending score order; *n_points (device,
 * int64) = number of runs.  The caller copies n_points entries to the host and forms the float32 curves there. */
int runia_ood_clf_curve_f64(const double* ind_scores, int64_t n_ind, const double* ood_scores, int64_t n_ood,
                            double* out3, unsigned* tps, unsigned* fps, int64_t* n_points, void* workspace,
                            size_t workspace_bytes, runia_stream_t stream);
int runia_ood_clf_curve_f32(const float* ind_scores, int64_t n_ind, const float* ood_scores, int64_t n_ood,
                            double* out3, unsigned* tps, unsigned* fps, int64_t* n_points, void* workspace,
                            size_t workspace_bytes, runia_stream_t stream);

/* ---- (e) multi-GPU: one-shot all-gather of the score shards (SURVEY section 5's fallback for a latency-bound gather) -- *
 * The path's only exchange is the (N / world,) score shard of every rank to every rank, one per postprocessor call
 * (the reference has no distributed code).  Instead of a ring, every rank WRITES its shard into every peer's
 * receive buffer over xGMI (buffers mapped through HIP IPC) and raises a flag; a second launch waits for the world flags of
 * the step and copies the gathered vector out.  Stream-ordered, no host synchronisation; a wait that never sees a peer
 * gives up after timeout_ms and sets a status word (runia_p2p_status) instead of hanging.
 *   runia_p2p_alloc   : this rank's receive buffer (fine-grained device memory, zeroed), world <= 16
 *   runia_p2p_export  : its 64-byte HIP IPC handle (send it to the peers over any host channel)
 *   runia_p2p_open    : map a peer's buffer from its handle;  runia_p2p_close / runia_p2p_free: undo
 *   runia_p2p_all_gather(local_shard, shard_bytes, out [world * shard_bytes], peer_buffers [world] (HOST array of the
 *                       mapped device pointers, own buffer at index rank), ..., seq = 1, 2, 3, ... (+1 per call on every
 *                       rank; slots alternate, a slot is rewritten two calls later), timeout_ms, stream) */
size_t runia_p2p_buffer_bytes(int world, size_t shard_capacity_bytes);
int runia_p2p_alloc(int world, size_t shard_capacity_bytes, void** buffer);
int runia_p2p_free(void* buffer);
int runia_p2p_export(void* buffer, void* handle64);
int runia_p2p_open(const void* handle64, void** peer_buffer);
int runia_p2p_close(void* peer_buffer);
int runia_p2p_all_gather(const void* local_shard, size_t shard_bytes, void* out, void* const* peer_buffers, int world,
                         int rank, size_t shard_capacity_bytes, uint64_t seq, int timeout_ms, runia_stream_t stream);
int runia_p2p_status(void* buffer, int* status);
/* runia_p2p_debug(1): the launches that follow ASSERT the ordering argument slot reuse rests on - a writer reads, over the
 * link, the step its peer last copied out of the slot and expects exactly seq - 2; a mismatch sets bit 1 (value 2) of the
 * status word.  The acknowledgements themselves are always recorded, so the check may be switched on at any step of a live
 * buffer and by the ranks independently.  Returns the previous setting; off by default. */
int runia_p2p_debug(int on);

/* ---- f4  remaining logits/features postprocessors (SURVEY 8f "next #4") ------- *
 * runia_linear_f32: out [N, C] = min(x, clip_max) @ w.T + bias on the f32 matrix cores - the final linear layer
 *   that ReAct / ASH / DICE re-apply to (transformed) features (inference/postprocessors.py:1193, 1441, 1466;
 *   RouteDICE.forward inference/funcs.py:180-189 with the masked weight).  x [N, D], w [C, D] K-contiguous rows,
 *   bias [C] or NULL, clip_max = +inf for no clipping.  Energy = runia_row_lse_msp_f32 on `out`.
 * runia_ash_s_f32: ASH-S pruning + sharpening of 2-D activations (ash_s_linear_layer, inference/funcs.py:234-261).
 * runia_gen_score_f32: generalized_entropy(softmax(logits), gamma, M) (inference/funcs.py:347-375). */
int runia_linear_f32(const float* x, const float* w, const float* bias, float* out, int64_t N, int64_t D,
                     int64_t C, float clip_max, runia_stream_t stream);
int runia_ash_s_f32(const float* x, float* y, int64_t N, int64_t D, int percentile, runia_stream_t stream);
int runia_gen_score_f32(const float* logits, float* score, int64_t N, int64_t C, int M, double gamma,
                        runia_stream_t stream);
/* runia_gen_entropy_f32: generalized_entropy(probs, gamma, M) on rows that already are probabilities - the reference's
 *   free function (inference/funcs.py:347-375) as its own tests call it (tests/unit_test_baselines.py).
 * runia_mcd_uncertainty_f32: get_predictive_uncertainty_score / get_mcd_pred_uncertainty_score (inference/funcs.py:
 *   430-465, 378-427): logits [N * n_mc, C] f32, the n_mc rows of an image consecutive -> pred_h [N] = H[mean_s softmax],
 *   mi [N] = pred_h - mean_s H[softmax]; probs (optional) [N * n_mc, C] receives the softmax rows (the first value the
 *   dataloader form returns).  One launch: a wave per image with the rows in registers up to C = 4096; wider heads
 *   (ImageNet-21k, LLM vocabularies) a workgroup per image with the rows re-read from L2 (n_mc <= 4096 there).  The GEN entry
 *   points take any C the same way.
 * runia_ash_s_rows_f32: ASH-S for rows of any length - ash_s_conv_layer (inference/funcs.py:194-227) on the flattened
 *   (B, C*H*W) maps and ash_s_linear_layer beyond 4096 features.  y = pruned row * exp(sum / kept sum); `pruned`
 *   (optional, may be x itself: the reference's view + scatter_ prunes its argument in place) = the pruned row.
 *   keep_all_when_k_is_zero: NumPy's x[:, -0:] semantics (linear form) instead of torch.topk(k = 0) (conv form). */
int runia_gen_entropy_f32(const float* probs, float* score, int64_t N, int64_t C, int M, double gamma,
                          runia_stream_t stream);
int runia_mcd_uncertainty_f32(const float* logits, float* probs, float* pred_h, float* mi, int64_t N, int n_mc, int64_t C,
                              runia_stream_t stream);
int runia_ash_s_rows_f32(const float* x, float* y, float* pruned, int64_t N, int64_t D, int percentile,
                         int keep_all_when_k_is_zero, runia_stream_t stream);
/* runia_tril_inverse_f64: inv [batch, D, D] = inverse of the lower-triangular factors tril [batch, D, D] (row-major f64; the strict
 *   upper triangle of inv is zero).  Setup of the class-wise Gaussians of GMMLatentSpace / DDU (inference/postprocessors.py:
 *   426-492, 694-786; torch.distributions.MultivariateNormal keeps scale_tril): the precision of a class is inv^T inv. */
int runia_tril_inverse_f64(const double* tril, double* inv, int64_t batch, int64_t D, runia_stream_t stream);
/* runia_cholesky_*: a [batch, D, D] row-major, in place: the lower triangle of every matrix is replaced by its Cholesky factor L
 *   (a + jitter I = L L^T), the strict upper triangle by zeros; info[b] = 0, or j + 1 when the pivot of column j was not positive
 *   (LAPACK potrf's convention; the matrix is then partly overwritten).  f32: the factorisation inside torch's
 *   MultivariateNormal(covariance_matrix=...) that gmm_fit's jitter ladder retries (inference/funcs.py:310-342); f64: the factor
 *   of a precision matrix for runia_md_score_tril_*.  One fixed summation order: same bits from run to run. */
int runia_cholesky_f32(float* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream);
int runia_cholesky_f64(double* a, int* info, int64_t batch, int64_t D, double jitter, runia_stream_t stream);
/* runia_select_hist_f32: one histogram pass of a radix select over a flat f32 array - hist [2048] (u32, cleared by the call) counts,
 *   among the elements whose order-preserving key k satisfies (k & prefix_mask) == prefix, the digit (k >> shift) & 2047.  Three
 *   passes (shift 21, 10, 0) give an exact order statistic; two neighbouring ones give np.percentile(train.flatten(), p), the
 *   clipping threshold of ReAct / DICE+ReAct (inference/postprocessors.py:1441, 1466).  n < 2^32; NaNs sort above +inf. */
int runia_select_hist_f32(const float* x, unsigned* hist, int64_t n, unsigned prefix, unsigned prefix_mask, int shift,
                          runia_stream_t stream);
/* runia_gmm_log_prob_f32: class-wise Gaussian log densities, all classes in one pass - replaces gmm.log_prob(x[:, None, :]) of
 *   the torch MultivariateNormal that gmm_fit builds (inference/funcs.py:265-344), as called by GMMLatentSpace.postprocess and
 *   DDU.postprocess (inference/postprocessors.py:490-491, 778-779), and the scipy logsumexp that follows it.
 *   x [N, D] f32; means [C, D] f32; w_tril [C, D, D] f32 row-major = L_c^-1, the inverse of torch's scale_tril, LOWER TRIANGULAR
 *   with exact zeros above the diagonal; consts [C] f64 = -D/2 log(2 pi) - sum log diag L_c.
 *   log_prob [N, C] f32 (optional) = consts[c] - 0.5 || w_c (x_n - means_c) ||^2, the difference formed in f32 as torch forms it,
 *   the products on the f32 matrix cores with only the k <= column half of w_c multiplied (D^2 flop per (row, class));
 *   lse [N] f32 (optional) = logsumexp over the classes of the f32 log_prob values (NaN in -> NaN).  At least one output.
 *   workspace: runia_gmm_log_prob_workspace_bytes(N, D, C) bytes, 8-byte aligned (per-column-tile sums of squares, f64); a smaller
 *   one is accepted as long as it holds 128 rows (the rows are then scored in chunks). */
size_t runia_gmm_log_prob_workspace_bytes(int64_t N, int64_t D, int C);
int runia_gmm_log_prob_f32(const float* x, const float* means, const float* w_tril, const double* consts, float* log_prob,
                           float* lse, void* workspace, size_t workspace_bytes, int64_t N, int64_t D, int C,
                           runia_stream_t stream);
/* runia_proj_norm_*: ViM residual norm || (x - u) @ NS ||_2 per row (inference/postprocessors.py:1106):
 *   x [N, D], u [D] (same dtype as x; f32 - f32 is rounded to f32 first, as NumPy), packed_ns = pack(NS [D, n]),
 *   norm [N] f64. */
int runia_proj_norm_f32(const float* x, const float* u, const double* packed_ns, double* norm, int64_t N,
                        int64_t D, int64_t n, runia_stream_t stream);
int runia_proj_norm_f64(const double* x, const double* u, const double* packed_ns, double* norm, int64_t N,
                        int64_t D, int64_t n, runia_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RUNIA_HIP_H */
