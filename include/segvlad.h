/*
 * segvlad.h -- C-ABI of the MI355X-native SegVLAD hot path (libsegvlad_hip.so, gfx950).
 *
 * The reference (AnyLoc/Revisit-Anything, all Python) exposes no FFI; the boundary it offers is
 * the function surface of func_vpr.py / place_rec_main.py.  Each entry point below replaces the
 * cited reference function(s); the Python module revisit_anything_amd.func_vpr maps the reference's
 * names/arguments onto these calls (INTEGRATION.md shows the ctypes stub a maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types.  Bulk data pointers may be DEVICE or HOST pointers
 *     (hipPointerGetAttributes decides; host data is staged through a context-owned buffer).
 *     Pointers documented "host" must be host memory (small launch-geometry metadata).
 *   - every function returns 0 (SEGVLAD_OK) or a negative error code; the message is available from
 *     segvlad_last_error(ctx).  No exception crosses the ABI.
 *   - one context per (device, stream).  A context is not re-entrant; distinct contexts may be
 *     driven from distinct host threads.  Work is enqueued on the context's HIP stream
 *     (segvlad_set_stream) and is asynchronous unless an output pointer is host memory.
 *   - results are freshly written into caller-owned buffers; inputs are never modified.
 *   - arithmetic: fp32 on device (the reference's fp64 aggregation is matched to <=1e-6 cosine);
 *     integer/bit outputs (incidence, labels away from audited ties, kNN ids away from ties,
 *     vote order) are exact.
 */
#ifndef SEGVLAD_H
#define SEGVLAD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct segvlad_ctx segvlad_ctx;

enum {
  SEGVLAD_OK = 0,
  SEGVLAD_ERR_ARG = -1,    /* bad argument (null pointer, negative size, unsupported shape)   */
  SEGVLAD_ERR_HIP = -2,    /* a HIP runtime call failed                                        */
  SEGVLAD_ERR_STATE = -3,  /* call order (e.g. segvlad_images before segvlad_set_vocab)        */
  SEGVLAD_ERR_LIMIT = -4,  /* documented implementation limit exceeded                        */
  SEGVLAD_ERR_NOMEM = -5,
  SEGVLAD_ERR_COMM = -6    /* RCCL: library not found, or a collective / communicator call failed */
};

#define SEGVLAD_VOTE_WT_BORDA_IM 0 /* get_matches(method="max_seg_topk_wt_borda_Im"), func_vpr.py:207-224 */
#define SEGVLAD_VOTE_COUNT 1       /* get_matches(method="max_seg_topk"),            func_vpr.py:118-125 */

/* ---- lifetime ------------------------------------------------------------------------------ */
int segvlad_version(void);
int segvlad_create(segvlad_ctx** out, int device_id);
int segvlad_destroy(segvlad_ctx* ctx);
const char* segvlad_last_error(const segvlad_ctx* ctx);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = default stream */
int segvlad_set_stream(segvlad_ctx* ctx, void* hip_stream);
int segvlad_synchronize(segvlad_ctx* ctx);

/* ---- vocabulary: torch.load(c_centers.pt) + F.normalize(c_centers)   place_rec_main.py:149-154,
 *      func_vpr.py:1145.  C is [K][D] fp32; both C and its row-normalised copy stay on device.   */
int segvlad_set_vocab(segvlad_ctx* ctx, const float* C, int K, int D);

/* ---- mask -> token incidence: nearest-upsample + argwhere + scatter  func_vpr.py:1088-1092 with
 *      the pixel->token map of place_rec_main.py:187-194 folded in.
 *      masks [S][Hm][Wm] bytes (non-zero = true); inc_bits [S][ceil(N/64)] u64, bit (t%64) of word
 *      (t/64) = token t covered; N = (H/patch)*(W/patch).  S may span a whole batch of images.   */
int segvlad_incidence(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                      uint64_t* inc_bits);

/*      Fused variant: one pass over the mask bytes yields the incidence rows AND the centroids of
 *      func_vpr.py:1314 (see segvlad_mask_centroids) -- what the batched pipeline calls.            */
int segvlad_incidence_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                                uint64_t* inc_bits, double* centroids);

/* ---- mask centroids: np.nonzero(mask).mean(1)[::-1]                  func_vpr.py:1314
 *      centroids [S][2] fp64 (x, y); an empty mask yields NaN (the reference raises ValueError).  */
int segvlad_mask_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, double* centroids);

/* ---- neighbourhood adjacency for a batch: getNbrsDelaunay + nbrMasksAGGFastSingle
 *      func_vpr.py:1241-1245, 1315-1345.  centroids [S_tot][2] fp64 (x, y) from segvlad_mask_centroids,
 *      seg_offsets [B+1] int32 HOST, order >= 1.  adj_out: concatenated per-image [S_b][S_b] byte
 *      matrices = (A1^order > 0), A1 = Delaunay neighbours + self loop; images with S_b <= 3 get the
 *      reference's special rows e0(+e1).  Computed on the device (empty-circle test per point pair;
 *      identical to Qhull for points in general position).  n_empty_out (HOST, may be NULL; passing it
 *      synchronises): low 16 bits = number of NaN centroids (= empty masks, for which the reference raises
 *      ValueError); bits 16.. = number of images holding a NON-GENERIC configuration (a duplicate centroid, or
 *      four centroids co-circular -- exactly or to within rounding -- with an empty circle), where the Delaunay triangulation is not unique
 *      and Qhull's choice cannot be reproduced: callers that need the reference's result bit for bit recompute
 *      such batches with Qhull (pipeline.py does). */
int segvlad_adjacency(segvlad_ctx* ctx, const double* centroids, const int32_t* seg_offsets, int B, int order,
                      uint8_t* adj_out, uint32_t* n_empty_out);

/*      The same, with one flag byte per image instead of the two counts: img_flags_out [B] (HOST or device; a host
 *      pointer synchronises): bit 0 = the image holds an empty mask (NaN centroid), bit 1 = it holds a non-generic
 *      centroid configuration (duplicate centroids, or four centroids co-circular exactly or to within rounding).
 *      A caller that needs the reference's result bit for bit recomputes exactly the flagged images with Qhull
 *      (pipeline.py does: ~0.2 ms of host work per flagged image instead of the whole batch). */
int segvlad_adjacency_flagged(segvlad_ctx* ctx, const double* centroids, const int32_t* seg_offsets, int B, int order,
                              uint8_t* adj_out, uint8_t* img_flags_out);

/* ---- segment VLAD for a batch of B images of identical token geometry
 *      seg_vlad_gpu_single(_img) -> vlad_single -> vlad_matmuls_per_cluster   func_vpr.py:1065-1210
 *      tokens      [B][D][N] fp32, as stored by the reference (D-major, N contiguous)
 *      inc_bits    [S_tot][ceil(N/64)] u64 from segvlad_incidence
 *      seg_offsets [B+1] int32 HOST: segments of image b are rows seg_offsets[b]..seg_offsets[b+1]
 *      adj         concatenated per-image [S_b][S_b] byte matrices (row-major, non-zero = 1), or
 *                  NULL for order 0 (identity)                                func_vpr.py:1192-1193
 *      out         [S_tot][K*D] fp32, cluster-major, intra-normalised then L2-normalised
 *      labels_out  [B][N] u8 or NULL      (argmax_k <x^, c^_k>, first max)    func_vpr.py:1146
 *      gap_out     [B][N] fp32 or NULL    (top-1 minus top-2 cosine: the tie audit)
 *      block_norms_out [S_tot][K] fp32 or NULL (||V[s,k,:]|| before intra-normalisation)          */
int segvlad_images(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                   const int32_t* seg_offsets, const uint8_t* adj, float* out, uint8_t* labels_out,
                   float* gap_out, float* block_norms_out);

/* ---- fused segvlad_images + segvlad_pca_apply: the per-batch pair of place_rec_main.py:259-270
 *      (seg_vlad_gpu_single per image, then apply_pca_transform_from_pkl on the batch) in one call.  The
 *      aggregation kernel emits the projection GEMM's input planes directly, so the K*D-wide fp32 descriptor is
 *      neither measured (max |x|), nor re-read, nor -- when desc_out is NULL -- written to HBM at all; with
 *      desc_out == NULL the call may not even form it (option "pca_path": the tokens' residuals are projected with
 *      their cluster's slice of the components and the segments aggregated in the P-dimensional space -- the same
 *      linear map in another order, fp32-class like the other form).
 *      y [S_tot][P] fp32 (l2norm != 0: rows normalised as normalizeFeat); desc_out [S_tot][K*D] fp32 or NULL;
 *      other arguments as segvlad_images.  Requires segvlad_set_vocab and segvlad_pca_set (KD == K*D).      */
int segvlad_images_pca(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                       const int32_t* seg_offsets, const uint8_t* adj, float* y, int l2norm, float* desc_out,
                       uint8_t* labels_out, float* gap_out);

/* ---- the describe stage of a batch in ONE call: the per-image chain of place_rec_main.py:244-263 (preload_masks -> masks ->
 *      nbrMasksAGGFastSingle -> seg_vlad_gpu_single) plus, with y != NULL, the per-batch apply_pca_transform_from_pkl (:269).
 *      = segvlad_incidence_centroids + segvlad_adjacency_flagged + segvlad_images / segvlad_images_pca with the same arguments
 *      and the same results; the mask branch runs on a second stream of the context beside the token-assignment pass.
 *      All bulk pointers are DEVICE memory, seg_offsets [B+1] HOST.  The intermediates come back in caller-owned buffers
 *      (inc_bits_out [S_tot][ceil(N/64)], centroids_out [S_tot][2] fp64, adj_out concatenated [S_b][S_b] bytes, img_flags_out [B]:
 *      bit 0 empty mask, bit 1 non-generic centroids -- see segvlad_adjacency_flagged: a caller that needs the reference's result
 *      bit for bit patches the flagged images' adjacency with Qhull and describes again through segvlad_images[_pca]).
 *      desc_out [S_tot][K*D] and / or y [S_tot][P] (one of them may be NULL).                                            */
int segvlad_describe(segvlad_ctx* ctx, const uint8_t* masks, int Hm, int Wm, int H, int W, int patch, const float* tokens, int B, int N,
                     const int32_t* seg_offsets, int order, uint64_t* inc_bits_out, double* centroids_out, uint8_t* adj_out,
                     uint8_t* img_flags_out, float* desc_out, float* y, int l2norm);
/*      The same stage as THREE calls, for a caller that patches flagged images with Qhull WHILE the device works:
 *      segvlad_describe_begin   enqueues the mask branch (side stream; its flags and centroids are also copied to pinned host
 *                               memory behind it) and the assignment pass (the context's stream), and returns.  pca != 0: the
 *                               projected form will be asked of segvlad_describe_end (checked here).
 *      segvlad_describe_flags   waits for the MASK BRANCH only -- the assignment pass keeps the device busy -- and hands the
 *                               per-image flags [B] (and, if asked for, the centroids [S_tot][2]) to HOST buffers.
 *      segvlad_describe_end     n_patch adjacency blocks recomputed by the caller (patch_images [n_patch] HOST, ascending image
 *                               indices; patch_blocks HOST: their [S_b][S_b] byte matrices, concatenated) are written over the
 *                               device's, then prep -> aggregation (-> projection) run: the result is that of segvlad_images[_pca]
 *                               with the patched adjacency.  Same tokens / inc_bits / seg_offsets / adj buffers as in begin.
 *      A begin must be followed by an end (flags is optional) before any other describe / incidence / adjacency / images /
 *      cluster_aggregate call on the context: those return SEGVLAD_ERR_STATE while a begin is open (round 6; the rule used to
 *      be documented only).  segvlad_describe_begin and segvlad_describe return SEGVLAD_ERR_LIMIT -- "this entry point cannot,
 *      the separate ones can" -- for an empty batch (B == 0 or no segments), for a PCA model without the fp16x3 form
 *      (option pca_arith=fp32, or K*D not a multiple of 32) and for an image with more segments than the in-LDS Delaunay holds.   */
int segvlad_describe_begin(segvlad_ctx* ctx, const uint8_t* masks, int Hm, int Wm, int H, int W, int patch, const float* tokens, int B,
                           int N, const int32_t* seg_offsets, int order, uint64_t* inc_bits_out, double* centroids_out, uint8_t* adj_out,
                           uint8_t* img_flags_out, int pca);
int segvlad_describe_flags(segvlad_ctx* ctx, uint8_t* flags_host, double* centroids_host);
int segvlad_describe_end(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits, const int32_t* seg_offsets,
                         uint8_t* adj, int n_patch, const int32_t* patch_images, const uint8_t* patch_blocks, float* desc_out, float* y,
                         int l2norm);

/* ---- vocabulary k-means, one Lloyd half-step over a batch of images: the fit that writes c_centers.pt
 *      (vlad_c_centers_pt_gen.py:86-158 -> utilities.py:749-791 VLAD.fit -> fast_pytorch_kmeans.KMeans(mode='cosine').fit).
 *      With the CURRENT centres in the context (segvlad_set_vocab): every token is assigned to the centre of largest cosine
 *      (normalised token against normalised centre, first maximum -- the assignment kernel of segvlad_images), and
 *        sums [K][D]  fp64 += sum of the NORMALISED tokens assigned to each centre      (DEVICE memory, accumulated into)
 *        counts [K]   int64 += number of tokens assigned to each centre                 (DEVICE memory, accumulated into)
 *      so that a caller walks its token set in batches, then sets centre k to sums[k] / counts[k] (centres that lost all their
 *      tokens keep their value; the means are NOT re-normalised, utilities.py:749-791).  tokens [B][D][N] fp32 as
 *      segvlad_images takes them; labels_out [B][N] bytes or NULL.  Deterministic (fixed summation order, no atomics on the
 *      sums).  Round 6: replaces the round-3 recovery of the sums from normalised VLAD descriptors.                              */
int segvlad_kmeans_step(segvlad_ctx* ctx, const float* tokens, int B, int N, double* sums, int64_t* counts, uint8_t* labels_out);

/* ---- the K-parametric entry: vlad_matmuls_per_cluster(num_c, masks, res, clus_labels, adjMat)
 *      func_vpr.py:1181-1210.  res [N][D] fp32 residuals (token-major, as the reference passes them),
 *      labels [N] u8 (< num_c <= 256), inc_bits [S][ceil(N/64)], adj [S][S] bytes or NULL,
 *      out [S][num_c*D] fp32.  Independent of the vocabulary held by the context.                  */
int segvlad_cluster_aggregate(segvlad_ctx* ctx, int num_c, const float* res, const uint8_t* labels, int N, int D,
                              const uint64_t* inc_bits, int S, const uint8_t* adj, float* out);

/* ---- PCA apply: pickle.load + PCA.transform                          func_vpr.py:1419-1443
 *      Y = ((X - mean) @ comps^T) / sqrt(expl_var) when whiten!=0.  comps [P][KD] fp32.          */
int segvlad_pca_set(segvlad_ctx* ctx, const float* mean, const float* comps, const float* expl_var, int P, int KD,
                    int whiten);
/*      X [n][KD] -> Y [n][P]; l2norm!=0 additionally applies normalizeFeat (func_vpr.py:1673-1676) */
int segvlad_pca_apply(segvlad_ctx* ctx, const float* X, int n, float* Y, int l2norm);

/* ---- row L2 normalisation: normalizeFeat                             func_vpr.py:1673-1676
 *      (no epsilon: an all-zero row becomes NaN exactly like the reference).  X may equal Y.      */
int segvlad_normalize_rows(segvlad_ctx* ctx, const float* X, int n, int d, float* Y);

/* ---- exact kNN: faiss.IndexFlatL2(d).add / .search                   place_rec_main.py:53-60
 *      The index keeps a device copy of the rows.  img_of_seg (imIndsRef / imInds1,
 *      place_rec_main.py:252) may be NULL if segvlad_vote is given the map explicitly.            */
int segvlad_db_reset(segvlad_ctx* ctx);
int segvlad_db_add(segvlad_ctx* ctx, const float* R, int n, int d, const int32_t* img_of_seg);
int segvlad_db_size(segvlad_ctx* ctx, int64_t* n_rows, int* d);
/*      d2_out [nq][k] fp32 ascending squared L2; idx_out [nq][k] int64 (ties -> lower id; slots
 *      beyond the database size hold (+inf, -1) like faiss).  1 <= k <= 1024.                      */
int segvlad_search(segvlad_ctx* ctx, const float* Q, int nq, int k, float* d2_out, int64_t* idx_out);

/* ---- merge of per-shard top-k lists (no reference counterpart: the reference is single-process).
 *      d2_parts/idx_parts [nq][parts*k] (shard-major within a row, global ids); output top-k by
 *      (distance, lower id).                                                                       */
int segvlad_merge_topk(segvlad_ctx* ctx, const float* d2_parts, const int64_t* idx_parts, int nq, int parts, int k,
                       float* d2_out, int64_t* idx_out);

/* ---- top-50 slice + "2 - d^2":  sims_50 = 2 - sims[:, :50]           place_rec_main.py:78-81     */
int segvlad_sims_from_d2(segvlad_ctx* ctx, const float* d2, const int64_t* idx, int nq, int k_in, int k_keep,
                         float* sims_out, int64_t* idx_out);

/* ---- global min / max of the kept similarities                        func_vpr.py:212-213
 *      minmax_out [2] fp32 = {min, max}.                                                          */
int segvlad_minmax(segvlad_ctx* ctx, const float* sims, int64_t count, float* minmax_out);

/* ---- image vote: get_matches + weighted_borda_count                   func_vpr.py:61-77, 207-224
 *      idx [nq][k] int64 reference-segment ids, sims [nq][k] fp32, img_of_seg [n_ref_seg] int32
 *      (NULL = the map given to segvlad_db_add; n_ref_seg is then ignored), qseg_offsets [n_img+1]
 *      int32 HOST.
 *      smin/smax: the GLOBAL extrema (func_vpr.py:212-213); pass NaN to have them computed from
 *      `sims`.  pred_out [n_img][n_top] int32 (-1 padded), score_out [n_img][n_top] fp64 or NULL
 *      (mode COUNT: the integer vote count as a double).  Ties: first appearance (rank-major, then
 *      segment) for WT_BORDA_IM; (count desc, image id asc) for COUNT.                             */
int segvlad_vote(segvlad_ctx* ctx, const int64_t* idx, const float* sims, const int32_t* img_of_seg,
                 int64_t n_ref_seg, const int32_t* qseg_offsets, int n_img, int k, float smin, float smax, int n_top, int mode,
                 int32_t* pred_out, double* score_out);

/* ---- instrumentation: with profiling on, every kernel group of a stage ("incidence", "adjacency",
 *      "assign", "prep", "aggregate", "pca", "describe" (segvlad_describe as a whole: its parts overlap), "knn_level0", "knn_gemm", "knn_select", "knn_fallback", "vote") is bracketed by a HIP event pair
 *      on the context stream.  segvlad_stage_ms returns the SUM of the elapsed times (ms) and the number
 *      of kernel launches recorded for the stage since the last segvlad_profile_reset; it returns
 *      SEGVLAD_ERR_STATE if the stage has not run.  Replaces the (discarded) time.time() pair of
 *      vlad_matmuls_per_cluster, func_vpr.py:1185,1207-1210.                                          */
int segvlad_set_profiling(segvlad_ctx* ctx, int on);
int segvlad_profile_reset(segvlad_ctx* ctx);
int segvlad_stage_ms(segvlad_ctx* ctx, const char* stage, float* ms_out, int* launches_out);

/* ---- switches (no reference counterpart: the reference has one arithmetic, fp32/fp64 torch + faiss).
 *      Read ONCE: the environment variables SEGVLAD_KNN_FILTER / SEGVLAD_KNN_FP32 / SEGVLAD_PCA_FP32 / SEGVLAD_PCA_PATH /
 *      SEGVLAD_KNN_HEURISTIC / SEGVLAD_SEARCH_STATS give a context its defaults at segvlad_create; this call overrides them.
 *      No kNN switch changes a result (every filter is followed by the exact fp32 refinement: tests assert
 *      bit-equality); the PCA variants are all fp32-class and agree to ~1e-5 relative (tests hold each to the same
 *      oracle tolerance), not bit for bit.
 *        "knn_filter"   auto | f16 | bf16x3 | fp32     arithmetic of the candidate filter GEMM
 *        "pca_arith"    auto | f16x3 | fp32            projection GEMM arithmetic
 *        "search_stats" 0 | 1                          record list occupancies (segvlad_search_stats)
 *        "knn_heuristic" 1 | 0                         low-rank, a-posteriori verified level thresholds (5-10x fewer
 *                                                      candidates per level) | rigorous k-th-rank thresholds only
 *        "pca_path"      auto | planes | project       form of segvlad_images_pca when no descriptor output is asked
 *                                                      for: "planes" projects the finished K*D-wide descriptors,
 *                                                      "project" projects every token's residual with its cluster's
 *                                                      slice of the components and aggregates the segments in the
 *                                                      P-dimensional space (same fp32-class result, N*D*P instead of
 *                                                      S*K*D*P flops per image); auto = the smaller product, decided PER
 *                                                      CALL from the batch (tokens per cluster, segments): the same
 *                                                      image can therefore come out ~1e-5 relative apart in two batches
 *                                                      of different size -- fix the form when runs must be comparable
 *                                                      digit for digit (bench.py does)
 *        "small_plan"    1 | 0                         searches of <= 128 query rows (ONE query image per pass: the
 *                                                      HBM-bound regime of SURVEY 8d) take their own plan -- one filter
 *                                                      level behind an exact sample of 2048..4096 rows, a workgroup per
 *                                                      list in the selects, refinement lists shared by workgroups, the
 *                                                      query scale left on the device | the plan of the batches
 *        "query_group"   0 | 1 .. 64                   a HINT, never a result: the query rows of the coming batches are the
 *                                                      segments of query images, this many consecutive rows per image (0 =
 *                                                      unknown).  The exact level of a batch search re-evaluates the refine
 *                                                      bands of a group of rows -- an image's rows with the hint, 32-row
 *                                                      blocks without -- as ONE fp32 GEMM over the union of their database
 *                                                      rows when that is the cheaper way (the segments of an image share most
 *                                                      of their neighbours: a cost model with measured constants decides per
 *                                                      group), row by row otherwise
 *        "refine_group"  1 | 0 | 2                     that grouped refinement | every row on its own (rounds 1-4) | every group
 *                                                      whose union fits, whatever the cost model says (tests); same bits
 *      Every other key is a DEVELOPMENT switch (kernel tuning, A/B variants, debugging, the tests' own hooks), documented in
 *      revisit-anything_amd/csrc/segvlad_dev.h and NOT part of this ABI: none changes a result, the shipped library holds
 *      only the kernels they default to and rejects the values that select another (SEGVLAD_ERR_ARG).
 *
 *      Environment, read ONCE by segvlad_create: SEGVLAD_GUARD=1 creates a GUARDED context (development / test runs):
 *      every device buffer of the context is allocated at its exact size between two fences of poison words, the back
 *      fence of a per-call scratch buffer right behind the bytes of the CURRENT request, and every call -- and
 *      segvlad_synchronize -- ends with a check of all fences: an out-of-bounds write fails the call with
 *      SEGVLAD_ERR_STATE and names the buffer.  (The library's scratch only ever grows; without the guard a request that
 *      an earlier, larger one already covers cannot fail.)  Results are unchanged; calls synchronise the device. */
int segvlad_set_option(segvlad_ctx* ctx, const char* key, const char* value);

/* ---- statistics of the last segvlad_search on this context (HOST array, up to 13 values):
 *      [0] filter levels run after the sampled exact level (0 = distance-matrix path)
 *      [1] filter arithmetic used (0 none, 1 f16, 2 bf16x3, 3 fp32)
 *      [2] query rows whose candidate list overflowed and that were redone, as one dense batch, on the
 *          exact distance-matrix path (the other rows keep their filtered result)
 *      [3] max and [4] sum of the last level's candidate-list lengths   (option search_stats = 1)
 *      [5] max and [6] sum of the exact-refinement list lengths         (option search_stats = 1)
 *      [7] number of query rows
 *      [8] query rows whose low-rank ("heuristic") level thresholds did not verify and that were redone
 *          with the rigorous k-th-rank thresholds (option knn_heuristic = 0 disables the low-rank thresholds)
 *      [9] query rows whose refine band held more rows than the first-tier list (512) and that were refined
 *          from their whole candidate list instead (second tier; temporally redundant databases)
 *      [10] groups of 32 consecutive query rows whose refine bands overlapped enough to be evaluated as one exact fp32 GEMM over
 *           the union of their rows, and [11] the sum of those unions' lengths            (option search_stats = 1)
 *      [12] database rows the last filter level did not evaluate again: the rows of the stride-16 level, whose survivors stayed in
 *           the candidate lists (0: every level started from empty lists)                                                         */
int segvlad_search_stats(segvlad_ctx* ctx, int64_t* stats_out, int n);

/* ---- row-sharded index over the GPUs of a node: one process (and one context) per GPU.
 *      No reference counterpart -- the reference is single-process (place_rec_main.py:53-60 builds ONE faiss index);
 *      this is SURVEY 8b/8e's own layer: rank r adds the rows [base_r, base_r + n_r) of the reference-segment matrix to
 *      ITS context (segvlad_db_add), every rank searches the full query batch against its shard, the per-shard top-k
 *      lists travel in ONE all-gather of packed 12-byte records {fp32 distance bits, int64 global id} over RCCL (xGMI
 *      inside a node), and every rank merges them -- by distance, ties by lower global id: bit for bit what one index
 *      over all rows returns.  RCCL is bound at run time (dlopen; a copy already in the process -- PyTorch-ROCm's -- is
 *      shared): without it these calls return SEGVLAD_ERR_COMM and everything else keeps working.
 *
 *      segvlad_comm_unique_id   rank 0 draws the 128-byte id (HOST); it reaches the other ranks by any host channel
 *                               (MPI, a file, torch.distributed's store ...)
 *      segvlad_comm_init        collective over all ranks: binds an RCCL communicator to the context (its stream carries
 *                               the collectives); world = 1 is valid
 *      segvlad_comm_info        rank / world of the context's communicator (world 0 = none) and which RCCL was bound
 *      segvlad_allgather_rows   [n_local][d] fp32 of every rank -> [world * n_local][d], rank order (equal slices): the
 *                               query descriptors when every rank describes a slice of the query images
 *      segvlad_search_sharded   global exact top-k (ascending (d2, id); (inf, -1) beyond the total row count), identical
 *                               on every rank.  id_base = global index of this shard's first row.  Collective: every
 *                               rank enters the all-gather or none does -- a rank whose LOCAL search fails still
 *                               contributes ((inf, -1) records and its status in a trailer record), and then EVERY rank
 *                               returns an error (the failing rank its own, the others SEGVLAD_ERR_COMM naming it); a
 *                               rank that cannot join any more (no memory for the exchange buffers) aborts the
 *                               communicator (ncclCommAbort) instead of leaving its peers waiting.
 *      segvlad_allgather_rows   n_local must be the same on every rank. */
#define SEGVLAD_COMM_ID_BYTES 128
int segvlad_comm_unique_id(void* id_out);
int segvlad_comm_init(segvlad_ctx* ctx, const void* id, int rank, int world);
int segvlad_comm_destroy(segvlad_ctx* ctx);
int segvlad_comm_info(segvlad_ctx* ctx, int* rank_out, int* world_out, char* origin_out, int origin_len);
int segvlad_allgather_rows(segvlad_ctx* ctx, const float* local_rows, int n_local, int d, float* all_rows);
int segvlad_search_sharded(segvlad_ctx* ctx, const float* Q, int nq, int k, int64_t id_base, float* d2_out, int64_t* idx_out);

#ifdef __cplusplus
}
#endif
#endif /* SEGVLAD_H */
