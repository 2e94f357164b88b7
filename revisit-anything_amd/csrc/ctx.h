// Context, scratch management and launch declarations shared by the SegVLAD HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/segvlad.h"

#define SV_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      return ctx->fail(SEGVLAD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                       __FILE__, __LINE__);                                                       \
    }                                                                                             \
  } while (0)

#define SV_TRY(expr)          \
  do {                        \
    int _r = (expr);          \
    if (_r != SEGVLAD_OK) return _r; \
  } while (0)

// squared L2 distance from the norms and the dot product; ONE definition so that every path (matrix GEMM,
// filtered GEMM, exact refinement) rounds identically: fma(-2, dot, ||q||^2 + ||r||^2), negative results set to zero --
// faiss's IndexFlatL2 (1.7.3, utils/distances.cpp, exhaustive_L2sqr_blas: `if (dis < 0) dis = 0`, "negative values can occur
// for identical vectors due to roundoff errors") does the same BEFORE its heap sees the value, so an exact duplicate is at
// distance 0 and `2 - d2` (place_rec_main.py:78-81) never exceeds 2 (round 6).  The comparison form keeps a NaN a NaN, as
// faiss's does.  max(0, .) is monotone and 1-Lipschitz: order statistics and the filters' margins |d2~ - d2| <= eps carry
// over to the clamped values unchanged.
#if defined(__HIPCC__)
__device__ __forceinline__ float sv_d2(float q2, float r2, float dot) {
  const float v = __fmaf_rn(-2.f, dot, q2 + r2);
  return v < 0.f ? 0.f : v;
}
#endif

// Layout of the fp16 operand planes of the split projection GEMM (gemm_f16x3_kernel): [row / 128][k / 32][128 rows][32 k],
// the four 16-byte chunks of a row stored at chunk ^ sv_x3_swz(row).  One (128-row, 32-k) block is 8 KiB of CONTIGUOUS
// memory that is, byte for byte, the bank-conflict-free LDS image the MFMA fragment reads expect: the global->LDS DMA
// of a k-tile is a linear copy of whole 128-byte lines.  (Row-major planes made every DMA piece touch 16 half-lines
// 196 KiB apart; each line was fetched twice, one k-tile apart, and the XCD's L2 -- 4 MiB against 4 MiB of such
// half-used lines in flight -- kept nothing for the workgroups sharing a tile: 26 GB fetched for 4.4 GB of operands.)
// Rows are padded to a multiple of 256 (the pad rows hold garbage: they only feed output rows / columns that are never stored).
//
// The swizzle term.  The fragments are read for v_mfma_f32_16x16x32_f16 (round 4: the shape that needs less energy per flop):
// lane l holds row (l & 15) of a 16-row tile and chunk (l >> 4) of the 64-byte row.  A ds_read_b128 is served in four groups of
// 16 lanes, {0-3, 12-15, 20-27} being one: rows 0-3 and 12-15 with chunk c, rows 4-11 with chunk c + 1; bank group (16-byte
// unit mod 16) of row r, physical chunk p = 4 (r & 3) + p.  With p = c ^ a[(r >> 2) & 3] the sixteen lanes of every group hit
// sixteen different bank groups iff {a0, a3, 1 ^ a1, 1 ^ a2} and {a1, a2, 1 ^ a0, 1 ^ a3} are both {0, 1, 2, 3}:
// a = (0, 3, 2, 1), i.e. (-(r >> 2)) & 3.  (Rounds 1-3 read 32-row fragments -- lane l: row l & 31, chunk l >> 5 -- for which
// a = (0, 1, 2, 3) was the conflict-free choice; that one is 2-way conflicted for the 16-row pattern.)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int sv_x3_swz(int64_t row) { return (int)((0 - (row >> 2)) & 3); }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline size_t sv_x3_off(int64_t row, int64_t k, int64_t Kd) {
  return ((size_t)(row >> 7) * (size_t)(Kd >> 5) + (size_t)(k >> 5)) * 4096 + (size_t)(row & 127) * 32 +
         (size_t)((((k & 31) >> 3) ^ sv_x3_swz(row)) << 3) + (size_t)(k & 7);
}
inline int64_t sv_x3_rows(int64_t n) { return (n + 255) & ~255ll; }

// grow-only device buffer
//
// Guard mode (SEGVLAD_GUARD=1 at segvlad_create; development / test runs): every buffer sits between two fences of
// SV_GUARD_BYTES poison bytes.  A grow-only buffer hides under-sized requests -- a request that an earlier, larger one
// already covers never fails (round 3 found a real out-of-bounds write that way, only on a fresh context) -- so in guard
// mode the BACK fence of a per-call scratch buffer sits right behind the bytes of the CURRENT request (re-placed by every
// reserve(), after a device synchronisation), and every API call ends with a check of all fences (sv_finish;
// segvlad_synchronize): a trampled fence fails the call with SEGVLAD_ERR_STATE and names the buffer.  Buffers whose
// contents outlive a call (`fixed`: database, models, the refinement's hand-over words) keep their back fence at the end
// of the allocation.
constexpr size_t SV_GUARD_BYTES = 1024;
constexpr uint32_t SV_GUARD_WORD = 0xA5C3F00Du;
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;          // usable bytes behind p
  const char* tag = "";
  bool guard = false;      // set by segvlad_create for every buffer of a guarded context
  bool fixed = false;      // guard mode: contents persist across calls -> the back fence stays at p + cap
  void* raw = nullptr;     // guard mode: the allocation (front fence | cap bytes | back fence)
  size_t req = 0;          // guard mode: bytes of the current request (the back fence starts at p + round_up(req, 4))
  int shrink = 0;          // guard mode, tests only (option "guard_undersize"): the fence is placed `shrink` bytes EARLY
  hipError_t reserve(size_t bytes) {
    if (guard) return reserve_guarded(bytes);
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr;
      cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    return hipSuccess;
  }
  hipError_t reserve_guarded(size_t bytes);   // api.hip
  size_t fence_off() const { return fixed ? cap : ((req > (size_t)shrink ? req - (size_t)shrink : 0) + 3) & ~(size_t)3; }
  void release() {
    if (raw) (void)hipFree(raw);
    else if (p) (void)hipFree(p);
    raw = nullptr;
    p = nullptr;
    cap = 0;
    req = 0;
  }
  template <class T>
  T* as() { return reinterpret_cast<T*>(p); }
};

// Accumulating stage timer: every StageScope records one (start, stop) event pair on the context
// stream; segvlad_stage_ms() sums the elapsed times of all pairs recorded since the last
// segvlad_profile_reset() (events are resolved lazily, nothing synchronises while timing).
struct StageTimer {
  std::vector<hipEvent_t> ev;  // pairs: ev[2i], ev[2i+1]
  int used = 0;                // number of pairs in use
  int launches = 0;
};

// Tuning / arithmetic switches.  Read ONCE (environment at segvlad_create, then segvlad_set_option); the hot
// entry points never call getenv.
struct SvOptions {
  int knn_filter = 0;     // 0 auto (fp16 when d % 64 == 0, else bf16x3 when d % 32 == 0, else fp32), 1 f16, 2 bf16x3, 3 fp32
  int pca_fp32 = 0;       // 1: plain fp32-MFMA projection instead of the fp16x3 split GEMM
  int f16_cfg = -1;       // kNN fp16 filter tile configuration (-1 = chosen from the shape)
  int f16_gm = -1;        // tile-block height of the XCD-aware order (-1 = by the number of query tiles, 0 = plain tm-fastest order)
  int f16_walk = -1;      // tile walk of the persistent fp16 filter: bit 0 = an XCD keeps its block of query tiles while it steps
                          // through the database blocks, bit 1 = odd steps run their k-tiles backwards (-1 = default = 3, see
                          // launch_f16_filter); never changes a result
  int f16_persist_wgs = 32;   // resident workgroups per XCD of the persistent batch filter (1 .. 32): fewer leave CUs free for the whole
                          // launch to kernels of OTHER streams (the next batch's describe stage, bench.py --pipeline); never changes a result
  int f16_mf = -1;        // MFMA shape of the persistent biased fp16 filter: 0 = 32 x 32 x 16, otherwise 16 x 16 x 32 (needs f16_epi != 0)
  int f16_deep_cfg = -1;  // deep rows (blocked accumulation): -1 / 5 = the persistent 256 x 256 ping-pong kernel with blocks flushed to a global
                          // scratch where the launch fills it (>= 1024 tiles), 4 elsewhere; 4 = 8 waves of 64 x 64 on 256 x 128 tiles, 16 x 16 x 32 MFMA, plain loop;
                          // 0 = 4 waves of 64 x 64 on 128 x 128 tiles, 32 x 32 x 16 MFMA (rounds 2-3); 1, 2, 3: measured variants
                          // (sv_launch_f16_filter)
  int f16_buf = -1;       // operand DMA as buffer_load ... lds: -1 = the deep-row kernel only (measured faster there, slower in the
                          // batch kernel), 1 = both, 0 = neither
  int tnk_gram = 1;       // fused VLAD -> PCA, project form: block norms of tasks with <= 64 tokens from their Gram matrix on the
                          // 16-bit matrix pipe (gram_norms_kernel); 0 = the fp32 block sums for every task
  int tnk_fork = 1;       // the two-tile Gram kernel on the context's side stream, beside the one-tile kernel (0: one after the other)
  int f16_dsplit = 0;     // batch kernel: pieces per phase whose DMA is issued from the MFMA segment instead of the load segment; -1: four of a phase's fragment reads issued from the previous MFMA segment; -2: a load segment's DMAs ahead of its fragment reads (A/B)
  int f16_small_mf = 0;   // 1: the batch filter's small (non-persistent) levels on the 16 x 16 x 32 shape + wave-private epilogue (A/B)
  int f16_pp = -1;        // main batch kernel: 0 = plain loop instead of the ping-pong loop (A/B)
  int f16_epi = -1;       // epilogue of the persistent biased fp16 filter: 0 = workgroup-level reservation (one global atomic per
                          // row and tile, two workgroup barriers), 1 = wave-private (one global atomic per survivor, no barrier)
  int x3_tile = 0;        // PCA split GEMM tile (0 = from the shape, 128, 256)
  int x3_gm = -1;         // PCA split GEMM XCD-aware block height (-1 = default of the kernel, 0 = plain order)
  int search_stats = 0;   // 1: segvlad_search records list occupancies (synchronises once per chunk)
  int knn_heuristic = 1;  // 1: low-rank (verified) thresholds in the level scheme; 0: rigorous k-th-rank thresholds only
  int assign_narrow = 0;  // 1: force the narrow assignment kernel
  int agg_kpb = 4;        // clusters per aggregation workgroup
  int debug_fail_search = 0;   // tests only: 1 = segvlad_search fails at once (SEGVLAD_ERR_STATE) -- the sharded entry's error path;
                               // 2 = segvlad_search_sharded fails BEHIND its local search, where it can only abort the communicator
  int debug_search = 0;   // 1: print per-level candidate statistics to stderr (synchronises); 7: token_norms_kernel waits for every
                          //    outstanding memory operation at every step (verification of its counted waits: same bits)
  int debug_small_tail = 0;   // tests only: bit 0 = every row of a device-driven pass is flagged for the tail's brute force, bit 1 = every
                              // row's band is sent to its second tier, bit 2 = the hand-over's sticky failure word is raised
  int batch_l0_f16 = 1;   // batch searches on the fp16 filter with guessed thresholds: the sampled level from the filter's own fp16 product
                          // (deep rows: through the filter kernel itself under +inf thresholds; 2 = the sample kernel there too, A/B)
                          // (sample_f16_batch_kernel) instead of the exact fp32 GEMM; 0 = rounds 2-5
  int level_carry = 1;    // batch searches, guessed thresholds, fp16 filter: the last level runs over the rows the stride-16 level has not
                          // seen and that level's survivors stay in the candidate lists (1/16 of the full-level GEMM saved); 0 = every
                          // level starts from empty lists and the last one covers every row (rounds 2-6a)
  int small_head = 1;     // single-image passes start with small_head_kernel (plane + scale + norms + flags + sample thresholds in one
                          // launch); 0 = query preparation -> exact sample level -> reduce + rank (rounds 3-5)
  int small_tail = 1;     // single-image passes end in small_tail_kernel (no read-back); 0 = the read-back of rounds 3-5
  int small_plan = 1;     // <= 128 queries (one query image per pass): one filter level behind an exact sample of 2048..4096
                          // rows (see segvlad_search); 0 = the deep plan of the batches
  int pj_f16 = 1;         // P-space aggregation: the tile sums on the 16-bit matrix pipe (0: fp32 MFMA, as before round 4)
  int pj_nw = 8;          // waves (32-column slices) per workgroup of the P-space aggregation: 8, or 4 (three workgroups per
                          // CU instead of one: measured SLOWER, 3.72 vs 3.40 ms for the PCA stage of 200 images)
  int refine_group = 1;   // batch searches: the refine bands of 32 consecutive query rows evaluated as ONE exact fp32 GEMM over the union of
                          // their rows when that is cheaper (refine_group_kernels.hip); 0 = every row on its own (rounds 1-4); 2 = every group whose union
                          // fits, whatever the cost model says (tests)
  int query_group = 0;    // hint: the query rows of a batch come in runs of this many rows per query image (1 .. 64; else unknown):
                          // the grouped refinement then takes an image's rows as one group instead of 32-row blocks
  int pca_path = 0;       // fused images_pca: 0 auto, 1 "planes" (descriptor planes x W), 2 "project" (project tokens, then aggregate)
};

// fp16 filter of the exact kNN: length of the accumulation blocks (0 = one running fp32 accumulator over the whole row).
// Rows of d >= 4096 (raw K*D descriptors) take the blocked kernel (option f16_cfg = 300 forces it for any d, any other
// explicit f16_cfg switches it off); the error constant below must describe the kernel that actually runs.
constexpr int SV_F16_KBLOCK = 1024;
inline int sv_f16_kblock(const SvOptions& o, int d) {
  if (o.f16_cfg == 300) return SV_F16_KBLOCK;
  return (o.f16_cfg < 0 && d >= 4096) ? SV_F16_KBLOCK : 0;
}
// Deep rows, default geometry (f16_deep_cfg -1 / 5): launches that fill the persistent 256 x 256 kernel run it with blocks of
// SV_F16_KFLUSH k-tiles of 64 flushed into a global scratch (knn_f16_filter_kernel, KFL) -- the block length the error constant
// of a search has to cover (the smaller levels run the register-blocked kernel, whose bound is smaller).
constexpr int SV_F16_KFLUSH = 64;
inline int sv_f16_eps_kblock(const SvOptions& o, int d) {
  const int kb = sv_f16_kblock(o, d);
  if (!kb) return 0;
  return (o.f16_deep_cfg < 0 || o.f16_deep_cfg == 5) ? SV_F16_KFLUSH * 64 : kb;
}
// |d2~ - d2| <= sv_f16_c_eps * ||q|| * ||r|| for the single-product fp16 filter (25 % slack included):
//   2^-10        both operands rounded to fp16 (relative 2^-11 each; products of two fp16 are exact in fp32),
//   2^-22        power-of-two scaling, sub-normal operands, second-order terms,
//   accumulation one running accumulator: <= 2 roundings per product, 2 d 2^-24;
//                blocked (kb > 0): 2 kb 2^-24 inside a block -- whatever order the matrix pipe sums a k-step in -- plus
//                (d / kb + 1) 2^-24 for the fp32 additions of the block sums,
//                relative to the largest running magnitude: sum_i |q_i r_i| <= ||q|| ||r||.  The batch kernels' BIAS starts
//                the accumulators at -||r||^2 / 2, so the running magnitude is <= ||q|| ||r|| + ||r||^2 / 2
//                <= bias_mult ||q|| max||r|| with bias_mult = 1 + max||r|| / (2 min||q||) over the query batch (1.5 for unit
//                vectors; segvlad_search measures it and keeps the unbiased kernel when it exceeds 5); bias_mult = 1 without;
// the factor 2 turns the error of the dot product into that of d2.
inline float sv_f16_c_eps(int d, int kb, float bias_mult) {
  const float acc = kb > 0 ? (2.f * (float)kb + (float)((d + kb - 1) / kb) + 1.f) / 16777216.f : 2.f * (float)d / 16777216.f;
  return 2.5f * (1.f / 1024.f + 1.f / 4194304.f + bias_mult * acc);
}

// statistics of the last segvlad_search (segvlad_search_stats)
struct SvSearchStats {
  int64_t levels = 0;           // filter levels after the sampled exact level (0 = matrix path)
  int64_t filter = 0;           // arithmetic of the filter levels: 0 none, 1 f16, 2 bf16x3, 3 fp32
  int64_t n_fallback = 0;       // query rows redone on the exact matrix path (list overflow)
  int64_t cand_max = 0;         // largest candidate list of the LAST level (search_stats only)
  int64_t cand_sum = 0;         // sum of the last level's candidate-list lengths (search_stats only)
  int64_t refine_max = 0;       // largest refine list (search_stats only)
  int64_t refine_sum = 0;       // sum of refine-list lengths (search_stats only)
  int64_t n_queries = 0;
  int64_t n_redo = 0;           // query rows whose heuristic thresholds did not verify and that were redone rigorously
  int64_t n_refine2 = 0;        // query rows whose refine band exceeded the first-tier list (SV_RCAP) and took the second tier
  int64_t grp_groups = 0;       // groups of 32 query rows whose bands were refined over the union of their rows (search_stats only)
  int64_t grp_union_sum = 0;    // sum of those unions' lengths (search_stats only)
  int64_t carry_rows = 0;       // database rows the last filter level did not compute again (taken over from the stride-16 level: level_carry)
};

struct segvlad_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // a second stream of the context's own, for kernels that share a launch slot with another one of the same call: forked off
  // `stream` behind an event and joined back before anything else is enqueued (sv_fork_side / sv_join_side); created on first use
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  uint32_t* h_pin = nullptr;          // 64 pinned bytes: scalars read back behind an event instead of a stream synchronisation
  hipEvent_t ev_scalars = nullptr;
  unsigned char* h_desc = nullptr;    // pinned: the per-image flags and the centroids of a segvlad_describe_begin
  size_t h_desc_cap = 0;
  hipEvent_t ev_desc = nullptr;       // behind their copies on the side stream
  int desc_B = 0, desc_S = 0;
  StageTimer* desc_timer = nullptr;   // the "describe" stage's open event pair (begin -> end)
  int desc_slot = 0;
  bool mask_branch_on_side = false;   // segvlad_describe: incidence / adjacency are in flight on `side`; images_impl joins before prep
  char err[512] = {0};
  bool profiling = false;
  bool scope_mute = false;   // set while a redo / fallback pass runs: its inner stages are part of "knn_redo" / "knn_fallback" only
  std::map<std::string, StageTimer> timers;
  SvOptions opt;
  SvSearchStats sstats;

  // vocabulary
  int K = 0, D = 0, Kpad = 0;
  float vocab_maxabs = 0.f;   // max |C_kd| (set_vocab): scale of the residual planes of the "project" form
  float vocab_norm_max = 0.f; // max_k ||C_k||_2 (set_vocab): bound of the projected residuals (16-bit P-space sums)
  // PCA model
  int P = 0, KD = 0, whiten = 0;
  float pca_w_scale = 0.f, pca_mean_maxabs = 0.f;
  bool pca_cproj_valid = false;           // cleared by segvlad_pca_set
  // database (exact kNN)
  int db_d = 0;
  int64_t db_n = 0;
  bool db_has_img = false;
  int64_t db_split_rows = 0;              // rows covered by the bf16 hi/lo planes (built lazily for the large-database search path)
  int64_t db_f16_rows = 0;                // rows covered by the fp16 image of the rows, scaled by a power of two (single-product filter)
  float db_f16_scale = 0.f, db_maxabs = 0.f;
  float db_rn_max = 0.f;
  int64_t db_rn_max_rows = 0;
  bool f16_bias_ok = false;   // this search's batch filter launches may use the biased-accumulator kernel (segvlad_search)
  const float* f16_scale_dev = nullptr;   // set by segvlad_search for the duration of a single-image search (see above)
  bool small_head_ran = false;   // this search's pass started with small_head_kernel (which also repairs a poisoned hand-over buffer)
  bool db_heur_off = false;   // set when > 25 % of a search's queries needed the rigorous redo (until the index changes)
  // device-driven single-image passes (small_pass_kernels.hip): the tail kernel's counters of the LAST such search live in device
  // memory and are fetched by segvlad_search_stats (the search itself never reads them back); its running totals reach the host
  // through two pinned words that segvlad_search looks at WITHOUT synchronising -- a database on which the low-rank thresholds
  // keep failing is switched to the rigorous plan one or two calls late instead of never
  const uint32_t* tail_stats_dev = nullptr;   // [4]: rows redone by brute force, second-tier rows, hand-over failures; null = none pending
  uint32_t tail_fail_base = 0;                // h_pin[8] when the index last changed
  int64_t tail_rows_since = 0;                // query rows sent through device-driven passes since then

  // Device buffers, as X-macro lists (segvlad_create tags them, segvlad_destroy releases them, guard mode walks them).
  //  persistent: contents outlive a call -- vocab: [K][D] raw centres; vocab_bt: normalised centres in MFMA-B order
  //  [D/2][Kpad/32][64]; pca_scale = 1/sqrt(var) or 1; pca_w1 / pca_w2: fp16 two-term split of comps * pca_w_scale (16-bit MFMA
  //  path); pca_cproj: [P] W mean ("project then aggregate"; rebuilt when stale); db_hi / db_lo / db_f16: 16-bit images of the rows;
  //  s_ref_keys / s_ref_tick: hand-over words of refine_exact_small_kernel (all ones / all zero between launches);
  //  s_tail_tick: tickets (all zero between launches) + running totals of small_tail_kernel
#define SV_PERSISTENT_BUFS(X)                                                                                                    \
  X(vocab) X(vocab_bt) X(pca_mean) X(pca_comps) X(pca_scale) X(pca_w1) X(pca_w2) X(pca_cproj) X(db_rows) X(db_norms) X(db_img)    \
  X(db_hi) X(db_lo) X(db_f16) X(s_ref_keys) X(s_ref_tick) X(s_tail_tick)
  //  scratch: grow-only, reused across calls, nothing in them is read after the call that wrote it; s_sh_*: exchange buffers of
  //  the row-sharded index (comm.hip)
#define SV_SCRATCH_BUFS(X)                                                                                                       \
  X(s_xt) X(s_labels) X(s_rnorm) X(s_gap) X(s_colmask) X(s_gscale) X(s_segimg) X(s_segoff) X(s_adjoff) X(s_dist) X(s_qnorm)      \
  X(s_misc) X(s_minmax) X(s_voteoff) X(s_cand_cnt) X(s_cand_d2) X(s_cand_id) X(s_thr_d2) X(s_thr_idx) X(s_flag) X(s_qh) X(s_ql)  \
  X(s_ref_cnt) X(s_ref_id) X(s_qscale) X(s_qf16) X(s_xh1) X(s_xh2) X(s_desc) X(s_tokorder) X(s_laboff) X(s_rnsorted) X(s_ovf)     \
  X(s_fb_q) X(s_fb_d2) X(s_fb_idx) X(s_fb_rows) X(s_rd_rows) X(s_rd_q) X(s_rd_d2) X(s_rd_idx) X(s_rd_flags) X(s_rd_p1) X(s_rd_p2)  \
  X(s_sel_todo) X(s_vote_keys) X(s_pz) X(s_rowbase) X(s_tilegrp) X(s_bn) X(s_l0part) X(s_ref_lim) X(s_sh_d2) X(s_sh_idx)          \
  X(s_sh_rec) X(s_sh_all) X(s_sh_d2c) X(s_sh_idc) X(s_grp_cnt) X(s_grp_ids) X(s_grp_rows) X(s_grp_keys) X(s_grp_work) X(s_grp_pos) X(s_tnk_redo) X(s_tail_part) X(s_km_part) X(s_km_cnt) X(s_kflush)
#define SV_DECL_BUF(n) DevBuf n;
  SV_PERSISTENT_BUFS(SV_DECL_BUF)
  SV_SCRATCH_BUFS(SV_DECL_BUF)
#undef SV_DECL_BUF
  template <class F>
  void for_each_buf(F&& f) {
#define SV_VISIT_BUF(n) f(n);
    SV_PERSISTENT_BUFS(SV_VISIT_BUF)
    SV_SCRATCH_BUFS(SV_VISIT_BUF)
#undef SV_VISIT_BUF
    for (auto& b : stage) f(b);
  }
  bool guard = false;          // SEGVLAD_GUARD=1 at segvlad_create
  char guard_hit[96] = {0};    // sticky: the first trampled fence found (buffer tag and which fence)
  // row-sharded index over several GPUs (comm.hip): an RCCL communicator bound at run time
  void* comm = nullptr;   // ncclComm_t
  int comm_rank = 0, comm_world = 1;
  std::vector<uint32_t> sh_flags_host;   // the ranks' status words of a sharded search (target of an asynchronous copy)
  // staging for host<->device pointers: a small ring, indexed by use inside one call
  std::vector<DevBuf> stage;
  struct Pending { void* host; void* dev; size_t bytes; };
  std::vector<Pending> pending_out;
  int stage_used = 0;

  int fail(int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
};

// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, bytes), remembered per (device, kernel): the driver call takes a
// lock and tens of microseconds; a launcher that repeats it before every launch leaves the GPU idle between kernels
hipError_t sv_max_dyn_lds(const void* fn, size_t bytes);

extern char sv_rccl_lib_override[256];   // comm.hip; filled by segvlad_create from SEGVLAD_RCCL_LIB
// comm.hip: destroys the context's communicator, if any (segvlad_destroy / segvlad_comm_destroy)
void sv_comm_release(segvlad_ctx* ctx);

// ---- pointer staging ---------------------------------------------------------------------------
bool sv_is_device_ptr(const void* p);
// returns a device pointer holding `bytes` of *p (copy enqueued on ctx->stream if p is host memory)
int sv_in(segvlad_ctx* ctx, const void* p, size_t bytes, const void** dev);
// returns a device pointer to write into; if p is host memory the D2H copy happens in sv_finish()
int sv_out(segvlad_ctx* ctx, void* p, size_t bytes, void** dev);
// flushes pending host outputs (synchronises the stream only if there are any) and resets staging; guard mode: also
// synchronises and checks every fence
int sv_finish(segvlad_ctx* ctx);
// guard mode: SEGVLAD_ERR_STATE if any fence of the context's buffers has been written (synchronises the device)
int sv_guard_check(segvlad_ctx* ctx);
int sv_fork_side(segvlad_ctx* ctx);   // side stream waits for everything enqueued on `stream` so far
int sv_join_side(segvlad_ctx* ctx);   // `stream` waits for everything enqueued on the side stream so far
void sv_begin(segvlad_ctx* ctx);

struct StageScope {
  segvlad_ctx* ctx;
  StageTimer* t = nullptr;
  int slot = 0;
  StageScope(segvlad_ctx* c, const char* name);
  ~StageScope();
  void count(int n = 1) { if (t) t->launches += n; }
};

// ---- kernel launchers (defined in the *_kernels.hip units) ---------------------------------------
// vlad_kernels.hip
int sv_launch_vocab_prepare(segvlad_ctx* ctx);
int sv_launch_incidence(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                        uint64_t* inc_bits, double* centroids /* may be null */);
int sv_launch_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, double* out);
int sv_launch_adjacency(segvlad_ctx* ctx, const double* cent, const int32_t* seg_off_dev, const int64_t* adj_off_dev,
                        int B, int S_max, int order, uint8_t* adj, uint32_t* n_bad, uint8_t* img_flags = nullptr);
int sv_launch_assign(segvlad_ctx* ctx, const float* tokens, int B, int N, float* xt, uint8_t* labels, float* rnorm,
                     float* gap);
int sv_launch_prep(segvlad_ctx* ctx, const uint8_t* labels, const uint64_t* inc_bits, const int32_t* seg_off_dev,
                   const int64_t* adj_off_dev, const uint8_t* adj, int B, int N, int K, int S_max, int SC,
                   uint64_t* colmask, float* gscale);
// centres == nullptr: the inputs are residuals already (x * rnorm - 0)
int sv_launch_aggregate(segvlad_ctx* ctx, const float* xt, const float* rnorm, const uint8_t* labels,
                        const uint64_t* colmask, const float* centres, int K, int D, const int32_t* seg_off_dev,
                        const float* gscale, int B, int N, int SC, float* out, float* block_norms,
                        const float* mean = nullptr, float xscale = 0.f, uint16_t* h1 = nullptr, uint16_t* h2 = nullptr);

// block norms + the fp16 planes of the token residuals x^ - C_k, grouped by cluster (rowbase from sv_launch_group_plan); after sv_launch_prep
int sv_launch_token_norms(segvlad_ctx* ctx, const float* xt, const uint64_t* colmask, const float* centres, int K, int D,
                          const int32_t* seg_off_dev, int B, int N, int SC, float* block_norms, float xscale, uint16_t* h1,
                          uint16_t* h2, const int32_t* rowbase, int64_t dummy_row /* a plane row nobody reads */);

// kmeans_kernels.hip: per-cluster sums of the normalised tokens + label counts of a batch, ACCUMULATED into sums / counts
int sv_launch_centroid_sums(segvlad_ctx* ctx, const float* xt, const float* rnorm, const uint8_t* labels, int B, int N, int D, int K,
                            double* sums, int64_t* counts);

// gemm_kernels.hip
int sv_launch_row_sumsq(segvlad_ctx* ctx, const float* X, int64_t n, int d, float* out);
int sv_launch_normalize_rows(segvlad_ctx* ctx, const float* X, int64_t n, int d, float* Y);
// C[M][N] = epilogue(A[M][Kd] . B[N][Kd]^T)
//   mode 0 (PCA):  A' = A - a_sub (a_sub [Kd] or null); C = acc * col_scale[n]
//   mode 1 (L2):   C = row_add[m] + col_add[n] - 2 acc
int sv_launch_gemm_nt(segvlad_ctx* ctx, int mode, const float* A, const float* Bm, float* C, int M, int N, int Kd,
                      int64_t ldc, const float* a_sub, const float* col_scale, const float* row_add,
                      const float* col_add);

// distance of every query to database rows 0, b_stride, 2*b_stride, ... (n_sample of them); column j of
// `dist` is sample j
int sv_launch_l2_strided(segvlad_ctx* ctx, const float* Q, const float* R, float* dist, int M, int n_sample, int Kd,
                         int64_t ldc, const float* qn, const float* rn, int b_stride, bool split_ok = false);
// the same launch with the K split's reduction left to the caller: *splits > 1 -> *parts = the partial dot products
// [*splits][M][ldc] (slices to be added in index order, then sv_d2 with the norms); *splits == 1 -> `dist` is finished
int sv_launch_l2_strided_parts(segvlad_ctx* ctx, const float* Q, const float* R, float* dist, int M, int n_sample, int Kd,
                               int64_t ldc, const float* qn, const float* rn, int b_stride, const float** parts, int* splits);
// the reduction of such a block fused with the rank select of every row (<= 4096 columns, <= 128 rows): thr_out[row] = the
// rank-th smallest distance of the row -- what splitk_reduce_d2_kernel + sv_launch_select_approx(mode 0, fixed_cnt) compute,
// value for value, in one launch instead of two (the single-image pass is a chain of dependent launches); also zeroes
// cand_cnt[row]; a row already flagged in fail_rows gets -inf
int sv_launch_l0_reduce_rank(segvlad_ctx* ctx, const float* parts, int splits, int M, int n_sample, int64_t ldc, const float* qn,
                             const float* rn, int b_stride, int rank, float* thr_out, uint32_t* cand_cnt, const uint32_t* fail_rows);
// same distances, but entries <= thr[m*thr_ld] are appended to (cand_d2, cand_id)[m][0..cap) via cand_cnt[m]
int sv_launch_l2_filter(segvlad_ctx* ctx, const float* Q, const float* R, int M, int n_sample, int Kd, const float* qn,
                        const float* rn, int b_stride, const float* thr, int64_t thr_ld, uint32_t* cand_cnt,
                        float* cand_d2, uint32_t* cand_id, int cap);

// knn_filter_kernels.hip
int sv_launch_split_bf16(segvlad_ctx* ctx, const float* X, int64_t n_elems, uint16_t* hi, uint16_t* lo);
int sv_launch_bf16_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Ql, const uint16_t* Rh, const uint16_t* Rl,
                          int M, int n_sample, int d, int b_stride, const float* qn, const float* rn, const float* thr,
                          int64_t thr_ld, float eps_mult, float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2,
                          uint32_t* cand_id, int cap);
// rank: which order statistic of the list is A (mode 0: the next level's threshold; mode 1: k).  check != 0 (heuristic
// thresholds): mode 0 flags a list shorter than rank, mode 1 flags A > thr_in[row * thr_in_ld] (the threshold the list was
// collected under).  Flagged rows (also: list overflow) are marked in fail_rows and counted once in *fail_count.
// mode 0 also ZEROES cand_cnt[row] (the next level's filter appends from zero).  fixed_cnt >= 0: every row is a list of that
// length (the sampled level's distance block) and cand_cnt is not read.
// mode 2 (batches, the level before the last one when that runs over the complement of this level's sample -- option level_carry):
// thr_out = min(A, thr_in) and the entries with d2~ <= thr_out + 2 eps STAY in the list, compacted to its front, cand_cnt[row] =
// their number: the last level appends behind them instead of computing the sample's rows a second time.
int sv_launch_select_approx(segvlad_ctx* ctx, uint32_t* cand_cnt, float* cand_d2, uint32_t* cand_id, int nq,
                            int cap, int rank, int mode, int check, const float* thr_in, int64_t thr_in_ld, const float* qn,
                            float c_eps, float rn_max, float* thr_out, uint32_t* ref_cnt, uint32_t* ref_id, int rcap,
                            uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows = nullptr, uint32_t* rovf_count = nullptr,
                            float* ref_lim = nullptr, int fixed_cnt = -1);
// rovf_rows / rovf_count / ref_lim (mode 1): a refine band longer than rcap marks its row there (with the band's upper limit)
// instead of in fail_rows: sv_launch_refine2_compact + sv_launch_refine_exact(..., rcap = cap, only_rows = rovf_rows)
// then refine it straight from the candidate list.
int sv_launch_refine2_compact(segvlad_ctx* ctx, const uint32_t* rovf_rows, const float* ref_lim, uint32_t* cand_cnt,
                              const float* cand_d2, uint32_t* cand_id, int nq, int cap);
// only_rows != null: rows whose flag is clear are skipped (their outputs stay as they are)
// fail_rows / fail_count / poison_dev (optional): the single-image refinement hands keys between workgroups through global
// memory and CHECKS the hand-over (refine_exact_small_kernel): a query whose keys did not all arrive is flagged there like
// a failed threshold check, and *poison_dev (set to a device word when that kernel ran, else to null) is non-zero afterwards;
// the caller then calls sv_refine_small_repair before the buffers' next use.
// fz / fused_done (round 6): when the shared-list kernel of a single-image pass runs, it finishes the rows the select flagged itself
// (small_pass_dev.h: SvSmallFinish) and *fused_done is set: the caller then launches no small_tail_kernel behind it.
struct SvSmallFinish;
int sv_launch_refine_exact(segvlad_ctx* ctx, const float* Q, const float* R, int nq, int d, const float* qn, const float* rn,
                           const uint32_t* ref_cnt, const uint32_t* ref_id, int rcap, int k, float* d2_out, int64_t* idx_out,
                           const uint32_t* only_rows = nullptr, uint32_t* fail_rows = nullptr, uint32_t* fail_count = nullptr,
                           const uint32_t** poison_dev = nullptr, const SvSmallFinish* fz = nullptr, bool* fused_done = nullptr);
int sv_small_words(segvlad_ctx* ctx);   // ctx->s_tail_tick: [129] tail tickets, [2] totals, pad, the head's 64-bit arrival counter
// tests only (option debug_small_tail): force the flags of a pass that needed none
int sv_launch_small_tail_debug(segvlad_ctx* ctx, int m, uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, float* ref_lim);
int sv_refine_small_repair(segvlad_ctx* ctx);
// small_pass_kernels.hip: the device-driven tail of a single-image pass (<= 128 rows): launched behind the refinement, returns at
// once when fail_count[0] == fail_count[1] == 0, else finishes the flagged rows on the device (second tier from the candidate
// lists; exact brute force for rows flagged for a redo) -- the search needs no read-back.  stats: [4] device words of this search.
int sv_launch_small_tail(segvlad_ctx* ctx, const float* Q, const float* R, const float* qn, const float* rn, int64_t n, int d, int m, int k,
                         uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, const float* ref_lim, const uint32_t* cand_cnt,
                         const float* cand_d2, const uint32_t* cand_id, int cap, float* d2_out, int64_t* idx_out, uint32_t* stats);
int sv_ensure_pinned_words(segvlad_ctx* ctx);
// the HEAD of such a pass in one launch: the queries' fp16 plane (scales_dev[0] = scale, [1] = 1 / (scale x db_scale)), their squared
// norms (row_sumsq's bits), the zeroed flag block, and thr_out[q] = the rank-th smallest APPROXIMATE distance (the filter's own fp16
// product) of query q to the n0 sample rows 0, stride, 2 stride, ... of the fp16 plane Rh; cand_cnt[q] = 0.
// cand_scratch: m x n0 floats.  sv_small_head_ok: the shapes it takes.
bool sv_small_head_ok(int m, int d, int n0, int rank);
// a BATCH search's sampled level from the same product: dist[q][j] = d2~(query q, database row j * stride), j < n0 (row stride ld)
int sv_launch_sample_f16_batch(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Rh, int m, int n0, int d, int64_t stride, float inv_scale,
                               const float* qn, const float* rn, float* dist, int64_t ld);
int sv_launch_small_head(segvlad_ctx* ctx, const float* X, int m, int d, const uint16_t* Rh, const float* rn, int64_t stride, int n0,
                         float db_scale, int rank, uint16_t* qplane, float* scales_dev, float* qn_out, uint32_t* zero, int zero_words,
                         float* cand_scratch, float* thr_out, uint32_t* cand_cnt);   // ctx->h_pin (16 zeroed words) + ctx->ev_scalars
// refine_group_kernels.hip: the same refinement for a batch, with the bands of 32 consecutive query rows evaluated over the
// union of their rows where they overlap (option refine_group); *launches = kernels launched
constexpr int SV_RG_UCAP = 2048;   // longest union a group may hold (longer: its rows keep the per-row kernels)
int sv_launch_refine_grouped(segvlad_ctx* ctx, const float* Q, const float* R, int nq, int d, const float* qn, const float* rn,
                             const uint32_t* ref_cnt, const uint32_t* ref_id, int rcap, int k, float* d2_out, int64_t* idx_out,
                             int* launches, const uint32_t* only_rows = nullptr, int live_groups_hint = 0);
int sv_refine_group_stats(segvlad_ctx* ctx, int nq, int64_t* groups, int64_t* grouped, int64_t* union_sum);
int sv_row_norm_max(segvlad_ctx* ctx, const float* norms, int64_t n, float* out_host);
int sv_row_norm_min(segvlad_ctx* ctx, const float* norms, int64_t n, float* out_host);
int sv_maxabs(segvlad_ctx* ctx, const float* x, int64_t n, float* out_host);
int sv_maxabs_and_norm_min(segvlad_ctx* ctx, const float* x, int64_t n, const float* norms, int64_t n_norms, float* maxabs_host,
                           float* norm_min_host);
int sv_maxabs_and_norm_min_begin(segvlad_ctx* ctx, const float* x, int64_t n, const float* norms, int64_t n_norms);
int sv_maxabs_and_norm_min_end(segvlad_ctx* ctx, int64_t n, int64_t n_norms, float* maxabs_host, float* norm_min_host);
// gemm_f16x3_kernels.hip
int sv_launch_split_f16x2(segvlad_ctx* ctx, const float* X, int64_t n_rows, int d, const float* sub, float scale, uint16_t* h1,
                          uint16_t* h2);
int sv_launch_gemm_f16x3(segvlad_ctx* ctx, const uint16_t* A1, const uint16_t* A2, const uint16_t* B1, const uint16_t* B2, int M,
                         int N, int Kd, float out_scale, const float* col_scale, float* C);
int sv_launch_gemm_f16x3_grouped(segvlad_ctx* ctx, const uint16_t* A1, const uint16_t* A2, const uint16_t* B1, const uint16_t* B2,
                                 int M_pad, int N, int Kd, int n_groups, const int32_t* tile_group, float out_scale, float* C);
// project_kernels.hip ("project then aggregate" form of segvlad_images_pca)
int sv_launch_group_plan(segvlad_ctx* ctx, const int32_t* lab_off, int B, int K, int32_t* rowbase, int32_t* tile_group, int max_tiles);
int sv_launch_project_consts(segvlad_ctx* ctx, const float* comps, const float* mean, int P, int64_t KD, float* wmu /*[P] = W mean*/);
int sv_launch_project_aggregate(segvlad_ctx* ctx, const float* Z, const float* wmu, const float* block_norms, const float* gscale,
                                const uint64_t* colmask, const int32_t* lab_off, const int32_t* rowbase, const int32_t* seg_off_dev,
                                int B, int N, int K, int P, int SC, int S_max, const float* col_scale, float* Y, float zscale);
int sv_launch_to_f16(segvlad_ctx* ctx, const float* X, int64_t n_elems, float scale, uint16_t* out);
// single-image searches: the query plane and its scales without a host round trip (scales_dev[0] = query scale,
// [1] = 1 / (query scale x db_scale)); ctx->f16_scale_dev != null makes the filter kernel read [1] instead of its argument
// qn_out != null (needs d % 4 == 0, 16-byte aligned rows): also the rows' squared norms, bit for bit sv_launch_row_sumsq's
int sv_launch_query_f16_small(segvlad_ctx* ctx, const float* X, int64_t n_elems, float db_scale, uint16_t* out, float* scales_dev,
                              float* qn_out, int nq, int d, uint32_t* zero, int zero_words);
int sv_launch_to_f16_devscale(segvlad_ctx* ctx, const float* X, int64_t n_elems, const float* scales_dev, uint16_t* out);
int sv_launch_f16_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Rh, int M, int n_sample, int d, int b_stride,
                         float inv_scale, const float* qn, const float* rn, const float* thr, int64_t thr_ld, float eps_mult,
                         float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2, uint32_t* cand_id, int cap, int skip = 0);
// skip = 16: the operand rows are the database rows that are NOT multiples of 16, in order (n_sample of them; b_stride = 1)
bool sv_f16_filter_skip_ok(const segvlad_ctx* ctx, int M, int64_t n_rows, int d);

// select_kernels.hip
// top-k of per-query candidate lists (LDS sort on (distance, id)); a list longer than cap flags its query row in
// ovf_rows (and counts it once in *ovf_count): that query is redone on the matrix path
int sv_launch_select_cand(segvlad_ctx* ctx, const uint32_t* cand_cnt, const float* cand_d2, const uint32_t* cand_id,
                          int nq, int cap, int k, float* d2_out, int64_t* idx_out, uint32_t* ovf_rows, uint32_t* ovf_count);
int sv_launch_select_topk(segvlad_ctx* ctx, const float* dist, int64_t ld, int nq, int64_t n, int k, float* d2_out,
                          int64_t* idx_out, int64_t out_ld, int64_t id_base);
int sv_launch_merge_topk(segvlad_ctx* ctx, const float* d2_parts, const int64_t* idx_parts, int nq, int cand, int k,
                         float* d2_out, int64_t* idx_out);
int sv_launch_sims(segvlad_ctx* ctx, const float* d2, const int64_t* idx, int nq, int k_in, int k_keep, float* sims,
                   int64_t* idx_out);
int sv_launch_minmax(segvlad_ctx* ctx, const float* sims, int64_t count, float* minmax_dev);

// vote_kernels.hip
int sv_launch_vote(segvlad_ctx* ctx, const int64_t* idx, const float* sims, const int32_t* img_of_seg,
                   int64_t n_ref_seg, const int32_t* qoff_dev, const int32_t* qoff_host, int n_img, int k,
                   const float* minmax_dev, int n_top, int mode, int32_t* pred, double* score);
