/*
 * segvlad_dev.h -- DEVELOPMENT switches of libsegvlad_hip.so (segvlad_set_option keys that are NOT part of the ABI declared in
 * include/segvlad.h).  Nothing here changes a result: every kNN switch is followed by the exact fp32 refinement
 * (tests/test_gpu_filter_variants.py holds each variant to the bits of the all-fp32 filter), the VLAD / PCA switches select
 * kernels that agree to ~1e-6 relative (tests/test_gpu_parity.py).
 *
 * Builds.  The SHIPPED library (revisit-anything_amd/build.py; lib/libsegvlad_hip.so) instantiates only the kernels the switches
 * default to; a value that selects another one is rejected with SEGVLAD_ERR_ARG.  The DEVELOPMENT build
 * (`python revisit-anything_amd/build.py --ablations`, -DSEGVLAD_ABLATIONS -> lib/libsegvlad_hip_abl.so, loaded with
 * SEGVLAD_LIB_PATH) holds every measured-and-not-kept variant (DESIGN.md 4 / 7.1 has the measurements), the timing ablations
 * with WRONG results (f16_cfg 10 .. 160, SEGVLAD_ASSIGN_ABL, SEGVLAD_AGG_ABL) and the phase timers.
 *
 *   fp16 candidate filter of the exact kNN (knn_filter_kernels.hip)                                shipped library accepts
 *     "f16_cfg"       tile configuration; -1 = from the shape: 250 batches, 300 deep rows (d >= 4096), 62 / 63 one query
 *                     image per pass                                                             -1, 250, 300, 62, 63
 *     "f16_mf"        MFMA shape of the batch kernels: 0 = 32 x 32 x 16, else 16 x 16 x 32       -1, 1
 *     "f16_epi"       epilogue: 0 = workgroup-level reservation, 1 = wave-private                -1, 1
 *     "f16_pp"        main loop of the batch kernel: 0 = plain, else ping-pong                   -1, 2
 *     "f16_deep_cfg"  deep-row geometry: -1 / 4 = 256 x 128 tiles, 8 waves, 16 x 16 x 32; 0 = rounds 2-3's kernel;
 *                     1, 2, 3 = measured variants                                                -1, 4
 *     "f16_buf"       operand DMA as buffer_load ... lds: -1 = the deep-row kernel only, 1 = both, 0 = neither   -1, 0
 *     "f16_dsplit"    placement of a phase's DMA / fragment reads (1, 2, -1, -2: measured variants)   0
 *     "f16_small_mf"  1 = the small (non-persistent) levels on the 16 x 16 x 32 shape            0
 *     "f16_walk"      tile walk of the persistent kernel: bit 0 = an XCD keeps its block of query tiles, bit 1 = odd steps run
 *                     their k-tiles backwards, bit 2 = rotated k start per workgroup (-1 = 3)    any (a launch parameter)
 *     "f16_gm"        tile-block height of the XCD-aware order                                    any (a launch parameter)
 *   fused VLAD -> PCA (vlad_kernels.hip, project_kernels.hip, gemm_f16x3_kernels.hip)
 *     "tnk_gram"      1 | 0   block norms from the Gram matrix of a task's residuals (16-bit matrix pipe) | fp32 block sums
 *     "tnk_fork"      1 | 0   the two-tile Gram launch on the context's side stream | in line
 *     "pj_f16"        1 | 0   P-space tile sums on the 16-bit matrix pipe | fp32 MFMA
 *     "pj_nw"         8 | 4   waves per workgroup of the P-space aggregation
 *     "x3_tile", "x3_gm", "agg_kpb", "assign_narrow"   tile / order / geometry of the projection GEMM, the descriptor
 *                     aggregation and the assignment kernel
 *     "f16_persist_wgs" 1..32 resident workgroups per XCD of the persistent batch filter (32 = every CU; fewer leave CUs to other streams)
 *     "level_carry"   1 | 0   batch searches (guessed thresholds): the last filter level skips the rows of the stride-16 level, whose
 *                             survivors stay in the candidate lists | every level from empty lists over all of its rows
 *     "batch_l0_f16"  1 | 0   batch searches (guessed thresholds): the sampled level from the filter's own fp16 product | the exact fp32 GEMM
 *                             (2: deep rows through sample_f16_batch_kernel instead of the filter kernel under +inf thresholds, A/B)
 *   single-image passes (small_pass_kernels.hip)
 *     "small_head"    1 | 0 | 3   the pass starts with small_head_kernel: plane, scale, norms, flags and the sample thresholds (from the
 *                     filter's own fp16 product) in one launch | query preparation -> exact fp32 sample level -> reduce + rank |
 *                     the same kernel with the query block loaded straight in MFMA fragment shape (measured slower: 22.7 vs 21 us)
 *     "small_tail"    1 | 2 | 0   no read-back, no host synchronisation in segvlad_search: flagged rows are finished on the DEVICE -- by the
 *                     refinement kernel's own workgroups where the shared-list refinement runs behind small_head_kernel (no extra launch),
 *                     else by small_tail_kernel, launched behind every pass | always small_tail_kernel | the read-back of rounds 3-5
 *   debugging and the tests' own hooks
 *     "debug_small_tail"   bit 0: every row of a device-driven pass is flagged for the exact brute force; bit 1: every row is sent through
 *                          the second tier; bit 2: the checked hand-over's sticky word is raised; bit 3: query row 1's hand-over is treated
 *                          as failed (the fused finish recomputes the band); bits 8..: grid of small_tail_kernel (A/B)
 *     "debug_search"       1 = per-level candidate statistics on stderr (synchronises); 7 = the Gram kernel waits for every
 *                          outstanding memory operation at every step (verification of its counted waits)
 *     "debug_fail_search"  segvlad_search fails at once (the sharded entry's error path)
 *     "guard_undersize"    "<buffer>:<bytes>": guard-mode tests (include/segvlad.h, SEGVLAD_GUARD)
 *   Environment read once by segvlad_create for these: SEGVLAD_F16_CFG, SEGVLAD_F16_GM, SEGVLAD_X3_TILE, SEGVLAD_X3_GM,
 *   SEGVLAD_ASSIGN_NARROW, SEGVLAD_DEBUG_SEARCH, SEGVLAD_AGG_KPB.
 */
#ifndef SEGVLAD_DEV_H
#define SEGVLAD_DEV_H
#include "../../include/segvlad.h"
#endif
