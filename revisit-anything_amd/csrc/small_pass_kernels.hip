// One query image per pass (<= 128 query rows) without a host round trip -- round 6.
//
//   small_head_kernel   the HEAD of the pass in one launch (rounds 3-5: three -- query preparation, an exact fp32 sample GEMM split over
//                       K, its reduction + rank select: 11 + 24 + 17 us of dependent launches in front of a 350-us filter).  What the
//                       filter needs from the head is (a) the queries' fp16 plane, its power-of-two scale, the rows' squared norms,
//                       zeroed flags -- and (b) per query row a THRESHOLD that admits roughly stride x rank rows.  The threshold is a
//                       guess that the pass verifies afterwards, so it does not have to come from exact
//                       distances: here every workgroup takes 32 rows of the strided sample, converts the queries itself (all of them
//                       sit in its registers: one read of <= 256 KiB from L2), multiplies them with its rows on the 16-bit matrix pipe
//                       (the filter's own product: d2~ = ||q||^2 + ||r||^2 - 2 q.r / (s_q s_db)), keeps the SMALLEST value per query of
//                       its 32 columns, and the workgroup that draws the last ticket ranks those minima (one per workgroup and query):
//                       thr[q] = the rank-th smallest of them -- the rank-th smallest approximate sample distance unless two of the
//                       smallest share a workgroup (then the next one: a little more generous, never tighter) --, the quantity a
//                       filter level hands to the next one in a batch search.  The plane, the scale, the norms (row_sumsq_kernel's arithmetic, lane for lane:
//                       the exact refinement uses them) and the zeroed flag block are written by the workgroups on the side.
//                       NOTHING downstream depends on the hand-over being complete: a threshold that is off (a candidate that had not
//                       landed when the last workgroup read it) fails the pass's own check and the row is redone exactly.
//   small_tail_kernel   the DEVICE-DRIVEN tail of the pass.  Rounds 3-5 ended every single-image search with an 8-byte read-back
//                       (how many rows failed their threshold check / outgrew the first-tier refine list?) and a host
//                       synchronisation: 10 % of the call, and the reason why back-to-back searches could not overlap their
//                       launch latencies.  This kernel is launched behind EVERY such pass and reads the two counters on the
//                       device: both zero (every pass of the benchmarks) -> it returns at once.  Otherwise it finishes the
//                       flagged rows itself, exactly:
//                         * a row whose refine band outgrew the first-tier list (rovf): the band's ids are compacted out of the
//                           row's candidate list (a superset of the band) and re-evaluated -- refine2_compact_kernel +
//                           refine_exact_kernel's arithmetic in one workgroup;
//                         * a row flagged for the redo (threshold check failed, candidate list overflow, a checked hand-over that
//                           failed): EXACT BRUTE FORCE over the whole index -- every workgroup keeps the k best (distance, id)
//                           keys of its slice of the rows, the workgroup that draws the last ticket of the row merges the G
//                           partial lists.  No thresholds, no lists: nothing left that could fail.
//                       Both evaluate a distance as the sequential chain acc = fma(q[j], r[j], acc), j = 0 .. d-1, then sv_d2 with
//                       the same norms, and order by (distance, id): the bits of refine_exact_kernel, i.e. of every other path.
//                       A sticky failure word of refine_exact_small_kernel's checked hand-over is also repaired here (its buffers
//                       back to all ones / all zero), where the host used to do it after the read-back.
//
// Rare path: written for clarity and exactness, not speed (one flagged row: ~2 ms on a 1 M x 1024 index).  A database on which
// the low-rank thresholds keep failing is noticed by the host through a pinned counter (no synchronisation: it reads whatever has
// arrived) and switched to the rigorous plan, as the read-back path did.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "ctx.h"
#include "knn_dev.h"
#include "small_pass_dev.h"


__global__ __launch_bounds__(256) void small_tail_kernel(const float* __restrict__ Q, const float* __restrict__ R,
                                                         const float* __restrict__ qn, const float* __restrict__ rn, int64_t n, int d,
                                                         int m, int k, uint32_t* __restrict__ fail_rows,
                                                         uint32_t* __restrict__ fail_count, uint32_t* __restrict__ rovf_rows,
                                                         const float* __restrict__ ref_lim, const uint32_t* __restrict__ cand_cnt,
                                                         const float* __restrict__ cand_d2, const uint32_t* __restrict__ cand_id, int cap,
                                                         float* __restrict__ d2_out, int64_t* __restrict__ idx_out,
                                                         uint64_t* __restrict__ part, uint32_t* __restrict__ tickets, int kp,
                                                         uint64_t* __restrict__ gkeys, int64_t gkeys_words, uint32_t* __restrict__ tick,
                                                         int tick_words, int tick_poison, uint32_t* __restrict__ stats,
                                                         uint32_t* __restrict__ totals, volatile uint32_t* __restrict__ host_totals) {
  const uint32_t n_fail = __hip_atomic_load(&fail_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t n_rovf = __hip_atomic_load(&fail_count[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_fail == 0u && n_rovf == 0u) return;   // every row of the pass is finished: the common case, ~2 us of launch
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);                          // [ST_CAP] sort scratch
  uint64_t* best = a + ST_CAP;                                              // [ST_QB][kp] (phase 2)   | the band's ids (phase 1)
  uint32_t* band = reinterpret_cast<uint32_t*>(best);                       // [ST_CAP]
  float* qs = reinterpret_cast<float*>(best + (size_t)ST_QB * 1024);        // [ST_QB][ST_KC]
  __shared__ int s_list[128];
  __shared__ int s_nlist, s_last;
  __shared__ uint32_t s_cnt;
  const int tid = threadIdx.x, G = gridDim.x;

  // ---- phase 0: a failed hand-over of refine_exact_small_kernel (its rows are flagged: phase 2 redoes them) ---------------------
  if (tick && __hip_atomic_load(&tick[tick_poison], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + tid; j < gkeys_words; j += (int64_t)G * 256)
      __hip_atomic_store(&gkeys[j], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the sticky word itself is cleared by the LAST workgroup to leave the kernel -- below -- so that every workgroup sees it)
  }

  // ---- phase 1: second refinement tier (band longer than the first-tier list): compact the band, refine it ----------------------
  if (n_rovf) {
    for (int row = blockIdx.x; row < m; row += G) {
      if (!rovf_rows[row] || fail_rows[row]) continue;   // (workgroup-uniform)
      const uint32_t c = min(cand_cnt[row], (uint32_t)cap);
      const float lim = ref_lim[row];
      if (tid == 0) s_cnt = 0u;
      __syncthreads();
      for (uint32_t j = tid; j < c; j += 256)
        if (cand_d2[(size_t)row * cap + j] <= lim) band[atomicAdd(&s_cnt, 1u)] = cand_id[(size_t)row * cap + j];
      __syncthreads();
      const int nb = (int)s_cnt;
      if (nb == 0) continue;   // (cannot happen for a row that outgrew the first tier; workgroup-uniform)
      int np2 = 2;
      while (np2 < nb) np2 <<= 1;
      for (int j = tid; j < np2; j += 256) a[j] = ~0ull;
      const float q2 = qn[row];
      for (int j0 = 0; j0 < nb; j0 += 256) {
        const int j = j0 + tid;
        const uint32_t id = j < nb ? band[j] : band[0];
        float acc[1] = {0.f};
        for (int c0 = 0; c0 < d; c0 += ST_KC) {
          const int kc = min(ST_KC, d - c0);
          __syncthreads();
          for (int t = tid; t < kc; t += 256) qs[t] = Q[(size_t)row * d + c0 + t];
          __syncthreads();
          chain_step<1>(R + (size_t)id * d + c0, kc, qs, acc);
        }
        if (j < nb) a[j] = ((uint64_t)f2key_(sv_d2(q2, rn[id], acc[0])) << 32) | id;
      }
      bitonic64(a, np2, tid);
      for (int j = tid; j < k; j += 256) {
        const bool in = j < nb;
        d2_out[(size_t)row * k + j] = in ? key2f_((uint32_t)(a[j] >> 32)) : INFINITY;
        idx_out[(size_t)row * k + j] = in ? (int64_t)(uint32_t)a[j] : -1;
      }
      __syncthreads();
    }
  }

  // ---- phase 2: exact brute force for the rows flagged for a redo ----------------------------------------------------------------
  if (n_fail) {
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    if (tid < m && tid < 128 && fail_rows[tid]) s_list[atomicAdd(&s_nlist, 1)] = tid;
    __syncthreads();
    const int nl = s_nlist;
    // (the order of s_list differs between workgroups; every row is handled on its own slot, so it does not matter -- sorted
    //  all the same, for the sweeps of ST_QB rows to group the same rows everywhere)
    if (tid == 0) {
      for (int i = 1; i < nl; ++i) {
        const int v = s_list[i];
        int j = i - 1;
        for (; j >= 0 && s_list[j] > v; --j) s_list[j + 1] = s_list[j];
        s_list[j + 1] = v;
      }
    }
    __syncthreads();
    const int64_t per = (n + G - 1) / G;
    const int64_t r_lo = (int64_t)blockIdx.x * per, r_hi = min(n, r_lo + per);
    int ns = 512;
    while (ns < kp + 256) ns <<= 1;
    for (int l0 = 0; l0 < nl; l0 += ST_QB) {
      const int nbq = min(ST_QB, nl - l0);
      for (int j = tid; j < ST_QB * kp; j += 256) best[j] = ~0ull;
      __syncthreads();
      for (int64_t base = r_lo; base < r_hi; base += 256) {
        const int64_t r = base + tid;
        const bool live = r < r_hi;
        const int64_t rr = live ? r : r_lo;
        float acc[ST_QB] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < d; c0 += ST_KC) {
          const int kc = min(ST_KC, d - c0);
          __syncthreads();
          for (int b = 0; b < nbq; ++b)
            for (int t = tid; t < kc; t += 256) qs[b * ST_KC + t] = Q[(size_t)s_list[l0 + b] * d + c0 + t];
          for (int b = nbq; b < ST_QB; ++b)
            for (int t = tid; t < kc; t += 256) qs[b * ST_KC + t] = 0.f;
          __syncthreads();
          chain_step<ST_QB>(R + (size_t)rr * d + c0, kc, qs, acc);
        }
        const float r2 = rn[rr];
        for (int b = 0; b < nbq; ++b) {
          const uint64_t key = live ? (((uint64_t)f2key_(sv_d2(qn[s_list[l0 + b]], r2, acc[b])) << 32) | (uint32_t)r) : ~0ull;
          merge_best(best + (size_t)b * kp, kp, k, key, a, ns, tid);
        }
      }
      // hand the slice's k best of every row of the sweep to the row's slot, draw a ticket; the last workgroup merges the G lists
      for (int b = 0; b < nbq; ++b) {
        const int row = s_list[l0 + b];
        uint64_t* mine = part + ((size_t)row * G + blockIdx.x) * kp;
        for (int j = tid; j < kp; j += 256) mine[j] = best[(size_t)b * kp + j];
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = (atomicAdd(&tickets[row], 1u) == (uint32_t)(G - 1)) ? 1 : 0;
        __syncthreads();
        if (!s_last) continue;
        __threadfence();
        uint64_t* bb = best + (size_t)b * kp;
        for (int j = tid; j < kp; j += 256) bb[j] = ~0ull;
        __syncthreads();
        const uint64_t* all = part + (size_t)row * G * kp;
        for (int64_t j0 = 0; j0 < (int64_t)G * kp; j0 += 256)
          merge_best(bb, kp, k, __hip_atomic_load(&all[j0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), a, ns, tid);
        for (int j = tid; j < k; j += 256) {
          const uint64_t v = bb[j];
          d2_out[(size_t)row * k + j] = v != ~0ull ? key2f_((uint32_t)(v >> 32)) : INFINITY;
          idx_out[(size_t)row * k + j] = v != ~0ull ? (int64_t)(uint32_t)v : -1;
        }
        if (tid == 0) tickets[row] = 0u;   // all zero again for the next pass
        __syncthreads();
      }
    }
  }

  // ---- the last workgroup out: statistics, the sticky word, the pinned totals ------------------------------------------------------
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(&tickets[128], 1u) == (uint32_t)(G - 1)) {
      tickets[128] = 0u;
      stats[0] = n_fail;
      stats[1] = n_rovf;
      if (tick && __hip_atomic_load(&tick[tick_poison], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        stats[2] = 1u;
        for (int j = 0; j < tick_words; ++j) __hip_atomic_store(&tick[j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      totals[0] += n_fail;
      totals[1] += n_rovf;
      if (host_totals) {
        host_totals[0] = totals[0];
        host_totals[1] = totals[1];
      }
    }
  }
}


// ---- the head: query plane + scale + norms + flags + sample thresholds in ONE launch ------------------------------------------------
typedef _Float16 sh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sh_f16x4 __attribute__((ext_vector_type(4)));
typedef float sh_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SH_T = 512, SH_W = SH_T / 64;   // threads, waves per workgroup
constexpr int SH_RW = 32;                     // sample rows per workgroup (two 16-column MFMA tiles)
constexpr int SH_NX = 32;                     // float4 of the query block per thread: <= 65 536 query floats
constexpr int SH_MT = 8;                      // 16-row query tiles (m <= 128)
constexpr int SH_PER = 2;                     // merge: workgroup minima per lane (<= 128 workgroups: a sample of <= 4096 rows)
constexpr int SH_PRE = 4;                     // k-steps whose database fragments are requested before anything else

// FR: 0 = the query block staged through LDS (any shape sv_small_head_ok admits); 1, 2, 4, 8 = the number of 16-row query tiles, padded
// to a power of two, of the FRAGMENT-DIRECT form (round 6, second version): the k-steps are split over the waves (wave v takes k-steps v,
// v + 8, ...), so the slices of the query block the waves multiply are DISJOINT -- every wave loads its own slice straight in MFMA
// fragment shape (lane (i, kq): 8 consecutive floats of row 16 t + i), converts it in registers and multiplies; no LDS image of the
// operand, no index arithmetic per element, no barrier between load and product (needs FR x ceil(d / 256) <= 16 fragments per lane).
template <int FR>
__global__ __launch_bounds__(SH_T) void small_head_kernel(const float* __restrict__ X, int m, int d, const _Float16* __restrict__ Rh,
                                                          const float* __restrict__ rn, int64_t stride, int n0, float db_scale, int rank,
                                                          _Float16* __restrict__ qplane, float* __restrict__ scales,
                                                          float* __restrict__ qn_out, uint32_t* __restrict__ zero, int zero_words,
                                                          float* __restrict__ cand, unsigned long long* __restrict__ ticket,
                                                          float* __restrict__ thr_out, uint32_t* __restrict__ cnt, int dbg,
                                                          uint32_t* __restrict__ rtick, int rtick_poison, uint64_t* __restrict__ gkeys,
                                                          int64_t gkeys_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // phase timing (development: -DSV_HEAD_TIMING and SV_HEAD_TIMING=1 in the environment print every workgroup's phase cycles)
#ifdef SV_HEAD_TIMING
  unsigned long long T[12];
  int ti = 0;
#define TICK() T[ti++] = __builtin_amdgcn_s_memtime()
#else
#define TICK()
#endif
  TICK();
  __shared__ uint32_t s_wmax[SH_W];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i16 = lane & 15, kq = lane >> 4;
  const int w = blockIdx.x, NW = gridDim.x;
  // a hand-over of the previous pass's refinement failed its check (never observed): its buffers back to all ones before THIS pass's
  // refinement uses them -- every workgroup reads the sticky word before the grid barrier, workgroup 0 clears it behind the barrier
  const bool repair = rtick && __hip_atomic_load(&rtick[rtick_poison], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
  if (repair)
    for (int64_t j = (int64_t)w * SH_T + tid; j < gkeys_words; j += (int64_t)NW * SH_T)
      __hip_atomic_store(&gkeys[j], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int mpad = (m + 15) & ~15, MT = mpad >> 4;
  const int lda = d + 8;                                   // halves per LDS row of the query operand (+16 B: the 16 rows of a fragment read hit 16 bank groups)
  _Float16* As = reinterpret_cast<_Float16*>(smem);        // [mpad][lda]
  const int d4 = d >> 2, n4 = m * d4, nks = d >> 5;

  // (0) this wave's first database fragments: k-steps wv, wv + 8, ... of the two 16-row tiles (rows beyond the sample re-read its last row)
  const int64_t row_a = min((int64_t)w * SH_RW + i16, (int64_t)n0 - 1), row_b = min((int64_t)w * SH_RW + 16 + i16, (int64_t)n0 - 1);
  const _Float16* ra = Rh + (size_t)(row_a * stride) * d + kq * 8;
  const _Float16* rb = Rh + (size_t)(row_b * stride) * d + kq * 8;
  sh_f16x8 bpre[SH_PRE][2];
#pragma unroll
  for (int s_ = 0; s_ < SH_PRE; ++s_) {
    const int ks = wv + SH_W * s_;
    if (ks < nks) {
      bpre[s_][0] = *reinterpret_cast<const sh_f16x8*>(ra + ks * 32);
      bpre[s_][1] = *reinterpret_cast<const sh_f16x8*>(rb + ks * 32);
    }
  }
  sh_f32x4 acc[SH_MT][2];
#pragma unroll
  for (int t = 0; t < SH_MT; ++t) {
    acc[t][0] = sh_f32x4{0.f, 0.f, 0.f, 0.f};
    acc[t][1] = sh_f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float inv_scale = 1.f;
  if constexpr (FR == 0) {
  // (1) the whole query block into registers, max |x| on the way
  float4 xv[SH_NX];
  uint32_t mx = 0;
#pragma unroll
  for (int j = 0; j < SH_NX; ++j) {
    const int idx = tid + SH_T * j;
    xv[j] = idx < n4 ? reinterpret_cast<const float4*>(X)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < SH_NX; ++j) {
    mx = max(max(mx, __float_as_uint(xv[j].x) & 0x7fffffffu), max(__float_as_uint(xv[j].y) & 0x7fffffffu, __float_as_uint(xv[j].z) & 0x7fffffffu));
    mx = max(mx, __float_as_uint(xv[j].w) & 0x7fffffffu);
  }
  TICK();
  mx = wave_max_u32_(mx);
  if (lane == 0) s_wmax[wv] = mx;
  // pad rows of the operand (m .. mpad): zero
  for (int j = tid; j < (mpad - m) * (lda >> 2); j += SH_T) reinterpret_cast<uint2*>(As + (size_t)m * lda)[j] = make_uint2(0u, 0u);
  __syncthreads();
#pragma unroll
  for (int v = 0; v < SH_W; ++v) mx = max(mx, s_wmax[v]);
  float scale = 1.f;
  {   // (query_f16_small_kernel's rule: the largest magnitude lands in [8192, 16384))
    const float maxabs = __uint_as_float(mx);
    if (maxabs > 0.f && isfinite(maxabs)) {
      int e;
      frexpf(maxabs, &e);
      scale = ldexpf(1.f, 14 - e);
    }
  }
  inv_scale = 1.f / (scale * db_scale);
  if (w == 0 && tid == 0) {
    scales[0] = scale;
    scales[1] = inv_scale;
  }
  if (w == (1 % NW))
    for (int j = tid; j < zero_words; j += SH_T) zero[j] = 0u;
  // (2) fp16 operand rows into LDS; every workgroup also writes its share of the plane the filter will read
  {
    int row = tid / d4, c4 = tid - row * d4;                 // float4 tid of the block; + 512 per step (no division per element)
    const int qstep = SH_T / d4, rstep = SH_T - qstep * d4;
#pragma unroll
    for (int j = 0; j < SH_NX; ++j) {
      const int idx = tid + SH_T * j;
      if (idx < n4) {
        sh_f16x4 h;
        h[0] = (_Float16)(xv[j].x * scale);
        h[1] = (_Float16)(xv[j].y * scale);
        h[2] = (_Float16)(xv[j].z * scale);
        h[3] = (_Float16)(xv[j].w * scale);
        *reinterpret_cast<sh_f16x4*>(As + (size_t)row * lda + c4 * 4) = h;
        if (j % NW == w) reinterpret_cast<sh_f16x4*>(qplane)[idx] = h;
      }
      row += qstep;
      c4 += rstep;
      if (c4 >= d4) {
        c4 -= d4;
        ++row;
      }
    }
  }
  __syncthreads();
  TICK();
  // (3) this wave's k-steps of the [mpad x 32] product
  auto kstep = [&](int ks, const sh_f16x8& b0, const sh_f16x8& b1) {
#pragma unroll
    for (int t = 0; t < SH_MT; ++t)
      if (t < MT) {
        const sh_f16x8 a = *reinterpret_cast<const sh_f16x8*>(As + (size_t)(t * 16 + i16) * lda + ks * 32 + kq * 8);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[t][1], 0, 0, 0);
      }
  };
#pragma unroll
  for (int s_ = 0; s_ < SH_PRE; ++s_) {
    const int ks = wv + SH_W * s_;
    if (ks < nks) kstep(ks, bpre[s_][0], bpre[s_][1]);
  }
  for (int ks = wv + SH_W * SH_PRE; ks < nks; ks += SH_W)
    kstep(ks, *reinterpret_cast<const sh_f16x8*>(ra + ks * 32), *reinterpret_cast<const sh_f16x8*>(rb + ks * 32));
  } else {
    // ---- fragment-direct: this wave's slice of the query block in MFMA A-fragment shape ----------------------------------------------
    constexpr int FRC = FR > 0 ? FR : 1, STEPS = 16 / FRC;
    const int steps_v = wv < nks ? (nks - wv + SH_W - 1) / SH_W : 0;   // k-steps wv, wv + 8, ... of this wave (<= STEPS: the launcher checks)
    float4 xf[FRC][STEPS][2];
    uint32_t mx = 0;
#pragma unroll
    for (int t = 0; t < FRC; ++t) {
      const int row = t * 16 + i16;
      const float* xr = X + (size_t)min(row, m - 1) * d + kq * 8;
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_)
        if (s_ < steps_v) {
          xf[t][s_][0] = *reinterpret_cast<const float4*>(xr + (wv + SH_W * s_) * 32);
          xf[t][s_][1] = *reinterpret_cast<const float4*>(xr + (wv + SH_W * s_) * 32 + 4);
        }
    }
#pragma unroll
    for (int t = 0; t < FRC; ++t)
#pragma unroll
      for (int s_ = 0; s_ < STEPS; ++s_)
        if (s_ < steps_v) {
          if (t * 16 + i16 >= m) xf[t][s_][0] = xf[t][s_][1] = make_float4(0.f, 0.f, 0.f, 0.f);   // pad rows of the last tile
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            mx = max(max(mx, __float_as_uint(xf[t][s_][h].x) & 0x7fffffffu), max(__float_as_uint(xf[t][s_][h].y) & 0x7fffffffu, __float_as_uint(xf[t][s_][h].z) & 0x7fffffffu));
            mx = max(mx, __float_as_uint(xf[t][s_][h].w) & 0x7fffffffu);
          }
        }
    mx = wave_max_u32_(mx);
    if (lane == 0) s_wmax[wv] = mx;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < SH_W; ++v) mx = max(mx, s_wmax[v]);
    float scale = 1.f;
    {
      const float maxabs = __uint_as_float(mx);
      if (maxabs > 0.f && isfinite(maxabs)) {
        int e;
        frexpf(maxabs, &e);
        scale = ldexpf(1.f, 14 - e);
      }
    }
    inv_scale = 1.f / (scale * db_scale);
    if (w == 0 && tid == 0) {
      scales[0] = scale;
      scales[1] = inv_scale;
    }
    if (w == (1 % NW))
      for (int j = tid; j < zero_words; j += SH_T) zero[j] = 0u;
#pragma unroll
    for (int s_ = 0; s_ < STEPS; ++s_)
      if (s_ < steps_v) {
        const int ks = wv + SH_W * s_;
        sh_f16x8 b0, b1;
        if (s_ < SH_PRE) {
          b0 = bpre[s_ < SH_PRE ? s_ : 0][0];
          b1 = bpre[s_ < SH_PRE ? s_ : 0][1];
        } else {
          b0 = *reinterpret_cast<const sh_f16x8*>(ra + ks * 32);
          b1 = *reinterpret_cast<const sh_f16x8*>(rb + ks * 32);
        }
#pragma unroll
        for (int t = 0; t < FRC; ++t) {
          sh_f16x8 a;
          a[0] = (_Float16)(xf[t][s_][0].x * scale);
          a[1] = (_Float16)(xf[t][s_][0].y * scale);
          a[2] = (_Float16)(xf[t][s_][0].z * scale);
          a[3] = (_Float16)(xf[t][s_][0].w * scale);
          a[4] = (_Float16)(xf[t][s_][1].x * scale);
          a[5] = (_Float16)(xf[t][s_][1].y * scale);
          a[6] = (_Float16)(xf[t][s_][1].z * scale);
          a[7] = (_Float16)(xf[t][s_][1].w * scale);
          // every workgroup writes its share of the plane the filter will read: fragment (wave, tile, step) by workgroup (...) % NW
          const int row = t * 16 + i16;
          if (row < m && ((wv * 16 + t * STEPS + s_) % NW) == w) *reinterpret_cast<sh_f16x8*>(qplane + (size_t)row * d + ks * 32 + kq * 8) = a;
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b0, acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b1, acc[t][1], 0, 0, 0);
        }
      }
  }
  __syncthreads();   // everybody is done with the operand: its LDS becomes the reduction block
  TICK();
  // (4) the waves' partial products -> LDS -> v = ||r||^2 - 2 q.r / (s_q s_db) per (query, column)
  float* red = reinterpret_cast<float*>(smem);                         // [SH_W][MT][2][64][4]
  float* val = red + (size_t)SH_W * SH_MT * 2 * 256;                     // [mpad][SH_RW]
#pragma unroll
  for (int t = 0; t < SH_MT; ++t)
    if (t < MT) {
      *reinterpret_cast<sh_f32x4*>(red + ((size_t)(wv * SH_MT + t) * 2 + 0) * 256 + lane * 4) = acc[t][0];
      *reinterpret_cast<sh_f32x4*>(red + ((size_t)(wv * SH_MT + t) * 2 + 1) * 256 + lane * 4) = acc[t][1];
    }
  __syncthreads();
  for (int o = tid; o < MT * 2 * 256; o += SH_T) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < SH_W; ++v) sum += red[(size_t)v * SH_MT * 2 * 256 + o];
    const int r = o & 3, ln = (o >> 2) & 63, rt = (o >> 8) & 1, t = o >> 9;
    const int q = t * 16 + 4 * (ln >> 4) + r, col = rt * 16 + (ln & 15);
    const int64_t srow = (int64_t)w * SH_RW + col;
    val[q * SH_RW + col] = srow < n0 ? fmaf(-2.f * inv_scale, sum, rn[srow * stride]) : INFINITY;
  }
  __syncthreads();
  TICK();
  // (5) per query the SMALLEST of this workgroup's 32 columns -> the query's slot of this workgroup.  (The rank-th smallest of
  //     the workgroups' minima is >= the rank-th smallest sample value -- equal unless two of the smallest share a workgroup -- so the
  //     threshold can only come out a little more generous, never tighter; and the last workgroup ranks 123 values per query, not 861.)
  for (int o = tid; o < m * 4; o += SH_T) {
    const int q = o >> 2, part = o & 3;
    const float4 u0 = *reinterpret_cast<const float4*>(val + q * SH_RW + part * 8), u1 = *reinterpret_cast<const float4*>(val + q * SH_RW + part * 8 + 4);
    float v = fminf(fminf(fminf(u0.x, u0.y), fminf(u0.z, u0.w)), fminf(fminf(u1.x, u1.y), fminf(u1.z, u1.w)));
    v = fminf(v, __shfl_xor(v, 1));
    v = fminf(v, __shfl_xor(v, 2));
    if (part == 0) __hip_atomic_store(&cand[(size_t)q * NW + w], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  TICK();
  // (6) squared norms of this workgroup's share of the rows: row_sumsq_kernel's arithmetic, lane for lane (gemm_kernels.hip) -- the
  //     exact refinement forms its distances with them
  if (wv == 0)
    for (int row = w; row < m; row += NW) {
      const float4* x4 = reinterpret_cast<const float4*>(X + (size_t)row * d);
      float s_ = 0.f;
      for (int j = lane; j < d4; j += 64) {
        const float4 v = x4[j];
        s_ = fmaf(v.x, v.x, s_);
        s_ = fmaf(v.y, v.y, s_);
        s_ = fmaf(v.z, v.z, s_);
        s_ = fmaf(v.w, v.w, s_);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s_ += __shfl_xor(s_, o);
      if (lane == 0) __hip_atomic_store(&qn_out[row], s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  TICK();
  // (7) grid barrier, then every workgroup ranks the minima of ITS queries (q = w, w + NW, ...: at most one or two, one wave each).
  //     A single last workgroup ranking all m queries -- seven per wave, each a dependent chain of ~3000 cycles -- took 10-16 us;
  //     spread over the grid it is one chain.  The barrier is a 64-bit arrival counter that only ever grows (no reset, no ABA):
  //     this workgroup's target is the end of its own generation.  All workgroups of the launch are resident together (<= 128
  //     workgroups, one per CU); should they ever not be -- or should anything else go wrong -- the spin is BOUNDED and the rows get
  //     a threshold of -inf instead: their candidate lists stay empty, the pass flags them and small_tail_kernel finishes them
  //     exactly.  Nothing can hang, nothing can come out wrong.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned long long mine = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (mine / (unsigned long long)NW + 1ull) * (unsigned long long)NW;
    int ok = 0;
    for (int spin = 0; spin < (1 << 20); ++spin) {
      if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    s_last = ok;
  }
  __syncthreads();
  const bool arrived = s_last != 0;
  if (repair && w == 0 && tid == 0) __hip_atomic_store(&rtick[rtick_poison], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  TICK();
  constexpr uint32_t PAD = 0xffffffffu;
  if (wv == 0)
    for (int q = w; q < m; q += NW) {
      uint32_t kk[SH_PER];
#pragma unroll
      for (int j = 0; j < SH_PER; ++j) {
        const int c = lane + 64 * j;
        kk[j] = c < NW ? f2key_(__hip_atomic_load(&cand[(size_t)q * NW + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : PAD;
      }
      const float q2 = __hip_atomic_load(&qn_out[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t wmin = PAD;
      for (int r = 0; r < rank; ++r) {
        uint32_t lmin = PAD;
#pragma unroll
        for (int j = 0; j < SH_PER; ++j) lmin = min(lmin, kk[j]);
        wmin = wave_min_u32_(lmin);
        const uint64_t owners = __builtin_amdgcn_ballot_w64(lmin == wmin);
        if (lane == (int)__builtin_ctzll(owners)) {   // ONE instance leaves (duplicates count one by one)
          bool gone = false;
#pragma unroll
          for (int j = 0; j < SH_PER; ++j) {
            const bool hit = !gone && kk[j] == wmin;
            kk[j] = hit ? PAD : kk[j];
            gone = gone || hit;
          }
        }
      }
      if (lane == 0) {
        thr_out[q] = (!arrived || wmin == PAD || NW < rank) ? -INFINITY : fmaxf(q2 + key2f_(wmin), 0.f);
        cnt[q] = 0u;   // the filter appends from zero
      }
    }
  TICK();
#ifdef SV_HEAD_TIMING
  if (dbg && tid == 0)
    printf("head wg %d: load %llu conv %llu mfma %llu red %llu min %llu qn %llu barrier %llu merge %llu cycles\n", w, T[1]-T[0], T[2]-T[1], T[3]-T[2], T[4]-T[3], T[5]-T[4], T[6]-T[5], T[7]-T[6], T[8]-T[7]);
#endif
#undef TICK
}

// true when the fused head takes this shape (else: query_f16_small_kernel + the exact sample level of rounds 3-5)
bool sv_small_head_ok(int m, int d, int n0, int rank) {
  const int nw = (n0 + SH_RW - 1) / SH_RW;
  return m >= 1 && m <= 16 * SH_MT && d % 64 == 0 && (int64_t)m * d <= (int64_t)SH_NX * SH_T * 4 && rank >= 1 && nw >= 2 &&
         nw <= 64 * SH_PER && nw >= 4 * rank;
}

int sv_small_words(segvlad_ctx* ctx) {   // [129] tail tickets, [2] tail totals, 1 pad, [2] the head's 64-bit arrival counter
  const size_t tcap = ctx->s_tail_tick.cap;
  SV_HIP(ctx->s_tail_tick.reserve((size_t)(129 + 2 + 1 + 2) * 4));
  if (ctx->s_tail_tick.cap != tcap) SV_HIP(hipMemsetAsync(ctx->s_tail_tick.p, 0, ctx->s_tail_tick.cap, ctx->stream));
  return SEGVLAD_OK;
}

int sv_launch_small_head(segvlad_ctx* ctx, const float* X, int m, int d, const uint16_t* Rh, const float* rn, int64_t stride, int n0,
                         float db_scale, int rank, uint16_t* qplane, float* scales_dev, float* qn_out, uint32_t* zero, int zero_words,
                         float* cand_scratch, float* thr_out, uint32_t* cand_cnt) {
  if (!sv_small_head_ok(m, d, n0, rank)) return ctx->fail(SEGVLAD_ERR_LIMIT, "small head: m=%d d=%d n0=%d rank=%d", m, d, n0, rank);
  SV_TRY(sv_small_words(ctx));
  const int nw = (n0 + SH_RW - 1) / SH_RW, mpad = (m + 15) & ~15;
  const size_t lds_a = (size_t)mpad * (d + 8) * 2, lds_r = ((size_t)SH_W * SH_MT * 2 * 256 + (size_t)mpad * SH_RW) * 4;
  // fragment-direct form when a lane's slice of the query block fits 16 fragments (FR tiles x ceil(d / 256) k-steps per wave)
  int fr = 1;
  while (fr * 16 < mpad) fr <<= 1;
  const int steps = ((d >> 5) + SH_W - 1) / SH_W;
  // (measured, m = 50, d = 1024, 1 M rows: the fragment-direct form 22.7 us against 20.8-21.4 us for the LDS-staged one -- its loads are
  //  16 rows x 32 B per quarter wave instead of whole lines, which costs what the LDS stage and its barrier saved: option
  //  small_head = 3 selects it, the default stays the staged form)
  const bool frag = fr * steps <= 16 && ctx->opt.small_head == 3;
  const size_t lds = frag ? lds_r : (lds_a > lds_r ? lds_a : lds_r);
  if (lds > 160 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "small head: %zu bytes of LDS", lds);
  auto kern = !frag ? small_head_kernel<0> : fr == 1 ? small_head_kernel<1> : fr == 2 ? small_head_kernel<2> : fr == 4 ? small_head_kernel<4> : small_head_kernel<8>;
  SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), lds));
  hipLaunchKernelGGL(kern, dim3(nw), dim3(SH_T), lds, ctx->stream, X, m, d, reinterpret_cast<const _Float16*>(Rh), rn, stride, n0,
                     db_scale, rank, reinterpret_cast<_Float16*>(qplane), scales_dev, qn_out, zero, zero_words, cand_scratch,
                     reinterpret_cast<unsigned long long*>(ctx->s_tail_tick.as<uint32_t>() + 132), thr_out, cand_cnt,
#ifdef SV_HEAD_TIMING
                     1,
#else
                     0,
#endif
                     ctx->s_ref_tick.as<uint32_t>(), 128, ctx->s_ref_keys.as<uint64_t>(), (int64_t)(ctx->s_ref_keys.cap / 8));
  ctx->small_head_ran = true;
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ---- the sampled level of a BATCH search from the filter's own fp16 product (round 6) --------------------------------------------------
// A batch search's coarsest level only has to hand a guessed threshold to the first filter level (the guesses are verified at the end,
// DESIGN.md 4 item 4), so -- like the single-image head above -- it needs no exact distances: dist[q][j] = d2~(q, sample row j) from one
// fp16 MFMA product of the query plane and the strided rows of the database plane, the rank select that follows is the one the exact
// level used.  10 000 queries x 244 (1 M rows) / 488 (a 125 k-row shard) sample rows: ~15 us instead of 115 / 162 us of fp32 MFMA
// (gemm_nt_kernel<1>, poorly filled at that width).  No LDS: a wave owns 16 query rows x 32 sample rows, its fragments come straight
// from global memory (the query plane is L2 resident, the sample is half a megabyte).
__global__ __launch_bounds__(256) void sample_f16_batch_kernel(const _Float16* __restrict__ Qh, const _Float16* __restrict__ Rh, int m, int n0,
                                                               int d, int64_t stride, float inv_scale, const float* __restrict__ qn,
                                                               const float* __restrict__ rn, float* __restrict__ dist, int64_t ld) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i16 = lane & 15, kq = lane >> 4;
  const int q0 = (int)blockIdx.y * 64 + wv * 16, c0 = (int)blockIdx.x * SH_RW;
  if (q0 >= m) return;
  const _Float16* qa = Qh + (size_t)min(q0 + i16, m - 1) * d + kq * 8;
  const _Float16* ra = Rh + (size_t)((int64_t)min(c0 + i16, n0 - 1) * stride) * d + kq * 8;
  const _Float16* rb = Rh + (size_t)((int64_t)min(c0 + 16 + i16, n0 - 1) * stride) * d + kq * 8;
  sh_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int nks = d >> 5;
  int ks = 0;
  for (; ks + 4 <= nks; ks += 4) {   // twelve 16-byte loads in flight per lane
    sh_f16x8 a[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = *reinterpret_cast<const sh_f16x8*>(qa + (ks + u) * 32);
      b0[u] = *reinterpret_cast<const sh_f16x8*>(ra + (ks + u) * 32);
      b1[u] = *reinterpret_cast<const sh_f16x8*>(rb + (ks + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b0[u], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b1[u], acc1, 0, 0, 0);
    }
  }
  for (; ks < nks; ++ks) {
    const sh_f16x8 a = *reinterpret_cast<const sh_f16x8*>(qa + ks * 32);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, *reinterpret_cast<const sh_f16x8*>(ra + ks * 32), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, *reinterpret_cast<const sh_f16x8*>(rb + ks * 32), acc1, 0, 0, 0);
  }
  // lane (i16, kq) holds rows 4 kq + r (queries), column i16 (sample row) of each tile
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = c0 + 16 * t + i16;
    if (col >= n0) continue;
    const float r2 = rn[(int64_t)col * stride];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * kq + r;
      if (q < m) dist[(size_t)q * ld + col] = sv_d2(qn[q], r2, (t == 0 ? acc0[r] : acc1[r]) * inv_scale);
    }
  }
}

int sv_launch_sample_f16_batch(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Rh, int m, int n0, int d, int64_t stride, float inv_scale,
                               const float* qn, const float* rn, float* dist, int64_t ld) {
  if (m <= 0 || n0 <= 0) return SEGVLAD_OK;
  if (d % 32) return ctx->fail(SEGVLAD_ERR_LIMIT, "sample_f16_batch: d=%d", d);
  hipLaunchKernelGGL(sample_f16_batch_kernel, dim3((n0 + SH_RW - 1) / SH_RW, (m + 63) / 64), dim3(256), 0, ctx->stream,
                     reinterpret_cast<const _Float16*>(Qh), reinterpret_cast<const _Float16*>(Rh), m, n0, d, stride, inv_scale, qn, rn, dist, ld);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// tests only (option debug_small_tail): force the tail's paths on a pass that needed none of them
__global__ void small_tail_debug_kernel(int bits, int m, uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, float* ref_lim,
                                        uint32_t* tick, int tick_poison) {
  const int row = threadIdx.x;
  if (row >= m) return;
  if ((bits & 1) && atomicExch(&fail_rows[row], 1u) == 0u) atomicAdd(&fail_count[0], 1u);
  if ((bits & 2) && !rovf_rows[row] && !(bits & 1)) {   // the whole candidate list as the band: a superset, same top k
    rovf_rows[row] = 1u;
    ref_lim[row] = INFINITY;
    atomicAdd(&fail_count[1], 1u);
  }
  if ((bits & 4) && row == 0 && tick) {
    tick[tick_poison] = 1u;
    if (atomicExch(&fail_rows[0], 1u) == 0u) atomicAdd(&fail_count[0], 1u);
  }
}

int sv_launch_small_tail_debug(segvlad_ctx* ctx, int m, uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, float* ref_lim) {
  if ((ctx->opt.debug_small_tail & 7) == 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(small_tail_debug_kernel, dim3(1), dim3(128), 0, ctx->stream, ctx->opt.debug_small_tail & 7, m, fail_rows, fail_count, rovf_rows,
                     ref_lim, ctx->s_ref_tick.as<uint32_t>(), 128);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// part: [m][G][kp] words, tickets: [129] words (all zero between launches), stats: [4] words of THIS search (zeroed by the pass's
// first kernel), totals: [2] words accumulated over the context's life, host_totals: their pinned mirror (or null)
int sv_launch_small_tail(segvlad_ctx* ctx, const float* Q, const float* R, const float* qn, const float* rn, int64_t n, int d, int m, int k,
                         uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, const float* ref_lim, const uint32_t* cand_cnt,
                         const float* cand_d2, const uint32_t* cand_id, int cap, float* d2_out, int64_t* idx_out, uint32_t* stats) {
  if (m <= 0) return SEGVLAD_OK;
  SV_TRY(sv_launch_small_tail_debug(ctx, m, fail_rows, fail_count, rovf_rows, const_cast<float*>(ref_lim)));
  if (m > 128 || k > 1024 || cap > ST_CAP || (d & 3)) return ctx->fail(SEGVLAD_ERR_LIMIT, "small tail: m=%d k=%d cap=%d d=%d", m, k, cap, d);
  const int G = (ctx->opt.debug_small_tail >> 8) > 0 ? (ctx->opt.debug_small_tail >> 8) : 64;   // (development: bits 8.. = grid size A/B)
  int kp = 256;
  while (kp < k) kp <<= 1;
  SV_HIP(ctx->s_tail_part.reserve((size_t)m * G * kp * 8));
  SV_TRY(sv_small_words(ctx));
  SV_TRY(sv_ensure_pinned_words(ctx));
  const size_t lds = (size_t)ST_CAP * 8 + (size_t)ST_QB * 1024 * 8 + (size_t)ST_QB * ST_KC * 4;
  SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(small_tail_kernel), lds));
  uint32_t* tickets = ctx->s_tail_tick.as<uint32_t>();
  hipLaunchKernelGGL(small_tail_kernel, dim3(G), dim3(256), lds, ctx->stream, Q, R, qn, rn, n, d, m, k, fail_rows, fail_count, rovf_rows,
                     ref_lim, cand_cnt, cand_d2, cand_id, cap, d2_out, idx_out, ctx->s_tail_part.as<uint64_t>(), tickets, kp,
                     ctx->s_ref_keys.as<uint64_t>(), (int64_t)(ctx->s_ref_keys.cap / 8), ctx->s_ref_tick.as<uint32_t>(),
                     ctx->s_ref_tick.p ? 129 : 0, 128, stats, tickets + 129, ctx->h_pin + 8);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
