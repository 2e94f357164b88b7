// Device helpers of the single-image pass's exact finish (small_pass_kernels.hip: small_tail_kernel; knn_filter_kernels.hip:
// refine_exact_small_kernel's fused finish) -- round 6.  Everything here evaluates a distance as the sequential chain
// acc = fma(q[j], r[j], acc), j = 0 .. d-1, then sv_d2 with the same norms, and orders by (distance, id): the bits of every exact path.
#pragma once
#include <math.h>
#include <stdint.h>

#include "ctx.h"
#include "knn_dev.h"

namespace {

constexpr int ST_QB = 4;        // flagged rows evaluated per sweep over a workgroup's slice of the index
constexpr int ST_KC = 1024;     // floats of a query row staged in LDS per step
constexpr int ST_CAP = 8192;    // longest candidate list (SV_CAP)

// exact chains of ONE index row against nb <= ST_QB staged query chunks (qs[b][0..kc)): acc[b] = fma(q[j], r[j], acc[b]) in j order
template <int NB>
__device__ __forceinline__ void chain_step(const float* __restrict__ rrow, int kc, const float* __restrict__ qs, float (&acc)[NB]) {
  const float4* rp = reinterpret_cast<const float4*>(rrow);
  const int n4 = kc >> 2;
  int t = 0;
  for (; t + 16 <= n4; t += 16) {
    float4 buf[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) buf[u] = rp[t + u];
    __builtin_amdgcn_sched_barrier(0);   // all sixteen loads are issued before the first fma
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + b * ST_KC + (t + u) * 4);
        acc[b] = fmaf(qv.x, buf[u].x, acc[b]);
        acc[b] = fmaf(qv.y, buf[u].y, acc[b]);
        acc[b] = fmaf(qv.z, buf[u].z, acc[b]);
        acc[b] = fmaf(qv.w, buf[u].w, acc[b]);
      }
    }
  }
  for (; t < n4; ++t) {
    const float4 rv = rp[t];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float4 qv = *reinterpret_cast<const float4*>(qs + b * ST_KC + t * 4);
      acc[b] = fmaf(qv.x, rv.x, acc[b]);
      acc[b] = fmaf(qv.y, rv.y, acc[b]);
      acc[b] = fmaf(qv.z, rv.z, acc[b]);
      acc[b] = fmaf(qv.w, rv.w, acc[b]);
    }
  }
}

// best[0..kp) (ascending, padded with all ones) <- the kp smallest of best U {mine of the 256 threads}; a: >= kp + 256 words of
// scratch, sort length ns = the power of two holding kp + 256.  Skipped (workgroup-uniformly) when nobody brings a key below
// best[k - 1].
__device__ __forceinline__ void merge_best(uint64_t* __restrict__ best, int kp, int k, uint64_t mine, uint64_t* __restrict__ a, int ns,
                                           int tid) {
  const bool better = mine < best[k - 1];
  if (!__syncthreads_or(better ? 1 : 0)) return;
  for (int j = tid; j < kp; j += 256) a[j] = best[j];
  a[kp + tid] = better ? mine : ~0ull;
  for (int j = kp + 256 + tid; j < ns; j += 256) a[j] = ~0ull;
  bitonic64(a, ns, tid);
  for (int j = tid; j < kp; j += 256) best[j] = a[j];
  __syncthreads();
}

}  // namespace

// what refine_exact_small_kernel needs to finish flagged rows itself (passed by value; on == 0: the kernel of rounds 3-5)
struct SvSmallFinish {
  int on = 0, debug = 0;
  uint32_t* rovf_rows = nullptr;
  const float* ref_lim = nullptr;
  const uint32_t* cand_cnt = nullptr;
  const float* cand_d2 = nullptr;
  const uint32_t* cand_id = nullptr;
  int cap = 0, kp = 0;
  int64_t n_db = 0, row_words = 0;     // rows of the index; words of `part2` per query row (>= max(parts * kp, cap))
  uint64_t* part2 = nullptr;           // exchange buffer between a row's workgroups
  uint32_t* stats = nullptr;           // [4] of this search: rows redone by brute force, second-tier rows, hand-overs recomputed
  uint32_t* totals = nullptr;          // [2] running totals of the context
  volatile uint32_t* host_totals = nullptr;   // their pinned mirror (or null)
};

namespace {

// ---- a flagged row finished by the row's OWN workgroups of the refinement kernel (P parts, part p of them) ------------------------------
// Exact brute force: part p scans rows [p n / P, (p + 1) n / P) of the index against query row `row`, keeps its k best keys and hands
// them to part p of the row's slot ([P][kp] words); returns with the slice's list in best[0 .. kp).  a: >= 2048 words of sort
// scratch, best: kp words, qs: ST_KC floats -- all LDS of the caller.
__device__ __forceinline__ void sp_brute_slice(const float* __restrict__ Q, const float* __restrict__ R, const float* __restrict__ qn,
                                               const float* __restrict__ rn, int64_t n, int d, int k, int kp, int64_t row, int p, int P,
                                               uint64_t* __restrict__ slot /* the ROW's [P][kp] words */, uint64_t* a, uint64_t* best, float* qs,
                                               int tid) {
  int ns = 512;
  while (ns < kp + 256) ns <<= 1;
  for (int j = tid; j < kp; j += 256) best[j] = ~0ull;
  __syncthreads();
  const int64_t per = (n + P - 1) / P;
  const int64_t r_lo = (int64_t)p * per, r_hi = min(n, r_lo + per);
  const float q2 = qn[row];
  for (int64_t base = r_lo; base < r_hi; base += 256) {
    const int64_t r = base + tid;
    const bool live = r < r_hi;
    const int64_t rr = live ? r : r_lo;
    float acc[1] = {0.f};
    for (int c0 = 0; c0 < d; c0 += ST_KC) {
      const int kc = min(ST_KC, d - c0);
      __syncthreads();
      for (int t = tid; t < kc; t += 256) qs[t] = Q[(size_t)row * d + c0 + t];
      __syncthreads();
      chain_step<1>(R + (size_t)rr * d + c0, kc, qs, acc);
    }
    const uint64_t key = live ? (((uint64_t)f2key_(sv_d2(q2, rn[rr], acc[0])) << 32) | (uint32_t)r) : ~0ull;
    merge_best(best, kp, k, key, a, ns, tid);
  }
  uint64_t* mine = slot + (size_t)p * kp;
  for (int j = tid; j < kp; j += 256) mine[j] = best[j];
}

// the P lists of a row merged into best[0 .. kp) (called by the workgroup that drew the row's last ticket, behind a __threadfence)
__device__ __forceinline__ void sp_brute_merge(const uint64_t* __restrict__ slot /* the ROW's [P][kp] words */, int k, int kp, int P, uint64_t* a,
                                               uint64_t* best, int tid) {
  int ns = 512;
  while (ns < kp + 256) ns <<= 1;
  for (int j = tid; j < kp; j += 256) best[j] = ~0ull;
  __syncthreads();
  const uint64_t* all = slot;
  for (int64_t j0 = 0; j0 < (int64_t)P * kp; j0 += 256) {
    const int64_t j = j0 + tid;
    merge_best(best, kp, k, j < (int64_t)P * kp ? __hip_atomic_load(&all[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull, a, ns, tid);
  }
}

// Second tier: part p evaluates the entries [p cap / P, (p + 1) cap / P) of the row's candidate list that lie in the band
// {d2~ <= lim} exactly and leaves their keys (all ones elsewhere) in keys_row[entry] (the ROW's cap words).
__device__ __forceinline__ void sp_tier2_slice(const float* __restrict__ Q, const float* __restrict__ R, const float* __restrict__ qn,
                                               const float* __restrict__ rn, int d, int64_t row, int p, int P, uint32_t c, float lim,
                                               const float* __restrict__ cand_d2, const uint32_t* __restrict__ cand_id, int cap,
                                               uint64_t* __restrict__ keys_row, float* qs, int tid) {
  const int per = (cap + P - 1) / P;
  const int j_lo = p * per, j_hi = min(cap, j_lo + per);
  const float q2 = qn[row];
  for (int j0 = j_lo; j0 < j_hi; j0 += 256) {
    const int j = j0 + tid;
    const bool in = j < j_hi && (uint32_t)j < c && cand_d2[(size_t)row * cap + j] <= lim;
    const uint32_t id = in ? cand_id[(size_t)row * cap + j] : 0u;
    float acc[1] = {0.f};
    for (int c0 = 0; c0 < d; c0 += ST_KC) {
      const int kc = min(ST_KC, d - c0);
      __syncthreads();
      for (int t = tid; t < kc; t += 256) qs[t] = Q[(size_t)row * d + c0 + t];
      __syncthreads();
      if (in) chain_step<1>(R + (size_t)id * d + c0, kc, qs, acc);
    }
    if (j < j_hi) keys_row[j] = in ? ((((uint64_t)f2key_(sv_d2(q2, rn[id], acc[0]))) << 32) | id) : ~0ull;
  }
}

}  // namespace
