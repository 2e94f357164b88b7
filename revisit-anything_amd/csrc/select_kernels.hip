// Exact top-k selection over a row of squared distances (faiss.IndexFlatL2.search semantics:
// ascending, place_rec_main.py:56-60), shard merge, "2 - d^2" slice (place_rec_main.py:78-81) and the
// global min/max (func_vpr.py:212-213).
//
// select_topk_kernel: one workgroup per query row.  MSB-first 8-bit radix select on the order-
// preserving integer image of the distances finds the k-th smallest key (histograms in LDS, with
// wave-aggregated atomics: distances share their exponent byte, so a naive LDS histogram would
// serialise 64-way); one more pass collects the k candidates, a bitonic sort on (key, index) orders
// them.  Ties are resolved towards the lower index, like a stable argsort.
#include "ctx.h"

__device__ __forceinline__ uint32_t f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// wave-aggregated histogram increment: lanes that share a bin elect one leader per round
__device__ __forceinline__ void hist_add(uint32_t* hist, bool active, uint32_t bin) {
  uint64_t todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const uint32_t lb = __shfl(bin, leader);
    const uint64_t same = __ballot(active && bin == lb) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

template <class T>
__device__ __forceinline__ void bitonic_sort_lds(T* a, int n /*pow2*/, int tid, int nthreads) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (n >> 1); t += nthreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const T x = a[lo], y = a[hi];
        if ((y < x) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

// STAGED: the row (n <= 12288 distances: the sampled level of the large-database search) is converted to keys once
// and kept in LDS, so the four radix passes and the collection pass do not go back to L2/HBM
template <bool STAGED>
__global__ __launch_bounds__(256) void select_topk_kernel(const float* __restrict__ dist, int64_t ld, int64_t n, int k,
                                                          int kpad, float* __restrict__ d2_out,
                                                          int64_t* __restrict__ idx_out, int64_t out_ld,
                                                          int64_t id_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* cand = reinterpret_cast<uint64_t*>(smem);  // [kpad]
  uint32_t* skey = reinterpret_cast<uint32_t*>(cand + kpad);  // [n] when STAGED
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_digit, s_krem, s_ceq, s_nless, s_neq;
  __shared__ uint32_t wcnt[4];
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* x = dist + row * ld;
  const int kk = (int)((int64_t)k < n ? k : n);
  if (STAGED) {
    for (int64_t j = tid; j < n; j += 256) skey[j] = f2key(x[j]);
    __syncthreads();
  }
  auto key_at = [&](int64_t j) -> uint32_t { return STAGED ? skey[j] : f2key(x[j]); };

  uint32_t prefix = 0, mask = 0, krem = (uint32_t)kk, ceq = 0;
  if (kk > 0) {
    for (int pass = 3; pass >= 0; --pass) {
      hist[tid] = 0;
      __syncthreads();
      const int shift = 8 * pass;
      for (int64_t j0 = 0; j0 < n; j0 += 256) {
        const int64_t j = j0 + tid;
        bool act = j < n;
        uint32_t key = act ? key_at(j) : 0u;
        act = act && ((key & mask) == prefix);
        hist_add(hist, act, (key >> shift) & 255u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t cum = 0, dsel = 255;
        for (uint32_t b = 0; b < 256; ++b) {
          const uint32_t h = hist[b];
          if (cum + h >= krem) {
            dsel = b;
            break;
          }
          cum += h;
        }
        s_digit = dsel;
        s_krem = krem - cum;
        s_ceq = hist[dsel];
      }
      __syncthreads();
      prefix |= s_digit << shift;
      mask |= 255u << shift;
      krem = s_krem;
      ceq = s_ceq;
      __syncthreads();
    }
  }
  // prefix = k-th smallest key; krem = how many elements equal to it belong to the top-k; ceq = how many exist
  for (int j = tid; j < kpad; j += 256) cand[j] = ~0ull;
  if (tid == 0) {
    s_nless = 0;
    s_neq = 0;
  }
  __syncthreads();
  if (kk > 0) {
    const uint32_t nless = (uint32_t)kk - krem;
    if (ceq == krem) {
      // common case: every element equal to the threshold is taken -> unordered collection
      for (int64_t j0 = 0; j0 < n; j0 += 256) {
        const int64_t j = j0 + tid;
        if (j < n) {
          const uint32_t key = key_at(j);
          if (key <= prefix) {
            const uint32_t slot = atomicAdd(&s_nless, 1u);
            cand[slot] = ((uint64_t)key << 32) | (uint32_t)j;
          }
        }
      }
    } else {
      // boundary ties: take the krem LOWEST indices among the equal elements (ordered scan)
      for (int64_t j0 = 0; j0 < n; j0 += 256) {
        const int64_t j = j0 + tid;
        const bool in = j < n;
        const uint32_t key = in ? key_at(j) : ~0u;
        if (in && key < prefix) {
          const uint32_t slot = atomicAdd(&s_nless, 1u);
          cand[slot] = ((uint64_t)key << 32) | (uint32_t)j;
        }
        const bool eq = in && key == prefix;
        // ordered rank of this equal element: previous chunks (s_neq) + lower threads of this chunk
        const uint64_t bal = __ballot(eq);
        if ((tid & 63) == 0) wcnt[tid >> 6] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = s_neq;
        for (int ww = 0; ww < (tid >> 6); ++ww) before += wcnt[ww];
        before += (uint32_t)__popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (eq && before < krem) cand[nless + before] = ((uint64_t)key << 32) | (uint32_t)j;
        __syncthreads();
        if (tid == 0) s_neq += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
      }
    }
  }
  __syncthreads();
  bitonic_sort_lds(cand, kpad, tid, 256);
  for (int j = tid; j < k; j += 256) {
    float d = INFINITY;
    int64_t id = -1;
    if (j < kk) {
      const uint64_t c = cand[j];
      d = key2f((uint32_t)(c >> 32));
      id = id_base + (int64_t)(uint32_t)c;
    }
    d2_out[row * out_ld + j] = d;
    idx_out[row * out_ld + j] = id;
  }
}

static int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

int sv_launch_select_topk(segvlad_ctx* ctx, const float* dist, int64_t ld, int nq, int64_t n, int k, float* d2_out,
                          int64_t* idx_out, int64_t out_ld, int64_t id_base) {
  if (nq <= 0) return SEGVLAD_OK;
  if (n >= (1ll << 32)) return ctx->fail(SEGVLAD_ERR_LIMIT, "select: more than 2^32-1 rows per shard");
  const int kpad = next_pow2(k < 2 ? 2 : k);
  if (n <= 12288)
    hipLaunchKernelGGL(select_topk_kernel<true>, dim3(nq), dim3(256), (size_t)kpad * 8 + (size_t)n * 4, ctx->stream, dist, ld, n, k,
                       kpad, d2_out, idx_out, out_ld, id_base);
  else
    hipLaunchKernelGGL(select_topk_kernel<false>, dim3(nq), dim3(256), (size_t)kpad * 8, ctx->stream, dist, ld, n, k, kpad, d2_out,
                       idx_out, out_ld, id_base);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// top-k of a per-query candidate list produced by the filtered GEMM epilogue (unordered, <= cap
// entries).  (key, id) pairs are sorted in LDS, so the result does not depend on the order in which
// the atomics handed out the slots.  A list that overflowed its capacity flags its query row
// (ovf_rows, counted once in *ovf_count) and the caller redoes that row alone on the matrix path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_cand_kernel(const uint32_t* __restrict__ cnt,
                                                          const float* __restrict__ cd2,
                                                          const uint32_t* __restrict__ cid, int cap, int k,
                                                          float* __restrict__ d2_out, int64_t* __restrict__ idx_out,
                                                          uint32_t* __restrict__ ovf_rows, uint32_t* __restrict__ ovf_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const uint32_t c = cnt[row];
  if (c > (uint32_t)cap) {
    if (tid == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      // this query is redone on the matrix path: a threshold of -inf keeps its list empty at the finer levels
      if (!idx_out) d2_out[row * k + (k - 1)] = -INFINITY;
    }
    return;
  }
  if (ovf_rows[row]) return;   // flagged at a coarser level: its result comes from the fallback
  int npad = 2;
  while (npad < (int)c) npad <<= 1;
  for (int j = tid; j < npad; j += 256)
    a[j] = (j < (int)c) ? (((uint64_t)f2key(cd2[row * cap + j]) << 32) | cid[row * cap + j]) : ~0ull;
  bitonic_sort_lds(a, npad, tid, 256);
  for (int j = tid; j < k; j += 256) {
    float d = INFINITY;
    int64_t id = -1;
    if (j < (int)c) {
      d = key2f((uint32_t)(a[j] >> 32));
      id = (int64_t)(uint32_t)a[j];
    }
    d2_out[row * k + j] = d;
    if (idx_out) idx_out[row * k + j] = id;
  }
}

int sv_launch_select_cand(segvlad_ctx* ctx, const uint32_t* cand_cnt, const float* cand_d2, const uint32_t* cand_id,
                          int nq, int cap, int k, float* d2_out, int64_t* idx_out, uint32_t* ovf_rows, uint32_t* ovf_count) {
  if (nq <= 0) return SEGVLAD_OK;
  const size_t lds = (size_t)cap * 8;
  if (lds > 128 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "select_cand: cap=%d exceeds the LDS sort", cap);
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(select_cand_kernel), (size_t)lds));
  hipLaunchKernelGGL(select_cand_kernel, dim3(nq), dim3(256), lds, ctx->stream, cand_cnt, cand_d2, cand_id, cap, k, d2_out,
                     idx_out, ovf_rows, ovf_count);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// merge of per-shard lists: sort (key, id) pairs; id -1 (empty slot) sorts last
// ------------------------------------------------------------------------------------------------
struct KeyId {
  uint32_t key;
  uint32_t pad;
  int64_t id;
  __device__ bool operator<(const KeyId& o) const { return key < o.key || (key == o.key && id < o.id); }
};

__global__ __launch_bounds__(256) void merge_topk_kernel(const float* __restrict__ d2p, const int64_t* __restrict__ idp,
                                                         int cand, int cpad, int k, int kpart, float* __restrict__ d2_out,
                                                         int64_t* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  KeyId* a = reinterpret_cast<KeyId*>(smem);
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  for (int j = tid; j < cpad; j += 256) {
    KeyId e;
    e.key = ~0u;
    e.pad = 0;
    e.id = INT64_MAX;
    if (j < cand) {
      const int64_t id = idp[row * cand + j];
      if (id >= 0) {
        e.key = f2key(d2p[row * cand + j]);
        e.id = id;
      }
    }
    a[j] = e;
  }
  // The parts are per-shard top-k lists: each already ascending by (distance, id).  Then the global rank of an entry is its own
  // position plus, for every other part, the number of entries in front of it there -- a binary search per part, no sort and no
  // barrier (round 5: the 512-entry bitonic sort took 0.35 ms per 10 000 queries x 8 shards, a fixed cost of every rank's
  // step).  Parts that are NOT sorted (the entry point promises nothing) keep the sort.
  if (kpart > 0) {
    __syncthreads();
    bool ok = true;
    for (int j = tid; j < cand; j += 256)
      if (j % kpart != 0 && a[j] < a[j - 1]) ok = false;
    if (__syncthreads_and(ok ? 1 : 0)) {
      const int parts = cand / kpart;
      for (int j = tid; j < k; j += 256) {   // slots nobody will claim
        if (j >= cand) {
          d2_out[row * k + j] = INFINITY;
          idx_out[row * k + j] = -1;
        }
      }
      for (int j = tid; j < cand; j += 256) {
        const KeyId e = a[j];
        const int r = j / kpart;
        int rank = j - r * kpart;
        for (int r2 = 0; r2 < parts && rank < k; ++r2) {
          if (r2 == r) continue;
          // entries of part r2 in front of e: strictly smaller, or equal and from an earlier part (equal entries: padding)
          const KeyId* lst = a + r2 * kpart;
          int lo = 0, hi = kpart;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const bool before = r2 < r ? !(e < lst[mid]) : (lst[mid] < e);
            if (before) lo = mid + 1;
            else hi = mid;
          }
          rank += lo;
        }
        if (rank < k) {
          const bool valid = e.id != INT64_MAX;
          d2_out[row * k + rank] = valid ? key2f(e.key) : INFINITY;
          idx_out[row * k + rank] = valid ? e.id : -1;
        }
      }
      return;
    }
  }
  bitonic_sort_lds(a, cpad, tid, 256);
  for (int j = tid; j < k; j += 256) {
    float d = INFINITY;
    int64_t id = -1;
    if (j < cand && a[j].id != INT64_MAX) {
      d = key2f(a[j].key);
      id = a[j].id;
    }
    d2_out[row * k + j] = d;
    idx_out[row * k + j] = id;
  }
}

int sv_launch_merge_topk(segvlad_ctx* ctx, const float* d2_parts, const int64_t* idx_parts, int nq, int cand, int k,
                         float* d2_out, int64_t* idx_out) {
  if (nq <= 0) return SEGVLAD_OK;
  const int cpad = next_pow2(cand < 2 ? 2 : cand);
  const size_t lds = (size_t)cpad * sizeof(KeyId);
  if (lds > 160 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "merge: %d candidates per query exceed the LDS sort (max 8192)", cand);
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(merge_topk_kernel), (size_t)lds));
  // (every caller hands parts of k entries each: cand = parts * k)
  hipLaunchKernelGGL(merge_topk_kernel, dim3(nq), dim3(256), lds, ctx->stream, d2_parts, idx_parts, cand, cpad, k,
                     (k > 0 && cand % k == 0) ? k : 0, d2_out, idx_out);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
__global__ void sims_kernel(const float* __restrict__ d2, const int64_t* __restrict__ idx, int64_t total, int k_in,
                            int k_keep, float* __restrict__ sims, int64_t* __restrict__ idx_out) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= total) return;
  const int64_t q = o / k_keep;
  const int r = (int)(o - q * k_keep);
  sims[o] = 2.0f - d2[q * k_in + r];
  idx_out[o] = idx[q * k_in + r];
}

int sv_launch_sims(segvlad_ctx* ctx, const float* d2, const int64_t* idx, int nq, int k_in, int k_keep, float* sims,
                   int64_t* idx_out) {
  const int64_t total = (int64_t)nq * k_keep;
  if (total <= 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(sims_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d2, idx, total, k_in,
                     k_keep, sims, idx_out);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// global min/max: integer atomics on the order-preserving key (exact, order-independent)
// ------------------------------------------------------------------------------------------------
__global__ void minmax_init_kernel(uint32_t* mm) {
  mm[0] = ~0u;
  mm[1] = 0u;
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, int64_t n, uint32_t* mm) {
  uint32_t lo = ~0u, hi = 0u;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
    const uint32_t key = f2key(x[j]);
    lo = min(lo, key);
    hi = max(hi, key);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
  }
  // one atomic pair per workgroup (thousands of waves hammering two addresses cost 0.18 ms for 500 k values)
  __shared__ uint32_t wl[4], wh[4];
  if ((threadIdx.x & 63) == 0) {
    wl[threadIdx.x >> 6] = lo;
    wh[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&mm[0], min(min(wl[0], wl[1]), min(wl[2], wl[3])));
    atomicMax(&mm[1], max(max(wh[0], wh[1]), max(wh[2], wh[3])));
  }
}
__global__ void minmax_final_kernel(const uint32_t* mm, float* out) {
  out[0] = key2f(mm[0]);
  out[1] = key2f(mm[1]);
}

int sv_launch_minmax(segvlad_ctx* ctx, const float* sims, int64_t count, float* minmax_dev) {
  SV_HIP(ctx->s_minmax.reserve(2 * sizeof(uint32_t)));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>();
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, ctx->stream, mm);
  if (count > 0) {
    int blocks = (int)((count + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sims, count, mm);
  }
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(1), 0, ctx->stream, mm, minmax_dev);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
