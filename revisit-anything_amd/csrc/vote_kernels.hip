// Similarity-weighted image vote (get_matches "max_seg_topk_wt_borda_Im" + weighted_borda_count,
// func_vpr.py:61-77, 207-224) and the integer variant ("max_seg_topk", func_vpr.py:118-125).
//
// One workgroup per query image.  The reference walks the (rank, segment) entries rank-major and
// adds python floats (fp64) into a dict keyed by reference IMAGE id, then sorts the keys by score,
// descending and stable (ties keep first-appearance order).  To reproduce the fp64 sums bit for bit
// the entries are sorted by (image id, visiting order) in LDS and every distinct image is summed
// sequentially, in visiting order, by one thread.  The weights are formed in fp32 exactly as NumPy
// does: (s - s_min) / (s_max - s_min) with the GLOBAL extrema.
//
// Fast path (images of <= 4096 entries): when every weight of a run is 0 or lies in [2^-17, 1] -- a multiple of 2^-40 --
// every partial sum of <= 4096 of them is exactly representable in fp64, so the run's sum does not depend on the order
// of the additions: the entries are then added with LDS fp64 atomics in parallel (a query image whose matches concentrate
// on one place has runs of hundreds of entries: the sequential walk was 0.3 of the kernel's 0.43 ms on the bench).  A run
// holding any other weight (or a NaN) is poisoned and summed sequentially as before.  Same bits either way.
#include "ctx.h"

struct Run {
  double score;
  uint32_t first;  // visiting order of the first appearance
  int32_t img;
};

__device__ __forceinline__ void bitonic_u64(uint64_t* a, int n, int tid, int nthreads) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (n >> 1); t += nthreads) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t x = a[lo], y = a[hi];
        if ((y < x) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

// better(a, b): a ranks before b
__device__ __forceinline__ bool better(double sa, uint32_t ta, double sb, uint32_t tb) {
  return sa > sb || (sa == sb && ta < tb);
}

constexpr int VOTE_THREADS = 1024;

// GLOBAL = false: the (image id, visiting order) keys of one query image are sorted in LDS (<= 16384 entries, i.e. 327
// segments at k = 50); images with more entries are skipped.  GLOBAL = true: the workgroup handles query image
// img_list[blockIdx.x] with its keys in a global scratch row (any size: SAM's automatic generator can return many hundreds
// of segments on cluttered images) -- same algorithm, same bits, slower.
template <bool GLOBAL>
__global__ __launch_bounds__(VOTE_THREADS) void vote_kernel(const int64_t* __restrict__ idx, const float* __restrict__ sims,
                                                   const int32_t* __restrict__ img_of_seg, int64_t n_ref_seg,
                                                   const int32_t* __restrict__ qoff, int k,
                                                   const float* __restrict__ minmax, int n_top, int mode,
                                                   Run* __restrict__ runs_scratch, int64_t runs_stride,
                                                   int32_t* __restrict__ pred, double* __restrict__ score, int use_wl,
                                                   int e_lds_cap, const int32_t* __restrict__ img_list,
                                                   uint64_t* __restrict__ gkeys, int64_t gkeys_stride, int fast) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = GLOBAL ? gkeys + (int64_t)blockIdx.x * gkeys_stride : reinterpret_cast<uint64_t*>(smem);
  __shared__ uint32_t s_nruns;
  __shared__ double r_score[VOTE_THREADS];
  __shared__ uint32_t r_tie[VOTE_THREADS];
  __shared__ int r_pos[VOTE_THREADS];
  const int tid = threadIdx.x;
  const int qi = GLOBAL ? img_list[blockIdx.x] : (int)blockIdx.x;
  const int q0 = qoff[qi], Sq = qoff[qi + 1] - q0;
  const int E = Sq * k;
  if (!GLOBAL && E > e_lds_cap) return;   // oversized image: the GLOBAL launch handles it
  int Epad = 2;
  while (Epad < E) Epad <<= 1;
  const float smin = minmax[0], smax = minmax[1];
  const float den = smax - smin;

  // visiting order o = rank * Sq + seg  (rank-major, then segment: func_vpr.py:216 zips the transposed lists)
  for (int o = tid; o < Epad; o += VOTE_THREADS) {
    uint64_t key = ~0ull;
    if (o < E) {
      const int rank = o / Sq, seg = o - rank * Sq;
      const int64_t m = idx[(int64_t)(q0 + seg) * k + rank];
      if (m >= 0 && m < n_ref_seg) key = ((uint64_t)(uint32_t)img_of_seg[m] << 32) | (uint32_t)o;
    }
    keys[o] = key;
  }
  if (tid == 0) s_nruns = 0;
  bitonic_u64(keys, Epad, tid, VOTE_THREADS);

  // the weights of the sorted entries, in parallel (the fp64 run sums below must stay sequential to reproduce the
  // reference's order of additions; fetching sims[] inside that serial walk was one exposed global load per entry)
  float* wl = reinterpret_cast<float*>(keys + Epad);   // [E]  (LDS mode only; use_wl = 0 in GLOBAL mode)
  if (mode == SEGVLAD_VOTE_WT_BORDA_IM && use_wl) {
    for (int p = tid; p < E; p += VOTE_THREADS) {
      const uint64_t key = keys[p];
      float wgt = 0.f;
      if (key != ~0ull) {
        const uint32_t o = (uint32_t)key;
        const int rank = o / Sq, seg = o - rank * Sq;
        wgt = (sims[(int64_t)(q0 + seg) * k + rank] - smin) / den;
      }
      wl[p] = wgt;
    }
    __syncthreads();
  }
  double* accd = reinterpret_cast<double*>(wl + Epad);   // [Epad] per-run sums, indexed by the run's first position (fast path)
  if (!GLOBAL && fast) {
    for (int p = tid; p < Epad; p += VOTE_THREADS) accd[p] = 0.0;
    __syncthreads();
    for (int p = tid; p < E; p += VOTE_THREADS) {
      const uint64_t key = keys[p];
      if (key == ~0ull) continue;
      const uint32_t img = (uint32_t)(key >> 32);
      int lo = 0, hi = p;   // first position of this image's run
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((uint32_t)(keys[mid] >> 32) < img) lo = mid + 1;
        else hi = mid;
      }
      const float wgt = (mode == SEGVLAD_VOTE_WT_BORDA_IM) ? wl[p] : 1.f;
      const bool exact = wgt == 0.f || (wgt >= 7.62939453125e-6f && wgt <= 1.f);   // 2^-17
      unsafeAtomicAdd(&accd[lo], exact ? (double)wgt : (double)NAN);
    }
    __syncthreads();
  }

  Run* runs = runs_scratch + (int64_t)qi * runs_stride;
  for (int p = tid; p < E; p += VOTE_THREADS) {
    const uint64_t key = keys[p];
    if (key == ~0ull) continue;
    const uint32_t img = (uint32_t)(key >> 32);
    if (p > 0 && (uint32_t)(keys[p - 1] >> 32) == img) continue;  // not a run start
    double acc = 0.0;
    uint32_t cnt = 0;
    int e = p;
    const bool summed = !GLOBAL && fast && accd[p] == accd[p];   // not poisoned
    if (summed) {
      acc = accd[p];
      e = E;   // skip the walk
    } else if (!GLOBAL && use_wl && mode == SEGVLAD_VOTE_WT_BORDA_IM) {
      // sequential sum in visiting order, with the run's end found first (binary search): a loop without a data-dependent
      // exit lets the LDS reads run ahead of the dependent fp64 additions (the walk below pays a key read, a compare and
      // a weight read per entry, back to back: ~200 cycles each)
      int lo = p + 1, hi = E;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint64_t km = keys[mid];
        if (km != ~0ull && (uint32_t)(km >> 32) == img) lo = mid + 1;
        else hi = mid;
      }
      const int e_end = lo;
#pragma unroll 8
      for (int q = p; q < e_end; ++q) acc += (double)wl[q];
      cnt = (uint32_t)(e_end - p);
      e = E;
    }
    while (e < E) {
      const uint64_t ke = keys[e];
      if (ke == ~0ull || (uint32_t)(ke >> 32) != img) break;
      if (mode == SEGVLAD_VOTE_WT_BORDA_IM) {
        if (use_wl) {
          acc += (double)wl[e];
        } else {   // > 8192 entries per image: no room for the weight array
          const uint32_t o = (uint32_t)ke;
          const int rank = o / Sq, seg = o - rank * Sq;
          acc += (double)((sims[(int64_t)(q0 + seg) * k + rank] - smin) / den);
        }
      }
      ++cnt;
      ++e;
    }
    Run r;
    r.score = (mode == SEGVLAD_VOTE_WT_BORDA_IM || summed) ? acc : (double)cnt;
    r.first = (mode == SEGVLAD_VOTE_WT_BORDA_IM) ? (uint32_t)key : img;  // COUNT ties: lower image id
    r.img = (int32_t)img;
    runs[atomicAdd(&s_nruns, 1u)] = r;
  }
  __syncthreads();
  const int R = (int)s_nruns;
  __threadfence_block();
  // n_top rounds of arg-best; a taken run gets score = -inf
  for (int round = 0; round < n_top; ++round) {
    double bs = -INFINITY;
    uint32_t bt = ~0u;
    int bp = -1;
    for (int j = tid; j < R; j += VOTE_THREADS) {
      const Run r = runs[j];
      if (r.score == -INFINITY) continue;
      if (bp < 0 || better(r.score, r.first, bs, bt)) {
        bs = r.score;
        bt = r.first;
        bp = j;
      }
    }
    r_score[tid] = bs;
    r_tie[tid] = bt;
    r_pos[tid] = bp;
    __syncthreads();
    for (int o = VOTE_THREADS / 2; o > 0; o >>= 1) {
      if (tid < o) {
        const int pb = r_pos[tid + o];
        if (pb >= 0 && (r_pos[tid] < 0 || better(r_score[tid + o], r_tie[tid + o], r_score[tid], r_tie[tid]))) {
          r_score[tid] = r_score[tid + o];
          r_tie[tid] = r_tie[tid + o];
          r_pos[tid] = pb;
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      const int bpos = r_pos[0];
      if (bpos >= 0) {
        pred[(int64_t)qi * n_top + round] = runs[bpos].img;
        if (score) score[(int64_t)qi * n_top + round] = runs[bpos].score;
        runs[bpos].score = -INFINITY;
      } else {
        pred[(int64_t)qi * n_top + round] = -1;
        if (score) score[(int64_t)qi * n_top + round] = 0.0;
      }
    }
    __syncthreads();
  }
}

int sv_launch_vote(segvlad_ctx* ctx, const int64_t* idx, const float* sims, const int32_t* img_of_seg,
                   int64_t n_ref_seg, const int32_t* qoff_dev, const int32_t* qoff_host, int n_img, int k,
                   const float* minmax_dev, int n_top, int mode, int32_t* pred, double* score) {
  if (n_img <= 0) return SEGVLAD_OK;
  constexpr int E_LDS = 16384;   // entries the in-LDS sort holds (8 B keys, 128 KiB)
  int maxS_small = 0, maxS = 0;
  std::vector<int32_t> big;
  for (int i = 0; i < n_img; ++i) {
    const int s = qoff_host[i + 1] - qoff_host[i];
    if (s < 0) return ctx->fail(SEGVLAD_ERR_ARG, "vote: qseg_offsets must be non-decreasing");
    if (s > maxS) maxS = s;
    if ((int64_t)s * k > E_LDS) big.push_back(i);
    else if (s > maxS_small) maxS_small = s;
  }
  const int64_t Emax = (int64_t)maxS * k;
  const int64_t stride = Emax > 0 ? Emax : 1;
  SV_HIP(ctx->s_misc.reserve((size_t)n_img * stride * sizeof(Run)));
  if ((int)big.size() < n_img) {
    const int64_t E = (int64_t)maxS_small * k;
    int Epad = 2;
    while (Epad < E) Epad <<= 1;
    size_t lds = (size_t)Epad * 12;   // sort keys + the entries' weights
    const int use_wl = lds <= 128 * 1024;
    if (!use_wl) lds = (size_t)Epad * 8;
    const int fast = Epad <= 4096 ? 1 : 0;   // + per-run fp64 sums (order-independent when exact, see the header)
    if (fast) lds = (size_t)Epad * 20;
    if (lds > 64 * 1024)
      SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(vote_kernel<false>), (size_t)lds));
    hipLaunchKernelGGL(vote_kernel<false>, dim3(n_img), dim3(VOTE_THREADS), lds, ctx->stream, idx, sims, img_of_seg, n_ref_seg, qoff_dev,
                       k, minmax_dev, n_top, mode, ctx->s_misc.as<Run>(), stride, pred, score, use_wl, E_LDS,
                       (const int32_t*)nullptr, (uint64_t*)nullptr, (int64_t)0, fast);
    SV_HIP(hipGetLastError());
  }
  if (!big.empty()) {   // oversized query images: keys sorted in a global scratch row each
    if (Emax > (1ll << 24)) return ctx->fail(SEGVLAD_ERR_LIMIT, "vote: %d segments x k=%d entries per query image", maxS, k);
    int64_t Epad = 2;
    while (Epad < Emax) Epad <<= 1;
    const size_t nb = big.size();
    SV_HIP(ctx->s_vote_keys.reserve(nb * (size_t)Epad * 8 + nb * 4 + 64));
    uint64_t* gk = ctx->s_vote_keys.as<uint64_t>();
    int32_t* list = reinterpret_cast<int32_t*>(gk + nb * (size_t)Epad);
    SV_HIP(hipMemcpyAsync(list, big.data(), nb * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(vote_kernel<true>, dim3((unsigned)nb), dim3(VOTE_THREADS), 0, ctx->stream, idx, sims, img_of_seg, n_ref_seg,
                       qoff_dev, k, minmax_dev, n_top, mode, ctx->s_misc.as<Run>(), stride, pred, score, 0, E_LDS,
                       (const int32_t*)list, gk, Epad, 0);
    SV_HIP(hipGetLastError());
    SV_HIP(hipStreamSynchronize(ctx->stream));   // big[] lives on this frame
  }
  return SEGVLAD_OK;
}
