// Grouped exact refinement of the kNN refine bands (round 5).
//
// The exact level of the search (faiss IndexFlatL2.search, place_rec_main.py:53-60) re-evaluates, per query ROW, its band
// {d2~ <= A_k + 2 eps} (>= k rows, <= SV_RCAP) with the sequential fp32 chain.  The rows of a query batch are the segments
// of query images, 50 consecutive rows per image, and the segments of an image share most of their neighbours (a place's
// reference segments): on the bench's raw 98 304-d workload (BASELINE configs[1]) the 50 bands of an image, 10 500 list
// slots, hold ~220 DISTINCT database rows; on the PCA'd 1 M-row index 1000-1500 of 13 500.  Read per row, those rows
// crossed HBM / L2 once per slot: 827 GB per 200 query images on raw descriptors.
//
// Here G consecutive query rows form a group (G = option "query_group" when it is 1 .. 64 -- the caller's hint that the query
// rows come in runs of G per image, so that groups coincide with images -- else 32):
//   refine_union_kernel        sorted union of the group's band lists (LDS hash set, then a sort of the distinct ids); a group whose union
//                              is not cheaper to evaluate than its bands row by row (a cost model with measured constants:
//                              nothing shared -- random queries --, or short bands against a long union), or longer than
//                              RG_UCAP, keeps the per-row kernels (its rows are flagged in `perrow`)
//   refine_group_gemm_kernel   exact distances of ALL 32 x U pairs of a grouped group on v_mfma_f32_32x32x2_f32: per output
//                              element the sequential chain acc = fma(q[k], r[k], acc), k = 0 .. d - 1, from acc = 0 --
//                              bit for bit the chain of refine_exact_kernel and of the distance-matrix path
//                              (gemm_kernels.hip, same instruction, same k order) -- then sv_d2 with the same norms.
//                              (The pairs outside a query's own band are computed and never read.)
//   refine_group_select_kernel per query row: the keys of its OWN band out of the group's key matrix (the union kernel leaves the
//                              union column of every band entry), (distance, id) sort, top k.
// Workgroup of the GEMM: 4 waves, wave w owns 32 union rows (MT = 1 or 2 accumulator tiles of 32 x 32: groups of <= 32 / <= 64
// query rows), the query rows are shared through LDS; operands staged k-major ([k][row], odd row stride) exactly like
// gemm_kernels.hip; global loads run THREE k-tiles ahead in three named register stages (a step's loads carry no condition in
// the steady-state loop: with one, the compiler waits for every outstanding load before it reuses a stage -- measured: 2 TB/s),
// one barrier per k-tile.  d % 32 == 0.
#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int RG_GMAX = 64;      // query rows per group, at most

__device__ __forceinline__ uint32_t rg_f2key(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rg_key2f(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ int rg_frag_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

template <class T, int NTH = 256>
__device__ __forceinline__ void rg_bitonic(T* a, int n, int tid) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (n >> 1); t += NTH) {
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

// ---- union of a group's band lists ---------------------------------------------------------------------------------------
constexpr int RG_UT = 1024;   // threads of the union kernel (the sort is its time)
__global__ __launch_bounds__(RG_UT) void refine_union_kernel(const uint32_t* __restrict__ ref_cnt, const uint32_t* __restrict__ ref_id,
                                                           int rcap, int m, int G, int d, int force, int ucap, int tcap /* hash slots: a power of two */,
                                                           uint32_t* __restrict__ grp_cnt,
                                                           uint32_t* __restrict__ grp_ids, uint32_t* __restrict__ perrow,
                                                           uint32_t* __restrict__ work /* [0] = count, then items */,
                                                           uint16_t* __restrict__ grp_pos /* [m][rcap]: union column of every band entry */,
                                                           const uint32_t* __restrict__ only_rows /* G = 1: rows to look at, or null */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* buf = reinterpret_cast<uint32_t*>(smem);   // [tcap] hash set of the group's band ids
  __shared__ uint32_t off[RG_GMAX + 1];
  __shared__ uint32_t wtot[RG_UT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int q0 = b * G;
  const int nrows = min(G, m - q0);
  if (only_rows && !only_rows[q0]) {   // (second tier: one-row groups, only the flagged rows exist)
    if (tid == 0) grp_cnt[b] = 0u;
    if (tid < nrows) perrow[q0 + tid] = 0u;
    return;
  }
  if (tid == 0) {
    uint32_t s = 0;
    for (int t = 0; t < G; ++t) {
      off[t] = s;
      uint32_t c = t < nrows ? ref_cnt[q0 + t] : 0u;
      if (c > (uint32_t)rcap) c = 0u;   // (never: an overflowing band leaves 0 and takes the second tier)
      s += c;
    }
    off[G] = s;
  }
  __syncthreads();
  const int total = (int)off[G];
  bool grouped = false;
  int U = 0;
  // The distinct ids through an LDS hash set (open addressing, atomicCAS), then ONLY those are sorted: the first version sorted
  // all band entries (13 500 -> 16 384 slots, 105 bitonic stages: ~150 us per group and 0.15-0.19 ms per 200 query images, as
  // much as the GEMM it feeds); the set's contents do not depend on the insertion order, the sort makes their order canonical.
  uint32_t* uniq = buf + tcap;                           // [ucap + RG_UT]: the distinct ids
  if (total > 0 && 4 * total <= 3 * tcap) {              // (load factor <= 0.75; beyond: the group stays with the per-row kernels)
    const uint32_t mask = (uint32_t)tcap - 1u;
    for (int j = tid; j < tcap; j += RG_UT) buf[j] = 0xffffffffu;
    __syncthreads();
    for (int t = w; t < nrows; t += RG_UT / 64) {
      const int c = (int)(off[t + 1] - off[t]);
      for (int j = l; j < c; j += 64) {
        const uint32_t id = ref_id[(size_t)(q0 + t) * rcap + j];
        uint32_t h = (id * 2654435761u) & mask;
        for (;;) {
          const uint32_t prev = atomicCAS(&buf[h], 0xffffffffu, id);
          if (prev == 0xffffffffu || prev == id) break;
          h = (h + 1u) & mask;
        }
      }
    }
    __syncthreads();
    // compaction of the occupied slots (ballot prefix per wave + the waves' totals)
    uint32_t base = 0;
    for (int j0 = 0; j0 < tcap; j0 += RG_UT) {
      const int j = j0 + tid;
      const uint32_t v = buf[j];
      const bool have = v != 0xffffffffu;
      const uint64_t mk = __builtin_amdgcn_ballot_w64(have);
      if (l == 0) wtot[w] = (uint32_t)__popcll(mk);
      __syncthreads();
      uint32_t o = base, all = 0;
      for (int x = 0; x < RG_UT / 64; ++x) {
        if (x < w) o += wtot[x];
        all += wtot[x];
      }
      const uint32_t pos = o + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      if (have && pos < (uint32_t)ucap) uniq[pos] = v;
      base += all;
      __syncthreads();
    }
    U = (int)base;
    if (U <= ucap) {
      int np2 = 64;
      while (np2 < U) np2 <<= 1;
      for (int j = U + tid; j < np2; j += RG_UT) uniq[j] = 0xffffffffu;
      rg_bitonic<uint32_t, RG_UT>(uniq, np2, tid);
      for (int j = tid; j < U; j += RG_UT) grp_ids[(size_t)b * ucap + j] = uniq[j];
    }
    // Is the union cheaper?  Measured on MI355X (ns, d = 1024; the first two terms grow with d): the per-row kernels ~0.39 per
    // (query, row) pair of the bands + 21 per query row; the union GEMM ~0.038 per slot of its 64 x 128 tiles, whatever their fill; the per-query
    // sort of a band's keys ~0.022 per slot of its power-of-two list (~1.5 slots per band entry).  (200-deep bands of an image's 50 segments on a 1 M-row index:
    // 10 900 pairs against 938 union rows -> grouped; the same image 50 deep on a 125 k-row shard: 2 500 pairs -> row by row.)
    const float dd = (float)d * (1.f / 1024.f);
    const float cost_grp = (float)((U + 127) >> 7) * 8192.f * 0.038f * dd + 1.5f * (float)total * 0.022f;
    // (round 6: the per-row kernels re-fitted on two band depths -- 10 000 rows x 270-row bands 1.26 ms, x 62-row bands (50-deep
    //  searches of a 125 k-row shard) 0.45 ms: 0.39 ns per pair + 21 ns per query row, not 0.46 ns per pair.  With the old constant
    //  the shard's groups were split between the two paths and paid both: select + refine 0.68 ms, 0.62 all per row, 0.48 all grouped.)
    const float cost_row = (float)total * 0.39f * dd + (float)nrows * 21.f;
    grouped = U <= ucap && (force || cost_grp < cost_row);
    if (grouped) {
      // where each band entry sits in the union: the per-query select then sorts a query's OWN band (<= rcap keys), not the union
      for (int t = w; t < nrows; t += RG_UT / 64) {
        const int c = (int)(off[t + 1] - off[t]);
        for (int j = l; j < c; j += 64) {
          const uint32_t id = ref_id[(size_t)(q0 + t) * rcap + j];
          int lo = 0, hi = U;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < id) lo = mid + 1;
            else hi = mid;
          }
          grp_pos[(size_t)(q0 + t) * rcap + j] = (uint16_t)lo;
        }
      }
    }
  }
  if (tid == 0) {
    grp_cnt[b] = grouped ? (uint32_t)U : 0u;
    if (grouped) {
      // the GEMM's work items (group, 128-row tile of its union), DENSE: a 2-D grid (tile, group) with early exits puts every
      // live workgroup on the XCDs its linear id selects -- tiles 0 and 1 of 16: two of the eight XCDs, measured 5x slower
      const uint32_t nt = (uint32_t)(U + 127) >> 7;
      const uint32_t at = atomicAdd(&work[0], nt);
      for (uint32_t t = 0; t < nt; ++t) work[1 + at + t] = ((uint32_t)b << 5) | t;
    }
  }
  if (tid < nrows) perrow[q0 + tid] = grouped ? 0u : 1u;
}

// ---- G query rows x the union's rows: exact fp32 distances as sortable keys -----------------------------------------------
// Every row is fetched in pieces of RG_KS floats (128 B): one load instruction of a wave covers eight rows x 128 contiguous
// bytes.  A super-tile of RG_KS k travels in ONE register stage ((MT + 4) x 4 / 4 sixteen-byte loads per thread, in flight
// while the previous super-tile is multiplied out of LDS) and is parked in a single LDS buffer between two barriers; with
// 27 KiB of LDS a CU holds five workgroups, which hide each other's load / store / barrier phases.  (Measured, 200 groups of
// 200-row unions: 512-byte pieces and one workgroup per CU 8.63 ms at d = 98 304 / 0.384 ms at d = 1024, 256-byte pieces
// 7.70 / 0.378, 128-byte pieces 7.51 / 0.362.)  The loads are inline asm with an explicit wait: a compiler-visible load with
// a condition on it makes every LDS access behind it wait for all of them.
// LDS image: ROW-major, row stride KS + 4 floats, and inside every group of 8 k the even k first, then the odd:
// [k0 k2 k4 k6 | k1 k3 k5 k7].  v_mfma_f32_32x32x2_f32 takes k = 2 s from lanes 0-31 and k = 2 s + 1 from lanes 32-63, so a
// lane's operands of FOUR consecutive k-steps are one aligned 16-byte read (rows 16 apart share a bank group, which the
// 16-lane service groups of a ds_read_b128 never hold together), and a loaded float4 goes out as two 8-byte writes.
// (Measured on the way, 10 000 x 200-row bands of 98 304-d rows: k-major [k][row] tiles fed by ds_read_b32 -- one or two reads,
//  a wait and one or two MFMAs per k-step, one wave per SIMD -- 63 ms with the matrix pipes 8 % busy: a chain of LDS latencies,
//  half of the LDS cycles bank conflicts of the transposing 4-byte stores; 128-byte row pieces three tiles ahead: 32-40 ms.)
typedef float rg_f32x4 __attribute__((ext_vector_type(4)));
typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
// The row pieces are requested by inline asm and waited for by a hand-placed s_waitcnt: the compiler takes the destination
// registers for defined when the asm statement ends, so nothing but today's register allocation keeps it from copying or spilling
// them between the request and the wait.  That form is therefore tied to the toolchain it was verified on -- ISA
// inspected, and tests/test_gpu_refine_group.py holds the kernel to the per-row kernels' bits: AMD clang 22 (ROCm 7.2).  Any
// other compiler gets plain loads, whose waits it places itself (measured slower: it waits for every outstanding load in front
// of every LDS access behind a conditional load -- but correct by construction).  -DSEGVLAD_RG_PLAIN_LOADS forces that form.
#if defined(__clang_major__) && __clang_major__ == 22 && !defined(SEGVLAD_RG_PLAIN_LOADS)
#define RG_ASM_LOADS 1
#define RG_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define RG_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define RG_ASM_LOADS 0
#define RG_GLOAD(dst, ptr) (dst) = *reinterpret_cast<const rg_f32x4*>(ptr)
#define RG_WAIT_LOADS() do { } while (0)
#endif
constexpr int RG_KS = 32;

template <int MT, int KS>
__global__ __launch_bounds__(256) void refine_group_gemm_kernel(const float* __restrict__ Q, const float* __restrict__ R, int d, int m,
                                                                int G, const float* __restrict__ qn, const float* __restrict__ rn,
                                                                const uint32_t* __restrict__ grp_cnt,
                                                                const uint32_t* __restrict__ grp_ids, int ucap,
                                                                uint64_t* __restrict__ keys, const uint32_t* __restrict__ work) {
  constexpr int LDR = KS + 4;         // row stride of the LDS image
  constexpr int LPR = KS / 4;         // lanes (16-byte loads) per row piece
  constexpr int RPI = 64 / LPR;       // rows per load instruction of a wave
  constexpr int NR = 32 * MT + 128;   // rows of a super-tile: the group's query rows, then 128 union rows
  constexpr int NLD = NR / (4 * RPI); // 16-byte loads per thread and super-tile
  static_assert(NR % (4 * RPI) == 0 && (32 * MT) % (4 * RPI) == 0, "whole instructions of query rows / of union rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                     // [NR][LDR]
  uint32_t* ids = reinterpret_cast<uint32_t*>(tile + NR * LDR);     // [128]
  if (blockIdx.x >= work[0]) return;
  const uint32_t item = work[1 + blockIdx.x];
  const int b = (int)(item >> 5), c0 = (int)(item & 31u) * 128;
  const int U = (int)grp_cnt[b];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int lp = l % LPR, lr = l / LPR;   // loader coordinates: 16-byte piece of a row, row of the instruction
  const int q0 = b * G;
  const int q_end = min(q0 + G, m);
  if (tid < 128) ids[tid] = grp_ids[(size_t)b * ucap + (c0 + tid < U ? c0 + tid : 0)];   // (columns beyond U: a valid row, never stored)
  __syncthreads();
  // load j of this thread: staged row rr = 4 RPI j + RPI w + lr, floats 4 lp .. + 3 of the super-tile
  const float* src[NLD];
#pragma unroll
  for (int j = 0; j < NLD; ++j) {
    const int rr = 4 * RPI * j + RPI * w + lr;
    src[j] = (rr < 32 * MT ? Q + (size_t)min(q0 + rr, q_end - 1) * d   // (rows beyond the group: never stored)
                           : R + (size_t)ids[rr - 32 * MT] * d) + 4 * lp;
  }
  rg_f32x4 v[NLD];
  auto gload = [&](int st) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) RG_GLOAD(v[j], src[j] + (size_t)st * KS);
  };
  // floats 4 lp .. 4 lp + 3 = k-group lp >> 1, half lp & 1: the even k to slots 2 (lp & 1) .. + 1, the odd k four slots further
  float* st_base = tile + (RPI * w + lr) * LDR + 8 * (lp >> 1) + 2 * (lp & 1);
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < NLD; ++j) asm volatile("" : "+v"(v[j]));   // (the values exist from HERE on: behind the caller's wait; the
                                                                   //  re-packing moves below are not memory operations)
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      float* p = st_base + 4 * RPI * j * LDR;
      rg_f32x2 ev, od;
      ev[0] = v[j][0];
      ev[1] = v[j][2];
      od[0] = v[j][1];
      od[1] = v[j][3];
      *reinterpret_cast<rg_f32x2*>(p) = ev;
      *reinterpret_cast<rg_f32x2*>(p + 4) = od;
    }
  };
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* a_frag = tile + i * LDR + 4 * kk;                       // + 32 t rows, + 8 g floats
  const float* b_frag = tile + (32 * MT + 32 * w + i) * LDR + 4 * kk;
  const int nst = d / KS;
  gload(0);
  RG_WAIT_LOADS();
  sstore();
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    if (st + 1 < nst) gload(st + 1);   // in flight while this super-tile is multiplied
    // KS / 8 groups of 8 k; the fragments of group g + 1 are requested before the MFMAs of group g are issued
    rg_f32x4 fa[2][MT], fb[2];
    fb[0] = *reinterpret_cast<const rg_f32x4*>(b_frag);
#pragma unroll
    for (int t = 0; t < MT; ++t) fa[0][t] = *reinterpret_cast<const rg_f32x4*>(a_frag + 32 * t * LDR);
#pragma unroll
    for (int g = 0; g < KS / 8; ++g) {
      if (g + 1 < KS / 8) {
        fb[(g + 1) & 1] = *reinterpret_cast<const rg_f32x4*>(b_frag + 8 * (g + 1));
#pragma unroll
        for (int t = 0; t < MT; ++t) fa[(g + 1) & 1][t] = *reinterpret_cast<const rg_f32x4*>(a_frag + 32 * t * LDR + 8 * (g + 1));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][t][u], fb[g & 1][u], acc[t], 0, 0, 0);
    }
    __syncthreads();   // every wave is done with the LDS image
    if (st + 1 < nst) {
      RG_WAIT_LOADS();
      sstore();
      __syncthreads();
    }
  }
  const int col = c0 + 32 * w + i;
  if (col < U) {
    const uint32_t id = ids[32 * w + i];
    const float r2 = rn[id];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + 32 * t + rg_frag_row(r, kk);
        if (q < q_end) keys[(size_t)q * ucap + col] = ((uint64_t)rg_f2key(sv_d2(qn[q], r2, acc[t][r])) << 32) | id;
      }
  }
}

// ---- per query row: the keys of ITS band out of the group's key matrix, (distance, id) sort, top k ----------------------------
__global__ __launch_bounds__(256) void refine_group_select_kernel(const uint32_t* __restrict__ grp_cnt, const uint64_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ ref_cnt,
                                                                  const uint16_t* __restrict__ grp_pos, int rcap, int G, int ucap, int k,
                                                                  float* __restrict__ d2_out, int64_t* __restrict__ idx_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* a = reinterpret_cast<uint64_t*>(smem);   // [np2 <= pow2(rcap)]
  const int64_t row = blockIdx.x;
  const int tid = threadIdx.x;
  if (grp_cnt[row / G] == 0) return;   // the per-row kernels' group
  int n = (int)ref_cnt[row];
  if (n > rcap) n = 0;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  for (int j = tid; j < np2; j += 256) a[j] = j < n ? keys[(size_t)row * ucap + grp_pos[(size_t)row * rcap + j]] : ~0ull;
  rg_bitonic<uint64_t>(a, np2, tid);
  for (int j = tid; j < k; j += 256) {
    float dd = INFINITY;
    int64_t id = -1;
    if (j < n) {
      dd = rg_key2f((uint32_t)(a[j] >> 32));
      id = (int64_t)(uint32_t)a[j];
    }
    d2_out[row * k + j] = dd;
    idx_out[row * k + j] = id;
  }
}

}   // namespace

static int rg_group_rows(const segvlad_ctx* ctx) {
  const int h = ctx->opt.query_group;
  return (h >= 1 && h <= RG_GMAX) ? h : 32;
}

// The refinement of a batch: groups whose bands overlap go through the union GEMM, the others (and every row of a search the
// grouping does not apply to) through the per-row kernels.  Same outputs as sv_launch_refine_exact, bit for bit.
int sv_launch_refine_grouped(segvlad_ctx* ctx, const float* Q, const float* R, int nq, int d, const float* qn, const float* rn,
                             const uint32_t* ref_cnt, const uint32_t* ref_id, int rcap, int k, float* d2_out, int64_t* idx_out,
                             int* launches, const uint32_t* only_rows, int live_groups_hint) {
  if (launches) *launches = 0;
  if (nq <= 0) return SEGVLAD_OK;
  const int ucap = SV_RG_UCAP;
  // only_rows (the second tier: a few flagged rows, each with a list of up to 8192 entries): one-row groups, always through the
  // union GEMM when the band fits it -- what is bought there is PARALLELISM (a list walked by one workgroup of the per-row kernel
  // is one latency chain: 4.6 ms for one 600-row band of 98 304-d rows), not shared rows
  const int G = only_rows ? 1 : rg_group_rows(ctx);
  // hash slots of the union kernel: load factor <= 0.75 for a group whose bands are all full, capped at 32 768 (128 KiB of LDS;
  // a group with more band entries than 0.75 x that stays with the per-row kernels)
  int tcap = 1024;
  while (tcap < 32768 && 3 * (int64_t)tcap < 4 * (int64_t)G * rcap) tcap <<= 1;
  const bool can = ctx->opt.refine_group != 0 && d % RG_KS == 0 && nq > 128 && (rcap <= 4096 || only_rows) &&
                   (reinterpret_cast<uintptr_t>(Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(R) & 15) == 0;
  if (!can) {
    if (launches) *launches = 1;
    return sv_launch_refine_exact(ctx, Q, R, nq, d, qn, rn, ref_cnt, ref_id, rcap, k, d2_out, idx_out, only_rows);
  }
  const int nb = (nq + G - 1) / G;
  SV_HIP(ctx->s_grp_cnt.reserve((size_t)nb * 4));
  SV_HIP(ctx->s_grp_ids.reserve((size_t)nb * ucap * 4));
  SV_HIP(ctx->s_grp_rows.reserve((size_t)nq * 4));
  SV_HIP(ctx->s_grp_keys.reserve((size_t)nq * ucap * 8));
  SV_HIP(ctx->s_grp_pos.reserve((size_t)nq * rcap * 2));
  uint32_t* gcnt = ctx->s_grp_cnt.as<uint32_t>();
  uint32_t* gids = ctx->s_grp_ids.as<uint32_t>();
  uint32_t* prow = ctx->s_grp_rows.as<uint32_t>();
  uint64_t* gkeys = ctx->s_grp_keys.as<uint64_t>();
  static_assert(SV_RG_UCAP / 128 <= 32, "a work item holds its tile in 5 bits");
  // (the GEMM's grid: every group could hold ucap / 128 tiles; a caller that knows how few groups are live -- the second tier --
  //  says so, instead of 16 early-exit workgroups per query row)
  const int max_items = ((live_groups_hint > 0 && live_groups_hint < nb) ? live_groups_hint : nb) * (ucap / 128);
  SV_HIP(ctx->s_grp_work.reserve((size_t)(max_items + 1) * 4));
  uint32_t* work = ctx->s_grp_work.as<uint32_t>();
  SV_HIP(hipMemsetAsync(work, 0, 4, ctx->stream));
  const size_t ulds = ((size_t)tcap + (size_t)ucap + RG_UT) * 4;
  if (ulds + 1024 > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(refine_union_kernel), ulds));
  hipLaunchKernelGGL(refine_union_kernel, dim3(nb), dim3(RG_UT), ulds, ctx->stream, ref_cnt, ref_id, rcap, nq, G, d,
                     (ctx->opt.refine_group == 2 || only_rows) ? 1 : 0, ucap, tcap, gcnt, gids, prow, work, ctx->s_grp_pos.as<uint16_t>(), only_rows);
  SV_HIP(hipGetLastError());
  {
    const int mt = G > 32 ? 2 : 1;
    const size_t glds = (size_t)(32 * mt + 128) * (RG_KS + 4) * 4 + 128 * 4;
    auto gk = mt == 2 ? refine_group_gemm_kernel<2, RG_KS> : refine_group_gemm_kernel<1, RG_KS>;
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(gk), glds));
    hipLaunchKernelGGL(gk, dim3(max_items), dim3(256), glds, ctx->stream, Q, R, d, nq, G, qn, rn, gcnt, gids, ucap, gkeys, work);
    SV_HIP(hipGetLastError());
  }
  {
    int rpad = 2;
    while (rpad < rcap) rpad <<= 1;
    if ((size_t)rpad * 8 + 1024 > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(refine_group_select_kernel), (size_t)rpad * 8));
    hipLaunchKernelGGL(refine_group_select_kernel, dim3(nq), dim3(256), (size_t)rpad * 8, ctx->stream, gcnt, gkeys, ref_cnt,
                       ctx->s_grp_pos.as<uint16_t>(), rcap, G, ucap, k, d2_out, idx_out);
    SV_HIP(hipGetLastError());
  }
  if (launches) *launches = 4;
  return sv_launch_refine_exact(ctx, Q, R, nq, d, qn, rn, ref_cnt, ref_id, rcap, k, d2_out, idx_out, prow);
}

// statistics for bench.py / the tests: groups of the last grouped refinement that took the union GEMM, and the sum of their
// union lengths (synchronises)
int sv_refine_group_stats(segvlad_ctx* ctx, int nq, int64_t* groups, int64_t* grouped, int64_t* union_sum) {
  const int nb = (nq + rg_group_rows(ctx) - 1) / rg_group_rows(ctx);
  std::vector<uint32_t> h(nb);
  SV_HIP(hipStreamSynchronize(ctx->stream));
  SV_HIP(hipMemcpy(h.data(), ctx->s_grp_cnt.p, (size_t)nb * 4, hipMemcpyDeviceToHost));
  int64_t g = 0, s = 0;
  for (uint32_t c : h) {
    if (c) ++g;
    s += c;
  }
  *groups = nb;
  *grouped = g;
  *union_sum = s;
  return SEGVLAD_OK;
}
