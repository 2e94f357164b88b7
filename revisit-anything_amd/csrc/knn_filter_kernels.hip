// Exact kNN, large-database path: 16-bit MFMA candidate FILTERS + exact fp32 refinement (gfx950).
// (This file was knn_bf16_kernels.hip until round 6: it started as the bf16x3 filter below; the default filter since round 2 is the
//  single-product fp16 kernel knn_f16_filter_kernel further down, the bf16x3 form serves d % 64 != 0.)
//
// fp32 MFMA runs at 1/16 of the bf16 rate on CDNA4.  Every fp32 value is split into two bf16 pieces
// x = hi + lo + e, |e| <= 2^-16 |x|, and the filter GEMM accumulates hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 (products of two bf16 are exact in fp32).  The result differs from the fp32
// fma-chain value of the exact path by at most
//        |dot~ - dot| <= (3*2^-16 + 4*d*2^-24) * ||q|| * ||r||         (dropped terms + accumulation)
// so a candidate filter with that margin can never drop a true neighbour.  The filter's survivors
// (about k per query after the threshold levels) are then re-evaluated with the SAME sequential fp32
// fma chain as the matrix path (gemm_nt_kernel<1>), sorted by (distance, id) and emitted: the final
// distances and ids are bit-identical to the all-fp32 path.
//
//   split_bf16_kernel        fp32 [n][d] -> hi, lo bf16 planes
//   knn_bf16_filter_kernel   128x128x32 tiles, 4 waves x (2x2) 32x32x16 MFMA tiles x 3 products;
//                            16-B coalesced global loads -> registers -> XOR-swizzled LDS (conflict-free
//                            ds_read_b128 fragments), double-buffered LDS, two register stages in flight;
//                            epilogue appends (d2~, id) with d2~ <= thr + margin to the candidate lists
//   select_approx_kernel     per query: sort candidates by d2~; intermediate level: A_k (k-th smallest);
//                            last level: the refine list {d2~ <= A_k + 2 eps}
//   refine_exact_kernel      per query: exact fp32 distances of the refine list, sort, top-k
//   select_wg_kernel / refine_exact_small_kernel   the same two steps for ONE query image per pass (<= 128 lists): a
//                            workgroup per list, refinement lists shared by workgroups
#include <stdlib.h>

#include <stdio.h>

#include "ctx.h"
#include "knn_dev.h"
#include "small_pass_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- fp32 -> (hi, lo) bf16, round-to-nearest-even ---------------------------------------------------
__device__ __forceinline__ uint16_t bf16_rne(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ X, int64_t n4, uint16_t* __restrict__ hi,
                                                         uint16_t* __restrict__ lo) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(X)[j];
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = bf16_rne(f[e]);
      l[e] = bf16_rne(f[e] - bf16_to_f32(h[e]));
    }
    reinterpret_cast<uint2*>(hi)[j] = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    reinterpret_cast<uint2*>(lo)[j] = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  }
}

int sv_launch_split_bf16(segvlad_ctx* ctx, const float* X, int64_t n_elems, uint16_t* hi, uint16_t* lo) {
  if (n_elems <= 0) return SEGVLAD_OK;
  const int64_t n4 = n_elems / 4;  // callers guarantee d % 8 == 0
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n4, hi, lo);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ---- filter GEMM ---------------------------------------------------------------------------------------
// Tile BM x BN x 32 (32 bf16 = 64 B = 4 chunks of 16 B per row and plane); WM x WN waves, each wave TM x TN
// MFMA tiles of 32x32.  Instantiated as 128x128 with 2x2 waves (2 workgroups per CU); a 256x256 variant
// measured the same time (the kernel is bound by HBM latency of the streamed DB operand, not by L2->LDS
// bandwidth), so the prefetch depth, not the tile, is the lever.
constexpr int FBK = 32;

// physical chunk of logical chunk c in row r: spreads a 16-lane ds_read_b128 group over all 16 slots
__device__ __forceinline__ int swz(int r, int c) { return c ^ ((r >> 2) & 3); }

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// Register stages: two 16-B loads per thread and plane in both tile configurations; THREE named stage sets
// rotate so that every k-tile's global loads are issued three k-tiles before they are written to LDS (the DB
// operand streams from HBM with ~2 us latency while a k-tile of bf16 MFMAs lasts only ~0.4-0.7 us).
// Named locals + macros: a struct or indexed array here is not promoted to registers by hipcc 7.2.
#define F_DECL(X) uint4 gah0##X, gah1##X, gal0##X, gal1##X, gbh0##X, gbh1##X, gbl0##X, gbl1##X
#define F_GLOAD(X, k0_)                                                          \
  do {                                                                           \
    const int64_t ko_ = (int64_t)(k0_) + 8 * (tid & 3);                          \
    gah0##X = *reinterpret_cast<const uint4*>(Qh + qa0 * d + ko_);               \
    gah1##X = *reinterpret_cast<const uint4*>(Qh + qa1 * d + ko_);               \
    gal0##X = *reinterpret_cast<const uint4*>(Ql + qa0 * d + ko_);               \
    gal1##X = *reinterpret_cast<const uint4*>(Ql + qa1 * d + ko_);               \
    gbh0##X = *reinterpret_cast<const uint4*>(Rh + rb0 * ldb + ko_);             \
    gbh1##X = *reinterpret_cast<const uint4*>(Rh + rb1 * ldb + ko_);             \
    gbl0##X = *reinterpret_cast<const uint4*>(Rl + rb0 * ldb + ko_);             \
    gbl1##X = *reinterpret_cast<const uint4*>(Rl + rb1 * ldb + ko_);             \
  } while (0)

#define F_SSTORE(X, S_)                                                          \
  do {                                                                           \
    unsigned char* s_ = (S_);                                                    \
    *reinterpret_cast<uint4*>(s_ + so0) = gah0##X;                               \
    *reinterpret_cast<uint4*>(s_ + so1) = gah1##X;                               \
    *reinterpret_cast<uint4*>(s_ + PA + so0) = gal0##X;                          \
    *reinterpret_cast<uint4*>(s_ + PA + so1) = gal1##X;                          \
    *reinterpret_cast<uint4*>(s_ + 2 * PA + so0) = gbh0##X;                      \
    *reinterpret_cast<uint4*>(s_ + 2 * PA + so1) = gbh1##X;                      \
    *reinterpret_cast<uint4*>(s_ + 2 * PA + PB + so0) = gbl0##X;                 \
    *reinterpret_cast<uint4*>(s_ + 2 * PA + PB + so1) = gbl1##X;                 \
  } while (0)

template <int BM, int BN, int WM, int WN, int DEPTH>
__global__ __launch_bounds__(64 * WM * WN, 2) void knn_bf16_filter_kernel(
    const uint16_t* __restrict__ Qh, const uint16_t* __restrict__ Ql, const uint16_t* __restrict__ Rh,
    const uint16_t* __restrict__ Rl, int M, int N, int d, int b_stride, int tiles_m, const float* __restrict__ qn,
    const float* __restrict__ rn, const float* __restrict__ thr, int64_t thr_ld, float eps_mult, float c_eps, float rn_max,
    uint32_t* __restrict__ cand_cnt, float* __restrict__ cand_d2, uint32_t* __restrict__ cand_id, int cap) {
  constexpr int T = 64 * WM * WN;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);  // MFMA tiles per wave
  constexpr int LA = BM * 4 / T, LB = BN * 4 / T;          // 16-B loads per thread, plane and k-tile
  constexpr int PA = BM * 64, PB = BN * 64;                // plane bytes
  constexpr int STAGE = 2 * PA + 2 * PB;                   // Ah, Al, Bh, Bl
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tile = blockIdx.x;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int wm = w / WN, wn = w % WN;
  const int64_t ldb = (int64_t)d * b_stride;
  const int ntiles = d / FBK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  static_assert(LA == 2 && LB == 2, "two 16-B loads per thread and plane");
  F_DECL(A);
  F_DECL(B);
  F_DECL(C);
  const int lr0 = tid >> 2, lr1 = lr0 + T / 4;
  // clamp: rows beyond the edge are loaded from the last valid row and never emitted
  const int64_t qa0 = (m0 + lr0 < M) ? (m0 + lr0) : (M - 1), qa1 = (m0 + lr1 < M) ? (m0 + lr1) : (M - 1);
  const int64_t rb0 = (n0 + lr0 < N) ? (n0 + lr0) : (N - 1), rb1 = (n0 + lr1 < N) ? (n0 + lr1) : (N - 1);
  const int so0 = lr0 * 64 + swz(lr0, tid & 3) * 16, so1 = lr1 * 64 + swz(lr1, tid & 3) * 16;
  F_GLOAD(A, 0);
  F_SSTORE(A, lds);
  if (ntiles > 1) F_GLOAD(A, FBK);
  if (DEPTH == 3) {
    if (ntiles > 2) F_GLOAD(B, 2 * FBK);
    if (ntiles > 3) F_GLOAD(C, 3 * FBK);
  }
  __syncthreads();
  int cur = 0;
  const int fa0 = wm * (32 * TM) + i, fb0 = wn * (32 * TN) + i;
  // multiply k-tile `kt` out of LDS buffer `cur`
  auto compute = [&]() {
    const unsigned char* S = lds + cur * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cl = 2 * ks + kk;  // logical 16-B chunk (8 consecutive k) of this lane
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int ra = fa0 + 32 * t;
        ah[t] = *reinterpret_cast<const bf16x8*>(S + ra * 64 + swz(ra, cl) * 16);
        al[t] = *reinterpret_cast<const bf16x8*>(S + PA + ra * 64 + swz(ra, cl) * 16);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int rb = fb0 + 32 * t;
        bh[t] = *reinterpret_cast<const bf16x8*>(S + 2 * PA + rb * 64 + swz(rb, cl) * 16);
        bl[t] = *reinterpret_cast<const bf16x8*>(S + 2 * PA + PB + rb * 64 + swz(rb, cl) * 16);
      }
      // three products per accumulator, small terms first; the TM*TN accumulators are independent, so
      // consecutive MFMAs never wait on each other
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = MFMA_BF16(al[mt], bh[nt], acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = MFMA_BF16(ah[mt], bl[nt], acc[mt][nt]);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = MFMA_BF16(ah[mt], bh[nt], acc[mt][nt]);
    }
  };
  // iteration kt: compute tile kt; write stage X (tile kt+1, loaded three k-tiles ago) to the other buffer;
  // barrier; refill X with tile kt+4
#define F_STEP(X)                                                     \
  do {                                                                \
    compute();                                                        \
    if (kt + 1 < ntiles) F_SSTORE(X, lds + (cur ^ 1) * STAGE);        \
    __syncthreads();                                                  \
    if (kt + 4 < ntiles) F_GLOAD(X, (kt + 4) * FBK);                  \
    cur ^= 1;                                                         \
    ++kt;                                                             \
  } while (0)
  int kt = 0;
  if (DEPTH == 3) {
    while (kt < ntiles) {
      F_STEP(A);
      if (kt >= ntiles) break;
      F_STEP(B);
      if (kt >= ntiles) break;
      F_STEP(C);
    }
  } else {
    for (; kt < ntiles; ++kt) {
      compute();
      if (kt + 1 < ntiles) F_SSTORE(A, lds + (cur ^ 1) * STAGE);
      __syncthreads();
      if (kt + 2 < ntiles) F_GLOAD(A, (kt + 2) * FBK);
      cur ^= 1;
    }
  }
#undef F_STEP

  // ---- epilogue: d2~ = (||q||^2 + ||r||^2) - 2 dot~ ; keep d2~ <= thr + eps_mult * eps(q) -----------------
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int64_t col = n0 + wn * (32 * TN) + nt * 32 + i;
    if (col >= N) continue;
    const float cn = rn[col * b_stride];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (32 * TM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row < M) {
          const float q2 = qn[row];
          const float v = sv_d2(q2, cn, acc[mt][nt][r]);
          const float lim = thr[row * thr_ld] + eps_mult * c_eps * sqrtf(q2 * rn_max);
          if (v <= lim) {
            const uint32_t slot = atomicAdd(&cand_cnt[row], 1u);
            if (slot < (uint32_t)cap) {
              cand_d2[row * cap + slot] = v;
              cand_id[row * cap + slot] = (uint32_t)(col * b_stride);
            }
          }
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int DEPTH>
static int launch_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Ql, const uint16_t* Rh, const uint16_t* Rl,
                         int M, int n_sample, int d, int b_stride, const float* qn, const float* rn, const float* thr,
                         int64_t thr_ld, float eps_mult, float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2,
                         uint32_t* cand_id, int cap) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (n_sample + BN - 1) / BN;
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  if (tiles > 0x7fffffffLL) return ctx->fail(SEGVLAD_ERR_LIMIT, "bf16 filter: too many tiles");
  const size_t lds = 2 * (size_t)(2 * BM * 64 + 2 * BN * 64);
  auto kern = knn_bf16_filter_kernel<BM, BN, WM, WN, DEPTH>;
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * WM * WN), lds, ctx->stream, Qh, Ql, Rh, Rl, M, n_sample, d, b_stride,
                     tiles_m, qn, rn, thr, thr_ld, eps_mult, c_eps, rn_max, cand_cnt, cand_d2, cand_id, cap);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

int sv_launch_bf16_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Ql, const uint16_t* Rh, const uint16_t* Rl,
                          int M, int n_sample, int d, int b_stride, const float* qn, const float* rn, const float* thr,
                          int64_t thr_ld, float eps_mult, float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2,
                          uint32_t* cand_id, int cap) {
  if (M <= 0 || n_sample <= 0) return SEGVLAD_OK;
  return launch_filter<128, 128, 2, 2, 1>(ctx, Qh, Ql, Rh, Rl, M, n_sample, d, b_stride, qn, rn, thr, thr_ld, eps_mult, c_eps, rn_max,
                                          cand_cnt, cand_d2, cand_id, cap);
}

// ---- fp16 single-product filter ----------------------------------------------------------------------------
// One fp16 MFMA product per fp32 fma: x~ = fl16(s * x) with a power-of-two scale s (exact), products of two
// fp16 are exact in fp32, so   |dot~ - dot| <= (2^-10 + 2^-22 + 2 d 2^-24) ||q|| ||r||   against the fp32 chain.
// The margin is ~24x wider than bf16x3's, which lets ~1.3-1.4x more candidates through (they are cheap: the
// exact refinement only sees the ~k survivors) for one third of the MFMA work.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// max |x| of a block of floats.  16-byte loads where the pointer allows, ONE atomic per workgroup: the first version's 4-byte
// loads and one atomicMax per WAVE (16 384 of them on one word for a 10 000 x 1024 query batch) took 190 us in front of every
// batch search (round 5 kernel trace) for 41 MB -- 10 us of reading.
__global__ __launch_bounds__(256) void maxabs_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t wmax[4];
  uint32_t m = 0;
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  int64_t head = 0;   // elements in front of the first 16-byte boundary
  if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) head = min<int64_t>(n, (int64_t)((16 - (reinterpret_cast<uintptr_t>(x) & 15)) >> 2));
  const int64_t n4 = (n - head) >> 2;
  const uint4* x4 = reinterpret_cast<const uint4*>(x + head);
  for (int64_t j = tid; j < n4; j += nth) {
    const uint4 v = x4[j];
    m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));  // |x| bit pattern orders like the value
  }
  if (tid < head) m = max(m, __float_as_uint(x[tid]) & 0x7fffffffu);
  const int64_t tail0 = head + 4 * n4;
  if (tid < n - tail0) m = max(m, __float_as_uint(x[tail0 + tid]) & 0x7fffffffu);
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

int sv_maxabs(segvlad_ctx* ctx, const float* x, int64_t n, float* out_host) {
  SV_HIP(ctx->s_minmax.reserve(32));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>() + 4;
  SV_HIP(hipMemsetAsync(mm, 0, 4, ctx->stream));
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (n > 0) hipLaunchKernelGGL(maxabs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, n, mm);
  uint32_t u = 0;
  SV_HIP(hipMemcpyAsync(&u, mm, 4, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  float f;
  memcpy(&f, &u, 4);
  *out_host = f;
  return SEGVLAD_OK;
}

__global__ __launch_bounds__(256) void to_f16_kernel(const float* __restrict__ X, int64_t n4, float scale,
                                                     _Float16* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(X)[j];
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h;
    h[0] = (_Float16)(v.x * scale);  // round-to-nearest-even conversion
    h[1] = (_Float16)(v.y * scale);
    h[2] = (_Float16)(v.z * scale);
    h[3] = (_Float16)(v.w * scale);
    reinterpret_cast<h4*>(out)[j] = h;
  }
}

// A handful of query rows (<= 128 x d floats): largest magnitude, the power-of-two scale that puts it in [8192, 16384) --
// the host's pow2_scale, bit for bit -- and the fp16 plane, in ONE workgroup and one launch; the scales stay on the device
// (scales[0] = query scale, scales[1] = 1 / (query scale x db_scale)): no memset, no atomics, no host round trip.
// qn_out != null: the rows' squared norms as well (one wave per row, 16 rows in flight: the arithmetic of row_sumsq_kernel,
// gemm_kernels.hip, operation for operation -- lane j sums the squares of float4 j, j + 64, ... with fmaf, then the xor
// butterfly -- so that the single-image pass sees the very bits the batch path computes), instead of a launch of its own in
// front of this one.
__global__ __launch_bounds__(1024) void query_f16_small_kernel(const float* __restrict__ X, int64_t n4, float db_scale,
                                                               _Float16* __restrict__ out, float* __restrict__ scales,
                                                               float* __restrict__ qn_out, int nq, int d,
                                                               uint32_t* __restrict__ zero, int zero_words) {
  __shared__ uint32_t wmax[16];
  __shared__ float s_scale;
  const int tid = threadIdx.x;
  // the search's flag block, zeroed here instead of by a fill launch of its own in front of the pass's dependent chain
  for (int j = tid; j < zero_words; j += 1024) zero[j] = 0u;
  if (qn_out) {
    const int lane = tid & 63;
    for (int row = tid >> 6; row < nq; row += 16) {
      const float4* x4 = reinterpret_cast<const float4*>(X + (int64_t)row * d);
      float s = 0.f;
      for (int j = lane; j < (d >> 2); j += 64) {
        const float4 v = x4[j];
        s = fmaf(v.x, v.x, s);
        s = fmaf(v.y, v.y, s);
        s = fmaf(v.z, v.z, s);
        s = fmaf(v.w, v.w, s);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) qn_out[row] = s;
    }
  }
  uint32_t m = 0;
  for (int64_t j = tid; j < n4; j += 1024) {
    const float4 v = reinterpret_cast<const float4*>(X)[j];
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, __float_as_uint(v.z) & 0x7fffffffu));
    m = max(m, __float_as_uint(v.w) & 0x7fffffffu);
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((tid & 63) == 0) wmax[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) m = max(m, wmax[w]);
    const float maxabs = __uint_as_float(m);
    float scale = 1.f;
    if (maxabs > 0.f && isfinite(maxabs)) {
      int e;
      frexpf(maxabs, &e);
      scale = ldexpf(1.f, 14 - e);
    }
    s_scale = scale;
    scales[0] = scale;
    scales[1] = 1.f / (scale * db_scale);
  }
  __syncthreads();
  const float scale = s_scale;
  for (int64_t j = tid; j < n4; j += 1024) {
    const float4 v = reinterpret_cast<const float4*>(X)[j];
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h;
    h[0] = (_Float16)(v.x * scale);
    h[1] = (_Float16)(v.y * scale);
    h[2] = (_Float16)(v.z * scale);
    h[3] = (_Float16)(v.w * scale);
    reinterpret_cast<h4*>(out)[j] = h;
  }
}

int sv_launch_query_f16_small(segvlad_ctx* ctx, const float* X, int64_t n_elems, float db_scale, uint16_t* out, float* scales_dev,
                              float* qn_out, int nq, int d, uint32_t* zero, int zero_words) {
  hipLaunchKernelGGL(query_f16_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, X, n_elems / 4, db_scale,
                     reinterpret_cast<_Float16*>(out), scales_dev, qn_out, nq, d, zero, zero_words);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

__global__ __launch_bounds__(256) void to_f16_devscale_kernel(const float* __restrict__ X, int64_t n4, const float* __restrict__ scales,
                                                              _Float16* __restrict__ out) {
  const float scale = scales[0];
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(X)[j];
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 h;
    h[0] = (_Float16)(v.x * scale);
    h[1] = (_Float16)(v.y * scale);
    h[2] = (_Float16)(v.z * scale);
    h[3] = (_Float16)(v.w * scale);
    reinterpret_cast<h4*>(out)[j] = h;
  }
}

int sv_launch_to_f16_devscale(segvlad_ctx* ctx, const float* X, int64_t n_elems, const float* scales_dev, uint16_t* out) {
  if (n_elems <= 0) return SEGVLAD_OK;
  const int64_t n4 = n_elems / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_f16_devscale_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n4, scales_dev,
                     reinterpret_cast<_Float16*>(out));
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

int sv_launch_to_f16(segvlad_ctx* ctx, const float* X, int64_t n_elems, float scale, uint16_t* out) {
  if (n_elems <= 0) return SEGVLAD_OK;
  const int64_t n4 = n_elems / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n4, scale,
                     reinterpret_cast<_Float16*>(out));
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N_>
__device__ __forceinline__ void wait_vm_lgkm0() {  // s_waitcnt needs a literal count
  static_assert(N_ == 0 || N_ == 1 || N_ == 2 || N_ == 3 || N_ == 4 || N_ == 8, "add the literal");
  if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if constexpr (N_ == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
  else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
}

// Operand tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging, no ds_write): one
// wave-instruction fills 1 KiB of the LDS image, lane l landing at base + 16 l, i.e. RP = 1024 / row_bytes rows
// of HBK fp16.  The LDS image must be lane-linear, so the bank-conflict swizzle is applied on the SOURCE side:
// the lane that owns physical chunk p of row r fetches logical chunk p ^ f(r), and the MFMA fragment reads apply
// the same involution (f(r) = (r >> 1) & 7 for 128-B rows, (r >> 2) & 3 for 64-B rows: a 16-lane ds_read_b128
// group then covers all 16 bank slots).
//   BM x BN tile, WM x WN waves, HBK k per tile, two A stages (queries: L2 resident) and NB B stages (database
//   rows stream from HBM/MALL; NB = 3 keeps their DMA two k-tiles ahead via a counted vmcnt).
// ABL == 9: phase timing (s_memtime of wave 0 at the phase boundaries, summed over workgroups; SEGVLAD_F16_CFG=90 prints it)
__device__ unsigned long long sv_f16_phase_cycles[8];
#define SV_PHASE(k)                                                                          \
  if (ABL == 9 || ABL == 12) {                                                                            \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                            \
    if (threadIdx.x == 0) atomicAdd(&sv_f16_phase_cycles[k], now_ - phase_t0);               \
    phase_t0 = now_;                                                                         \
  }

// Accumulator element for the epilogue.  FROM_AGPR (the 128 x 128 wave tiles: 256 accumulator registers per lane, held in
// AGPRs by the MFMA loop): read through an explicit v_accvgpr_read so that the value STAYS in its AGPR until this use --
// left to itself the register allocator copies all 256 to VGPRs at the loop exit and spills half of them to scratch.
template <bool FROM_AGPR>
__device__ __forceinline__ float acc_elem(const f32x16& v, int r) {
  if (FROM_AGPR) {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(v[r]));
    return x;
  }
  return v[r];
}

// fire-and-forget fp32 add at L2 (`global_atomic_add_f32` without return: no register, no wait)
__device__ __forceinline__ void sv_atomic_add_noret(float* p, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v);
#else
  (void)p;
  (void)v;
#endif
}

// POL: cache policy of the operand DMA (never changes a result): bit 0 = database rows (B) non-temporal, bit 1 = queries (A)
// PP : 0 = every wave runs the k-tile as one segment (one barrier per k-tile); PP > 0 = "ping-pong": the k-tile is cut
//      into PP phases of [load segment: LDS fragment reads + DMA issue][barrier][MFMA segment][barrier], and the second
//      half of the waves (the SIMD partners of the first half: waves w and w + NW/2 share a SIMD) runs one barrier
//      behind, so that on every SIMD one wave feeds the matrix pipe while its partner reads LDS and issues DMA.
// KBT: > 0 = BLOCKED accumulation (deep rows, e.g. raw K*D = 98 304-d descriptors): every KBT k-tiles the MFMA accumulators
//      are added into a second register set and cleared, so that the fp32 accumulation error of a dot product is bounded
//      by (2 KBT HBK + d / (KBT HBK)) 2^-24 sum|q_i r_i| instead of 2 d 2^-24 sum|q_i r_i| -- whatever the matrix pipe's
//      internal summation order is (see sv_f16_c_eps).  Needs the plain loop (PP == 0) and 2 x TM x TN x 16 accumulators.
// BIAS: the accumulators START at -||r||^2 / 2 (in the scaled domain) instead of 0, so that at the end of the k-loop they
//      hold dot - ||r||^2 / 2 -- the quantity the epilogue screens -- and the per-element subtraction of pass 1 (four of its
//      seven VALU instructions per four elements) disappears; d2~ = ||q||^2 - 2 acc / scale.  The column norms are fetched
//      ahead of the tile (before the head DMA; PERSIST: before the previous tile's epilogue).  The running sums are up to
//      1.5 x larger in magnitude, which sv_f16_c_eps accounts for.
// EPI : epilogue.  0 = workgroup-level (one global atomic per row and tile); 1 = wave-private (see the epilogue)
// MF  : MFMA shape.  0 = v_mfma_f32_32x32x16_f16 (wave tile = TM x TN tiles of 32 x 32); 1 = v_mfma_f32_16x16x32_f16 (the same
//      64 x 128 wave tile as 4 x 8 tiles of 16 x 16, a k-step of 32): the same flops, LDS fragment bytes and accumulator
//      registers, but a QUARTER of the accumulator read-modify-write traffic per flop inside the matrix pipe.  The chip is
//      power-limited on this kernel (tools/ubench/mfma_peak: random operands sustain 1.71-1.76 PF in the 32 x 32 x 16 shape
//      and 1.92-2.01 PF in the 16 x 16 x 32 shape, MFMA only), so the shape that needs less energy per flop is the faster one.
// The buffer-resource type and its two builtins exist in the DEVICE pass only; the host pass, which parses the kernel bodies
// too (and silently drops a kernel's launch stub when its body does not type-check there), sees inert stand-ins.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t sv_rsrc_t;
#define SV_BUF_RSRC(base) __builtin_amdgcn_make_buffer_rsrc((void*)(base), 0, (int)0xffffffffu, 0x00020000)
#define SV_BUF_LOAD_LDS(rs, ldsptr, voff, soff, aux) __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (ldsptr), 16, (voff), (soff), 0, (aux))
#else
typedef int sv_rsrc_t;
#define SV_BUF_RSRC(base) 0
#define SV_BUF_LOAD_LDS(rs, ldsptr, voff, soff, aux) ((void)(rs), (void)(ldsptr), (void)(voff), (void)(soff))
#endif

// KFL : > 0 = blocked accumulation WITHOUT a second register set (deep rows on the 256 x 256 ping-pong kernel, whose 128 accumulators
//      per lane leave no room for one): every KFL k-tiles a wave ADDS its accumulators into its own slice of a global scratch
//      (kscr: [workgroup][wave][128][64 lanes] fp32, all zero between tiles) with non-returning `global_atomic_add_f32` -- fire and
//      forget: no temporary registers, no wait, the fp32 additions happen in L2, one address is only ever touched by one lane, in
//      program order -- and clears them; behind the last k-tile the rest is added too, the totals are read back INTO the accumulator
//      registers (system-coherent loads: the lines of the previous tile may sit in this CU's vector cache) and the slice is zeroed
//      for the next tile.  Error: sv_f16_c_eps with kb = KFL x HBK (the L2's additions are the "block sums" of that bound).
// BUF : the operand DMA as `buffer_load_dwordx4 ... lds` -- an SGPR resource per operand and tile, ONE never-rewritten 32-bit
//      VGPR offset per piece, the k-offset in an SGPR -- instead of `global_load_lds_dwordx4` on a 64-bit per-lane pointer that
//      every piece re-forms in the same VGPR pair (a write-after-read stall behind the previous piece's address read).  Needs
//      every piece offset of a tile below 4 GiB (the launcher checks).  tools/ubench/mfma_peak: 1317 -> 1345 TF for the loop
//      of this kernel's byte : flop ratio; the deep-row kernel 126.4 -> 125.2 ms; the ping-pong batch kernel 18.19 -> 18.70 ms
//      (slower: its default stays global_load_lds).
template <int BM, int BN, int WM, int WN, int HBK, int NB, int ABL = 0, bool PERSIST = false, int POL = 0, int PP = 0, int KBT = 0,
          bool BIAS = false, int EPI = 0, int MF = 0, bool BUF = false, int DSPLIT = 0, int SKIP = 0, int KFL = 0>
__global__ __launch_bounds__(64 * WM * WN) void knn_f16_filter_kernel(
    const uint16_t* __restrict__ Qh, const uint16_t* __restrict__ Rh, int M, int N, int d, int b_stride, int tiles_m, int gm,
    int seq_total, int walk,
    float inv_scale, const float* __restrict__ qn, const float* __restrict__ rn, const float* __restrict__ thr,
    int64_t thr_ld, float eps_mult, float c_eps, float rn_max, uint32_t* __restrict__ cand_cnt,
    float* __restrict__ cand_d2, uint32_t* __restrict__ cand_id, int cap, const float* __restrict__ inv_scale_dev,
    float* __restrict__ kscr) {
  // single-image searches leave the query scale on the device (no host round trip in front of the pass): [0] = scale,
  // [1] = 1 / (query scale x database scale)
  if (inv_scale_dev) inv_scale = inv_scale_dev[1];
  constexpr int NW = WM * WN;
  constexpr int AUXA = (POL & 2) ? 2 : 0, AUXB = (POL & 1) ? 2 : 0;   // aux = 2: "nt" (streaming) hint
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr bool ACC_A = TM * TN > 8;   // 256 accumulator registers per lane: they live in AGPRs (see acc_elem)
  static_assert(KBT == 0 || ((PP == 0 || MF == 1) && !ACC_A), "blocked accumulation: two accumulator sets in VGPRs; plain loop, or the 16 x 16 x 32 loops");
  constexpr int RB = HBK * 2;            // row bytes per k-tile
  constexpr int CH = RB / 16;            // 16-B chunks per row (4 or 8)
  constexpr int RP = 1024 / RB;          // rows per 1-KiB DMA piece
  constexpr int KS = HBK / 16;           // MFMA k-steps per tile
  constexpr int PA = BM * RB, PB = BN * RB;
  constexpr int JA = BM / RP / NW, JB = BN / RP / NW;  // DMA pieces per wave and operand
  static_assert(JA * RP * NW == BM && JB * RP * NW == BN && (PP > 0 || (JA + JB) % (MF ? 2 : KS) == 0), "tile/wave geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned long long phase_t0 = (ABL == 9 || ABL == 12) ? __builtin_amdgcn_s_memtime() : 0ull;
  // XCD-aware tile order.  Workgroup ids are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2); the 32
  // workgroups an XCD runs side by side form one gm x (32/gm) block of tiles, so that they share their query and
  // database rows in that L2 while they march over k (tm-fastest order made every XCD fetch every database row).
  // PERSIST: 8 x 32 workgroups stay resident and walk their XCD's sequence 32 positions at a time; the head of the next
  // tile (A(0), B(0), B(1)) is requested before the epilogue of the current one, which hides the ~2 us a fresh
  // workgroup spends waiting for its first operands (6 % of the kernel: one workgroup per CU, nothing else covers it).
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: LDS DMA bases live in M0
  const int wm = w / WN, wn = w % WN;
  // KFL: this wave's slice of the flush scratch, as a scalar (wave-uniform) base address -- formed where it is used (a few SALU
  // instructions every 64 k-tiles) rather than kept live through the main loop
  auto kscr_base = [&]() -> float* {
    const unsigned long long ka = (unsigned long long)(size_t)(kscr + ((size_t)blockIdx.x * NW + (size_t)w) * (128 * 64));
    return reinterpret_cast<float*>((size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ka >> 32)) << 32) |
                                             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ka)));
  };
  const int64_t ldb = (int64_t)d * b_stride;
  // SKIP > 0 ("the complement of a sample"): operand row j is database row j + j / (SKIP - 1) + 1 -- the rows that are NOT multiples
  // of SKIP, in order.  The level before the last one has already filtered the multiples of SKIP (its stride-SKIP sample) with the same
  // arithmetic; its survivors stay in the candidate lists (select mode 0 with carry) and the last level does not compute them again.
  auto grow = [&](int64_t j) -> int64_t {
    if constexpr (SKIP > 0) return j + (int64_t)((uint32_t)j / (uint32_t)(SKIP - 1)) + 1;
    else return j * b_stride;
  };
  const int64_t ldr = (int64_t)d;   // fp16 elements per database row: operand row j starts at Rh + grow(j) * ldr
  const int ntiles = d / HBK;
  const int tiles_n = (N + BN - 1) / BN;
  auto swz = [](int r, int c) { return CH == 8 ? (c ^ ((r >> 1) & 7)) : (c ^ ((r >> 2) & 3)); };
  // walk bit 0 ("query-block-resident"): an XCD keeps ONE block of gm query tiles while it steps through its share of the
  // database blocks (blocks xcd, xcd + 8, ... -- the direction alternates from one query block to the next), instead of
  // meeting a different query block at every step: the gm query tiles (gm x 512 KiB of fp16 rows at d = 1024) are then
  // re-referenced by every step.  walk bit 1 ("serpentine k"): odd steps run their k-tiles from the END of the rows to the
  // start.  Under an LRU-like L2 a cyclic re-reference of a working set larger than the cache (4 MiB of query tiles + 2 MiB
  // of database tiles per step against 4 MiB of L2 per XCD) never hits; reversing the direction makes the most recently
  // used half of the resident block hit (the filter's result does not depend on the accumulation order: the margin
  // sv_f16_c_eps bounds ANY order).
  auto tile_of = [&](int sq, int& tm_, int& tn_) -> bool {   // sq: position in this XCD's sequence
    const int xcd = blockIdx.x & 7, within = sq & 31;
    const int gn = 32 / gm, sm_cnt = (tiles_m + gm - 1) / gm;
    if (walk & 1) {
      // sn_cnt database blocks = 8 nbf + rem: every XCD sweeps nbf of them per query block; the sm_cnt x rem left-over
      // (query block, database block) pairs are dealt round-robin to the XCDs at the end of the sequence (all of a query
      // block's left-overs on XCD 0 would lengthen that XCD's sequence by sm_cnt steps: +0.7 % on the bench shape)
      const int sn_cnt = (tiles_n + gn - 1) / gn, nbf = sn_cnt >> 3, rem = sn_cnt - 8 * nbf;
      const int s_ = sq >> 5;
      int qb, nb;
      if (s_ < sm_cnt * nbf) {
        qb = s_ / nbf;
        int j = s_ - qb * nbf;
        if (qb & 1) j = nbf - 1 - j;
        nb = j * 8 + xcd;
      } else {
        const int idx = (s_ - sm_cnt * nbf) * 8 + xcd;
        if (rem == 0 || idx >= sm_cnt * rem) return false;
        qb = idx / rem;
        nb = 8 * nbf + (idx - qb * rem);
      }
      tm_ = qb * gm + within % gm;
      tn_ = nb * gn + within / gm;
    } else {
      const int st = (sq >> 5) * 8 + xcd;
      tm_ = (st % sm_cnt) * gm + within % gm;
      tn_ = (st / sm_cnt) * gn + within / gm;
    }
    return tm_ < tiles_m && tn_ < tiles_n;
  };
  // serpentine k: element offset of k-tile x of a step that runs backwards (wave-uniform)
  // walk bit 2 ("rotated k start"): workgroup (i_m, i_n) of the XCD's gm x gn block starts its k-loop at k-tile
  // (i_m ntiles / gm + i_n) mod ntiles and wraps around.  The gm workgroups that share a database tile (and the gn that share a
  // query tile) otherwise ask for the same k-slice at the same moment: one L2 miss, everybody waiting out its HBM latency;
  // rotated, a slice is fetched by the first workgroup that reaches it and is an L2 HIT for the others, one to two k-tiles
  // later.  The k-order of a step: rot, rot + 1, ... (mod ntiles), or -- serpentine, odd steps -- rot - 1, rot - 2, ...: the
  // reversed step starts on the slice the previous one ended on.  (rev_of packs both: bit 0 = reversed, bits 1.. = rot.)
  auto rev_of = [&](int sq) -> int {
    int rot = 0;
    if (walk & 4) {
      const int within = sq & 31, im = within % gm, in_ = within / gm;
      rot = (im * (ntiles >= gm ? ntiles / gm : 1) + in_) % ntiles;
    }
    return 2 * rot + (((walk & 2) && ((sq >> 5) & 1)) ? 1 : 0);
  };
  auto kofs = [&](int rev_, int x) -> int {
    const int rot = rev_ >> 1;
    int i = (rev_ & 1) ? rot - 1 - x : rot + x;
    if (i < 0) i += ntiles;
    if (i >= ntiles) i -= ntiles;
    return i * HBK;
  };
  int tm, tn;
  int seq = (int)(blockIdx.x >> 3);
  // PERSIST: resident workgroups per XCD = the stride of a workgroup through its XCD's sequence (bits 8.. of `walk`; 32 = one per
  // CU.  Fewer leave CUs to kernels of other streams -- the describe stage of the next batch -- for the whole launch: option f16_persist_wgs)
  const int pstep = (walk >> 8) > 0 ? (walk >> 8) : 32;
  if (PERSIST) {
    while (seq < seq_total && !tile_of(seq, tm, tn)) seq += pstep;
    if (seq >= seq_total) return;
  } else if (gm > 0) {
    if (!tile_of(seq, tm, tn)) return;
  } else {
    tm = blockIdx.x % tiles_m;
    tn = blockIdx.x / tiles_m;
  }
  // LDS stages.  Plain: A stages at 0, PA; B stages behind them.  PERSIST (five 32-KiB slots): the first stages of both
  // operands and B's second sit in slots 3, 4, 2 -- outside the epilogue's scratch (slots 0, 1) -- so that the next
  // tile's head can land while the epilogue runs.
  static_assert(!PERSIST || (PA == PB && NB == 3 && 5 * PA <= 160 * 1024), "persistent layout: five equal slots");
  auto a_off = [](int st_) { return PERSIST ? (st_ == 0 ? 3 * PA : 0) : st_ * PA; };
  auto b_off = [](int st_) { return PERSIST ? (st_ == 0 ? 4 * PA : (st_ == 1 ? 2 * PA : PA)) : 2 * PA + st_ * PB; };
  const int lrow_p = l / CH, lch = l % CH;
  // head of a tile: A(0), B(0) and (NB == 3) B(1), by global->LDS DMA
  // BUF: byte offset of this lane's 16 bytes of piece j inside its tile (rows beyond the operand clamp to its last row: they
  // are never emitted), and the tile's buffer resource
  auto voff_a = [&](int64_t m0_, int j) -> unsigned {
    const int row = (w * JA + j) * RP + lrow_p;
    const int rr = (m0_ + row < M) ? row : (int)((int64_t)M - 1 - m0_);
    return (unsigned)rr * (unsigned)(d * 2) + 16u * (unsigned)swz(row, lch);
  };
  auto voff_b = [&](int64_t n0_, int j) -> unsigned {
    const int row = (w * JB + j) * RP + lrow_p;
    const int rr = (n0_ + row < N) ? row : (int)((int64_t)N - 1 - n0_);
    if constexpr (SKIP > 0) return (unsigned)(grow(n0_ + rr) - grow(n0_)) * (unsigned)(d * 2) + 16u * (unsigned)swz(row, lch);
    return (unsigned)rr * (unsigned)(ldb * 2) + 16u * (unsigned)swz(row, lch);
  };
  auto rsrc_of = [](const uint16_t* base) -> sv_rsrc_t { return SV_BUF_RSRC(base); };
  auto issue_head = [&](int tm_, int tn_, int rev_) {
    const int64_t m0_ = (int64_t)tm_ * BM, n0_ = (int64_t)tn_ * BN;
    const int k0_ = kofs(rev_, 0), k1_ = kofs(rev_, 1);
    if constexpr (BUF) {
      const sv_rsrc_t ra = rsrc_of(Qh + m0_ * d), rb_ = rsrc_of(Rh + grow(n0_) * ldr);
#pragma unroll
      for (int j = 0; j < JA; ++j)
        SV_BUF_LOAD_LDS(ra, (lptr_t)(lds + a_off(0) + (w * JA + j) * 1024), voff_a(m0_, j), 2 * k0_, AUXA);
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const unsigned vo = voff_b(n0_, j);
        SV_BUF_LOAD_LDS(rb_, (lptr_t)(lds + b_off(0) + (w * JB + j) * 1024), vo, 2 * k0_, AUXB);
      }
      if (NB == 3 && ntiles > 1) {
#pragma unroll
        for (int j = 0; j < JB; ++j)
          SV_BUF_LOAD_LDS(rb_, (lptr_t)(lds + b_off(1) + (w * JB + j) * 1024), voff_b(n0_, j), 2 * k1_, AUXB);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int row = (w * JA + j) * RP + lrow_p;
      const int64_t qa = (m0_ + row < M) ? (m0_ + row) : (int64_t)(M - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Qh + qa * d + 8 * swz(row, lch) + k0_), (lptr_t)(lds + a_off(0) + (w * JA + j) * 1024), 16,
                                       0, AUXA);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int row = (w * JB + j) * RP + lrow_p;
      const int64_t rb = (n0_ + row < N) ? (n0_ + row) : (int64_t)(N - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(Rh + grow(rb) * ldr + 8 * swz(row, lch) + k0_), (lptr_t)(lds + b_off(0) + (w * JB + j) * 1024), 16,
                                       0, AUXB);
    }
    if (NB == 3 && ntiles > 1) {
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int row = (w * JB + j) * RP + lrow_p;
        const int64_t rb = (n0_ + row < N) ? (n0_ + row) : (int64_t)(N - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(Rh + grow(rb) * ldr + 8 * swz(row, lch) + k1_),
                                         (lptr_t)(lds + b_off(1) + (w * JB + j) * 1024), 16, 0, AUXB);
      }
    }
  };
  // BIAS: this lane's column norms of the (next) tile, requested BEFORE the tile's head so that the head's wait covers them
  static_assert(MF == 0 || (PP >= 0 && BIAS && EPI == 1 && BM / (32 * WM) == 2 && (BN / (32 * WN) == 4 || BN / (32 * WN) == 2) && HBK == 64),
                "16 x 16 x 32 MFMA: the biased kernels with the wave-private epilogue, wave tiles of 64 rows x 128 or 64 columns");
  constexpr int TN16 = (BN / WN) / 16;                // MF = 1: column tiles of 16 per wave (8, or 4 with blocked accumulation)
  constexpr int CW = MF ? 16 : 32;                    // columns (and rows) per MFMA tile
  constexpr int NCN = (BN / WN) / CW;                 // column tiles per wave = column norms per lane
  float cnn[NCN];
  auto load_cn = [&](int tn_) {
#pragma unroll
    for (int nt = 0; nt < NCN; ++nt) {
      const int64_t colj = (int64_t)tn_ * BN + wn * (32 * TN) + nt * CW + (threadIdx.x & (CW - 1));
      cnn[nt] = (colj < N) ? rn[grow(colj)] : INFINITY;  // +inf: columns beyond N never pass
    }
  };
  if (BIAS) load_cn(tn);
  int rev = PERSIST ? rev_of(seq) : 0;   // (only the persistent walk has steps to alternate / rotate)
  issue_head(tm, tn, rev);
  if (NB == 3 && ntiles > 1)
    wait_vm_lgkm0<JB>();
  else
    wait_vm_lgkm0<0>();
  // (the barrier that publishes the head to the other waves is the one at the top of the tile loop)

  for (;;) {   // PERSIST: one iteration per tile; otherwise a single pass
  __builtin_amdgcn_s_barrier();   // head of this tile landed for every wave; (PERSIST) every wave left the previous epilogue
  // the lane id is re-derived from an opaque copy in every iteration: otherwise the hundreds of constant addresses of
  // the unrolled epilogue are hoisted out of the tile loop and spill (713 VGPRs)
  int lane_opaque = (int)threadIdx.x;
  asm volatile("" : "+v"(lane_opaque));
  const int tid = lane_opaque, l = tid & 63, i = l & 31, kk = l >> 5;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  float cn[NCN];
  if (BIAS) {
#pragma unroll
    for (int nt = 0; nt < NCN; ++nt) cn[nt] = cnn[nt];
  }
  f32x16 acc[MF ? 1 : TM][MF ? 1 : TN];
  f32x4v acc16[MF ? 4 : 1][MF ? TN16 : 1];   // MF = 1: element j of tile (mt, nt) = row mt*16 + 4*(lane>>4) + j, column nt*16 + (lane&15)
  f32x4v accb16[(MF && KBT > 0) ? 4 : 1][(MF && KBT > 0) ? TN16 : 1];   // blocked accumulation: the sum of the finished k-blocks (+ the bias)
  if constexpr (MF == 0) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const float a0_ = BIAS ? -(cn[b] * (0.5f / inv_scale)) : 0.f;   // (the same product the epilogue forms: cnh)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = a0_;
      }
  } else {
#pragma unroll
    for (int b = 0; b < TN16; ++b) {
      const float a0_ = -(cn[b] * (0.5f / inv_scale));
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (KBT > 0) {   // the bias lives in the sum of the blocks; every block starts at zero
            accb16[a][b][r] = a0_;
            acc16[a][b][r] = 0.f;
          } else {
            acc16[a][b][r] = a0_;
          }
        }
    }
  }
  f32x16 accb[KBT > 0 ? TM : 1][KBT > 0 ? TN : 1];   // blocked accumulation: the sum of the finished k-blocks
  if (KBT > 0) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[KBT > 0 ? a : 0][KBT > 0 ? b : 0][r] = 0.f;
  }

  // epilogue inputs, requested now so that their latency hides under the main loop (one workgroup per CU: nothing
  // else would cover it): row tid's ||q||^2 and threshold, this lane's column norms
  static_assert(BM <= 64 * NW, "one thread per query row stages the epilogue's row record");
  float pre_q2 = 0.f, pre_thr = 0.f;
  if (EPI == 1) {   // wave-private epilogue: lane l stages row l of THIS wave's 64 query rows
    const int64_t qrow = m0 + wm * (32 * TM) + l;
    if (qrow < M) {
      pre_q2 = qn[qrow];
      pre_thr = thr[qrow * thr_ld];
    }
  } else if (tid < BM && m0 + tid < M) {
    pre_q2 = qn[m0 + tid];
    pre_thr = thr[(m0 + tid) * thr_ld];
  }
  if constexpr (!BIAS) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      const int64_t colj = n0 + wn * (32 * TN) + nt * 32 + i;
      cn[nt] = (colj < N) ? rn[grow(colj)] : INFINITY;  // +inf: columns beyond N never pass
    }
  }

  // per-lane source rows of this wave's DMA pieces (clamped: edge rows are never emitted)
  const uint16_t* srcA[BUF ? 1 : JA];
  const uint16_t* srcB[BUF ? 1 : JB];
  unsigned voA[BUF ? JA : 1], voB[BUF ? JB : 1];
  const sv_rsrc_t rsA = rsrc_of(Qh + m0 * d), rsB = rsrc_of(Rh + grow(n0) * ldr);
  if constexpr (BUF) {
#pragma unroll
    for (int j = 0; j < JA; ++j) voA[j] = voff_a(m0, j);
#pragma unroll
    for (int j = 0; j < JB; ++j) voB[j] = voff_b(n0, j);
  } else {
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      const int row = (w * JA + j) * RP + lrow_p;
      const int64_t qa = (m0 + row < M) ? (m0 + row) : (int64_t)(M - 1);
      srcA[j] = Qh + qa * d + 8 * swz(row, lch);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int row = (w * JB + j) * RP + lrow_p;
      const int64_t rb = (n0 + row < N) ? (n0 + row) : (int64_t)(N - 1);
      srcB[j] = Rh + grow(rb) * ldr + 8 * swz(row, lch);
    }
  }
  constexpr int BAHEAD = NB - 1;  // how many k-tiles ahead the B DMA runs (A always runs one ahead)
  // DMA piece p of iteration kt: pieces 0..JA-1 belong to A(kt+1), JA..JA+JB-1 to B(kt+BAHEAD)
  auto dma_piece = [&](int piece, int kt, int ia_next, int ib_next) {
    if (ABL == 3) return;  // ablation: no DMA in the loop
    if (ABL == 13 && piece < JA) return;    // ablation: B only
    if (ABL == 14 && piece >= JA) return;   // ablation: A only
    if constexpr (BUF) {
      if (piece < JA) {
        if (kt + 1 < ntiles)
          SV_BUF_LOAD_LDS(rsA, (lptr_t)(lds + a_off(ia_next) + (w * JA + piece) * 1024), voA[piece], 2 * kofs(rev, kt + 1), AUXA);
      } else {
        const int j = piece - JA;
        if (kt + BAHEAD < ntiles)
          SV_BUF_LOAD_LDS(rsB, (lptr_t)(lds + b_off(ib_next) + (w * JB + j) * 1024), voB[j], 2 * kofs(rev, kt + BAHEAD), AUXB);
      }
    } else if (piece < JA) {
      if (kt + 1 < ntiles)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA[piece] + kofs(rev, kt + 1)),
                                         (lptr_t)(lds + a_off(ia_next) + (w * JA + piece) * 1024), 16, 0, AUXA);
    } else {
      const int j = piece - JA;
      if (kt + BAHEAD < ntiles)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + kofs(rev, kt + BAHEAD)),
                                         (lptr_t)(lds + b_off(ib_next) + (w * JB + j) * 1024), 16, 0, AUXB);
    }
  };
  SV_PHASE(0)  // prologue: first tiles landed

  int ia = 0, ib = 0;
  const int fa0 = wm * (32 * TM) + i, fb0 = wn * (32 * TN) + i;
  if constexpr (PP > 0) {
    constexpr int PPn = PP > 0 ? PP : 1;
    constexpr int PH = KS / PPn;            // MFMA k-steps per phase
    constexpr int DPP = (JA + JB) / PPn;    // DMA pieces per phase and wave
    static_assert(PP == 0 || (KS % PPn == 0 && (JA + JB) % PPn == 0 && PH >= 1), "phase geometry");
    const bool lag = w >= NW / 2;           // wave-uniform (w comes from readfirstlane)
    if (lag) __builtin_amdgcn_s_barrier(); // the second half of the waves runs one barrier (= one segment) behind
    // DSPLIT < 0 (APF): four of a phase's twelve fragments are read at the END of the previous phase's MFMA segment (16 more
    // live registers) -- a load segment is then 8 fragment reads + the DMA issue, and the partner's 32 MFMAs have less to
    // cover.  Phase 1 takes its A fragments that way (same stages as phase 0: landed).  Phase 0 of the NEXT k-tile takes its
    // first four B fragments: B(kt+1) was requested a whole k-tile earlier, and a vmcnt(JA) added to phase 0's load segment
    // (behind the issue of A(kt+1)) makes every wave's pieces of it certain one barrier before the earliest such read --
    // the A stage of the next tile would not do, the lagging half only waits for it one segment after the leading half's
    // MFMA segment that would read it.
    constexpr bool APF = DSPLIT == -1;
    // DSPLIT == -2 (DFIRST): a load segment issues its DMA pieces FIRST and its twelve fragment reads behind them (raw
    // ds_read_b128: the compiler would put s_waitcnt vmcnt(0) in front of any LDS read it can see behind a DMA) -- asks
    // whether a DMA's issue is cheaper with no LDS read of the wave in flight (A/B).
    constexpr bool DFIRST = DSPLIT == -2;
    constexpr int DSP = DSPLIT > 0 ? DSPLIT : 0;
    static_assert(!APF || (MF == 1 && PPn == 2 && TN16 >= 4 && NB == 3), "fragment prefetch: the 16 x 16 x 32 ping-pong loop");
    f16x8 apf[4];
    if constexpr (APF) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rb = wn * (16 * TN16) + 16 * t + (l & 15);
        apf[t] = *reinterpret_cast<const f16x8*>(lds + b_off(ib) + rb * RB + swz(rb, l >> 4) * 16);
      }
    }
    // (KFL: the k-tiles run in blocks of KFL; the flush sits BETWEEN two runs of the inner loop, not behind a branch inside it -- with
    //  the branch inside, the 128 accumulators met a zeroed copy of themselves at the loop's back edge and 40 registers spilled)
    int kt = 0;
    for (int kstop = (KFL > 0 && KFL < ntiles) ? KFL : ntiles;;) {
    for (; kt < kstop; ++kt) {
      const int ibn = (ib + BAHEAD >= NB) ? ib + BAHEAD - NB : ib + BAHEAD;
      const unsigned char* SA = lds + a_off(ia);
      const unsigned char* SB = lds + b_off(ib);
#pragma unroll
      for (int ph = 0; ph < PPn; ++ph) {
        // ---- load segment: fragments of this phase's k-steps, this phase's share of the DMA pieces ----
        f16x8 a[MF ? 1 : PH][MF ? 4 : TM], b[MF ? 1 : PH][MF ? TN16 : TN];
        if constexpr (MF == 0) {
#pragma unroll
          for (int k2 = 0; k2 < PH; ++k2) {
            const int cl = 2 * (ph * PH + k2) + kk;
#pragma unroll
            for (int t = 0; t < TM; ++t) {
              const int ra = fa0 + 32 * t;
              a[k2][t] = *reinterpret_cast<const f16x8*>(SA + ra * RB + swz(ra, cl) * 16);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
              const int rb = fb0 + 32 * t;
              b[k2][t] = *reinterpret_cast<const f16x8*>(SB + rb * RB + swz(rb, cl) * 16);
            }
          }
        } else {
          // one k-step of 32 per phase: lane l holds row / column (l & 15) of a 16-wide tile, k-chunk (l >> 4) of the step's four
          // (the source-side swizzle of the 128-byte rows is conflict-free for this pattern too: a 16-lane ds_read_b128 group
          //  covers rows {0-3, 12-15} of one chunk and rows {4-11} of the next, 16 distinct bank groups)
          static_assert(MF == 0 || (PH == 2 && CH == 8), "a phase = 32 k of 128-byte rows");
          const int cl = 4 * ph + (l >> 4);
          if constexpr (DFIRST) {
            static_assert(!DFIRST || TN16 == 8, "DMA-first load segment: 64 x 128 wave tiles");
#pragma unroll
            for (int pz = 0; pz < DPP; ++pz) dma_piece(ph * DPP + pz, kt, ia ^ 1, ibn);
            // (rows 16 t apart: the swizzle term (row >> 1) & 7 does not depend on t -> one lane address per operand, t in the offset)
            const int ra0 = wm * 64 + (l & 15), rb0 = wn * (16 * TN16) + (l & 15);
            const unsigned aa = (unsigned)(size_t)(lptr_t)lds + a_off(ia) + ra0 * RB + swz(ra0, cl) * 16;
            const unsigned ab = (unsigned)(size_t)(lptr_t)lds + b_off(ib) + rb0 * RB + swz(rb0, cl) * 16;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\t"
                         "ds_read_b128 %3, %4 offset:6144"
                         : "=&v"(a[0][0]), "=&v"(a[0][1]), "=&v"(a[0][2]), "=&v"(a[0][3]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\tds_read_b128 %2, %4 offset:4096\n\t"
                         "ds_read_b128 %3, %4 offset:6144"
                         : "=&v"(b[0][0]), "=&v"(b[0][1]), "=&v"(b[0][2]), "=&v"(b[0][3]) : "v"(ab));
            asm volatile("ds_read_b128 %0, %4 offset:8192\n\tds_read_b128 %1, %4 offset:10240\n\tds_read_b128 %2, %4 offset:12288\n\t"
                         "ds_read_b128 %3, %4 offset:14336"
                         : "=&v"(b[0][4 % TN16]), "=&v"(b[0][5 % TN16]), "=&v"(b[0][6 % TN16]), "=&v"(b[0][7 % TN16]) : "v"(ab));
          } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int ra = wm * 64 + 16 * t + (l & 15);
            if (APF && ph == 1) a[0][t] = apf[t];
            else a[0][t] = *reinterpret_cast<const f16x8*>(SA + ra * RB + swz(ra, cl) * 16);
          }
#pragma unroll
          for (int t = 0; t < TN16; ++t) {
            const int rb = wn * (16 * TN16) + 16 * t + (l & 15);
            if (APF && ph == 0 && t < 4) b[0][t] = apf[t];
            else b[0][t] = *reinterpret_cast<const f16x8*>(SB + rb * RB + swz(rb, cl) * 16);
          }
          }
        }
        // DSPLIT > 0: the last DSPLIT pieces of a phase are issued from its MFMA segment (between the MFMAs) instead of its
        // load segment -- the load segment (12 fragment reads + the DMA issue) is what the partner's 32 MFMAs have to cover.
        // The pieces of the LAST phase that move are B(kt+2)'s (pieces >= JA): they are issued behind the k-tile's wait, so
        // the wait leaves only the JB - DSPLIT of them that are in flight by then (older ones retire first: in order).
        static_assert(DSP == 0 || (MF == 1 && PPn == 2 && DPP > DSP && JB >= DSP && NB == 3), "split DMA issue: the 16 x 16 x 32 ping-pong loop");
        if constexpr (!DFIRST) {
#pragma unroll
          for (int pz = 0; pz < DPP - DSP; ++pz) dma_piece(ph * DPP + pz, kt, ia ^ 1, ibn);
        }
        if (ph == PPn - 1) {
          // last load segment of the k-tile: this wave's pieces of A(kt+1) and B(kt+1) have landed (B(kt+2), the
          // youngest JB DMA instructions, may still fly).  The barriers between here and the first read of tile kt+1
          // (one for the leading half, two for the lagging half) make that true for every wave's pieces.
          if (NB == 3 && kt + 2 < ntiles)
            wait_vm_lgkm0<JB - DSP>();
          else
            wait_vm_lgkm0<0>();
        } else if (APF && ph == 0) {
          if (kt + 1 < ntiles) wait_vm_lgkm0<JA>();   // B(kt+1) landed (A(kt+1), just issued, may fly)
          else wait_vm_lgkm0<0>();
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if constexpr (DFIRST && MF == 1) {   // (the raw reads' registers are final only behind the wait above)
          asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]));
#pragma unroll
          for (int t = 0; t < TN16; ++t) asm volatile("" : "+v"(b[0][t]));
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment ----
        __builtin_amdgcn_s_setprio(1);
        if constexpr (MF == 0) {
#pragma unroll
          for (int k2 = 0; k2 < PH; ++k2)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
              for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = MFMA_F16(a[k2][mt], b[k2][nt], acc[mt][nt]);
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < TN16; ++nt)
              acc16[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][mt], b[0][nt], acc16[mt][nt], 0, 0, 0);
            if constexpr (DSP > 0) {   // one moved piece behind each of the first DSPLIT rows of MFMAs
              if (mt < DSP) {
                __builtin_amdgcn_sched_barrier(0);
                dma_piece(ph * DPP + (DPP - DSP) + mt, kt, ia ^ 1, ibn);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if constexpr (APF) {
            __builtin_amdgcn_sched_barrier(0);
            if (ph == 0) {           // phase 1's A fragments
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int ra = wm * 64 + 16 * t + (l & 15);
                apf[t] = *reinterpret_cast<const f16x8*>(SA + ra * RB + swz(ra, 4 + (l >> 4)) * 16);
              }
            } else if (kt + 1 < ntiles) {   // the next k-tile's first four B fragments of phase 0
              const unsigned char* SN = lds + b_off((ib + 1 >= NB) ? 0 : ib + 1);
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int rb = wn * (16 * TN16) + 16 * t + (l & 15);
                apf[t] = *reinterpret_cast<const f16x8*>(SN + rb * RB + swz(rb, l >> 4) * 16);
              }
            }
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      ia ^= 1;
      ib = (ib + 1 >= NB) ? 0 : ib + 1;
      if constexpr (MF == 1 && KBT > 0) {   // close a k-block (this wave's registers only: no synchronisation; wave-uniform)
        if ((kt + 1) % (KBT > 0 ? KBT : 1) == 0 || kt + 1 == ntiles) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN16; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                accb16[mt][nt][r] += acc16[mt][nt][r];
                acc16[mt][nt][r] = (kt + 1 == ntiles) ? accb16[mt][nt][r] : 0.f;   // last block: acc = the total
              }
        }
      }
    }
    if (KFL == 0 || kt >= ntiles) break;
    if constexpr (KFL > 0) {   // close a k-block into the wave's global scratch slice (see KFL)
      static_assert(KFL == 0 || (MF == 1 && PP > 0 && KBT == 0 && PERSIST && TN16 == 8), "flushed blocks: the persistent 16 x 16 x 32 ping-pong kernel");
      float* const kscr_w = kscr_base() + l;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN16; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sv_atomic_add_noret(kscr_w + 64 * ((mt * TN16 + nt) * 4 + r), acc16[mt][nt][r]);
            acc16[mt][nt][r] = 0.f;
          }
    }
    kstop = (kt + (KFL > 0 ? KFL : 1) < ntiles) ? kt + (KFL > 0 ? KFL : 1) : ntiles;
    }
    if (!lag) __builtin_amdgcn_s_barrier();  // the leading half waits for the lagging half's last MFMA segment
    if constexpr (KFL > 0) {
      if (ntiles > KFL) {   // the last block joins the flushed ones; totals back into the accumulators; the slice is left zero
        float* const kscr_w = kscr_base() + l;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sv_atomic_add_noret(kscr_w + 64 * ((mt * TN16 + nt) * 4 + r), acc16[mt][nt][r]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)   // (agent-scope load: from L2, not from a line the previous tile left in this CU's vector cache)
              acc16[mt][nt][r] = __hip_atomic_load(kscr_w + 64 * ((mt * TN16 + nt) * 4 + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 16 * TN16; ++j) __hip_atomic_store(kscr_w + 64 * j, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else if constexpr (PP < 0) {
    // One wave per SIMD (4 waves of 128 x 128: 0.5 LDS fragment reads per MFMA instead of 0.75): nothing else hides a
    // wave's LDS latency, so the loop is software-pipelined -- the fragments of k-step s+1 are read (into the other half
    // of a register double buffer) before the MFMAs of step s are issued, and the ONE barrier of a k-tile sits before
    // its last MFMA group: every wave has then finished reading the tile's LDS stages (they may be overwritten), the
    // next tile's operands have landed for every wave, and its first fragments are fetched under that MFMA group.
    static_assert(PP >= 0 || (KS % 2 == 0 && TM == 4 && TN == 4 && CH == 8), "register double buffer; 128 x 128 wave tiles of 128-B rows");
    // The fragment reads are raw ds_read_b128 (inline asm) with counted s_waitcnt lgkmcnt: while a global->LDS DMA is
    // pending the compiler only ever waits with lgkmcnt(0) / vmcnt(0) before an LDS read it can see (it cannot tell the
    // DMA'd stages from the fragments' addresses), which would serialise the prefetch behind the MFMA group it is meant
    // to overlap.  The waits below carry the fragment registers as operands so that no MFMA can be scheduled above them.
    constexpr int DPS = (JA + JB) / KS;   // DMA pieces per k-step and wave
    f16x8 fa[2][4], fb[2][4];
    // row ra = fa0 + 32 t: the swizzle term (ra >> 1) & 7 does not depend on t -> one lane address per k-step, t in the
    // instruction's immediate offset (32 rows x 128 B = 4096)
    unsigned offa[KS], offb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      offa[ks] = (unsigned)(fa0 * RB + swz(fa0, 2 * ks + kk) * 16);
      offb[ks] = (unsigned)(fb0 * RB + swz(fb0, 2 * ks + kk) * 16);
    }
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
#define SV_RD4(dst, addr)                                                                                      \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\tds_read_b128 %2, %4 offset:8192\n\t" \
                 "ds_read_b128 %3, %4 offset:12288"                                                            \
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3])                                  \
                 : "v"(addr))
#define SV_RD1(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#define SV_WAIT_FRAGS(N, A, B)                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                   \
                 : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]))
    {
      const unsigned aa = lds0 + a_off(ia) + offa[0], ab = lds0 + b_off(ib) + offb[0];
      SV_RD4(fa[0], aa);
      SV_RD4(fb[0], ab);
    }
    // DMA schedule: the stages of tile kt are free from that tile's barrier on, so the pieces of "slot 0" of the NEXT tile's
    // quota (its first A pieces) are issued right behind the barrier, one k-step earlier than their tile starts
    {
      const int ibn0 = (ib + BAHEAD >= NB) ? ib + BAHEAD - NB : ib + BAHEAD;
#pragma unroll
      for (int pz = 0; pz < DPS; ++pz) dma_piece(pz, 0, ia ^ 1, ibn0);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
      const int ibn = (ib + BAHEAD >= NB) ? ib + BAHEAD - NB : ib + BAHEAD;
      const int ib1 = (ib + 1 >= NB) ? 0 : ib + 1;
      const int ibn1 = (ib1 + BAHEAD >= NB) ? ib1 + BAHEAD - NB : ib1 + BAHEAD;
      const unsigned SAo = lds0 + a_off(ia), SBo = lds0 + b_off(ib);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        const bool last = ks + 1 == KS;
        bool fetched = true;
        unsigned aa, ab;
        if (!last) {
          aa = SAo + offa[ks + 1];
          ab = SBo + offb[ks + 1];
        } else {
          // A(kt+1) and B(kt+1) have landed (B(kt+2), the youngest JB DMA instructions, may still fly) and this wave's
          // reads of tile kt are complete (they were waited for at the top of the previous step ... and below)
          if (NB == 3 && kt + 2 < ntiles)
            wait_vm_lgkm0<JB>();
          else
            wait_vm_lgkm0<0>();
          __builtin_amdgcn_s_barrier();
          fetched = kt + 1 < ntiles;
          aa = lds0 + a_off(ia ^ 1) + offa[0];
          ab = lds0 + b_off(ib1) + offb[0];
        }
        // this step's fragments were requested under the previous MFMA group
        SV_WAIT_FRAGS(0, fa[cur], fb[cur]);
        __builtin_amdgcn_sched_barrier(0);
        // 16 MFMAs; the next step's 8 fragment reads and this step's DMA pieces are issued in the gaps between them (the
        // matrix pipe takes a new MFMA every 32 cycles: a lone wave that issues its loads in a block leaves it idle)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int mt = j >> 2, nt = j & 3;
          acc[mt][nt] = MFMA_F16(fa[cur][mt], fb[cur][nt], acc[mt][nt]);
          __builtin_amdgcn_sched_barrier(0);
          if (fetched) {
            if (j == 0) SV_RD1(fa[nxt][0], aa, 0);
            if (j == 1) SV_RD1(fa[nxt][1], aa, 4096);
            if (j == 2) SV_RD1(fa[nxt][2], aa, 8192);
            if (j == 3) SV_RD1(fa[nxt][3], aa, 12288);
            if (j == 4) SV_RD1(fb[nxt][0], ab, 0);
            if (j == 5) SV_RD1(fb[nxt][1], ab, 4096);
            if (j == 6) SV_RD1(fb[nxt][2], ab, 8192);
            if (j == 7) SV_RD1(fb[nxt][3], ab, 12288);
          }
          if (j >= 8 && j < 8 + DPS) {
            if (!last) dma_piece((ks + 1) * DPS + (j - 8), kt, ia ^ 1, ibn);
            else if (fetched) dma_piece(j - 8, kt + 1, ia, ibn1);   // A(kt+2) -> tile kt's A stage (free since the barrier)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      ia ^= 1;
      ib = ib1;
    }
#undef SV_RD1
#undef SV_RD4
#undef SV_WAIT_FRAGS
  } else if constexpr (MF == 1) {
    // the plain loop (one barrier per k-tile) on the 16 x 16 x 32 shape: two k-steps of 32 per k-tile
    static_assert(MF == 0 || PP != 0 || (KS == 4 && (JA + JB) % 2 == 0), "two k-steps of 32 per 64-deep k-tile");
    for (int kt = 0; kt < ntiles; ++kt) {
      const int ibn = (ib + BAHEAD >= NB) ? ib + BAHEAD - NB : ib + BAHEAD;
      const unsigned char* SA = lds + a_off(ia);
      const unsigned char* SB = lds + b_off(ib);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int cl = 4 * k2 + (l >> 4);
        f16x8 a[4], b[TN16];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ra = wm * 64 + 16 * t + (l & 15);
          a[t] = *reinterpret_cast<const f16x8*>(SA + ra * RB + swz(ra, cl) * 16);
        }
#pragma unroll
        for (int t = 0; t < TN16; ++t) {
          const int rb = wn * (16 * TN16) + 16 * t + (l & 15);
          b[t] = *reinterpret_cast<const f16x8*>(SB + rb * RB + swz(rb, cl) * 16);
        }
#pragma unroll
        for (int pz = 0; pz < (JA + JB) / 2; ++pz) dma_piece(k2 * ((JA + JB) / 2) + pz, kt, ia ^ 1, ibn);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN16; ++nt) acc16[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt], b[nt], acc16[mt][nt], 0, 0, 0);
      }
      if (NB == 3 && kt + 2 < ntiles)
        wait_vm_lgkm0<JB>();
      else
        wait_vm_lgkm0<0>();
      __builtin_amdgcn_s_barrier();
      ia ^= 1;
      ib = (ib + 1 >= NB) ? 0 : ib + 1;
      if constexpr (KBT > 0) {
        if ((kt + 1) % (KBT > 0 ? KBT : 1) == 0 || kt + 1 == ntiles) {   // close a k-block (wave-uniform)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN16; ++nt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                accb16[mt][nt][r] += acc16[mt][nt][r];
                acc16[mt][nt][r] = (kt + 1 == ntiles) ? accb16[mt][nt][r] : 0.f;   // last block: acc = the total
              }
        }
      }
    }
  } else {
  for (int kt = 0; kt < ntiles; ++kt) {
      const int ibn = (ib + BAHEAD >= NB) ? ib + BAHEAD - NB : ib + BAHEAD;
      const unsigned char* SA = lds + a_off(ia);
      const unsigned char* SB = lds + b_off(ib);
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int cl = 2 * ks + kk;  // logical 16-B chunk (8 consecutive k) of this lane
        f16x8 a[TM], b[TN];
  #pragma unroll
        for (int t = 0; t < TM; ++t) {
          const int ra = fa0 + 32 * t;
          a[t] = *reinterpret_cast<const f16x8*>(SA + ra * RB + swz(ra, cl) * 16);
        }
  #pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int rb = fb0 + 32 * t;
          b[t] = *reinterpret_cast<const f16x8*>(SB + rb * RB + swz(rb, cl) * 16);
        }
  #pragma unroll
        for (int pz = 0; pz < (JA + JB) / KS; ++pz) dma_piece(ks * ((JA + JB) / KS) + pz, kt, ia ^ 1, ibn);
  #pragma unroll
        for (int mt = 0; mt < TM; ++mt)
  #pragma unroll
          for (int nt = 0; nt < TN; ++nt) {
            if (ABL == 2) {  // ablation: no MFMA (operands stay live)
              acc[mt][nt][0] += (float)a[mt][0] + (float)b[nt][0];
            } else {
              acc[mt][nt] = MFMA_F16(a[mt], b[nt], acc[mt][nt]);
            }
          }
      }
      // this wave's pieces of A(kt+1) and B(kt+1) have landed (with NB = 3, B(kt+2) -- the youngest JB DMA
      // instructions -- may still fly: vmcnt retires in order) and its LDS reads are done; after the barrier that
      // holds for every wave, so the stages of tile kt may be overwritten
      if (NB == 3 && kt + 2 < ntiles)
        wait_vm_lgkm0<JB>();
      else
        wait_vm_lgkm0<0>();
      __builtin_amdgcn_s_barrier();
      ia ^= 1;
      ib = (ib + 1 >= NB) ? 0 : ib + 1;
      if (KBT > 0 && ((kt + 1) % (KBT > 0 ? KBT : 1) == 0 || kt + 1 == ntiles)) {   // close a k-block (wave-uniform)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              accb[KBT > 0 ? mt : 0][KBT > 0 ? nt : 0][r] += acc[mt][nt][r];
              acc[mt][nt][r] = (kt + 1 == ntiles) ? accb[KBT > 0 ? mt : 0][KBT > 0 ? nt : 0][r] : 0.f;   // last block: acc = the total
            }
      }
    }
  }

  SV_PHASE(1)  // main loop
  if ((ABL >= 1 && ABL <= 3) || ABL == 16) {  // ablation: no epilogue (accumulators stay live; 16: DMA skeleton only)
    float t = 0.f;
    if (ABL != 16) {
      if constexpr (MF == 0) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN; ++nt) t += acc[mt][nt][0] + acc[mt][nt][15];
      } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < TN16; ++nt) t += acc16[mt][nt][0] + acc16[mt][nt][3];
      }
    }
    if (t == 12345.678f) cand_cnt[0] = 1;
    if (!PERSIST) return;
    // persistent ablations: on to the workgroup's next tile (its head requested here, waited for at once)
    int tm_n = 0, tn_n = 0, sq_n = seq + pstep;
    while (sq_n < seq_total && !tile_of(sq_n, tm_n, tn_n)) sq_n += pstep;
    if (sq_n >= seq_total) return;
    if (BIAS) load_cn(tn_n);
    if (ABL != 3) issue_head(tm_n, tn_n, rev_of(sq_n));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seq = sq_n;
    tm = tm_n;
    tn = tn_n;
    rev = rev_of(seq);
    continue;
  }
  // The epilogue's row record {||q||^2, exact limit, screening bound} of this thread's row, in registers and BEFORE the next
  // tile's head is requested: the compiler waits for the two global loads behind it (issued before the main loop, long
  // since landed) with s_waitcnt vmcnt(0) at their first use -- placed behind the head's DMA instructions that wait sat out
  // the head's whole flight (~2 us per tile, with nothing else to do: the very latency the early request is meant to hide).
  //   v <= lim  <=>  acc - cn * half_scale >= (q2 - lim) * half_scale; the slack (relative 2^-17 of the largest possible
  //   magnitude) makes rounding of this shortcut only ever ADD candidates
  const float half_scale = 0.5f / inv_scale;
  const float rmax_hs = rn_max * half_scale;
  float st_q2 = 0.f, st_lim = -INFINITY, st_tau = INFINITY;
  if (EPI == 1 ? (m0 + wm * (32 * TM) + l < M) : (tid < BM && m0 + tid < M)) {
    st_q2 = pre_q2;
    st_lim = pre_thr + eps_mult * c_eps * sqrtf(st_q2 * rn_max);
    const float base = (st_q2 - st_lim) * half_scale;
    st_tau = base - 7.7e-6f * (fabsf(base) + rmax_hs);
  }
  asm volatile("" : "+v"(st_q2), "+v"(st_lim), "+v"(st_tau));   // (keeps the computation -- and its wait -- on this side of the DMA issue)
  // PERSIST: the next tile of this workgroup; its head is requested now and lands under the epilogue
  int tm_next = 0, tn_next = 0, seq_next = seq_total;
  if (PERSIST) {
    seq_next = seq + pstep;
    while (seq_next < seq_total && !tile_of(seq_next, tm_next, tn_next)) seq_next += pstep;
    if (seq_next < seq_total) {
      if (BIAS) load_cn(tn_next);
      issue_head(tm_next, tn_next, rev_of(seq_next));
    }
  }
  // ---- epilogue: keep d2~ <= thr + eps_mult * eps(q) -----------------------------------------------------------
  // The epilogue is VALU-issue bound (s_memtime phase timing, SEGVLAD_F16_CFG=90: every instruction of the sparse
  // per-survivor paths is paid by the whole wave), so it is organised around instruction count and everything
  // per-survivor happens on DENSE lanes:
  //  * per-row quantities {||q||^2, exact limit, screening bound} and the tile's column norms are staged once per
  //    workgroup in LDS (their global loads were issued before the main loop);
  //  * pass 1 screens every accumulator element with one fma + compare against the row's bound; the (rare) waves
  //    that see a hit compact the raw {accumulator, row | column} pairs by ballot/mbcnt into a wave-private LDS
  //    list -- three instructions per element on the common path, no atomics, no exact arithmetic;
  //  * pass 2a walks that list 64 records at a time (all lanes busy): exact d2~, exact limit test, LDS count per row;
  //  * ONE global atomic per row and workgroup reserves a slot range in that query's candidate list;
  //  * pass 2b walks the list again and stores the survivors at range + LDS ticket.
  // (History: one returning global atomic per survivor parked the wave ~1.5 us each; re-walking the 128 accumulator
  // elements under a per-lane bitmap cost 20 % of the kernel; ~100-instruction exact-test bodies per hit row 25 %.)
  // A wave whose 64 x 128 block holds more than LCAP hits (databases are spatially coherent: the 50 segments of a
  // query image against the rows of the same place) abandons its list and walks its accumulators directly
  // (per-hit LDS atomics: slow, but only for the handful of dense blocks).  The list order is arbitrary: every
  // consumer ranks or sorts it.
  // records per wave: a quarter of its elements at most; PERSIST keeps the whole scratch inside LDS slots 0 and 1
  constexpr int LCAP = PERSIST ? 896 : ((TM * TN * 256 < 2048) ? TM * TN * 256 : 2048);
  static_assert(!PERSIST || (size_t)BM * 24 + (size_t)BN * 4 + (size_t)NW * (LCAP + 1) * 8 <= 2 * (size_t)PA, "epilogue scratch");
  if constexpr (EPI == 1) {
    // ---- wave-private epilogue (EPI = 1; persistent + biased kernels) ------------------------------------------------
    // Nothing in it is shared between waves, so nothing in it waits for another wave: every wave stages the records of ITS
    // 64 query rows, screens its 64 x 128 block into its own LDS list, tests the list densely and takes ONE returning global
    // atomic per ROW WITH SURVIVORS (ranks inside a row from a wave-private LDS counter): the reservation's round trip
    // overlaps the wait for the next tile's head, which the wave has to sit out anyway, and the stores are left in flight.
    // A block that fills its list (spatially coherent databases: the 50 segments of a query image against the 200 rows of the
    // same place are thousands of hits in ONE 64 x 128 block) flushes it in place and goes on screening -- still one round
    // trip per flush.  (First version: one returning global atomic per survivor, 64 at a time -- fine at ~6 survivors per
    // block, 150 round trips for such a block.  A re-entrant screening pass -- flush, then jump back in -- turns the pass
    // into a loop whose invariants the compiler hoists and spills: 80 dwords.)
    static_assert(BIAS && TM == 2 && (TN == 4 || (MF == 1 && TN == 2)) && !ACC_A, "wave-private epilogue: 64-row wave tiles of the biased kernels");
    constexpr int LCAPE = 864;   // records per wave list
    constexpr int WSZ = (1024 + (LCAPE + 1) * 8 + 15) & ~15;
    static_assert((size_t)NW * WSZ <= (PERSIST ? 2 * (size_t)PA : 2 * (size_t)PA + (size_t)NB * PB), "epilogue scratch");
    // Every LDS access between the head's DMA instructions and the wait inside the first flush is RAW (inline asm): the
    // compiler cannot tell DMA'd LDS bytes from any other LDS address and puts s_waitcnt vmcnt(0) in front of every LDS
    // instruction it can see while a DMA is pending -- the wave would sit out the head's flight before its first epilogue
    // instruction (which is what EPI = 0 does).  One wave's LDS instructions execute in order, so a write followed by a read
    // of the same bytes needs no wait in between; reads are waited for with explicit lgkmcnt.
    // per wave: [64] {||q||^2, exact limit} | +512: [64] screening bounds | +768: [64] survivors per row, then the row's first
    // global slot | +1024: [LCAPE + 1] hits {accumulator bits -> d2~, row << 16 | rank in the row << 8 | column}
    const unsigned wb_a = (unsigned)(size_t)(lptr_t)(lds + (size_t)w * WSZ);
    {
      const float2 qr = make_float2(st_q2, st_lim);
      asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3 offset:512\n\tds_write_b32 %2, %4 offset:768" ::"v"(wb_a + 8u * (unsigned)l), "v"(qr),
                   "v"(wb_a + 4u * (unsigned)l), "v"(st_tau), "v"(0u)
                   : "memory");
    }
    SV_PHASE(2)
    // this lane's screening bounds: 32 x 32 tiles: element r of tile mt is row mt*32 + 8*(r>>2) + 4*kk + (r&3) -- 8 groups of 4
    // consecutive rows; 16 x 16 tiles: element j of tile mt is row mt*16 + 4*(lane>>4) + j -- 4 groups of 4
    float4 tqw[MF ? 1 : 2][4];
    if constexpr (MF == 0) {
      const unsigned ta = wb_a + 512u + 16u * (unsigned)kk;
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:32\n\tds_read_b128 %2, %8 offset:64\n\tds_read_b128 %3, %8 offset:96\n\t"
          "ds_read_b128 %4, %8 offset:128\n\tds_read_b128 %5, %8 offset:160\n\tds_read_b128 %6, %8 offset:192\n\tds_read_b128 %7, %8 offset:224\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(tqw[0][0]), "=&v"(tqw[0][1]), "=&v"(tqw[0][2]), "=&v"(tqw[0][3]), "=&v"(tqw[MF ? 0 : 1][0]), "=&v"(tqw[MF ? 0 : 1][1]),
            "=&v"(tqw[MF ? 0 : 1][2]), "=&v"(tqw[MF ? 0 : 1][3])
          : "v"(ta)
          : "memory");
    } else {
      const unsigned ta = wb_a + 512u + 16u * (unsigned)(l >> 4);
      asm volatile(
          "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(tqw[0][0]), "=&v"(tqw[0][1]), "=&v"(tqw[0][2]), "=&v"(tqw[0][3])
          : "v"(ta)
          : "memory");
    }
    uint32_t wave_cnt = 0;   // wave-uniform
    const uint32_t colbase = (uint32_t)(wn * (32 * TN) + (MF ? (l & 15) : i)), rowsel = (uint32_t)(4 * (MF ? (l >> 4) : kk));
    const int64_t rowbase = m0 + wm * 64;
    // dense exact test of the list, one returning global atomic per row with survivors, the stores left in flight
    auto flush = [&]() {
      const uint32_t n_w = wave_cnt;
      for (uint32_t tb = 0; tb < n_w; tb += 64u) {   // exact test; a survivor draws its rank inside its row
        const uint32_t t = tb + (uint32_t)l;
        if (t < n_w) {
          uint2 rec;
          float2 rr;
          asm volatile("ds_read_b64 %0, %1 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(rec) : "v"(wb_a + 8u * t) : "memory");
          const unsigned lrow = rec.y >> 16;
          asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(rr) : "v"(wb_a + 8u * lrow) : "memory");
          const float v = __fmaf_rn(-2.f, __uint_as_float(rec.x) * inv_scale, rr.x);
          if (v <= rr.y && v < INFINITY) {   // +inf: padding columns beyond N (admitted by the screen when thr = +inf)
            uint32_t rank;
            asm volatile("ds_add_rtn_u32 %0, %1, %2 offset:768\n\ts_waitcnt lgkmcnt(0)" : "=v"(rank) : "v"(wb_a + 4u * lrow), "v"(1u) : "memory");
            rec.x = __float_as_uint(v);
            rec.y |= rank << 8;              // (< 128 survivors per row and block; the column keeps bits 0-7)
          } else {
            rec.y = 0xffffffffu;             // screened in by the slack only
          }
          asm volatile("ds_write_b64 %0, %1 offset:1024" ::"v"(wb_a + 8u * t), "v"(rec) : "memory");
        }
      }
      uint32_t c;
      asm volatile("ds_read_b32 %0, %1 offset:768\n\ts_waitcnt lgkmcnt(0)" : "=v"(c) : "v"(wb_a + 4u * (unsigned)l) : "memory");
      uint32_t base = 0u;
      if (c) base = atomicAdd(&cand_cnt[rowbase + l], c);   // (rows >= M screen with +inf: no survivors)
      // the reservations -- and, the first time round, this wave's DMA pieces of the next tile's head (requested before the
      // screening pass; loads retire in order): the barrier at the top of the tile loop makes that true for every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("ds_write_b32 %0, %1 offset:768" ::"v"(wb_a + 4u * (unsigned)l), "v"(base) : "memory");
      for (uint32_t tb = 0; tb < n_w; tb += 64u) {
        const uint32_t t = tb + (uint32_t)l;
        if (t < n_w) {
          uint2 rec;
          asm volatile("ds_read_b64 %0, %1 offset:1024\n\ts_waitcnt lgkmcnt(0)" : "=v"(rec) : "v"(wb_a + 8u * t) : "memory");
          if (rec.y != 0xffffffffu) {
            const unsigned lrow = rec.y >> 16;
            uint32_t b0;
            asm volatile("ds_read_b32 %0, %1 offset:768\n\ts_waitcnt lgkmcnt(0)" : "=v"(b0) : "v"(wb_a + 4u * lrow) : "memory");
            const uint32_t slot = b0 + ((rec.y >> 8) & 0xffu);
            if (slot < (uint32_t)cap) {
              const int64_t row = rowbase + (int64_t)lrow;
              cand_d2[row * cap + slot] = __uint_as_float(rec.x);
              cand_id[row * cap + slot] = (uint32_t)grow(n0 + (int64_t)(rec.y & 0xffu));
            }
          }
        }
      }
      asm volatile("ds_write_b32 %0, %1 offset:768" ::"v"(wb_a + 4u * (unsigned)l), "v"(0u) : "memory");
      wave_cnt = 0u;
    };
    if constexpr (MF == 1) {
      // 16 screening steps of 8 elements (one row of the wave tile across its 8 column tiles)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 tq4 = tqw[0][mt];
          const float tau = j == 0 ? tq4.x : j == 1 ? tq4.y : j == 2 ? tq4.z : tq4.w;
          float best;
          asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(acc16[mt][0][j]), "v"(acc16[mt][1][j]), "v"(acc16[mt][2][j]));
          if constexpr (TN16 == 8) {
            asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(best), "v"(acc16[mt][3][j]), "v"(acc16[mt][TN16 == 8 ? 4 : 0][j]));
            asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(best), "v"(acc16[mt][TN16 == 8 ? 5 : 0][j]), "v"(acc16[mt][TN16 == 8 ? 6 : 0][j]));
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(acc16[mt][TN16 == 8 ? 7 : 0][j]));
          } else {
            asm volatile("v_max_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(acc16[mt][3][j]));
          }
          if (__builtin_amdgcn_ballot_w64(best >= tau) != 0ull) {
            const uint32_t rc = (((uint32_t)(mt * 16 + j) + rowsel) << 16) | colbase;
#pragma unroll
            for (int nt = 0; nt < TN16; ++nt) {
              const bool hit = acc16[mt][nt][j] >= tau;
              const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
              if (mk != 0ull) {
                const uint32_t pos = wave_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                if (hit) {
                  const uint2 rec = make_uint2(__float_as_uint(acc16[mt][nt][j]), rc + (uint32_t)(nt * 16));
                  asm volatile("ds_write_b64 %0, %1 offset:1024" ::"v"(wb_a + 8u * pos), "v"(rec) : "memory");
                }
                wave_cnt += (uint32_t)__popcll(mk);
              }
            }
            if (wave_cnt > (uint32_t)(LCAPE - 64 * TN16)) flush();   // (a step adds up to TN16 x 64 records)
          }
        }
    } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 tq4 = tqw[mt][r >> 2];
        const float tau = (r & 3) == 0 ? tq4.x : (r & 3) == 1 ? tq4.y : (r & 3) == 2 ? tq4.z : tq4.w;
        float best;
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(best) : "v"(acc[mt][0][r]), "v"(acc[mt][1][r]), "v"(acc[mt][2][r]));
        asm volatile("v_max_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(acc[mt][3][r]));
        if (__builtin_amdgcn_ballot_w64(best >= tau) != 0ull) {
          const uint32_t rc = (((uint32_t)(mt * 32 + (r & 3) + 8 * (r >> 2)) + rowsel) << 16) | colbase;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            const bool hit = acc[mt][nt][r] >= tau;
            const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
            if (mk != 0ull) {
              const uint32_t pos = wave_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
              if (hit) {
                const uint2 rec = make_uint2(__float_as_uint(acc[mt][nt][r]), rc + (uint32_t)(nt * 32));
                asm volatile("ds_write_b64 %0, %1 offset:1024" ::"v"(wb_a + 8u * pos), "v"(rec) : "memory");
              }
              wave_cnt += (uint32_t)__popcll(mk);
            }
          }
          // the next step could overflow the list (spatially coherent databases): flush it here and go on
          if (wave_cnt > (uint32_t)(LCAPE - 256)) flush();
        }
      }
    }
    SV_PHASE(3)
    flush();
    SV_PHASE(4)
  }
  if constexpr (EPI == 0) {
  float4* rrec = reinterpret_cast<float4*>(lds);                         // [BM] {||q||^2, exact limit, screening bound, -}
  uint32_t* rowcnt = reinterpret_cast<uint32_t*>(rrec + BM);             // [BM] survivors per row -> next free global slot
  float* cnl = reinterpret_cast<float*>(rowcnt + BM);                    // [BN] column norms
  float* taul = cnl + BN;                                                // [BM] screening bounds, contiguous (16-B reads)
  uint2* wlist = reinterpret_cast<uint2*>(taul + BM) + (size_t)w * (LCAP + 1);  // this wave's hit list (+1 dump slot)
  if (tid < BM) {
    const int j = tid;
    rrec[j] = make_float4(st_q2, st_lim, st_tau, 0.f);
    rowcnt[j] = 0u;
    taul[j] = st_tau;
  }
  float cnh[TN];
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    cnh[nt] = BIAS ? 0.f : cn[nt] * half_scale;                  // BIAS: already inside the accumulators
    if (!BIAS && wm == 0) cnl[wn * (32 * TN) + nt * 32 + i] = cn[nt];   // both half-waves hold the same value
  }
  __syncthreads();
  SV_PHASE(2)  // row records staged
  uint32_t wave_cnt = 0;  // wave-uniform: only updated under wave-uniform control flow
  uint32_t dbg_bodies = 0;
  // this lane's 16 * TM screening bounds, fetched up front with 16-B reads (accumulator element r of tile mt belongs to
  // row mt*32 + 8*(r>>2) + 4*kk + (r&3): four consecutive rows per (mt, r>>2)) -- a per-iteration LDS read put ~100
  // cycles of latency into each of the 32 screening steps (two waves per SIMD cannot hide it)
  // (TM > 2 -- the 128 x 128 wave tiles -- fetches them per 32-row tile instead: 64 registers would not fit)
  constexpr bool TQ_UPFRONT = TM <= 2;
  float4 tq[TQ_UPFRONT ? TM : 1][4];
  if (TQ_UPFRONT) {
#pragma unroll
    for (int mt = 0; mt < (TQ_UPFRONT ? TM : 1); ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        tq[mt][g] = *reinterpret_cast<const float4*>(taul + wm * (32 * TM) + mt * 32 + 8 * g + 4 * kk);
  }
#pragma unroll
  for (int mt = 0; mt < TM; ++mt) {
    if (!TQ_UPFRONT) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        tq[0][g] = *reinterpret_cast<const float4*>(taul + wm * (32 * TM) + mt * 32 + 8 * g + 4 * kk);
      __builtin_amdgcn_sched_barrier(0);   // one tile's bounds at a time
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t lrow16 = (uint32_t)(wm * (32 * TM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) << 16;
      const float4 tq4 = tq[TQ_UPFRONT ? mt : 0][r >> 2];
      const float tau = (r & 3) == 0 ? tq4.x : (r & 3) == 1 ? tq4.y : (r & 3) == 2 ? tq4.z : tq4.w;
      float av[TN], dd[TN];
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        av[nt] = acc_elem<ACC_A>(acc[mt][nt], r);
        dd[nt] = BIAS ? av[nt] : av[nt] - cnh[nt];
      }
      float best = dd[0];
#pragma unroll
      for (int nt = 1; nt < TN; ++nt) best = fmaxf(best, dd[nt]);
      if (ABL < 12 && __builtin_amdgcn_ballot_w64(best >= tau) != 0ull) {   // ABL >= 12: timing ablations (wrong results)
        if (ABL == 9 || ABL == 12) ++dbg_bodies;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const bool hit = dd[nt] >= tau;
          const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
          if (mk != 0ull) {
            uint32_t pos = wave_cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
            pos = pos < (uint32_t)LCAP ? pos : (uint32_t)LCAP;   // beyond the list: the dump slot (the block turns dense below)
            if (hit)
              wlist[pos] = make_uint2(__float_as_uint(av[nt]), lrow16 | (uint32_t)(wn * (32 * TN) + nt * 32 + i));
            wave_cnt += (uint32_t)__popcll(mk);
          }
        }
      }
    }
  }
  if ((ABL == 9 || ABL == 12) && tid == 0) atomicAdd(&sv_f16_phase_cycles[6], (unsigned long long)dbg_bodies);
  SV_PHASE(3)  // pass 1
  // pass 2a: exact test of this wave's hits, dense (the list is wave-private: no barrier needed before reading it)
  const uint32_t n_w = wave_cnt <= (uint32_t)LCAP ? wave_cnt : 0u;   // a dense block abandons its (truncated) list
  for (uint32_t t = (uint32_t)l; t < n_w; t += 64u) {
    uint2 rec = wlist[t];
    const int lrow = (int)(rec.y >> 16);
    const float4 rr = rrec[lrow];
    const float v = BIAS ? __fmaf_rn(-2.f, __uint_as_float(rec.x) * inv_scale, rr.x)
                         : sv_d2(rr.x, cnl[rec.y & 0xffffu], __uint_as_float(rec.x) * inv_scale);
    if (v <= rr.y && v < INFINITY) {   // +inf: padding columns beyond N (admitted by the screen when thr = +inf)
      atomicAdd(&rowcnt[lrow], 1u);
      rec.x = __float_as_uint(v);
    } else {
      rec.y = 0xffffffffu;   // screened in by the slack only
    }
    wlist[t] = rec;
  }
  const bool dense = wave_cnt > (uint32_t)LCAP;   // wave-uniform
  if (dense) {
    // (the scale is laundered through an empty asm so that the compiler does not keep 128 values of pass 1 or of this
    //  walk alive for the next one: that spilled the common path)
    float isc = inv_scale;
    asm volatile("" : "+v"(isc));
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = wm * (32 * TM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const float4 rr = rrec[lrow];
          const float v = BIAS ? __fmaf_rn(-2.f, acc_elem<ACC_A>(acc[mt][nt], r) * isc, rr.x)
                               : sv_d2(rr.x, cn[nt], acc_elem<ACC_A>(acc[mt][nt], r) * isc);
          if (v <= rr.y && v < INFINITY) atomicAdd(&rowcnt[lrow], 1u);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the 32 row-record loads from being hoisted (register pressure)
      }
  }
  // PERSIST: this wave's DMA pieces of the next tile's head have landed by now (requested before pass 1; loads retire
  // in order); the barrier below makes that true for every wave, so the next tile starts without a memory wait
  if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  SV_PHASE(4)  // pass 2a
  if (tid < BM) {
    const uint32_t c = rowcnt[tid];
    rowcnt[tid] = (c > 0u) ? atomicAdd(&cand_cnt[m0 + tid], c) : 0u;  // rows >= M never count
  }
  __syncthreads();
  for (uint32_t t = (uint32_t)l; t < n_w; t += 64u) {
    const uint2 rec = wlist[t];
    if (rec.y != 0xffffffffu) {
      const int lrow = (int)(rec.y >> 16);
      const uint32_t slot = atomicAdd(&rowcnt[lrow], 1u);
      if (slot < (uint32_t)cap) {
        const int64_t row = m0 + lrow;
        cand_d2[row * cap + slot] = __uint_as_float(rec.x);
        cand_id[row * cap + slot] = (uint32_t)grow(n0 + (int64_t)(rec.y & 0xffffu));
      }
    }
  }
  if (dense) {
    float isc = inv_scale;
    asm volatile("" : "+v"(isc));
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lrow = wm * (32 * TM) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          const float4 rr = rrec[lrow];
          {
            const float v = BIAS ? __fmaf_rn(-2.f, acc_elem<ACC_A>(acc[mt][nt], r) * isc, rr.x)
                                 : sv_d2(rr.x, cn[nt], acc_elem<ACC_A>(acc[mt][nt], r) * isc);
            if (v <= rr.y && v < INFINITY) {
              const uint32_t slot = atomicAdd(&rowcnt[lrow], 1u);
              if (slot < (uint32_t)cap) {
                const int64_t row = m0 + lrow;
                cand_d2[row * cap + slot] = v;
                cand_id[row * cap + slot] = (uint32_t)grow(n0 + wn * (32 * TN) + nt * 32 + i);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  }   // EPI == 0
  SV_PHASE(5)  // reservation + pass 2b
  if (!PERSIST || seq_next >= seq_total) break;
  seq = seq_next;
  tm = tm_next;
  tn = tn_next;
  rev = rev_of(seq);
  }   // tile loop
}

template <int BM, int BN, int WM, int WN, int HBK, int NB, int ABL = 0, bool PERSIST = false, int POL = 0, int PP = 0, int KBT = 0,
          bool BIAS = false, int EPI = 0, int MF = 0, bool BUF = false, int DSPLIT = 0, int SKIP = 0, int KFL = 0>
static int launch_f16_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Rh, int M, int n_sample, int d, int b_stride,
                             float inv_scale, const float* qn, const float* rn, const float* thr, int64_t thr_ld,
                             float eps_mult, float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2, uint32_t* cand_id,
                             int cap) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (n_sample + BN - 1) / BN;
  int64_t tiles = (int64_t)tiles_m * tiles_n;
  // tile-block height of the XCD-aware order (0 = plain tm-fastest order).  Every database tile is fetched once per
  // block ROW of query tiles: with >= 16 query tiles (one pass over 10 000 queries has 40) blocks of 8 x 4 halve that
  // re-read against 4 x 8 (10 -> 5 fetches of the fp16 plane per launch) at the same speed; 16 x 2 is 14 % slower (the 16
  // query tiles no longer stay in the XCD's L2).
  int gm = ctx->opt.f16_gm >= 0 ? ctx->opt.f16_gm : (tiles_m >= 16 ? 8 : 4);
  if (PERSIST && gm <= 0) gm = 4;
  int seq_total = 0;
  // tile walk of the persistent kernel (see the kernel): bit 0 = query-block-resident order, bit 1 = serpentine k
  // default 3: measured on 10 000 x 1 M x 1024 (rocprofv3 FETCH_SIZE, calibrated; tools/pmc_walk.sh): L2 fills of the full-level
  // launch 42.7 GB (walk 0) / 45.1 GB (2: serpentine alone) / 26.3 GB (3), at the same speed (18.39 / 18.34 ms per search's filter
  // launches, interleaved A/B)
  // deep rows (KFL): serpentine k alone.  Measured 10 000 x 46 875 x 98 304 (config2's filter launches per step, tools/walk_sweep_cfg2.sh):
  // walk 2 -> 93.6 ms, 0 -> 94.4-94.9, 3 -> 100.9-101.7, 1 -> 103.6, 4 -> 102.9, 5 / 6 / 7 -> 110.9 / 111.6 / 116.6; L2 fills 260-290 GB per
  // launch either way (a query block is 8 x 50 MB there: "resident" means nothing at that depth, and the resident order makes the 32
  // workgroups of an XCD meet a NEW database block at every step)
  int walk = PERSIST ? (ctx->opt.f16_walk >= 0 ? (ctx->opt.f16_walk & 7) : (KFL > 0 ? 2 : 3)) : 0;
  const int pwgs = (ctx->opt.f16_persist_wgs >= 1 && ctx->opt.f16_persist_wgs <= 32) ? ctx->opt.f16_persist_wgs : 32;
  if (PERSIST) walk |= pwgs << 8;
  if (gm > 0) {
    gm = gm >= 32 ? 32 : gm >= 16 ? 16 : gm >= 8 ? 8 : gm >= 4 ? 4 : gm >= 2 ? 2 : 1;
    while (gm > 1 && gm / 2 >= tiles_m) gm >>= 1;
    const int gn = 32 / gm;
    int64_t st = (int64_t)((tiles_m + gm - 1) / gm) * ((tiles_n + gn - 1) / gn);
    if (walk & 1) {   // per XCD: query blocks x its share of the database blocks, + its share of the left-over pairs (see the kernel)
      const int64_t smc = (tiles_m + gm - 1) / gm, snc = (tiles_n + gn - 1) / gn, nbf = snc / 8, rem = snc - 8 * nbf;
      st = (smc * nbf + (smc * rem + 7) / 8) * 8;
    }
    tiles = (st + 7) / 8 * 8 * 32;
    if (tiles / 8 > 0x7fffffffLL) return ctx->fail(SEGVLAD_ERR_LIMIT, "f16 filter: too many tiles");
    seq_total = (int)(tiles / 8);
    if (PERSIST) tiles = 8 * pwgs;   // 8 XCDs x 32 (option f16_persist_wgs) resident workgroups, each walks its XCD's sequence
  }
  if (tiles > 0x7fffffffLL) return ctx->fail(SEGVLAD_ERR_LIMIT, "f16 filter: too many tiles");
  size_t lds = 2 * (size_t)BM * HBK * 2 + (size_t)NB * BN * HBK * 2;  // two A stages + NB B stages
  if (!PERSIST && EPI == 0) {  // epilogue: row records + per-row counters + one survivor list per wave
    constexpr int TMl = BM / (32 * WM), TNl = BN / (32 * WN);
    constexpr int LCAPl = (TMl * TNl * 256 < 2048) ? TMl * TNl * 256 : 2048;
    const size_t elds = (size_t)BM * 24 + (size_t)BN * 4 + (size_t)WM * WN * (LCAPl + 1) * 8;
    if (lds < elds) lds = elds;
  }
  auto kern = knn_f16_filter_kernel<BM, BN, WM, WN, HBK, NB, ABL, PERSIST, POL, PP, KBT, BIAS, EPI, MF, BUF, DSPLIT, SKIP, KFL>;
  float* kscr = nullptr;
  if (KFL > 0) {   // one slice of 128 x 64 fp32 per resident wave, all zero between tiles (the kernel leaves it so); zeroed here as well
    const size_t bytes = (size_t)tiles * (WM * WN) * 128 * 64 * 4;
    SV_HIP(ctx->s_kflush.reserve(bytes));
    SV_HIP(hipMemsetAsync(ctx->s_kflush.p, 0, bytes, ctx->stream));
    kscr = ctx->s_kflush.as<float>();
  }
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * WM * WN), lds, ctx->stream, Qh, Rh, M, n_sample, d, b_stride, tiles_m,
                     gm, seq_total, walk, inv_scale, qn, rn, thr, thr_ld, eps_mult, c_eps, rn_max, cand_cnt, cand_d2, cand_id, cap,
                     ctx->f16_scale_dev, kscr);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// Can the LAST level of a batch search run over the complement of the stride-16 sample (skip = 16)?  Only the two default batch
// kernels have that form; the answer mirrors sv_launch_f16_filter's choice for (M, n_rows, d) under the context's options.
// deep rows: does this launch take the persistent 256 x 256 kernel with flushed blocks (KFL)?
static bool deep_flush_ok(const segvlad_ctx* ctx, int M, int64_t n_rows) {
  const SvOptions& o = ctx->opt;
  return (o.f16_deep_cfg < 0 || o.f16_deep_cfg == 5) && ctx->f16_bias_ok && o.f16_mf != 0 && o.f16_epi != 0 && M > 128 &&
         (int64_t)((M + 255) / 256) * ((n_rows + 255) / 256) >= 1024;
}

bool sv_f16_filter_skip_ok(const segvlad_ctx* ctx, int M, int64_t n_rows, int d) {
  const SvOptions& o = ctx->opt;
  if (!o.level_carry || M <= 128 || n_rows <= 0 || !ctx->f16_bias_ok || o.f16_mf == 0 || o.f16_epi == 0) return false;
  if (sv_f16_kblock(o, d)) return o.f16_deep_cfg < 0 || o.f16_deep_cfg == 4 || o.f16_deep_cfg == 5;
  if (o.f16_cfg >= 0 && o.f16_cfg != 250) return false;
  if (o.f16_pp == 0 || o.f16_dsplit != 0 || o.f16_buf == 1) return false;   // (the A/B variants of the batch kernel)
  return (int64_t)((M + 255) / 256) * ((n_rows + 255) / 256) >= 1024;
}

int sv_launch_f16_filter(segvlad_ctx* ctx, const uint16_t* Qh, const uint16_t* Rh, int M, int n_sample, int d, int b_stride,
                         float inv_scale, const float* qn, const float* rn, const float* thr, int64_t thr_ld, float eps_mult,
                         float c_eps, float rn_max, uint32_t* cand_cnt, float* cand_d2, uint32_t* cand_id, int cap, int skip) {
  if (M <= 0 || n_sample <= 0) return SEGVLAD_OK;
#define SV_F16_ARGS ctx, Qh, Rh, M, n_sample, d, b_stride, inv_scale, qn, rn, thr, thr_ld, eps_mult, c_eps, rn_max, cand_cnt, cand_d2, cand_id, cap
  if (skip) {
    // operand row j = database row j + j / 15 + 1 (n_sample = the number of rows that are not multiples of 16): see the kernel's SKIP
    if (skip != 16 || b_stride != 1 || !sv_f16_filter_skip_ok(ctx, M, n_sample, d))
      return ctx->fail(SEGVLAD_ERR_STATE, "f16 filter: no complement-of-sample form for this configuration");
    if (sv_f16_kblock(ctx->opt, d)) {
      if (deep_flush_ok(ctx, M, n_sample))
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, 0, 16, SV_F16_KFLUSH>(SV_F16_ARGS);
      if (ctx->opt.f16_buf != 0 && (int64_t)256 * d * 2 * 2 + 4096 < (int64_t)0xffffffffLL)
        return launch_f16_filter<256, 128, 4, 2, 64, 3, 0, false, 0, 0, 16, true, 1, 1, true, 0, 16>(SV_F16_ARGS);
      return launch_f16_filter<256, 128, 4, 2, 64, 3, 0, false, 0, 0, 16, true, 1, 1, false, 0, 16>(SV_F16_ARGS);
    }
    return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, 0, 16>(SV_F16_ARGS);
  }
  // tile configuration: option "f16_cfg" (default chosen from measurements)
  // r02 measurements (10 000 x 1 M x 1024, random unit vectors, filter launches only): 0 -> 22.5 ms, 50 (ping-pong) -> 21.7,
  // 200 (persistent) -> 21.7, 250 (persistent + ping-pong) -> 21.5; HBK = 32 variants (4, 1) 24.1 / 25.3
  // 4 waves of 128 x 128 (a third fewer LDS fragment reads per MFMA; accumulators in AGPRs, epilogue through acc_elem):
  // 5 (compiler-scheduled loop) -> 21.8 ms, 55 (software-pipelined: reads / DMA issued between the MFMAs, one barrier per
  // k-tile) -> 21.9, 255 (55 + persistent) -> 22.1, against 20.9-21.1 for 250 in the same sessions: with one wave per SIMD
  // nothing covers the per-tile barrier and the epilogue, and the kernel sits at the same power-limited clock either way.
  // streaming regime (M <= 128, e.g. one 50-segment query image per pass over 1 M rows): 2 -> 0.41 ms, 3 -> 0.51 ms
  // deep rows (raw K*D descriptors, d >= 4096): blocked accumulation (configuration 300), which is what keeps the
  // filter's error margin -- and with it the refine band -- as tight as at d = 1024 (sv_f16_c_eps)
  const int c = sv_f16_kblock(ctx->opt, d) ? 300 : ctx->opt.f16_cfg >= 0 ? ctx->opt.f16_cfg : (M > 128 ? 250 : M > 64 ? 63 : 62);
  // BUF kernels: every piece offset of a 256-row tile (row pitch d * b_stride fp16 for the database side) in 32 bits
  const bool buf_ok = ctx->opt.f16_buf != 0 && (int64_t)256 * d * b_stride * 2 + 4096 < (int64_t)0xffffffffLL;
  switch (c) {
    case 300:   // blocked accumulation (deep rows): two accumulator sets, a k-block of SV_F16_KBLOCK = 16 k-tiles of 64
      static_assert(SV_F16_KBLOCK == 16 * 64, "k-block = KBT x HBK");
      // round 4: 8 waves of 64 x 64 on a 256 x 128 tile, 16 x 16 x 32 MFMA, the plain loop (one barrier per k-tile), biased
      // accumulators (the bias sits in the block-sum set), wave-private epilogue -- 1.5 x the operand bytes per flop of the
      // 256 x 256 kernel instead of 2 x, two waves per SIMD instead of one.  Measured, 10 000 x 50 000 x 98 304 (filter launches):
      // rounds 2-3's 4 waves of 64 x 64 on 128 x 128 tiles with the 32 x 32 x 16 shape 147 ms (f16_deep_cfg = 0; also the
      // fallback when the norms are too unbalanced for the bias), this kernel 126 ms; the same with the ping-pong loop 165 ms
      // (1: half the MFMAs per phase of the 64 x 128 wave tiles behind the same barriers), 128 x 128 tiles with the 16 x 16 x 32
      // shape 234 ms ping-pong / 198 ms plain (2 / 3: one wave per SIMD)
      // round 6b: launches that fill the persistent 256 x 256 ping-pong kernel (the batch kernel of the 1024-d searches) take it, with the
      // accumulation blocked by FLUSHING: every 64 k-tiles the accumulators are added into a global scratch slice and cleared (KFL) --
      // the second register set that kept the 256-row tiles out of reach is gone; measured 10 000 x 46 875 x 98 304: see DESIGN.md
      if (deep_flush_ok(ctx, M, n_sample))
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, 0, 0, SV_F16_KFLUSH>(SV_F16_ARGS);
      if (ctx->f16_bias_ok && ctx->opt.f16_mf != 0 && ctx->opt.f16_epi != 0 && M > 128) {
        if (ctx->opt.f16_deep_cfg < 0 || ctx->opt.f16_deep_cfg == 4 || ctx->opt.f16_deep_cfg == 5) {   // the default: 780 vs 667 TF algorithmic at 10 000 x 50 000 x 98 304
          if (buf_ok) return launch_f16_filter<256, 128, 4, 2, 64, 3, 0, false, 0, 0, 16, true, 1, 1, true>(SV_F16_ARGS);
          return launch_f16_filter<256, 128, 4, 2, 64, 3, 0, false, 0, 0, 16, true, 1, 1>(SV_F16_ARGS);
        }
#ifdef SEGVLAD_ABLATIONS   // measured and not kept (development build; the shipped library holds the default only)
        if (ctx->opt.f16_deep_cfg == 1) return launch_f16_filter<256, 128, 4, 2, 64, 3, 0, false, 0, 2, 16, true, 1, 1>(SV_F16_ARGS);
        if (ctx->opt.f16_deep_cfg == 2) return launch_f16_filter<128, 128, 2, 2, 64, 3, 0, false, 0, 2, 16, true, 1, 1>(SV_F16_ARGS);
        if (ctx->opt.f16_deep_cfg == 3) return launch_f16_filter<128, 128, 2, 2, 64, 3, 0, false, 0, 0, 16, true, 1, 1>(SV_F16_ARGS);   // plain loop
#endif
      }
      return launch_f16_filter<128, 128, 2, 2, 64, 3, 0, false, 0, 0, 16>(SV_F16_ARGS);
#ifdef SEGVLAD_ABLATIONS   // development builds only: the configurations measured on the way (rounds 1-4, DESIGN.md 4 / 7.1: correct
                           // results, not kept), timing ablations (WRONG results) and phase timing
    case 0: return launch_f16_filter<256, 256, 4, 2, 64, 3>(SV_F16_ARGS);  // 160 KiB LDS, 1 workgroup / CU
    case 200:   // persistent workgroups that request the next tile's head before their epilogue
      if ((int64_t)((M + 255) / 256) * ((n_sample + 255) / 256) >= 1024)
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true>(SV_F16_ARGS);
      return launch_f16_filter<256, 256, 4, 2, 64, 3>(SV_F16_ARGS);
    case 10: return launch_f16_filter<256, 256, 4, 2, 64, 3, 1>(SV_F16_ARGS);  // ablations of config 0 (WRONG results)
    case 20: return launch_f16_filter<256, 256, 4, 2, 64, 3, 2>(SV_F16_ARGS);
    case 30: return launch_f16_filter<256, 256, 4, 2, 64, 3, 3>(SV_F16_ARGS);
    case 130: return launch_f16_filter<256, 256, 4, 2, 64, 3, 13>(SV_F16_ARGS);   // DMA only, B pieces only
    case 140: return launch_f16_filter<256, 256, 4, 2, 64, 3, 14>(SV_F16_ARGS);   // DMA only, A pieces only
    case 121: return launch_f16_filter<256, 256, 4, 2, 64, 3, 15>(SV_F16_ARGS);   // DMA only (no phase timing)
    case 160: return launch_f16_filter<256, 256, 4, 2, 64, 3, 16>(SV_F16_ARGS);   // DMA only, no epilogue
    case 94: ctx->f16_bias_ok = true; return launch_f16_filter<256, 256, 4, 2, 64, 3, 1, true, 0, 2, 0, true, 1, 1>(SV_F16_ARGS);   // default kernel, no epilogue
    case 95: ctx->f16_bias_ok = true; return launch_f16_filter<256, 256, 4, 2, 64, 3, 3, true, 0, 2, 0, true, 1, 1>(SV_F16_ARGS);   // + no DMA in the loop
    case 91:
    case 92:
    case 93: {  // phase timing of the default batch kernel (persistent + ping-pong + bias), epilogue 0 / 1 / 1 + 16x16x32 MFMA
      unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c8[8];
      SV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sv_f16_phase_cycles), z, sizeof(z)));
      ctx->f16_bias_ok = true;
      const int rc = c == 91   ? launch_f16_filter<256, 256, 4, 2, 64, 3, 9, true, 0, 2, 0, true, 0>(SV_F16_ARGS)
                     : c == 92 ? launch_f16_filter<256, 256, 4, 2, 64, 3, 9, true, 0, 2, 0, true, 1>(SV_F16_ARGS)
                               : launch_f16_filter<256, 256, 4, 2, 64, 3, 9, true, 0, 2, 0, true, 1, 1>(SV_F16_ARGS);
      SV_HIP(hipStreamSynchronize(ctx->stream));
      SV_HIP(hipMemcpyFromSymbol(c8, HIP_SYMBOL(sv_f16_phase_cycles), sizeof(c8)));
      double tot = 0;
      for (int k = 0; k < 6; ++k) tot += (double)c8[k];
      fprintf(stderr, "[f16 filter phases, cfg %d] M=%d n=%d: head-wait %.1f%% main %.1f%% stage %.1f%% pass1 %.1f%% pass2a %.1f%% rest %.1f%% (%.3g cycles/WG-sum)\n",
              c, M, n_sample, 100 * c8[0] / tot, 100 * c8[1] / tot, 100 * c8[2] / tot, 100 * c8[3] / tot, 100 * c8[4] / tot,
              100 * c8[5] / tot, tot);
      return rc;
    }
    case 90:
    case 120: {  // phase timing of config 0 (debug: synchronises and prints; 110 / 120 also ablate pass 1)
      unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c8[8];
      SV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sv_f16_phase_cycles), z, sizeof(z)));
      const int rc = c == 90 ? launch_f16_filter<256, 256, 4, 2, 64, 3, 9>(SV_F16_ARGS)
                             : launch_f16_filter<256, 256, 4, 2, 64, 3, 12>(SV_F16_ARGS);
      SV_HIP(hipStreamSynchronize(ctx->stream));
      SV_HIP(hipMemcpyFromSymbol(c8, HIP_SYMBOL(sv_f16_phase_cycles), sizeof(c8)));
      double tot = 0;
      for (int k = 0; k < 6; ++k) tot += (double)c8[k];
      fprintf(stderr, "[f16 filter phases] M=%d n=%d: prologue %.1f%% main %.1f%% stage %.1f%% pass1 %.1f%% pass2a %.1f%% reserve+pass2b %.1f%% (%.3g cycles/WG-sum)\n",
              M, n_sample, 100 * c8[0] / tot, 100 * c8[1] / tot, 100 * c8[2] / tot, 100 * c8[3] / tot, 100 * c8[4] / tot,
              100 * c8[5] / tot, tot);
      fprintf(stderr, "[f16 filter phases]   wave 0: %.2f hit bodies per tile\n",
              (double)c8[6] / ((double)((M + 255) / 256) * ((n_sample + 255) / 256)));
      return rc;
    }
    case 50: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 2>(SV_F16_ARGS);  // ping-pong, 2 phases per k-tile
    case 51: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 4>(SV_F16_ARGS);  // ping-pong, 4 phases per k-tile
    case 52: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 1>(SV_F16_ARGS);  // ping-pong, 1 phase per k-tile
#endif
    case 250:   // the default for batches: biased accumulators (see BIAS) when segvlad_search found the norms balanced enough
      if (!ctx->f16_bias_ok) goto unbiased_250;
      if ((int64_t)((M + 255) / 256) * ((n_sample + 255) / 256) >= 1024) {
#ifdef SEGVLAD_ABLATIONS   // the A/B variants of the batch kernel (development build)
        if (ctx->opt.f16_epi == 0)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true>(SV_F16_ARGS);      // persistent + ping-pong
        if (ctx->opt.f16_mf == 0)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1>(SV_F16_ARGS);   // + wave-private epilogue
        if (ctx->opt.f16_pp == 0)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 0, 0, true, 1, 1>(SV_F16_ARGS);   // plain loop (A/B)
        // (buffer_load lds is SLOWER in this kernel -- 18.70 vs 18.19 ms -- although faster in the micro-benchmark's loop and in
        //  the deep-row kernel, 125.2 vs 126.4 ms: only on request here)
        if (ctx->opt.f16_dsplit == -1)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, -1>(SV_F16_ARGS);   // fragment prefetch (A/B)
        if (ctx->opt.f16_dsplit == -2)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, -2>(SV_F16_ARGS);   // DMA-first load segment (A/B)
        if (ctx->opt.f16_dsplit == 2)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, 2>(SV_F16_ARGS);   // split DMA issue (A/B)
        if (ctx->opt.f16_dsplit == 1)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, false, 1>(SV_F16_ARGS);
        if (buf_ok && ctx->opt.f16_buf == 1)
          return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1, true>(SV_F16_ARGS);   // + buffer_load lds
#endif
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2, 0, true, 1, 1>(SV_F16_ARGS);  // + 16 x 16 x 32 MFMA
      }
#ifdef SEGVLAD_ABLATIONS
      if (ctx->opt.f16_small_mf == 1)   // (A/B: the small levels on the new shape + wave-private epilogue, non-persistent)
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 2, 0, true, 1, 1>(SV_F16_ARGS);
#endif
      return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 2, 0, true>(SV_F16_ARGS);
    case 251:   // 250 without the bias (A/B; norms too unbalanced for the biased margin)
    unbiased_250:
      if ((int64_t)((M + 255) / 256) * ((n_sample + 255) / 256) >= 1024)
        return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, true, 0, 2>(SV_F16_ARGS);
      return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 0, 2>(SV_F16_ARGS);
#ifdef SEGVLAD_ABLATIONS
    case 40: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 1>(SV_F16_ARGS);  // database rows non-temporal
    case 41: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 2>(SV_F16_ARGS);  // queries non-temporal
    case 42: return launch_f16_filter<256, 256, 4, 2, 64, 3, 0, false, 3>(SV_F16_ARGS);  // both
    case 1: return launch_f16_filter<256, 256, 4, 2, 32, 2>(SV_F16_ARGS);  //  64 KiB LDS, 2 workgroups / CU
    case 2: return launch_f16_filter<128, 128, 2, 2, 64, 3>(SV_F16_ARGS);  //  80 KiB
    case 4: return launch_f16_filter<256, 256, 4, 2, 32, 3>(SV_F16_ARGS);  //  80 KiB
    case 60: return launch_f16_filter<64, 128, 1, 2, 64, 3>(SV_F16_ARGS);   //  64 KiB, 2 waves: one query image (<= 64 rows) per pass
#endif
    case 62: return launch_f16_filter<64, 128, 1, 2, 64, 3, 0, false, 1>(SV_F16_ARGS);   // 60 + database rows non-temporal
    case 63: return launch_f16_filter<128, 128, 2, 2, 64, 3, 0, false, 1>(SV_F16_ARGS);  // 2 + database rows non-temporal
#ifdef SEGVLAD_ABLATIONS
    case 5: return launch_f16_filter<256, 256, 2, 2, 64, 3>(SV_F16_ARGS);  // 4 waves of 128 x 128: 0.5 LDS fragment / MFMA
    case 55: return launch_f16_filter<256, 256, 2, 2, 64, 3, 0, false, 0, -1>(SV_F16_ARGS);  // + software-pipelined loop
    case 255:
      if ((int64_t)((M + 255) / 256) * ((n_sample + 255) / 256) >= 1024)
        return launch_f16_filter<256, 256, 2, 2, 64, 3, 0, true, 0, -1>(SV_F16_ARGS);       // + persistent
      return launch_f16_filter<256, 256, 2, 2, 64, 3, 0, false, 0, -1>(SV_F16_ARGS);
    case 15: return launch_f16_filter<256, 256, 2, 2, 64, 3, 1>(SV_F16_ARGS);
    case 155: return launch_f16_filter<256, 256, 2, 2, 64, 3, 1, false, 0, -1>(SV_F16_ARGS);
    case 6: return launch_f16_filter<256, 256, 2, 2, 32, 3>(SV_F16_ARGS);
    case 7: return launch_f16_filter<256, 128, 4, 1, 32, 3>(SV_F16_ARGS);  // 56 KiB, 4 waves: 2 independent workgroups / CU
    case 17: return launch_f16_filter<256, 128, 4, 1, 32, 3, 1>(SV_F16_ARGS);
    case 8: return launch_f16_filter<256, 128, 4, 1, 32, 2>(SV_F16_ARGS);
    case 9: return launch_f16_filter<128, 256, 2, 2, 32, 3>(SV_F16_ARGS);
    default: return launch_f16_filter<128, 128, 2, 2, 32, 2>(SV_F16_ARGS); //  32 KiB
#else
    default: return ctx->fail(SEGVLAD_ERR_ARG, "fp16 filter configuration %d exists in development builds only (SEGVLAD_BUILD_ABLATIONS=1)", c);
#endif
  }
#undef SV_F16_ARGS
}

// ---- candidate handling ----------------------------------------------------------------------------------
// wave-aggregated LDS histogram increment (keys cluster on few digits: a plain atomicAdd would serialise)
__device__ __forceinline__ void hist_add_(uint32_t* hist, bool active, uint32_t bin) {
  uint64_t todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const uint32_t lb = __shfl(bin, leader);
    const uint64_t same = __ballot(active && bin == lb) & todo;
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[lb], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

// Candidate lists are only RANKED here, never sorted: an MSB-first radix select over the keys held in LDS
// yields A_k, the k-th smallest approximate distance (+inf if fewer than k candidates).
//   mode 0: thr_out[q] = A_k
//   mode 1: refine list = ids with d2~ <= A_k + 2 eps(q) (unordered; at most rcap, more -> the row is flagged in ovf_rows)
__global__ __launch_bounds__(256) void select_approx_kernel(uint32_t* __restrict__ cnt, float* __restrict__ cd2,
                                                            uint32_t* __restrict__ cid, int cap, int k, int mode, int check,
                                                            const float* __restrict__ thr_in, int64_t thr_in_ld,
                                                            const float* __restrict__ qn, float c_eps, float rn_max,
                                                            float* __restrict__ thr_out, uint32_t* __restrict__ ref_cnt,
                                                            uint32_t* __restrict__ ref_id, int rcap,
                                                            uint32_t* __restrict__ ovf_rows,
                                                            uint32_t* __restrict__ ovf_count,
                                                            const uint32_t* __restrict__ todo,
                                                            uint32_t* __restrict__ rovf_rows,
                                                            uint32_t* __restrict__ rovf_count,
                                                            float* __restrict__ ref_lim) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);  // [cap]
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_digit, s_krem, s_n;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  if (todo && !todo[row]) return;   // already ranked by select_small_kernel
  __shared__ uint32_t s_c;
  if (tid == 0) {
    s_c = cnt[row];
    if (mode != 1) cnt[row] = 0u;   // the next level's filter appends from zero (no memset launch between the levels); mode 2: below
  }
  __syncthreads();
  const uint32_t c = s_c;
  // the threshold this list was collected under (read before thr_out -- possibly the same word -- is overwritten)
  const float t_in = ((check && mode == 1) || mode == 2) ? thr_in[row * thr_in_ld] : 0.f;
  auto flag_row = [&]() {
    // this query is redone later (rigorous thresholds / exact matrix path): a threshold of -inf keeps its candidate
    // list empty at the finer levels, an empty refine list makes the refinement a no-op
    if (tid == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      if (mode == 1) ref_cnt[row] = 0;
      else thr_out[row] = -INFINITY;
    }
  };
  if (c > (uint32_t)cap || ovf_rows[row] || (check && (int)c < k)) {   // overflow / flagged at a coarser level / too few
    flag_row();
    return;
  }
  for (int j = tid; j < (int)c; j += 256) keys[j] = f2key_(cd2[row * cap + j]);
  float ak = INFINITY;
  if ((int)c >= k) {
    uint32_t prefix = 0, mask = 0, krem = (uint32_t)k;
    for (int pass = 3; pass >= 0; --pass) {
      hist[tid] = 0;
      __syncthreads();
      const int shift = 8 * pass;
      for (int j0 = 0; j0 < (int)c; j0 += 256) {
        const int j = j0 + tid;
        const uint32_t key = (j < (int)c) ? keys[j] : 0u;
        hist_add_(hist, (j < (int)c) && ((key & mask) == prefix), (key >> shift) & 255u);
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
      }
      __syncthreads();
      prefix |= s_digit << shift;
      mask |= 255u << shift;
      krem = s_krem;
      __syncthreads();
    }
    ak = key2f_(prefix);
  } else {
    __syncthreads();
  }
  if (mode == 0) {
    if (tid == 0) thr_out[row] = ak;
    return;
  }
  if (mode == 2) {   // carry (see select_small_body): in place, 256 entries at a time -- a chunk's survivors land below its own start + 256
    const float t2 = fminf(ak, t_in);
    const float lim2 = t2 + 2.f * c_eps * sqrtf(qn[row] * rn_max);
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int j0 = 0; j0 < (int)c; j0 += 256) {
      const int j = j0 + tid;
      const float v = (j < (int)c) ? key2f_(keys[j]) : INFINITY;
      const uint32_t id = (j < (int)c) ? cid[row * cap + j] : 0u;
      __syncthreads();   // the chunk is in registers
      if (j < (int)c && v <= lim2) {
        const uint32_t pos = atomicAdd(&s_n, 1u);
        cd2[row * cap + pos] = v;
        cid[row * cap + pos] = id;
      }
      __syncthreads();   // (the next chunk's reads start at j0 + 256 >= every position written so far)
    }
    if (tid == 0) {
      thr_out[row] = t2;
      cnt[row] = s_n;
    }
    return;
  }
  // heuristic thresholds: the list holds every row with d2~ <= t_in + 2 eps; the refine set {d2~ <= A_k + 2 eps} is
  // contained in it iff A_k <= t_in
  if (check && !(ak <= t_in)) {
    flag_row();
    return;
  }
  const float flim = ak + 2.f * c_eps * sqrtf(qn[row] * rn_max);
  const uint32_t klim = f2key_(flim);
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int j = tid; j < (int)c; j += 256) {
    if (keys[j] <= klim) {
      const uint32_t slot = atomicAdd(&s_n, 1u);
      if (slot < (uint32_t)rcap) ref_id[row * rcap + slot] = cid[row * cap + j];
    }
  }
  __syncthreads();
  if (tid == 0) {
    if (s_n > (uint32_t)rcap) {
      // the band holds more rows than the first-tier refine list: second tier (refine2_compact_kernel + a refinement
      // pass straight from the candidate list), or -- without one -- the exact matrix path
      if (rovf_rows) {
        rovf_rows[row] = 1u;
        ref_lim[row] = flim;
        atomicAdd(rovf_count, 1u);
      } else if (atomicExch(&ovf_rows[row], 1u) == 0u) {
        atomicAdd(ovf_count, 1u);
      }
      ref_cnt[row] = 0;
    } else {
      ref_cnt[row] = s_n;
    }
  }
}

// The same ranking for lists of up to 4096 candidates (every list of the low-rank level scheme, and nearly every list of
// the rigorous one), one WAVE per query instead of one 256-thread workgroup: the keys live in registers (4, 16 or 64 per
// lane, by the list's length), the rank-th smallest is found by a binary MSB-first radix select whose per-bit counts are
// per-lane sums + one DPP wave reduction, the refine list is compacted by ballots.  No LDS, no barriers.  Longer lists are
// left to select_approx_kernel (todo[row] = 1).
// select_small_body: the ranking itself for lists of at most 64 * PER keys (PER register slots per lane, loops fully
// unrolled: a run-time bound on the slot loops cost a scalar branch per slot and bit -- 40 us per launch).
template <int PER>
__device__ __forceinline__ void select_small_body(uint32_t* __restrict__ cnt, float* __restrict__ cd2, uint32_t* __restrict__ cid, int64_t row, int l,
                                                  uint32_t c, int cap, int k, int mode, int check, float t_in,
                                                  const float* __restrict__ qn, float c_eps, float rn_max,
                                                  float* __restrict__ thr_out, uint32_t* __restrict__ ref_cnt,
                                                  uint32_t* __restrict__ ref_id, int rcap, uint32_t* __restrict__ ovf_rows,
                                                  uint32_t* __restrict__ ovf_count, uint32_t* __restrict__ rovf_rows,
                                                  uint32_t* __restrict__ rovf_count, float* __restrict__ ref_lim) {
  auto flag_row = [&]() {
    if (l == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      if (mode == 1) ref_cnt[row] = 0;
      else thr_out[row] = -INFINITY;
    }
  };
  constexpr uint32_t PAD = 0xffffffffu;   // padding sorts last (a real key is never all ones: NaN-free)
  uint32_t key[PER], cidv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = l + 64 * i;
    key[i] = PAD;
    cidv[i] = 0u;
    if (j < (int)c) {
      key[i] = f2key_(cd2[row * cap + j]);
      // mode 1: the ids travel with the keys (fetched behind a ballot branch, slot by slot, each was a round trip of its own:
      // 12 of the 19 us of a pass's last select)
      if (mode != 0) cidv[i] = cid[row * cap + j];
    }
  }
  float ak = INFINITY;
  if ((int)c >= k) {
    // The keys of a list share their leading bits (distances of one query: same sign, a handful of exponents), and after a
    // dozen more only one key still matches the prefix: the bit loop starts below the common prefix of the list's smallest and
    // largest key and stops as soon as a single candidate is left (32 bits x 2 PER VALU instructions were 7 of the ~10 us of a
    // 4096-key launch).  All of it is wave-uniform.
    uint32_t kmn = 0xffffffffu, kmx = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      kmn = min(kmn, key[i]);
      kmx = key[i] != PAD ? max(kmx, key[i]) : kmx;
    }
    kmn = wave_min_u32_(kmn);
    kmx = wave_max_u32_(kmx);
    const uint32_t diff = kmn ^ kmx;
    uint32_t prefix = kmn, mask = 0xffffffffu, rem = (uint32_t)k, m = c;   // diff == 0: every key is kmn
    int bit = -1;
    if (diff) {
      bit = 31 - __builtin_clz(diff);
      mask = (bit == 31) ? 0u : ~((2u << bit) - 1u);
      prefix = kmn & mask;
    }
    for (; bit >= 0 && m > 1u; --bit) {
      const uint32_t b = 1u << bit;
      // keys that match the prefix so far and have this bit clear: counted per lane (a compare + an add per key slot), then
      // ONE wave sum per bit by DPP row reductions + four readlanes.  (The butterfly of six ds_bpermute shuffles it replaces
      // was ~8 us of LDS-crossbar latency per launch; a ballot + scalar popcount per slot stalls on the VALU -> SALU
      // hand-over of every compare: 42 us for 64 slots.)
      uint32_t zl = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) zl += ((key[i] & (mask | b)) == prefix) ? 1u : 0u;
      const uint32_t zeros = wave_sum_u32_(zl);
      if (rem > zeros) {
        rem -= zeros;
        m -= zeros;
        prefix |= b;
      } else {
        m = zeros;
      }
      mask |= b;
    }
    if (bit >= 0) {   // one key left under the prefix: it is the answer, whatever its remaining bits
      uint32_t v = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) v |= (key[i] != PAD && (key[i] & mask) == prefix) ? key[i] : 0u;
      prefix = wave_max_u32_(v);
    }
    ak = key2f_(prefix);
  }
  if (mode == 0) {
    if (l == 0) thr_out[row] = ak;
    return;
  }
  if (mode == 2) {
    // carry: the next level runs over the rows this level has NOT seen (the complement of its sample), under the threshold
    // t2 = min(A_k, t_in).  This list holds every sampled row with d2~ <= t_in + 2 eps, hence every one with d2~ <= t2 + 2 eps --
    // exactly the rows the next level's filter would append for the sample: they are compacted to the front and the counter is
    // left at their number.  (A threshold below A_k is as good a guess as A_k: the last level's check is against the value stored.)
    const float t2 = fminf(ak, t_in);
    const float lim2 = t2 + 2.f * c_eps * sqrtf(qn[row] * rn_max);
    uint32_t kept = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const float v = key2f_(key[i]);
      const bool hit = key[i] != PAD && v <= lim2;
      const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
      if (mk != 0ull) {
        const uint32_t pos = kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (hit) {   // (every key of the list is in registers: writing in place races with nothing)
          cd2[row * cap + pos] = v;
          cid[row * cap + pos] = cidv[i];
        }
        kept += (uint32_t)__popcll(mk);
      }
    }
    if (l == 0) {
      thr_out[row] = t2;
      cnt[row] = kept;
    }
    return;
  }
  if (check && !(ak <= t_in)) {
    flag_row();
    return;
  }
  const float flim = ak + 2.f * c_eps * sqrtf(qn[row] * rn_max);
  const uint32_t klim = f2key_(flim);
  uint32_t total = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool hit = key[i] <= klim;   // (klim is a finite float's key: the padding never hits)
    const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
    if (mk != 0ull) {
      const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      if (hit && pos < (uint32_t)rcap) ref_id[row * rcap + pos] = cidv[i];
      total += (uint32_t)__popcll(mk);
    }
  }
  if (l == 0) {
    if (total > (uint32_t)rcap) {
      if (rovf_rows) {   // second tier (see select_approx_kernel)
        rovf_rows[row] = 1u;
        ref_lim[row] = flim;
        atomicAdd(rovf_count, 1u);
      } else if (atomicExch(&ovf_rows[row], 1u) == 0u) {
        atomicAdd(ovf_count, 1u);
      }
      ref_cnt[row] = 0;
    } else {
      ref_cnt[row] = total;
    }
  }
}

// k-th smallest of a workgroup's keys (PER per thread, padding = all ones), c >= 1 real keys among them; +inf if c < k.
// xs: [2][4] LDS exchange slots.  Every thread returns the same value.
template <int PER>
__device__ __forceinline__ float wg_kth_smallest_(const uint32_t (&key)[PER], uint32_t c, int k, uint32_t (*xs)[4], int tid) {
  constexpr uint32_t PAD = 0xffffffffu;
  const int w = tid >> 6;
  int turn = 0;
  auto exchange = [&](uint32_t v_wave) {   // v_wave: this wave's (uniform) partial; returns the four partials
    if ((tid & 63) == 0) xs[turn][w] = v_wave;
    __syncthreads();
    const uint4 r = make_uint4(xs[turn][0], xs[turn][1], xs[turn][2], xs[turn][3]);
    turn ^= 1;
    return r;
  };
  float ak = INFINITY;
  if ((int)c >= k) {
    uint32_t kmn = PAD, kmx = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      kmn = min(kmn, key[i]);
      kmx = key[i] != PAD ? max(kmx, key[i]) : kmx;
    }
    {
      const uint4 r = exchange(wave_min_u32_(kmn));
      kmn = min(min(r.x, r.y), min(r.z, r.w));
    }
    {
      const uint4 r = exchange(wave_max_u32_(kmx));
      kmx = max(max(r.x, r.y), max(r.z, r.w));
    }
    const uint32_t diff = kmn ^ kmx;
    uint32_t prefix = kmn, mask = 0xffffffffu, rem = (uint32_t)k, m = c;   // diff == 0: every key is kmn
    int bit = -1;
    if (diff) {
      bit = 31 - __builtin_clz(diff);
      mask = (bit == 31) ? 0u : ~((2u << bit) - 1u);
      prefix = kmn & mask;
    }
    for (; bit >= 0 && m > 1u; --bit) {
      const uint32_t b = 1u << bit;
      uint32_t zl = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) zl += ((key[i] & (mask | b)) == prefix) ? 1u : 0u;
      const uint4 r = exchange(wave_sum_u32_(zl));
      const uint32_t zeros = r.x + r.y + r.z + r.w;
      if (rem > zeros) {
        rem -= zeros;
        m -= zeros;
        prefix |= b;
      } else {
        m = zeros;
      }
      mask |= b;
    }
    if (bit >= 0) {   // one key left under the prefix: it is the answer, whatever its remaining bits
      uint32_t v = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) v |= (key[i] != PAD && (key[i] & mask) == prefix) ? key[i] : 0u;
      const uint4 r = exchange(wave_max_u32_(v));
      prefix = max(max(r.x, r.y), max(r.z, r.w));
    }
    ak = key2f_(prefix);
  }
  return ak;
}

// One query image per pass (<= 128 lists): a whole workgroup per list instead of a wave -- 32 keys per thread (lists of up
// to 8192 entries, the capacity of the candidate lists; longer ones are flagged for the exact path), the same
// binary MSB-first radix select below the common prefix of the list's smallest and largest key, stopping when one key is
// left; the per-bit count is a DPP wave sum + a four-entry LDS exchange (one barrier per bit: the exchange slots alternate).
// The wave kernel's 64-keys-per-lane instantiation is ~8000 straight-line instructions that a pass runs through ONCE --
// instruction fetch, not arithmetic: 29 us for a 3906-entry sample row; this kernel takes ~10.
// PERK: keys per thread.  32 covers the candidate lists' capacity (8192); 16 (lists of <= 4096 entries -- every list the
// single-image plan produces in practice) halves the slot loops of the load and of every radix step: the kernel picks the
// body by the list's length (workgroup-uniform).
template <int PERK>
__device__ __forceinline__ void select_wg_body(uint32_t* __restrict__ cnt, const float* __restrict__ cd2,
                                                        const uint32_t* __restrict__ cid, int cap, int k, int mode, int check,
                                                        const float* __restrict__ thr_in, int64_t thr_in_ld,
                                                        const float* __restrict__ qn, float c_eps, float rn_max,
                                                        float* __restrict__ thr_out, uint32_t* __restrict__ ref_cnt,
                                                        uint32_t* __restrict__ ref_id, int rcap, uint32_t* __restrict__ ovf_rows,
                                                        uint32_t* __restrict__ ovf_count, uint32_t* __restrict__ rovf_rows,
                                                        uint32_t* __restrict__ rovf_count, float* __restrict__ ref_lim, int fixed_cnt,
                                                        const uint32_t c, const float (&pre_d2)[16], const uint32_t (&pre_id)[16],
                                                        const uint32_t flagged, const float t_in) {
  constexpr uint32_t PAD = 0xffffffffu;
  constexpr int PER = PERK;   // 32: 8192 keys, the candidate lists' capacity (SV_CAP)
  __shared__ uint32_t xs[2][4];
  __shared__ uint32_t s_n;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  uint32_t key[PER], cidv[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int j = tid + 256 * i;
    key[i] = PAD;
    cidv[i] = 0u;
    if (j < (int)c && c <= (uint32_t)(256 * PER)) {
      // (the first 16 slots per thread were requested by the kernel together with the list's length: one round trip, not two)
      key[i] = f2key_(i < 16 ? pre_d2[i < 16 ? i : 0] : cd2[row * cap + j]);
      if (mode == 1) cidv[i] = i < 16 ? pre_id[i < 16 ? i : 0] : cid[row * cap + j];
    }
  }
  if (tid == 0) s_n = 0u;
  __syncthreads();   // every thread has read cnt[row]
  if (tid == 0 && mode == 0) cnt[row] = 0u;   // the next level's filter appends from zero
  if (c > (uint32_t)cap || c > (uint32_t)(256 * PER) || flagged || (check && (int)c < k)) {
    if (tid == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      if (mode == 1) ref_cnt[row] = 0;
      else thr_out[row] = -INFINITY;
    }
    return;
  }
  const float ak = wg_kth_smallest_<PER>(key, c, k, xs, tid);
  if (mode == 0) {
    if (tid == 0) thr_out[row] = ak;
    return;
  }
  if (check && !(ak <= t_in)) {   // see select_approx_kernel
    if (tid == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      ref_cnt[row] = 0;
    }
    return;
  }
  const float flim = ak + 2.f * c_eps * sqrtf(qn[row] * rn_max);
  const uint32_t klim = f2key_(flim);
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (key[i] <= klim) {   // (a finite float's key: the padding never hits)
      const uint32_t pos = atomicAdd(&s_n, 1u);
      if (pos < (uint32_t)rcap) ref_id[row * rcap + pos] = cidv[i];
    }
  __syncthreads();
  if (tid == 0) {
    const uint32_t total = s_n;
    if (total > (uint32_t)rcap) {
      if (rovf_rows) {   // second tier (see select_approx_kernel)
        rovf_rows[row] = 1u;
        ref_lim[row] = flim;
        atomicAdd(rovf_count, 1u);
      } else if (atomicExch(&ovf_rows[row], 1u) == 0u) {
        atomicAdd(ovf_count, 1u);
      }
      ref_cnt[row] = 0;
    } else {
      ref_cnt[row] = total;
    }
  }
}

__global__ __launch_bounds__(256) void select_wg_kernel(uint32_t* __restrict__ cnt, const float* __restrict__ cd2,
                                                        const uint32_t* __restrict__ cid, int cap, int k, int mode, int check,
                                                        const float* __restrict__ thr_in, int64_t thr_in_ld,
                                                        const float* __restrict__ qn, float c_eps, float rn_max,
                                                        float* __restrict__ thr_out, uint32_t* __restrict__ ref_cnt,
                                                        uint32_t* __restrict__ ref_id, int rcap, uint32_t* __restrict__ ovf_rows,
                                                        uint32_t* __restrict__ ovf_count, uint32_t* __restrict__ rovf_rows,
                                                        uint32_t* __restrict__ rovf_count, float* __restrict__ ref_lim, int fixed_cnt) {
  // Everything the list's length decides is REQUESTED before the length is known: the first 16 slots of every thread (lists of
  // <= 4096 entries -- every list the single-image plan produces in practice -- are complete with them; the slots lie inside the
  // row's `cap` entries whatever the length, entries beyond it are never looked at), the row's flag, its threshold.  The length
  // used to be a round trip of its own in front of them (round 6: ~1.5 us of a 20-us kernel that is a chain of such trips).
  const int64_t row = blockIdx.x;
  float pre_d2[16];
  uint32_t pre_id[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = threadIdx.x + 256 * i;
    const bool in = j < cap;
    pre_d2[i] = in ? cd2[row * cap + j] : 0.f;
    pre_id[i] = (in && mode == 1) ? cid[row * cap + j] : 0u;
  }
  const uint32_t flagged = ovf_rows[row];
  const float t_in = (check && mode == 1) ? thr_in[row * thr_in_ld] : 0.f;
  const uint32_t c = fixed_cnt >= 0 ? (uint32_t)fixed_cnt : cnt[row];
  if (c <= 4096u)
    select_wg_body<16>(cnt, cd2, cid, cap, k, mode, check, thr_in, thr_in_ld, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                       ovf_count, rovf_rows, rovf_count, ref_lim, fixed_cnt, c, pre_d2, pre_id, flagged, t_in);
  else
    select_wg_body<32>(cnt, cd2, cid, cap, k, mode, check, thr_in, thr_in_ld, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                       ovf_count, rovf_rows, rovf_count, ref_lim, fixed_cnt, c, pre_d2, pre_id, flagged, t_in);
}

// The sampled exact level of a single-image pass: the K-split partial dot products of <= 128 query rows against <= 4096
// sample rows are reduced (slices added in index order, sv_d2 with the norms: splitk_reduce_d2_kernel's arithmetic, value for
// value) and the row's rank-th smallest distance is selected in the same workgroup -- one launch instead of two in a pass
// that is a chain of dependent launches.
__global__ __launch_bounds__(256) void l0_reduce_rank_kernel(const float* __restrict__ part, int splits, int M, int N, int64_t ldc,
                                                             const float* __restrict__ row_add, const float* __restrict__ col_add,
                                                             int b_stride, int rank, float* __restrict__ thr_out,
                                                             uint32_t* __restrict__ cnt, const uint32_t* __restrict__ ovf_rows) {
  constexpr uint32_t PAD = 0xffffffffu;
  constexpr int PER = 16;
  __shared__ uint32_t xs[2][4];
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int64_t mn = (int64_t)M * ldc;
  const float q2 = row_add[row];
  // slice-major: the 16 loads of a slice are in flight together, two slices per round trip (one load after the other down a
  // column is 8 dependent round trips per key: 35 us for this kernel)
  float sum[PER], rn[PER];
  const float* p0 = part + row * ldc + tid;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const bool in = tid + 256 * i < N;
    sum[i] = in ? p0[256 * i] : 0.f;
    rn[i] = in ? col_add[(int64_t)(tid + 256 * i) * b_stride] : 0.f;
  }
  int t = 1;
  for (; t + 1 < splits; t += 2) {
    float a[PER], b[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const bool in = tid + 256 * i < N;
      a[i] = in ? p0[(int64_t)t * mn + 256 * i] : 0.f;
      b[i] = in ? p0[(int64_t)(t + 1) * mn + 256 * i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) sum[i] = (sum[i] + a[i]) + b[i];   // (index order, as splitk_reduce_d2_kernel adds them)
  }
  if (t < splits) {
#pragma unroll
    for (int i = 0; i < PER; ++i) sum[i] += (tid + 256 * i < N) ? p0[(int64_t)t * mn + 256 * i] : 0.f;
  }
  uint32_t key[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) key[i] = (tid + 256 * i < N) ? f2key_(sv_d2(q2, rn[i], sum[i])) : PAD;
  if (tid == 0) cnt[row] = 0u;   // the next level's filter appends from zero
  if (ovf_rows[row] || N < rank) {   // (workgroup-uniform)
    if (tid == 0) thr_out[row] = -INFINITY;
    return;
  }
  const float ak = wg_kth_smallest_<PER>(key, (uint32_t)N, rank, xs, tid);
  if (tid == 0) thr_out[row] = ak;
}

int sv_launch_l0_reduce_rank(segvlad_ctx* ctx, const float* parts, int splits, int M, int n_sample, int64_t ldc, const float* qn,
                             const float* rn, int b_stride, int rank, float* thr_out, uint32_t* cand_cnt, const uint32_t* fail_rows) {
  if (M <= 0) return SEGVLAD_OK;
  if (n_sample > 4096) return ctx->fail(SEGVLAD_ERR_LIMIT, "l0_reduce_rank: rows of at most 4096 columns");
  hipLaunchKernelGGL(l0_reduce_rank_kernel, dim3(M), dim3(256), 0, ctx->stream, parts, splits, M, n_sample, ldc, qn, rn, b_stride, rank,
                     thr_out, cand_cnt, fail_rows);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

__global__ __launch_bounds__(256) void select_small_kernel(uint32_t* __restrict__ cnt, float* __restrict__ cd2,
                                                           uint32_t* __restrict__ cid, int nq, int cap, int k, int mode, int check,
                                                           const float* __restrict__ thr_in, int64_t thr_in_ld,
                                                           const float* __restrict__ qn, float c_eps, float rn_max,
                                                           float* __restrict__ thr_out, uint32_t* __restrict__ ref_cnt,
                                                           uint32_t* __restrict__ ref_id, int rcap, uint32_t* __restrict__ ovf_rows,
                                                           uint32_t* __restrict__ ovf_count, uint32_t* __restrict__ todo,
                                                           uint32_t* __restrict__ rovf_rows, uint32_t* __restrict__ rovf_count,
                                                           float* __restrict__ ref_lim, int fixed_cnt) {
  constexpr int PER = 64;   // up to 4096 keys per wave, in registers
  const int l = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nq) return;
  const uint32_t c = fixed_cnt >= 0 ? (uint32_t)fixed_cnt : cnt[row];   // fixed_cnt: every row is a list of that length
  if (todo && c > (uint32_t)(64 * PER) && c <= (uint32_t)cap && !ovf_rows[row]) {   // long list: the workgroup kernel ranks it
    if (l == 0) todo[row] = 1u;
    return;
  }
  if (l == 0) {
    if (todo) todo[row] = 0u;
    if (mode != 1) cnt[row] = 0u;   // the next level's filter appends from zero (no memset launch between the levels); mode 2: see the body
  }
  const float t_in = ((check && mode == 1) || mode == 2) ? thr_in[row * thr_in_ld] : 0.f;
  auto flag_row = [&]() {
    if (l == 0) {
      if (atomicExch(&ovf_rows[row], 1u) == 0u) atomicAdd(ovf_count, 1u);
      if (mode == 1) ref_cnt[row] = 0;
      else thr_out[row] = -INFINITY;
    }
  };
  // todo == null (a handful of queries: the workgroup kernel is not even launched): a list beyond the wave's 4096 keys is
  // treated like an overflowing one -- the query is redone on the exact path
  if (c > (uint32_t)cap || (!todo && c > (uint32_t)(64 * PER)) || ovf_rows[row] || (check && (int)c < k)) {
    flag_row();
    return;
  }
  if (c <= 256u)
    select_small_body<4>(cnt, cd2, cid, row, l, c, cap, k, mode, check, t_in, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                         ovf_count, rovf_rows, rovf_count, ref_lim);
  else if (c <= 1024u)
    select_small_body<16>(cnt, cd2, cid, row, l, c, cap, k, mode, check, t_in, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                          ovf_count, rovf_rows, rovf_count, ref_lim);
  else if (c <= 2048u)
    select_small_body<32>(cnt, cd2, cid, row, l, c, cap, k, mode, check, t_in, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                          ovf_count, rovf_rows, rovf_count, ref_lim);
  else
    select_small_body<64>(cnt, cd2, cid, row, l, c, cap, k, mode, check, t_in, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, ovf_rows,
                          ovf_count, rovf_rows, rovf_count, ref_lim);
}

int sv_launch_select_approx(segvlad_ctx* ctx, uint32_t* cand_cnt, float* cand_d2, uint32_t* cand_id, int nq,
                            int cap, int rank, int mode, int check, const float* thr_in, int64_t thr_in_ld, const float* qn,
                            float c_eps, float rn_max, float* thr_out, uint32_t* ref_cnt, uint32_t* ref_id, int rcap,
                            uint32_t* fail_rows, uint32_t* fail_count, uint32_t* rovf_rows, uint32_t* rovf_count, float* ref_lim,
                            int fixed_cnt) {
  if (nq <= 0) return SEGVLAD_OK;
  if (mode == 2 && nq <= 128) return ctx->fail(SEGVLAD_ERR_STATE, "select: the carrying form exists for batches only");
  if (nq <= 128) {   // one query image per pass: a workgroup per list
    hipLaunchKernelGGL(select_wg_kernel, dim3(nq), dim3(256), 0, ctx->stream, cand_cnt, cand_d2, cand_id, cap, rank, mode, check, thr_in,
                       thr_in_ld, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, fail_rows, fail_count, rovf_rows, rovf_count,
                       ref_lim, fixed_cnt);
    SV_HIP(hipGetLastError());
    return SEGVLAD_OK;
  }
  const bool wave_only = fixed_cnt >= 0 && fixed_cnt <= 4096;
  uint32_t* todo = nullptr;
  if (!wave_only) {
    SV_HIP(ctx->s_sel_todo.reserve((size_t)nq * 4));
    todo = ctx->s_sel_todo.as<uint32_t>();
  }
  hipLaunchKernelGGL(select_small_kernel, dim3((nq + 3) / 4), dim3(256), 0, ctx->stream, cand_cnt, cand_d2, cand_id, nq, cap, rank, mode,
                     check, thr_in, thr_in_ld, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, fail_rows, fail_count, todo,
                     rovf_rows, rovf_count, ref_lim, fixed_cnt);
  if (wave_only) {   // every list is ranked (or flagged) by the wave kernel
    SV_HIP(hipGetLastError());
    return SEGVLAD_OK;
  }
  const size_t lds = (size_t)cap * 4;
  hipLaunchKernelGGL(select_approx_kernel, dim3(nq), dim3(256), lds, ctx->stream, cand_cnt, cand_d2, cand_id, cap, rank, mode, check,
                     thr_in, thr_in_ld, qn, c_eps, rn_max, thr_out, ref_cnt, ref_id, rcap, fail_rows, fail_count, todo,
                     rovf_rows, rovf_count, ref_lim);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// Second refinement tier.  A query whose band {d2~ <= A_k + 2 eps} holds more rows than the first-tier list (SV_RCAP) --
// temporally redundant databases: every reference segment comes with its ~30 near-duplicates from the neighbouring video
// frames, so whole clumps of rows sit inside the band -- keeps its candidate list (<= cap entries, a superset of the band):
// this kernel compacts the band's ids to the front of that list, in place, and the exact refinement then runs straight
// from it (rcap = cap).  Only the flagged rows do any work; nobody is sent to the distance-matrix path for this.
__global__ __launch_bounds__(256) void refine2_compact_kernel(const uint32_t* __restrict__ rovf_rows, const float* __restrict__ ref_lim,
                                                              uint32_t* __restrict__ cnt, const float* __restrict__ cd2,
                                                              uint32_t* __restrict__ cid, int cap) {
  __shared__ uint32_t wtot[4];
  const int64_t row = blockIdx.x;
  if (!rovf_rows[row]) return;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const uint32_t c = cnt[row];
  const float lim = ref_lim[row];
  uint32_t base = 0;
  for (uint32_t j0 = 0; j0 < c; j0 += 256) {
    const uint32_t j = j0 + tid;
    const bool hit = j < c && cd2[row * cap + j] <= lim;
    const uint32_t id = hit ? cid[row * cap + j] : 0u;
    const uint64_t mk = __builtin_amdgcn_ballot_w64(hit);
    if (l == 0) wtot[w] = (uint32_t)__popcll(mk);
    __syncthreads();   // every read of this chunk precedes its writes (which land at positions <= the reads': in place is safe)
    uint32_t off = base;
    for (int x = 0; x < w; ++x) off += wtot[x];
    const uint32_t pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
    if (hit) cid[row * cap + pos] = id;
    base += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  }
  if (tid == 0) cnt[row] = base;
}

int sv_launch_refine2_compact(segvlad_ctx* ctx, const uint32_t* rovf_rows, const float* ref_lim, uint32_t* cand_cnt,
                              const float* cand_d2, uint32_t* cand_id, int nq, int cap) {
  if (nq <= 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(refine2_compact_kernel, dim3(nq), dim3(256), 0, ctx->stream, rovf_rows, ref_lim, cand_cnt, cand_d2, cand_id, cap);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// exact distances of the refine list: the sequential fp32 fma chain over k = 0..d-1 (bit-identical to the
// v_mfma_f32_32x32x2_f32 chain of the matrix path), then (distance, id) sort and top-k.  One thread per candidate row; the
// query row is cached in LDS (QLDS; d up to ~38k).  The row is walked with THIRTY-TWO 16-byte loads in flight per thread:
// a 50-query pass has fewer waves than the chip has SIMDs, so the loop is pure load latency -- one round trip per
// 32 x 16 B (an 8-deep register double buffer still paid one round trip per 128 B: 78 us per pass).
// only_rows != null: rows whose flag is clear are left untouched (second refinement tier).
template <bool QLDS>
__global__ __launch_bounds__(256) void refine_exact_kernel(const float* __restrict__ Q, const float* __restrict__ R, int d,
                                                           const float* __restrict__ qn, const float* __restrict__ rn,
                                                           const uint32_t* __restrict__ ref_cnt,
                                                           const uint32_t* __restrict__ ref_id, int rcap, int rpad, int k,
                                                           float* __restrict__ d2_out, int64_t* __restrict__ idx_out,
                                                           const uint32_t* __restrict__ only_rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);                                     // [d] when QLDS
  uint64_t* a = reinterpret_cast<uint64_t*>(smem + (QLDS ? (size_t)d * 4 : 0));   // [rpad]
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  if (only_rows && !only_rows[row]) return;
  const int n = (int)ref_cnt[row];
  int np2 = 2;   // sort length: the smallest power of two holding the list (<= rpad)
  while (np2 < n) np2 <<= 1;
  if (QLDS)
    for (int j = tid; j < d; j += 256) qs[j] = Q[row * d + j];
  for (int j = tid; j < np2; j += 256) a[j] = ~0ull;
  __syncthreads();
  const float q2 = qn[row];
  const float4* qp = QLDS ? reinterpret_cast<const float4*>(qs) : reinterpret_cast<const float4*>(Q + row * d);
  const int n4 = d >> 2;
  for (int j = tid; j < n; j += 256) {
    const uint32_t id = ref_id[row * rcap + j];
    const float4* rp = reinterpret_cast<const float4*>(R + (size_t)id * d);
    float acc = 0.f;
    int t = 0;
    if ((n4 & 31) == 0) {
      for (; t < n4; t += 32) {   // 512 B of the row in flight per lane: 8 round trips for a 1024-d row
        float4 buf[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) buf[u] = rp[t + u];
        __builtin_amdgcn_sched_barrier(0);   // all 32 loads are issued before the first fma (the scheduler otherwise
                                             // sinks them next to their uses to save registers -- and pays the latency 32 times)
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const float4 qv = qp[t + u];
          acc = fmaf(qv.x, buf[u].x, acc);
          acc = fmaf(qv.y, buf[u].y, acc);
          acc = fmaf(qv.z, buf[u].z, acc);
          acc = fmaf(qv.w, buf[u].w, acc);
        }
      }
    } else {
      for (; t < n4; ++t) {
        const float4 rv = rp[t];
        const float4 qv = qp[t];
        acc = fmaf(qv.x, rv.x, acc);
        acc = fmaf(qv.y, rv.y, acc);
        acc = fmaf(qv.z, rv.z, acc);
        acc = fmaf(qv.w, rv.w, acc);
      }
    }
    const float v = sv_d2(q2, rn[id], acc);
    a[j] = ((uint64_t)f2key_(v) << 32) | id;
  }
  bitonic64(a, np2, tid);
  for (int j = tid; j < k; j += 256) {
    float dd = INFINITY;
    int64_t id = -1;
    if (j < n) {
      dd = key2f_((uint32_t)(a[j] >> 32));
      id = (int64_t)(uint32_t)a[j];
    }
    d2_out[row * k + j] = dd;
    idx_out[row * k + j] = id;
  }
}

// Deep rows (raw K*D descriptors: d = 98 304 is 384 KiB per row, far beyond the LDS and -- one row per lane -- beyond what
// L1 can keep of 256 private streams): the same sequential chain, with the candidate rows fetched COALESCED (KC * 4 bytes
// of a row per step: C4 lanes x 16 B) into an LDS tile [64 rows][KC (+4 pad)], double buffered, which the 64 lanes of wave 0
// then walk ONE ROW EACH (conflict-free ds_read_b128: the row stride is 4 banks mod 64); all four waves load.
// What bounds it is bytes in flight: one workgroup per CU (the tile) with one step of 32 KiB outstanding ran at 1.4-1.8 TB/s
// -- piece size, a time skew between the rows and the 3 * 2^17-byte row pitch made no difference, and
// tools/ubench/gather_bw.hip reaches 7 TB/s on the same addresses with 256 KiB per CU in flight.  So the loads run DEPTH
// steps ahead in a register ring (DEPTH x NP float4 per thread: 128 KiB per CU at DEPTH = 4), and only the step that is due
// is written to the LDS tile.  d % KC == 0.
template <int KC, int DEPTH>
__global__ __launch_bounds__(256) void refine_exact_wide_kernel(const float* __restrict__ Q, const float* __restrict__ R, int d,
                                                                const float* __restrict__ qn, const float* __restrict__ rn,
                                                                const uint32_t* __restrict__ ref_cnt,
                                                                const uint32_t* __restrict__ ref_id, int rcap, int rpad, int k,
                                                                float* __restrict__ d2_out, int64_t* __restrict__ idx_out,
                                                                const uint32_t* __restrict__ only_rows) {
  constexpr int ROWS = 64, LDR = KC + 4, C4 = KC / 4;    // C4 16-byte pieces per row and step
  constexpr int NP = ROWS * C4 / 256;                    // pieces per thread and step
  constexpr int RSTEP = 256 / C4;                        // tile rows between two pieces of a thread
  static_assert(C4 <= 64 && 64 % C4 == 0 && NP * 256 == ROWS * C4, "a wave covers whole rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                                 // [2][ROWS][LDR]
  float* qs = tile + 2 * ROWS * LDR;                                            // [2][KC]
  uint32_t* ids = reinterpret_cast<uint32_t*>(qs + 2 * KC);                     // [ROWS]
  uint64_t* a = reinterpret_cast<uint64_t*>(ids + ROWS);                        // [rpad]
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  if (only_rows && !only_rows[row]) return;
  const int n = (int)ref_cnt[row];
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  for (int j = tid; j < np2; j += 256) a[j] = ~0ull;
  const float q2 = qn[row];
  const float* qrow = Q + row * d + (tid < C4 ? tid * 4 : 0);
  const int nch = d / KC;
  const int seg = tid % C4, lrow0 = tid / C4;   // piece u of this thread: tile row lrow0 + RSTEP u, 16-byte segment seg
  for (int base = 0; base < n; base += ROWS) {
    const int cnt = (n - base < ROWS) ? (n - base) : ROWS;
    __syncthreads();   // the previous pass is done with ids[] and the tile
    if (tid < cnt) ids[tid] = ref_id[row * rcap + base + tid];
    __syncthreads();
    // (no per-piece predication: a branch around every load makes the compiler wait for each one.  Tile rows beyond the
    //  list re-read candidate 0 -- L2 hits -- and are never looked at; every thread carries a query piece, lanes >= C4 a
    //  duplicate of piece 0 that is never stored)
    const float* src[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
      const int lr = lrow0 + RSTEP * u;
      src[u] = R + (size_t)ids[lr < cnt ? lr : 0] * d + seg * 4;
    }
    // The ring is four NAMED register sets and the step is a macro instantiated once per set: hipcc 7.2 sends a
    // [DEPTH][NP] array that is indexed through a lambda parameter to scratch memory (seen in the ISA: scratch_load/_store
    // around every piece, vmcnt(0) after every load).
    static_assert(DEPTH == 4 && NP == 8, "four named ring sets of eight named pieces");
#define SV_WG_DECL(X) float4 g##X##0, g##X##1, g##X##2, g##X##3, g##X##4, g##X##5, g##X##6, g##X##7, q##X
    SV_WG_DECL(A);
    SV_WG_DECL(B);
    SV_WG_DECL(C);
    SV_WG_DECL(D);
#define SV_WG_LOAD(X, c_)                                                          \
    do {                                                                           \
      const size_t o_ = (size_t)(c_) * KC;                                         \
      g##X##0 = *reinterpret_cast<const float4*>(src[0] + o_);                     \
      g##X##1 = *reinterpret_cast<const float4*>(src[1] + o_);                     \
      g##X##2 = *reinterpret_cast<const float4*>(src[2] + o_);                     \
      g##X##3 = *reinterpret_cast<const float4*>(src[3] + o_);                     \
      g##X##4 = *reinterpret_cast<const float4*>(src[4] + o_);                     \
      g##X##5 = *reinterpret_cast<const float4*>(src[5] + o_);                     \
      g##X##6 = *reinterpret_cast<const float4*>(src[6] + o_);                     \
      g##X##7 = *reinterpret_cast<const float4*>(src[7] + o_);                     \
      q##X = *reinterpret_cast<const float4*>(qrow + o_);                          \
    } while (0)
#define SV_WG_STORE(X, buf_)                                                       \
    do {                                                                           \
      float* t_ = tile + ((size_t)(buf_) * ROWS + lrow0) * LDR + seg * 4;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 0 * LDR) = g##X##0;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 1 * LDR) = g##X##1;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 2 * LDR) = g##X##2;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 3 * LDR) = g##X##3;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 4 * LDR) = g##X##4;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 5 * LDR) = g##X##5;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 6 * LDR) = g##X##6;          \
      *reinterpret_cast<float4*>(t_ + (size_t)RSTEP * 7 * LDR) = g##X##7;          \
      if (tid < C4) *reinterpret_cast<float4*>(qs + (buf_) * KC + tid * 4) = q##X; \
    } while (0)
    // one step: multiply step c out of tile buffer c & 1, move step c + 1 (ring set XN) into the other buffer, request
    // step c + 1 + DEPTH into the set that just became free
#define SV_WG_MUL(c)                                                                                       \
    if (tid < cnt) {                                                                                       \
      const float* tr = tile + ((size_t)((c) & 1) * ROWS + tid) * LDR;                                     \
      const float* qb = qs + ((c) & 1) * KC;                                                               \
      _Pragma("unroll 8") for (int s4 = 0; s4 < C4; ++s4) {                                                \
        const float4 rv = *reinterpret_cast<const float4*>(tr + s4 * 4);                                   \
        const float4 qv = *reinterpret_cast<const float4*>(qb + s4 * 4);                                   \
        acc = fmaf(qv.x, rv.x, acc);                                                                       \
        acc = fmaf(qv.y, rv.y, acc);                                                                       \
        acc = fmaf(qv.z, rv.z, acc);                                                                       \
        acc = fmaf(qv.w, rv.w, acc);                                                                       \
      }                                                                                                    \
    }
    // steady state (no conditions on the loads: the compiler's wait counts then leave the three younger steps in flight)
#define SV_WG_STEP_FULL(c_, XN)                                                                            \
    do {                                                                                                   \
      const int c = (c_);                                                                                  \
      SV_WG_MUL(c)                                                                                         \
      SV_WG_STORE(XN, (c + 1) & 1);                                                                        \
      SV_WG_LOAD(XN, c + 1 + DEPTH);                                                                       \
      __syncthreads();                                                                                     \
    } while (0)
#define SV_WG_STEP(c_, XN)                                                                                 \
    do {                                                                                                   \
      const int c = (c_);                                                                                  \
      if (c < nch) {                                                                                       \
        if (tid < cnt) {                                                                                   \
          const float* tr = tile + ((size_t)(c & 1) * ROWS + tid) * LDR;                                   \
          const float* qb = qs + (c & 1) * KC;                                                             \
          _Pragma("unroll 8") for (int s4 = 0; s4 < C4; ++s4) {                                            \
            const float4 rv = *reinterpret_cast<const float4*>(tr + s4 * 4);                               \
            const float4 qv = *reinterpret_cast<const float4*>(qb + s4 * 4);                               \
            acc = fmaf(qv.x, rv.x, acc);                                                                   \
            acc = fmaf(qv.y, rv.y, acc);                                                                   \
            acc = fmaf(qv.z, rv.z, acc);                                                                   \
            acc = fmaf(qv.w, rv.w, acc);                                                                   \
          }                                                                                                \
        }                                                                                                  \
        if (c + 1 < nch) {                                                                                 \
          SV_WG_STORE(XN, (c + 1) & 1);   /* waits for the loads of step c + 1 only */                     \
          if (c + 1 + DEPTH < nch) SV_WG_LOAD(XN, c + 1 + DEPTH);                                          \
        }                                                                                                  \
        __syncthreads();                                                                                   \
      }                                                                                                    \
    } while (0)
    float acc = 0.f;
    // set A holds steps 0, 4, 8, ...; B 1, 5, ...; C 2, 6, ...; D 3, 7, ...
    SV_WG_LOAD(A, 0);
    if (1 < nch) SV_WG_LOAD(B, 1);
    if (2 < nch) SV_WG_LOAD(C, 2);
    if (3 < nch) SV_WG_LOAD(D, 3);
    SV_WG_STORE(A, 0);
    if (4 < nch) SV_WG_LOAD(A, 4);
    __syncthreads();
    int c0 = 0;
    for (; c0 + 3 + 1 + DEPTH < nch; c0 += 4) {   // every load of these four steps exists
      SV_WG_STEP_FULL(c0, B);
      SV_WG_STEP_FULL(c0 + 1, C);
      SV_WG_STEP_FULL(c0 + 2, D);
      SV_WG_STEP_FULL(c0 + 3, A);
    }
    for (; c0 < nch; c0 += 4) {                    // the last steps: nothing (or not everything) left to request
      SV_WG_STEP(c0, B);
      SV_WG_STEP(c0 + 1, C);
      SV_WG_STEP(c0 + 2, D);
      SV_WG_STEP(c0 + 3, A);
    }
#undef SV_WG_STEP
#undef SV_WG_STEP_FULL
#undef SV_WG_MUL
#undef SV_WG_STORE
#undef SV_WG_LOAD
#undef SV_WG_DECL
    if (tid < cnt) {
      const uint32_t id = ids[tid];
      a[base + tid] = ((uint64_t)f2key_(sv_d2(q2, rn[id], acc)) << 32) | id;
    }
  }
  bitonic64(a, np2, tid);
  for (int j = tid; j < k; j += 256) {
    float dd = INFINITY;
    int64_t id = -1;
    if (j < n) {
      dd = key2f_((uint32_t)(a[j] >> 32));
      id = (int64_t)(uint32_t)a[j];
    }
    d2_out[row * k + j] = dd;
    idx_out[row * k + j] = id;
  }
}

// A handful of queries (one query image per pass): fewer lists than CUs, and a list walked by ONE workgroup is one memory
// round trip after the other (59 us for 240 rows of 1024 floats).  Here a list is dealt to `parts` workgroups, 32 rows each:
// a workgroup requests 1024 floats of each of its 32 rows in ONE burst (thread t: 16 bytes of row 2 j + (t >> 7), in both
// 512-float halves: 32 coalesced loads in flight per thread, one round trip per 1024 floats), parks one half at a time in an
// LDS tile [32][516], and 32 lanes walk one row each (the same sequential fp32 chain; the query's floats are LDS
// broadcasts).  The keys go to global memory as device-scope stores; the workgroup that takes the last ticket of its query
// reads them back, sorts them and writes the top k.  d % 1024 == 0.  tick[] is all zero before and after.
// Measured (50 queries x ~240 rows, 1 M x 1024 index): 59 us -> 34-38 us; what is left is a chain of ~7 dependent memory
// round trips (list length + ids, rows, key stores, ticket, key loads, results) around 5 us of arithmetic.
constexpr int SV_TICK_ROWS = 128, SV_TICK_POISON = SV_TICK_ROWS;   // tick[0..127]: one ticket counter per query; [128]: sticky failure word
__global__ __launch_bounds__(256) void refine_exact_small_kernel(const float* __restrict__ Q, const float* __restrict__ R, int d,
                                                                 const float* __restrict__ qn, const float* __restrict__ rn,
                                                                 const uint32_t* __restrict__ ref_cnt,
                                                                 const uint32_t* __restrict__ ref_id, int rcap, int rpad, int k,
                                                                 float* __restrict__ d2_out, int64_t* __restrict__ idx_out, int parts,
                                                                 uint64_t* __restrict__ gkeys, uint32_t* __restrict__ tick,
                                                                 uint32_t* __restrict__ fail_rows, uint32_t* __restrict__ fail_count,
                                                                 SvSmallFinish fz) {
  constexpr int ROWS = 32, KC = 512, LDR = KC + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tile = reinterpret_cast<float*>(smem);                  // [ROWS][LDR]
  uint64_t* a = reinterpret_cast<uint64_t*>(tile + ROWS * LDR);  // [rpad]
  __shared__ uint32_t ids[ROWS];
  __shared__ int last;
  const int tid = threadIdx.x;
#ifdef SV_REFINE_TIMING
  unsigned long long T[12];
  int ti = 0;
#define RTICK() T[ti++] = __builtin_amdgcn_s_memtime()
#else
#define RTICK()
#endif
  RTICK();
  // part-major: the first workgroups of the grid are part 0 of EVERY list, then part 1, ... -- the parts that hold rows (a band of ~270
  // rows: parts 0-8 of 16) are all resident in the first round of workgroups, the empty ones come last and leave at once.  Row-major
  // (rounds 3-5) put all 16 parts of lists 0-31 into the 512 resident slots and made lists 32-49 wait for them (round 6: 40 -> 33 us).
  const int nlists = (int)(gridDim.x / parts);
  const int64_t row = blockIdx.x % nlists;
  const int base = (int)(blockIdx.x / nlists) * ROWS;
  // the list's length and this workgroup's slice of it are requested together (the slice lies inside the list's rcap slots
  // whatever the length; entries beyond it are not looked at)
  const uint32_t idv = tid < ROWS ? ref_id[row * rcap + base + tid] : 0u;
  // fz.on (round 6, second step: the pass WITHOUT small_tail_kernel -- every kernel boundary of this chain costs 4-5 us, whatever the
  // kernel does): the rows the select flagged are finished HERE, by the row's own `parts` workgroups -- a band that outgrew the
  // first tier: every part evaluates its slice of the candidate list, the last one sorts; a row flagged for the redo: exact brute
  // force, every part a slice of the index, the last one merges (small_pass_dev.h; the same chain, sv_d2, (distance, id) order).
  // The two flags are requested with the list's length: no extra round trip on the common path.
  const uint32_t f_fail = fz.on ? fail_rows[row] : 0u, f_rovf = fz.on ? fz.rovf_rows[row] : 0u;
  const int n = (int)ref_cnt[row];
  if (f_fail | f_rovf) {   // (workgroup-uniform)
    const int p = (int)(blockIdx.x / nlists);
    uint64_t* a2 = reinterpret_cast<uint64_t*>(smem);                 // <= 8192 words of sort scratch
    float* qs2 = reinterpret_cast<float*>(smem + 65536);              // 1024 floats
    uint64_t* best2 = a2 + 2048;                                      // (brute force: the sort scratch is 2048 words)
    uint64_t* slot = fz.part2 + (size_t)row * fz.row_words;           // this row's words of the exchange buffer
    if (f_fail) sp_brute_slice(Q, R, qn, rn, fz.n_db, d, k, fz.kp, row, p, parts, slot, a2, best2, qs2, tid);
    else sp_tier2_slice(Q, R, qn, rn, d, row, p, parts, min(fz.cand_cnt[row], (uint32_t)fz.cap), fz.ref_lim[row], fz.cand_d2, fz.cand_id, fz.cap, slot,
                        qs2, tid);
    __threadfence();
    __syncthreads();
    if (tid == 0) last = (__hip_atomic_fetch_add(&tick[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(parts - 1));
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (f_fail) {
      sp_brute_merge(slot, k, fz.kp, parts, a2, best2, tid);
      for (int j = tid; j < k; j += 256) {
        const uint64_t v = best2[j];
        d2_out[row * k + j] = v != ~0ull ? key2f_((uint32_t)(v >> 32)) : INFINITY;
        idx_out[row * k + j] = v != ~0ull ? (int64_t)(uint32_t)v : -1;
      }
    } else {
      const int c = (int)min(fz.cand_cnt[row], (uint32_t)fz.cap);
      int np2 = 2;
      while (np2 < c) np2 <<= 1;
      for (int j = tid; j < np2; j += 256)
        a2[j] = j < c ? __hip_atomic_load(&slot[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      bitonic64(a2, np2, tid);
      for (int j = tid; j < k; j += 256) {
        const uint64_t v = j < np2 ? a2[j] : ~0ull;
        d2_out[row * k + j] = v != ~0ull ? key2f_((uint32_t)(v >> 32)) : INFINITY;
        idx_out[row * k + j] = v != ~0ull ? (int64_t)(uint32_t)v : -1;
      }
    }
    if (tid == 0) {
      __hip_atomic_store(&tick[row], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&fz.stats[f_fail ? 0 : 1], 1u);
      const uint32_t tot = atomicAdd(&fz.totals[f_fail ? 0 : 1], 1u) + 1u;
      if (fz.host_totals) fz.host_totals[f_fail ? 0 : 1] = tot;   // (the pinned mirror: the latest writer's total)
    }
    return;
  }
  const int cnt = min(ROWS, n - base);
  RTICK();   // T1: list length + ids
  if (cnt > 0) {
    if (tid < ROWS) ids[tid] = idv;
    __syncthreads();
    if (tid >= cnt && tid < ROWS) ids[tid] = ids[0];   // rows beyond the list re-read its first one
    __syncthreads();
    const int h = tid >> 7, off = (tid & 127) * 4;
    // (named registers: hipcc 7.2 sends a float4 g[..] filled in an unrolled loop to scratch memory here, with a vmcnt(0)
    //  behind every load)
#define SV_RS_J(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define SV_RS_SRC(j) const float* src##j = R + (size_t)ids[2 * j + h] * d + off;
#define SV_RS_LOAD(j)                                                       \
  const float4 gA##j = *reinterpret_cast<const float4*>(src##j + c0);        \
  const float4 gB##j = *reinterpret_cast<const float4*>(src##j + c0 + KC);
#define SV_RS_STORE_A(j) *reinterpret_cast<float4*>(tile + (2 * j + h) * LDR + off) = gA##j;
#define SV_RS_STORE_B(j) *reinterpret_cast<float4*>(tile + (2 * j + h) * LDR + off) = gB##j;
#define SV_RS_WALK(c_)                                                      \
  if (tid < cnt) {                                                          \
    const float* tr = tile + tid * LDR;                                     \
    const float* qb = qs + (c_);                                            \
    _Pragma("unroll 16") for (int s4 = 0; s4 < KC / 4; ++s4) {              \
      const float4 rv = *reinterpret_cast<const float4*>(tr + s4 * 4);      \
      const float4 qv = *reinterpret_cast<const float4*>(qb + s4 * 4);      \
      acc = fmaf(qv.x, rv.x, acc);                                          \
      acc = fmaf(qv.y, rv.y, acc);                                          \
      acc = fmaf(qv.z, rv.z, acc);                                          \
      acc = fmaf(qv.w, rv.w, acc);                                          \
    }                                                                       \
  }
    SV_RS_J(SV_RS_SRC)
    // (the query's 1024 floats of the step sit in LDS beside the tile, read as broadcasts: through the scalar cache every
    //  batch of 64 floats was a cold ~0.7 us miss in front of its fmas)
    float* qs = reinterpret_cast<float*>(a + rpad);   // [2 KC]
    const float* qsrc = Q + row * d + tid * 4;
    float acc = 0.f;
    for (int c0 = 0; c0 < d; c0 += 2 * KC) {
      SV_RS_J(SV_RS_LOAD)
      const float4 qv4 = *reinterpret_cast<const float4*>(qsrc + c0);
      if (c0) __syncthreads();   // the walkers are done with the previous half
      SV_RS_J(SV_RS_STORE_A)
      *reinterpret_cast<float4*>(qs + tid * 4) = qv4;
      __syncthreads();
      SV_RS_WALK(0)
      __syncthreads();
      SV_RS_J(SV_RS_STORE_B)
      __syncthreads();
      SV_RS_WALK(KC)
    }
#undef SV_RS_WALK
#undef SV_RS_STORE_B
#undef SV_RS_STORE_A
#undef SV_RS_LOAD
#undef SV_RS_SRC
#undef SV_RS_J
    RTICK();   // T2: rows loaded and walked
    if (tid < cnt) {
      const uint32_t id = ids[tid];
      __hip_atomic_store(&gkeys[row * rcap + base + tid], ((uint64_t)f2key_(sv_d2(qn[row], rn[id], acc)) << 32) | id, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // The keys are device-scope (write-through) stores and device-scope loads; each wave waits for its stores to be
  // acknowledged before the barrier that precedes the ticket.  (A __threadfence() on either side is an L2 write-back +
  // invalidate on this eight-L2 part.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) last = (__hip_atomic_fetch_add(&tick[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)(parts - 1));
  __syncthreads();
  RTICK();   // T3: key stores acknowledged + ticket
  if (!last) return;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  // Ordering.  Every access to gkeys / tick is an agent-scope atomic (sc1: performed at the memory side, never served from
  // an XCD's own L2), the writers wait for their key stores to be ACKNOWLEDGED (vmcnt(0)) before the barrier in front of the
  // ticket, and the reader issues its loads after its ticket returned: on this hardware that is a release / acquire chain
  // through the ticket.  The C++ model does not promise it for relaxed atomics, and a formal acq_rel ticket costs an L2
  // write-back + invalidate per workgroup (see above) -- so the protocol is CHECKED instead of trusted: gkeys holds all ones
  // wherever no key of this launch has landed (the launcher fills a new buffer so, the reader puts the fill back behind
  // every key it takes; a key is never all ones: finite distance, 32-bit id); a slot still all ones is re-read a bounded
  // number of times, and a slot that never fills FLAGS its query (fail_rows: the caller redoes flagged rows on another
  // path, exactly) and raises the sticky word tick[SV_TICK_POISON], on which the host re-initialises both buffers before
  // their next use (a key landing after the reader gave up would otherwise pass for a key of the next launch).
  int holes = 0;
  for (int j = tid; j < np2; j += 256) {
    uint64_t v = ~0ull;
    if (j < n) {
      for (int spin = 0; spin < 4096; ++spin) {
        v = __hip_atomic_load(&gkeys[row * rcap + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != ~0ull) break;
      }
      if (v == ~0ull) ++holes;
      __hip_atomic_store(&gkeys[row * rcap + j], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a[j] = v;
  }
  if (tid == 0) __hip_atomic_store(&tick[row], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (fz.on && (fz.debug & 8) && row == 1) holes = 1;   // (tests: the path below has never been taken by the hardware)
  if (__syncthreads_or(holes)) {   // (never observed)
    if (tid == 0) __hip_atomic_store(&tick[SV_TICK_POISON], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!fz.on) {   // the row's output is left to the redo (the caller's read-back, or small_tail_kernel)
      if (tid == 0 && fail_rows && atomicExch(&fail_rows[row], 1u) == 0u) atomicAdd(fail_count, 1u);
      return;
    }
    // fused finish: nobody comes after this kernel -- this workgroup re-evaluates the row's whole band itself (<= rcap rows, a thread
    // per row: the same chain); the sticky word makes the NEXT pass's head refill the hand-over buffers (a key that lands late must
    // not pass for a key of that pass)
    float* qs2 = reinterpret_cast<float*>(smem);   // (the row tile is free: every part has drawn its ticket)
    for (int j0 = 0; j0 < n; j0 += 256) {
      const int j = j0 + tid;
      const uint32_t id = ref_id[row * rcap + min(j, n - 1)];
      float acc2[1] = {0.f};
      for (int c0 = 0; c0 < d; c0 += ST_KC) {
        const int kc = min(ST_KC, d - c0);
        __syncthreads();
        for (int t = tid; t < kc; t += 256) qs2[t] = Q[row * d + c0 + t];
        __syncthreads();
        chain_step<1>(R + (size_t)id * d + c0, kc, qs2, acc2);
      }
      if (j < n) a[j] = ((uint64_t)f2key_(sv_d2(qn[row], rn[id], acc2[0])) << 32) | id;
    }
    __syncthreads();
    if (tid == 0) atomicAdd(&fz.stats[2], 1u);
  }
  RTICK();   // T4: keys read back
  // (distance, id) order WITHOUT a sort: the keys are distinct (the id is part of them), so a key's place in the sorted list is the
  // number of smaller keys -- every thread counts that for its own one or two keys against the n keys in LDS (broadcast reads, no
  // barrier) and writes its result straight to that place.  The bitonic sort of 512 words was 45 barrier-separated stages: 9.6 us of
  // the last workgroup's 25 (round 6); the count is ~2.
  for (int j = tid; j < n; j += 256) {
    const uint64_t v = a[j];
    int place = 0;
#pragma unroll 8
    for (int t = 0; t < n; ++t) place += (a[t] < v) ? 1 : 0;
    if (place < k) {
      d2_out[row * k + place] = key2f_((uint32_t)(v >> 32));
      idx_out[row * k + place] = (int64_t)(uint32_t)v;
    }
  }
  for (int j = n + tid; j < k; j += 256) {   // a list shorter than k pads with (inf, -1)
    d2_out[row * k + j] = INFINITY;
    idx_out[row * k + j] = -1;
  }
  RTICK();   // T5: placed
#ifdef SV_REFINE_TIMING
  if (tid == 0 && row == 0 && cnt > 0)
    printf("refine row0 last wg: ids %llu rows+walk %llu store+ticket %llu readback %llu sort %llu cycles (n=%d)\n", T[1]-T[0], T[2]-T[1], T[3]-T[2], T[4]-T[3], T[5]-T[4], n);
#endif
#undef RTICK
}

int sv_refine_small_repair(segvlad_ctx* ctx) {
  // the hand-over buffers of refine_exact_small_kernel back to their initial state (all ones / all zero)
  if (ctx->s_ref_tick.p) SV_HIP(hipMemsetAsync(ctx->s_ref_tick.p, 0, ctx->s_ref_tick.cap, ctx->stream));
  if (ctx->s_ref_keys.p) SV_HIP(hipMemsetAsync(ctx->s_ref_keys.p, 0xff, ctx->s_ref_keys.cap, ctx->stream));
  return SEGVLAD_OK;
}

int sv_launch_refine_exact(segvlad_ctx* ctx, const float* Q, const float* R, int nq, int d, const float* qn, const float* rn,
                           const uint32_t* ref_cnt, const uint32_t* ref_id, int rcap, int k, float* d2_out, int64_t* idx_out,
                           const uint32_t* only_rows, uint32_t* fail_rows, uint32_t* fail_count, const uint32_t** poison_dev,
                           const SvSmallFinish* fz, bool* fused_done) {
  if (poison_dev) *poison_dev = nullptr;
  if (fused_done) *fused_done = false;
  if (nq <= 0) return SEGVLAD_OK;
  int rpad = 2;
  while (rpad < rcap) rpad <<= 1;
  size_t lds = (size_t)d * 4 + (size_t)rpad * 8;
  if (nq <= SV_TICK_ROWS && d % 1024 == 0 && rcap <= 1024 && rcap % 32 == 0 && !only_rows) {   // one query image: lists shared by workgroups
    const int parts = rcap / 32;
    const size_t tick_cap = ctx->s_ref_tick.cap;
    SV_HIP(ctx->s_ref_tick.reserve((size_t)(SV_TICK_ROWS + 1) * 4));
    if (ctx->s_ref_tick.cap != tick_cap) SV_HIP(hipMemsetAsync(ctx->s_ref_tick.p, 0, ctx->s_ref_tick.cap, ctx->stream));
    const size_t keys_cap = ctx->s_ref_keys.cap;
    SV_HIP(ctx->s_ref_keys.reserve((size_t)nq * rcap * 8));
    if (ctx->s_ref_keys.cap != keys_cap) SV_HIP(hipMemsetAsync(ctx->s_ref_keys.p, 0xff, ctx->s_ref_keys.cap, ctx->stream));
    lds = (size_t)(32 * 516 + 1024) * 4 + (size_t)rpad * 8;
    if (lds > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(refine_exact_small_kernel), lds));
    SvSmallFinish f;   // (off)
    if (fz && fz->on && fail_rows && fused_done && k <= 1024) {
      // the flagged rows' exchange words: a row's `parts` lists of kp keys (brute force) or its `cap` candidate keys (second tier)
      f = *fz;
      f.kp = 256;
      while (f.kp < k) f.kp <<= 1;
      f.row_words = std::max<int64_t>((int64_t)parts * f.kp, f.cap);
      SV_HIP(ctx->s_tail_part.reserve((size_t)nq * f.row_words * 8));
      SV_TRY(sv_small_words(ctx));
      SV_TRY(sv_ensure_pinned_words(ctx));
      f.part2 = ctx->s_tail_part.as<uint64_t>();
      f.totals = ctx->s_tail_tick.as<uint32_t>() + 129;
      f.host_totals = ctx->h_pin + 8;
      f.debug = ctx->opt.debug_small_tail;
      *fused_done = true;
    }
    hipLaunchKernelGGL(refine_exact_small_kernel, dim3(nq * parts), dim3(256), lds, ctx->stream, Q, R, d, qn, rn, ref_cnt, ref_id, rcap,
                       rpad, k, d2_out, idx_out, parts, ctx->s_ref_keys.as<uint64_t>(), ctx->s_ref_tick.as<uint32_t>(), fail_rows, fail_count, f);
    SV_HIP(hipGetLastError());
    if (poison_dev) *poison_dev = ctx->s_ref_tick.as<uint32_t>() + SV_TICK_POISON;
    return SEGVLAD_OK;
  }
#define SV_REFINE_ARGS dim3(nq), dim3(256), lds, ctx->stream, Q, R, d, qn, rn, ref_cnt, ref_id, rcap, rpad, k, d2_out, idx_out, only_rows
  if (lds <= 160 * 1024) {
    if (lds > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(refine_exact_kernel<true>), lds));
    hipLaunchKernelGGL(refine_exact_kernel<true>, SV_REFINE_ARGS);
  } else if (d % 128 == 0) {
    // [2][64 rows][132] floats + [2][128] query floats + [64] ids + the sort keys (<= 64 KiB for a second-tier list)
    lds = (size_t)(2 * 64 * 132 + 2 * 128 + 64) * 4 + (size_t)rpad * 8;
    if (lds > 160 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "refine: a %d-entry list of %d-d rows exceeds the LDS", rcap, d);
    auto kern = refine_exact_wide_kernel<128, 4>;
    if (lds > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, SV_REFINE_ARGS);
  } else {
    lds = (size_t)rpad * 8;
    if (lds > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(refine_exact_kernel<false>), lds));
    hipLaunchKernelGGL(refine_exact_kernel<false>, SV_REFINE_ARGS);
  }
#undef SV_REFINE_ARGS
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ---- max of the database row norms (margin scale) -------------------------------------------------------------
__global__ __launch_bounds__(256) void max_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) m = max(m, f2key_(x[j]));
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

__global__ __launch_bounds__(256) void min_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
  uint32_t m = 0xffffffffu;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) m = min(m, f2key_(x[j]));
  for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) atomicMin(out, m);
}

// smallest of n non-negative values (the query batch's smallest squared norm), read back to the host (synchronises)
int sv_row_norm_min(segvlad_ctx* ctx, const float* norms, int64_t n, float* out_host) {
  SV_HIP(ctx->s_minmax.reserve(32));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>() + 6;
  SV_HIP(hipMemsetAsync(mm, 0xff, 4, ctx->stream));
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (n > 0) hipLaunchKernelGGL(min_kernel, dim3(blocks), dim3(256), 0, ctx->stream, norms, n, mm);
  uint32_t key = 0;
  SV_HIP(hipMemcpyAsync(&key, mm, 4, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
  float f;
  memcpy(&f, &u, 4);
  *out_host = (n > 0 && key != 0xffffffffu) ? f : 0.f;
  return SEGVLAD_OK;
}

// max |x| of a block and the smallest of a list of row norms behind ONE read-back (a batch search needs both before its first
// filter launch: two synchronisations in a row otherwise)
int sv_maxabs_and_norm_min(segvlad_ctx* ctx, const float* x, int64_t n, const float* norms, int64_t n_norms, float* maxabs_host,
                           float* norm_min_host) {
  SV_HIP(ctx->s_minmax.reserve(32));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>() + 4;   // [4] = max |x| bits (init 0), [5] = min key (init all ones)
  static const uint32_t init[2] = {0u, 0xffffffffu};
  SV_HIP(hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, ctx->stream));
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (n > 0) hipLaunchKernelGGL(maxabs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, n, mm);
  int nb = (int)((n_norms + 255) / 256);
  if (nb > 1024) nb = 1024;
  if (n_norms > 0) hipLaunchKernelGGL(min_kernel, dim3(nb), dim3(256), 0, ctx->stream, norms, n_norms, mm + 1);
  uint32_t h[2] = {0u, 0u};
  SV_HIP(hipMemcpyAsync(h, mm, 8, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(maxabs_host, &h[0], 4);
  const uint32_t u = (h[1] & 0x80000000u) ? (h[1] & 0x7fffffffu) : ~h[1];
  float f;
  memcpy(&f, &u, 4);
  *norm_min_host = (n_norms > 0 && h[1] != 0xffffffffu) ? f : 0.f;
  return SEGVLAD_OK;
}

int sv_ensure_pinned_words(segvlad_ctx* ctx) {
  if (!ctx->h_pin) {
    SV_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pin), 64, hipHostMallocDefault));
    for (int j = 0; j < 16; ++j) ctx->h_pin[j] = 0u;
    SV_HIP(hipEventCreateWithFlags(&ctx->ev_scalars, hipEventDisableTiming));
  }
  return SEGVLAD_OK;
}

// The same two scalars WITHOUT the wait in between: _begin enqueues the reductions and the copy into pinned words of the
// context and records an event behind them; _end blocks on that event only.  What the caller enqueues between the two runs on
// the device while the host waits.
int sv_maxabs_and_norm_min_begin(segvlad_ctx* ctx, const float* x, int64_t n, const float* norms, int64_t n_norms) {
  SV_HIP(ctx->s_minmax.reserve(32));
  SV_TRY(sv_ensure_pinned_words(ctx));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>() + 4;   // [4] = max |x| bits (init 0), [5] = min key (init all ones)
  static const uint32_t init[2] = {0u, 0xffffffffu};
  SV_HIP(hipMemcpyAsync(mm, init, 8, hipMemcpyHostToDevice, ctx->stream));
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  if (n > 0) hipLaunchKernelGGL(maxabs_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x, n, mm);
  int nb = (int)((n_norms + 255) / 256);
  if (nb > 1024) nb = 1024;
  if (n_norms > 0) hipLaunchKernelGGL(min_kernel, dim3(nb), dim3(256), 0, ctx->stream, norms, n_norms, mm + 1);
  SV_HIP(hipGetLastError());
  SV_HIP(hipMemcpyAsync(ctx->h_pin, mm, 8, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipEventRecord(ctx->ev_scalars, ctx->stream));
  return SEGVLAD_OK;
}

int sv_maxabs_and_norm_min_end(segvlad_ctx* ctx, int64_t n, int64_t n_norms, float* maxabs_host, float* norm_min_host) {
  SV_HIP(hipEventSynchronize(ctx->ev_scalars));
  const uint32_t h0 = ctx->h_pin[0], h1 = ctx->h_pin[1];
  memcpy(maxabs_host, &h0, 4);
  if (n <= 0) *maxabs_host = 0.f;
  const uint32_t u = (h1 & 0x80000000u) ? (h1 & 0x7fffffffu) : ~h1;
  float f;
  memcpy(&f, &u, 4);
  *norm_min_host = (n_norms > 0 && h1 != 0xffffffffu) ? f : 0.f;
  return SEGVLAD_OK;
}

int sv_row_norm_max(segvlad_ctx* ctx, const float* norms, int64_t n, float* out_host) {
  SV_HIP(ctx->s_minmax.reserve(16));
  uint32_t* mm = ctx->s_minmax.as<uint32_t>() + 2;
  SV_HIP(hipMemsetAsync(mm, 0, 4, ctx->stream));
  int blocks = (int)((n + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  if (n > 0) hipLaunchKernelGGL(max_kernel, dim3(blocks), dim3(256), 0, ctx->stream, norms, n, mm);
  uint32_t key = 0;
  SV_HIP(hipMemcpyAsync(&key, mm, 4, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
  float f;
  memcpy(&f, &u, 4);
  *out_host = (n > 0 && key != 0) ? f : 0.f;
  return SEGVLAD_OK;
}
