// PCA projection on the 16-bit matrix pipe with fp32-class accuracy (gfx950).
//
// fp32 MFMA runs at 1/16 of the fp16 rate.  Each fp32 operand is scaled by a power of two and split into TWO
// fp16 values, x*s = h1 + h2 + e with |e| <= 2^-22 |x*s| (two 11-bit significands), and the product is
// accumulated as h2.g1 + h1.g2 + h1.g1 on v_mfma_f32_16x16x32_f16 (fp16 x fp16 products are exact in fp32, the
// accumulator is fp32; rounds 1-3 used the 32 x 32 x 16 shape: the chip is power limited on these kernels and the 16 x 16 x 32
// shape needs ~12 % less energy per flop -- tools/ubench/mfma_peak, knn_f16_filter_kernel).  The dropped terms are <= 3*2^-22 ||a|| ||b||: BELOW the fp32 chain's own accumulation
// error for K = 98304 (sqrt(K) 2^-24 typical), i.e. the result is at least as accurate as the fp32-MFMA GEMM it
// replaces, at 3/16 of the matrix-pipe time.  (The kNN filter can afford a single product because it only needs
// a rigorous bound; here the value itself is the output, hence the two-term split.)
//
//   split_f16x2_kernel   (X - sub) * scale -> h1, h2 planes in the blocked layout of ctx.h (sv_x3_off); sub = PCA mean on
//                        the A side, none on the W side
//   gemm_f16x3_kernel    C = (A1+A2).(B1+B2)^T * col_scale: BM x BN x 32 tiles, global->LDS DMA with source-side
//                        swizzle (see knn_filter_kernels.hip), two stages per operand, optional split-K
#include <stdlib.h>

#include "ctx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void split_f16x2_kernel(const float* __restrict__ X, int64_t n_rows, int d,
                                                          const float* __restrict__ sub, float scale,
                                                          _Float16* __restrict__ h1, _Float16* __restrict__ h2) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const int64_t n4 = n_rows * (int64_t)(d >> 2);
  const int d4 = d >> 2;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(X)[j];
    if (sub) {
      const float4 s = reinterpret_cast<const float4*>(sub)[j % d4];
      v.x -= s.x;
      v.y -= s.y;
      v.z -= s.z;
      v.w -= s.w;
    }
    const float f[4] = {v.x * scale, v.y * scale, v.z * scale, v.w * scale};
    h4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = (_Float16)f[e];
      b[e] = (_Float16)(f[e] - (float)a[e]);
    }
    const size_t o = sv_x3_off(j / d4, (j % d4) * 4, d);   // blocked plane layout (ctx.h)
    *reinterpret_cast<h4*>(h1 + o) = a;
    *reinterpret_cast<h4*>(h2 + o) = b;
  }
}

int sv_launch_split_f16x2(segvlad_ctx* ctx, const float* X, int64_t n_rows, int d, const float* sub, float scale, uint16_t* h1,
                          uint16_t* h2) {
  if (n_rows <= 0) return SEGVLAD_OK;
  int64_t blocks = (n_rows * (d / 4) + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n_rows, d, sub, scale,
                     reinterpret_cast<_Float16*>(h1), reinterpret_cast<_Float16*>(h2));
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

template <int N_>
__device__ __forceinline__ void wait_vm_lgkm0_() {
  if (N_ == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if (N_ == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else if (N_ == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
}

// C[M][N] (+ split-K partials) = sum_k (A1+A2)[m][k] (B1+B2)[n][k], k in this block's slice.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f16x3_kernel(const uint16_t* __restrict__ A1,
                                                                  const uint16_t* __restrict__ A2,
                                                                  const uint16_t* __restrict__ B1,
                                                                  const uint16_t* __restrict__ B2, int M, int N, int Kd,
                                                                  int tiles_m, int gm, int n_splits, int k_per_split, float out_scale,
                                                                  const float* __restrict__ col_scale,
                                                                  float* __restrict__ C, int64_t ldc,
                                                                  const int32_t* __restrict__ tile_group, int nkb_b) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);   // 16 x 16 MFMA tiles per wave (4 x 8 for the 64 x 128 wave tile)
  static_assert(TM == 4 && (TN == 8 || TN == 4), "wave tiles of 64 rows, 128 or 64 columns");
  constexpr int HBK = 32, RB = 64, RP = 16;           // 64-B rows per plane and k-tile, 16 rows per 1-KiB DMA piece
  constexpr int PA = BM * RB, PB = BN * RB;            // one plane
  constexpr int JA = BM / RP / NW, JB = BN / RP / NW;  // DMA pieces per wave and PLANE
  static_assert(JA * RP * NW == BM && JB * RP * NW == BN, "tile/wave geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // XCD-aware order (see knn_f16_filter_kernel): the 32 workgroups one XCD runs side by side are a gm x (32/gm) block
  // of output tiles of ONE k-split, sharing their A and B slices in that XCD's L2
  int tm, tn, split;
  if (gm > 0) {
    const int b = blockIdx.x, xcd = b & 7, s = b >> 3, within = s & 31, u = (s >> 5) * 8 + xcd;
    const int gn = 32 / gm, sm_cnt = (tiles_m + gm - 1) / gm, tiles_n = (N + BN - 1) / BN;
    split = u % n_splits;
    const int r = u / n_splits;
    tm = (r % sm_cnt) * gm + within % gm;
    tn = (r / sm_cnt) * gn + within / gm;
    if (tm >= tiles_m || tn >= tiles_n) return;
  } else if (tile_group) {
    // grouped, XCD-aware: workgroup b runs on XCD b % 8.  The 32 workgroups an XCD runs side by side are the tiles_n column
    // tiles of 32 / tiles_n CONSECUTIVE row tiles ("chunk"; nearly always one cluster): every A row block is fetched into
    // ONE L2 and used by its tiles_n column tiles there, and the chunk's row tiles stream the same slice of B together.
    // (plain tn-fastest order: A read by tiles_n XCDs, 10 GB of HBM traffic per launch at the bench shape instead of ~3)
    const int tiles_n = (N + BN - 1) / BN;
    const int b = blockIdx.x, xcd = b & 7, s = b >> 3;
    int rows_per_chunk = 32 / tiles_n;
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    const int per = rows_per_chunk * tiles_n;          // workgroups of one chunk
    const int chunk = (s / per) * 8 + xcd, within = s % per;
    tn = within % tiles_n;
    tm = chunk * rows_per_chunk + within / tiles_n;
    split = 0;
    if (tm >= tiles_m) return;
  } else {
    tm = blockIdx.x % tiles_m;
    tn = blockIdx.x / tiles_m;
    split = blockIdx.y;
  }
  // GROUPED mode (tile_group != null; the token projection of segvlad_images_pca): the rows of A are grouped, every
  // BM-row tile belongs to ONE group g = tile_group[tm] (-1: unused tile) and is multiplied with k-blocks
  // [g * Kd/32, (g+1) * Kd/32) of B, whose row blocks hold nkb_b k-blocks: C[rows of g] = A_g . B[:, g-th column slice]^T
  int group = 0;
  if (tile_group) {
    group = tile_group[tm];
    if (group < 0) return;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int tid = threadIdx.x, l = tid & 63, i = l & 15, kq = l >> 4;   // lane: row / column i of a 16-wide tile, k-chunk kq
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WN, wn = w % WN;
  const int kbeg = split * k_per_split;
  const int kend = (kbeg + k_per_split < Kd) ? kbeg + k_per_split : Kd;
  const int ntiles = (kend - kbeg) / HBK;
  auto swz = [](int r, int c) { return c ^ sv_x3_swz(r); };

  f32x4 acc[TM][TN];   // element j of tile (mt, nt): row mt*16 + 4*kq + j, column nt*16 + i
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

  // Element offsets of this lane's 16-B chunk in the blocked planes (ctx.h: sv_x3_off): piece p of a tile = rows
  // 16 p .. 16 p + 15, which are 1 KiB of contiguous memory inside their (128-row, 32-k) block, already in the swizzled
  // order of the LDS image -> the DMA is a linear copy (lane l fetches bytes 16 l .. 16 l + 15 of the piece).  The
  // planes are padded to whole 256-row tiles, so edge tiles need no clamping.
  const size_t nkb = (size_t)(Kd >> 5);
  size_t offA[JA], offB[JB];
#pragma unroll
  for (int j = 0; j < JA; ++j) {
    const int64_t r0 = m0 + (w * JA + j) * RP;
    offA[j] = ((size_t)(r0 >> 7) * nkb + (size_t)(kbeg >> 5)) * 4096 + (size_t)(r0 & 127) * 32 + (size_t)l * 8;
  }
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int64_t r0 = n0 + (w * JB + j) * RP;
    offB[j] = ((size_t)(r0 >> 7) * (tile_group ? (size_t)nkb_b : nkb) + (size_t)(kbeg >> 5) + (size_t)group * nkb) * 4096 +
              (size_t)(r0 & 127) * 32 + (size_t)l * 8;
  }
  // LDS: stage s = [A1 | A2 | B1 | B2], two stages
  constexpr int STAGE = 2 * PA + 2 * PB;
  auto dma_tile = [&](int kt, int buf) {
    unsigned char* S = lds + buf * STAGE;
    const size_t k0 = (size_t)kt * 4096;   // one 32-deep k-block further: 4096 elements inside the 128-row block
#pragma unroll
    for (int j = 0; j < JA; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)(A1 + offA[j] + k0), (lptr_t)(S + (w * JA + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(A2 + offA[j] + k0), (lptr_t)(S + PA + (w * JA + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      __builtin_amdgcn_global_load_lds((gptr_t)(B1 + offB[j] + k0), (lptr_t)(S + 2 * PA + (w * JB + j) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(B2 + offB[j] + k0), (lptr_t)(S + 2 * PA + PB + (w * JB + j) * 1024), 16, 0, 0);
    }
  };
  dma_tile(0, 0);
  wait_vm_lgkm0_<0>();
  __builtin_amdgcn_s_barrier();
  int cur = 0;
  const int fa0 = wm * (16 * TM) + i, fb0 = wn * (16 * TN) + i;
  for (int kt = 0; kt < ntiles; ++kt) {
    if (kt + 1 < ntiles) dma_tile(kt + 1, cur ^ 1);  // lands while tile kt is multiplied
    const unsigned char* S = lds + cur * STAGE;
    // the k-tile of 32 is ONE k-step of the 16 x 16 x 32 MFMA: chunk kq of every row
    f16x8 a1[TM], a2[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int ra = fa0 + 16 * t;
      a1[t] = *reinterpret_cast<const f16x8*>(S + ra * RB + swz(ra, kq) * 16);
      a2[t] = *reinterpret_cast<const f16x8*>(S + PA + ra * RB + swz(ra, kq) * 16);
    }
#pragma unroll
    for (int h = 0; h < TN / 4; ++h) {   // four column tiles at a time (16 fragment registers per plane)
      f16x8 b1[4], b2[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int rb = fb0 + 16 * (4 * h + t);
        b1[t] = *reinterpret_cast<const f16x8*>(S + 2 * PA + rb * RB + swz(rb, kq) * 16);
        b2[t] = *reinterpret_cast<const f16x8*>(S + 2 * PA + PB + rb * RB + swz(rb, kq) * 16);
      }
      // small terms first; the accumulators are independent, so consecutive MFMAs never wait on each other
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][4 * h + nt] = MFMA_F16(a2[mt], b1[nt], acc[mt][4 * h + nt]);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][4 * h + nt] = MFMA_F16(a1[mt], b2[nt], acc[mt][4 * h + nt]);
#pragma unroll
      for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][4 * h + nt] = MFMA_F16(a1[mt], b1[nt], acc[mt][4 * h + nt]);
    }
    wait_vm_lgkm0_<0>();
    __builtin_amdgcn_s_barrier();
    cur ^= 1;
  }

  if (n_splits > 1) C += (int64_t)split * M * ldc;
#pragma unroll
  for (int nt = 0; nt < TN; ++nt) {
    const int64_t col = n0 + wn * (16 * TN) + nt * 16 + i;
    if (col >= N) continue;
    const float cs = (n_splits > 1) ? 1.f : out_scale * (col_scale ? col_scale[col] : 1.f);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * (16 * TM) + mt * 16 + 4 * kq + r;
        if (row < M) C[row * ldc + col] = acc[mt][nt][r] * cs;
      }
  }
}

// out = (sum_s part[s]) * out_scale * col_scale, slices added in index order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_scale_kernel(const float* __restrict__ part, int splits, int64_t mn,
                                                                  int N, float out_scale,
                                                                  const float* __restrict__ col_scale,
                                                                  float* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= mn) return;
  float s = part[j];
  for (int t = 1; t < splits; ++t) s += part[(int64_t)t * mn + j];
  const int col = (int)(j % N);
  out[j] = s * out_scale * (col_scale ? col_scale[col] : 1.f);
}

template <int BM, int BN, int WM, int WN>
static int launch_x3(segvlad_ctx* ctx, const uint16_t* A1, const uint16_t* A2, const uint16_t* B1, const uint16_t* B2, int M,
                     int N, int Kd, float out_scale, const float* col_scale, float* C) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  // split-K so that >= ~2 waves of workgroups fill the chip evenly
  const int slots = 256 * ((BM == 256) ? 1 : 2);
  // split-K count: the launch runs in ceil(tiles * s / slots) rounds of K / s each; take the s that minimises
  // rounds / s (a full last round) plus a small charge per slice for the partial-sum traffic
  int splits = 1;
  {
    double best = 1e30;
    for (int sct = 1; sct <= 32; ++sct) {
      const int64_t rounds = (tiles * sct + slots - 1) / slots;
      const double cost = (double)rounds / sct + 0.004 * sct * (double)tiles / slots;
      if (cost < best - 1e-12) {
        best = cost;
        splits = sct;
      }
    }
  }
  int k_per_split = (((Kd + splits - 1) / splits) + 31) / 32 * 32;
  splits = (Kd + k_per_split - 1) / k_per_split;
  float* dst = C;
  if (splits > 1) {
    SV_HIP(ctx->s_dist.reserve((size_t)splits * M * N * sizeof(float)));
    dst = ctx->s_dist.as<float>();
  }
  const size_t lds = 2 * (size_t)(2 * BM * 64 + 2 * BN * 64);
  auto kern = gemm_f16x3_kernel<BM, BN, WM, WN>;
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
  // tile-block height of the XCD-aware order (0 = plain order, k-split in grid.y); measured: the plain order is ~8 %
  // faster here (probe_pca.py)
  int gm = ctx->opt.x3_gm > 0 ? ctx->opt.x3_gm : 0;
  dim3 grid((unsigned)tiles, (unsigned)splits);
  if (gm > 0) {
    gm = gm >= 32 ? 32 : gm >= 16 ? 16 : gm >= 8 ? 8 : gm >= 4 ? 4 : gm >= 2 ? 2 : 1;
    while (gm < 32 && 32 / gm > tiles_n) gm <<= 1;   // no wider than the output
    while (gm > 1 && gm / 2 >= tiles_m) gm >>= 1;
    const int gn = 32 / gm;
    const int64_t units = (int64_t)((tiles_m + gm - 1) / gm) * ((tiles_n + gn - 1) / gn) * splits;
    grid = dim3((unsigned)((units + 7) / 8 * 8 * 32), 1);
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), lds, ctx->stream, A1, A2, B1, B2, M, N, Kd, tiles_m, gm, splits, k_per_split,
                     out_scale, col_scale, dst, (int64_t)N, (const int32_t*)nullptr, 0);
  SV_HIP(hipGetLastError());
  if (splits > 1) {
    const int64_t mn = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_scale_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, ctx->stream, dst, splits, mn,
                       N, out_scale, col_scale, C);
    SV_HIP(hipGetLastError());
  }
  return SEGVLAD_OK;
}

// C[M][N] = ((A1+A2) . (B1+B2)^T) * out_scale * col_scale[n]; Kd % 32 == 0
int sv_launch_gemm_f16x3(segvlad_ctx* ctx, const uint16_t* A1, const uint16_t* A2, const uint16_t* B1, const uint16_t* B2, int M,
                         int N, int Kd, float out_scale, const float* col_scale, float* C) {
  if (M <= 0 || N <= 0) return SEGVLAD_OK;
  const bool big = ctx->opt.x3_tile ? (ctx->opt.x3_tile == 256) : (M >= 1024);  // 256x256 tiles: 8.1 vs 12.6 ms at 10000 x 98304 x 1024
  if (big) return launch_x3<256, 256, 4, 2>(ctx, A1, A2, B1, B2, M, N, Kd, out_scale, col_scale, C);
  return launch_x3<128, 128, 2, 2>(ctx, A1, A2, B1, B2, M, N, Kd, out_scale, col_scale, C);
}

// Grouped projection (segvlad_images_pca, "project then aggregate"): Z[rows of group g][N] = A_g . B[:, g*Kd .. (g+1)*Kd)^T for
// every group g, in ONE launch.  A planes: [M_pad][Kd] blocked, rows grouped, every 256-row tile inside one group
// (tile_group[tile], -1 = unused); B planes: [N][groups * Kd] blocked (the PCA components: group g = cluster g's columns).
int sv_launch_gemm_f16x3_grouped(segvlad_ctx* ctx, const uint16_t* A1, const uint16_t* A2, const uint16_t* B1, const uint16_t* B2,
                                 int M_pad, int N, int Kd, int n_groups, const int32_t* tile_group, float out_scale, float* C) {
  if (M_pad <= 0 || N <= 0) return SEGVLAD_OK;
  constexpr int BM = 256, BN = 256;
  const int tiles_m = M_pad / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * (size_t)(2 * BM * 64 + 2 * BN * 64);
  auto kern = gemm_f16x3_kernel<BM, BN, 4, 2>;
  SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
  // XCD-aware order (see the kernel): chunks of 32 / tiles_n row tiles, chunk c on XCD c % 8
  const int rows_per_chunk = 32 / tiles_n > 0 ? 32 / tiles_n : 1;
  const int chunks = (tiles_m + rows_per_chunk - 1) / rows_per_chunk;
  const int64_t grid = (int64_t)((chunks + 7) / 8) * 8 * rows_per_chunk * tiles_n;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, 1), dim3(512), lds, ctx->stream, A1, A2, B1, B2, M_pad, N, Kd, tiles_m, 0,
                     1, Kd, out_scale, (const float*)nullptr, C, (int64_t)N, tile_group, n_groups * (Kd >> 5));
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
