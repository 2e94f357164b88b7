// "Project then aggregate": the fused segment-VLAD -> PCA call (segvlad_images_pca without the descriptor output)
// reformulated so that the K*D-wide descriptor is never formed (gfx950).
//
// The reference computes, per segment s, the blocks V_sk = sum_{t in cluster k, t covered by s} r_t with r_t = x^_t - C_k(t),
// normalises each block and the whole vector, and projects: y_s = (g_s * concat_k(V_sk / ||V_sk||) - mu) W^T / sqrt(lambda)
// (func_vpr.py:1181-1210, 1419-1443).  The projection is linear, so with W_k = W[:, kD:(k+1)D], z_t = W_k(t) r_t and
// a_sk = g_s / ||V_sk||:
//
//     y_s = ( sum_t [t covered by s] a_{s,k(t)} z_t  -  W mu ) / sqrt(lambda)
//
// i.e. every TOKEN's residual is projected once with its cluster's slice of the components -- 2 N D P flops per image
// instead of 2 S K D P for the descriptor (1530 x 1536 instead of 50 x 98 304 rows x columns: 2.1 x fewer) -- and the
// segments are aggregated in the P-dimensional space.  Only the block norms ||V_sk|| still come from the D-space
// (token_norms_kernel, vlad_kernels.hip, which also emits the fp16 planes of the residuals, grouped by cluster).
// Projecting the RESIDUAL (not the token, with the centre term subtracted afterwards) keeps the sums free of cancellation:
// the result is as accurate as projecting the finished descriptor.
//
//   group_plan_kernel          lab_off [B][K+1] -> rowbase [B][K] (grouped row of (image, cluster)), tile_group [tiles]
//   project_consts_kernel      W mu ([P])                                         (once per PCA model)
//   project_aggregate_kernel   the weighted sums above on fp32 MFMA (exact fp32 chains), mean term and whitening scale fused
#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ int pj_frag_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

// ---- grouping plan ---------------------------------------------------------------------------------------------------
// One workgroup.  Cluster k's tokens of all images occupy rows [base_k, base_k + M_k) of the grouped planes, base_k a
// multiple of 256 (a GEMM row tile never straddles two clusters); inside, image b's tokens of that cluster start at
// rowbase[b][k], in the label-grouped order of prep_kernel.  Thread (k, c) owns cluster k for the c-th slice of the images:
// slice sums -> exclusive scan over the slices (per cluster) and over the padded cluster totals -> slice walk.
__global__ __launch_bounds__(1024) void group_plan_kernel(const int32_t* __restrict__ lab_off, int B, int K,
                                                          int32_t* __restrict__ rowbase, int32_t* __restrict__ tile_group,
                                                          int max_tiles) {
  __shared__ int32_t part[1024], base[257];
  const int nsl = 1024 / K;                       // image slices (K <= 256 -> >= 4)
  const int k = threadIdx.x % K, c = threadIdx.x / K;
  const int per = (B + nsl - 1) / nsl, b0 = c * per, b1 = min(B, b0 + per);
  const bool act = c < nsl;
  for (int t = threadIdx.x; t < max_tiles; t += blockDim.x) tile_group[t] = -1;
  int sum = 0;
  if (act)
    for (int b = b0; b < b1; ++b) sum += lab_off[(size_t)b * (K + 1) + k + 1] - lab_off[(size_t)b * (K + 1) + k];
  part[threadIdx.x] = act ? sum : 0;
  __syncthreads();
  if ((int)threadIdx.x < K) {                     // exclusive scan over this cluster's slices; total M_k
    int run = 0;
    for (int q = 0; q < nsl; ++q) {
      const int v = part[q * K + threadIdx.x];
      part[q * K + threadIdx.x] = run;
      run += v;
    }
    base[threadIdx.x + 1] = (run + 255) & ~255;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    base[0] = 0;
    for (int q = 0; q < K; ++q) base[q + 1] += base[q];
  }
  __syncthreads();
  if (act) {
    int run = base[k] + part[threadIdx.x];
    for (int b = b0; b < b1; ++b) {
      rowbase[(size_t)b * K + k] = run;
      run += lab_off[(size_t)b * (K + 1) + k + 1] - lab_off[(size_t)b * (K + 1) + k];
    }
  }
  __syncthreads();   // the -1 fill above is complete
  if ((int)threadIdx.x < K)
    for (int t = base[threadIdx.x] >> 8; t < (base[threadIdx.x + 1] >> 8) && t < max_tiles; ++t) tile_group[t] = threadIdx.x;
}

int sv_launch_group_plan(segvlad_ctx* ctx, const int32_t* lab_off, int B, int K, int32_t* rowbase, int32_t* tile_group,
                         int max_tiles) {
  if (K > 256) return ctx->fail(SEGVLAD_ERR_LIMIT, "group_plan: K=%d > 256", K);
  hipLaunchKernelGGL(group_plan_kernel, dim3(1), dim3(1024), 0, ctx->stream, lab_off, B, K, rowbase, tile_group, max_tiles);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ---- constant of a PCA model ------------------------------------------------------------------------------------------------
// wmu[p] = sum_j comps[p][j] * mean[j]
__global__ __launch_bounds__(256) void project_consts_kernel(const float* __restrict__ comps, const float* __restrict__ mean,
                                                             int P, int64_t KD, float* __restrict__ wmu) {
  const int p = blockIdx.x;
  const float* w = comps + (size_t)p * KD;
  double acc = 0.0;   // one-off, tiny: fp64 so that the constant carries no rounding of its own
  for (int64_t j = threadIdx.x; j < KD; j += 256) acc += (double)w[j] * (double)mean[j];
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) wmu[p] = (float)red[0];
}

int sv_launch_project_consts(segvlad_ctx* ctx, const float* comps, const float* mean, int P, int64_t KD, float* wmu) {
  hipLaunchKernelGGL(project_consts_kernel, dim3(P), dim3(256), 0, ctx->stream, comps, mean, P, KD, wmu);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ---- aggregation in the projected space ---------------------------------------------------------------------------------
// Workgroup = (256 output columns, image); 8 waves, wave w owns columns [32 w, 32 w + 32) for up to 64 segments (two
// 32x32 accumulators).  The image's projected tokens are walked cluster by cluster in tiles of <= 32 rows (a tile never
// spans clusters: the segment weights a_sk are per-tile constants held in registers) staged through a DOUBLE-BUFFERED LDS
// tile: the tile list is built once, the global loads of tile t+1 are issued before the MFMAs of tile t and stored behind
// them -- one barrier per tile and no exposed load latency (a cluster holds ~24 tokens of an image: ~64 tiles per image,
// each previously paying a full load round trip between two barriers).  MFMA 32x32x2 f32 with A = the segment's weight
// where its column-mask bit is set, else 0, B = z.  The mean term W mu and the whitening scale are applied in the epilogue.
constexpr int PJ_T = 32;   // rows per tile (an image has at most K + N / 32 tiles: `maxt`)

// F16 (round 4): the sums of a tile on the 16-bit matrix pipe.  Inside a tile (one cluster) the weights factor out,
// y_s += a_sk * (sum_t m_st z_t): the inner sum has 0 / 1 coefficients -- exact in fp16 -- and z * zscale = h + l (|rest| <= 2^-22
// |z|; zscale a power of two from a rigorous bound on |z|), so it is two v_mfma_f32_32x32x16_f16 per 16 rows and segment half
// (A = the mask bits as 1.0 / 0.0, B = l, then h) instead of eight v_mfma_f32_32x32x2f32 at a sixteenth of the rate; the weight
// is applied to the tile's 32 x 32 sums (one fma per accumulator element, the 16 weights of a lane's rows as four 16-byte reads
// of a [cluster][segment] table).  The mask fragments are the same for the workgroup's eight waves: 256 threads build them once
// per tile (from the tokens' mask words, while the tile is fetched) and every wave reads its 16 bytes per (k-step, half).
template <int NW, bool F16 = false>   // waves per workgroup = 32-column slices: 8 (256 columns, 82 KiB of LDS: one workgroup per CU; default) or
                    // 4 (128 columns, 49 KiB: three per CU; option pj_nw = 4 -- measured slower: every workgroup repeats the
                    // per-tile mask loads, barriers and the weight table)
__global__ __launch_bounds__(64 * NW) void project_aggregate_kernel(const float* __restrict__ Z, const float* __restrict__ wmu,
                                                                const float* __restrict__ bn, const float* __restrict__ gscale,
                                                                const uint64_t* __restrict__ colmask,
                                                                const int32_t* __restrict__ lab_off,
                                                                const int32_t* __restrict__ rowbase,
                                                                const int32_t* __restrict__ seg_off, int N, int K, int P, int SC,
                                                                int maxt, const float* __restrict__ col_scale,
                                                                float* __restrict__ Y, float zscale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  static_assert(!F16 || NW == 8, "16-bit tile sums: the 8-wave workgroup");
  const int KS = K + 1 + (K & 1);                                  // row stride of ag, always odd: conflict-free column reads
  float* ag = reinterpret_cast<float*>(smem);                      // [64][KS]  a_sk = g_s / ||V_sk||   (0 for empty blocks); F16: [K][64]
  constexpr int NC = 32 * NW, NT_ = 64 * NW, C4W = NC / 4;         // columns, threads, float4 per staged row
  float* zt = ag + 64 * KS;                                        // [2][PJ_T][NC] staged rows
  uint64_t* mk = reinterpret_cast<uint64_t*>(zt + 2 * PJ_T * NC);  // [2][PJ_T] column masks of the staged tokens
  int32_t* tl_k = reinterpret_cast<int32_t*>(mk + 2 * PJ_T);       // [maxt] cluster of tile t
  int32_t* tl_j = tl_k + maxt;                                     // [maxt] first token (label-grouped order) of tile t
  int32_t* tl_z = tl_j + maxt;                                     // [maxt] first row of Z of tile t
  int32_t* tl_n = tl_z + maxt;                                     // [maxt] rows in tile t
  // F16: [2 buffers][k-step][segment half][64 lanes] 16-byte mask fragments, behind the lists (+ K + 1 tile offsets)
  uint4* af = reinterpret_cast<uint4*>((reinterpret_cast<uintptr_t>(tl_n + maxt + K + 1) + 15) & ~(uintptr_t)15);
  __shared__ int s_tiles;
  const int b = blockIdx.y, p0 = blockIdx.x * NC;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  const int SCb = (S + 63) >> 6;
  const int32_t* lo = lab_off + (size_t)b * (K + 1);
  const int pcol = p0 + 32 * w + i;
  // tile list, built by K threads at once: thread k fetches its cluster's extent (three independent global loads -- the
  // one-thread loop this replaces paid K dependent round trips, ~60 us per workgroup, a quarter of the kernel), thread 0
  // turns the per-cluster tile counts into offsets (LDS only), thread k writes its cluster's tiles
  int32_t* tl_pre = tl_n + maxt;                                   // [K + 1] first tile of cluster k
  int my_o0 = 0, my_n = 0, my_g0 = 0;
  if (tid < K) {
    my_o0 = lo[tid];
    my_n = lo[tid + 1] - my_o0;
    my_g0 = rowbase[(size_t)b * K + tid];
    tl_pre[tid + 1] = (my_n + PJ_T - 1) / PJ_T;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    tl_pre[0] = 0;
    for (int k = 0; k < K; ++k) {
      run += tl_pre[k + 1];
      tl_pre[k + 1] = run;
    }
    s_tiles = run < maxt ? run : maxt;
  }
  __syncthreads();
  if (tid < K) {
    int t = tl_pre[tid];
    for (int j0 = 0; j0 < my_n && t < maxt; j0 += PJ_T, ++t) {
      tl_k[t] = tid;
      tl_j[t] = my_o0 + j0;
      tl_z[t] = my_g0 + j0;
      tl_n[t] = min(PJ_T, my_n - j0);
    }
  }
  __syncthreads();
  const int tiles = s_tiles;

  for (int sc = 0; sc < SCb; ++sc) {
    const int Sc = min(64, S - 64 * sc);
    __syncthreads();   // the previous chunk's readers of ag / zt are done
    for (int idx = tid; idx < 64 * K; idx += NT_) {
      const int s = idx / K, k = idx - s * K;
      float a = 0.f;
      if (s < Sc) {
        const float nrm = bn[(size_t)(s0 + 64 * sc + s) * K + k];
        a = nrm > 1e-12f ? gscale[s0 + 64 * sc + s] / nrm : 0.f;   // an all-zero block contributes nothing (reference: 0 / 1e-12)
      }
      if (F16) ag[k * 64 + s] = a;
      else ag[s * KS + k] = a;
    }
    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const bool two = Sc > 32;

    // Two register sets: the rows of tile t+1 AND t+2 are in flight while tile t is multiplied (a tile is ~24 rows = 0.6 us
    // of MFMA work; with one tile in flight every tile waited out most of a ~1.5 us load round trip).  The sets are named
    // and the loop is unrolled by two (an array indexed by t & 1 would go to scratch memory).
    float4 va[4], vb[4];
    uint64_t ma = 0ull, mb = 0ull;
    uint4 fa = make_uint4(0u, 0u, 0u, 0u), fb = make_uint4(0u, 0u, 0u, 0u);   // F16: this thread's mask fragment of the tile in flight
#define SV_PJ_FETCH(V, M, t_)                                                                                   \
    do {                                                                                                        \
      const int nt_ = tl_n[t_];                                                                                 \
      const float* src_ = Z + (size_t)tl_z[t_] * P;                                                             \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        const int idx4 = tid + NT_ * q, row = idx4 / C4W, c4 = idx4 % C4W;                                      \
        V[q] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
        if (row < nt_ && p0 + 4 * c4 < P) V[q] = *reinterpret_cast<const float4*>(src_ + (size_t)row * P + p0 + 4 * c4); \
      }                                                                                                         \
      if (tid < PJ_T) M = tid < nt_ ? colmask[((size_t)b * N + tl_j[t_] + tid) * SC + sc] : 0ull;               \
    } while (0)
    // F16: thread (k-step, half, lane) = (tid >> 7, (tid >> 6) & 1, tid & 63) packs the mask bits of its lane's 8 rows
    // (16 ks + 8 kk + e) for segment 32 half + i as fp16 1.0 / 0.0
#define SV_PJ_FETCH_F(Fv, t_)                                                                                   \
    do {                                                                                                        \
      if (F16 && tid < 256) {                                                                                   \
        const int nt_ = tl_n[t_], ks_ = tid >> 7, hf_ = (tid >> 6) & 1, ln_ = tid & 63;                         \
        const int sh_ = 32 * hf_ + (ln_ & 31), j0_ = 16 * ks_ + 8 * (ln_ >> 5);                                 \
        const uint64_t* mrow_ = colmask + ((size_t)b * N + tl_j[t_] + j0_) * SC + sc;                           \
        uint32_t d_[4];                                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
          const uint64_t m0_ = (j0_ + 2 * q < nt_) ? mrow_[(size_t)(2 * q) * SC] : 0ull;                        \
          const uint64_t m1_ = (j0_ + 2 * q + 1 < nt_) ? mrow_[(size_t)(2 * q + 1) * SC] : 0ull;                \
          d_[q] = (((m0_ >> sh_) & 1ull) ? 0x3C00u : 0u) | (((m1_ >> sh_) & 1ull) ? 0x3C000000u : 0u);          \
        }                                                                                                       \
        Fv = make_uint4(d_[0], d_[1], d_[2], d_[3]);                                                            \
      }                                                                                                         \
    } while (0)
#define SV_PJ_STASH(V, M, buf_)                                                                                 \
    do {                                                                                                        \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        const int idx4 = tid + NT_ * q, row = idx4 / C4W, c4 = idx4 % C4W;                                      \
        *reinterpret_cast<float4*>(zt + (size_t)(buf_) * PJ_T * NC + row * NC + 4 * c4) = V[q];                 \
      }                                                                                                         \
      if (tid < PJ_T) mk[(buf_) * PJ_T + tid] = M;                                                              \
    } while (0)
#define SV_PJ_STASH_F(Fv, buf_)                                                                                 \
    do {                                                                                                        \
      if (F16 && tid < 256) af[(buf_) * 256 + tid] = Fv;                                                        \
    } while (0)
#define SV_PJ_MUL(t_)                                                                                           \
    do {                                                                                                        \
      const int cur_ = (t_) & 1, k_ = tl_k[t_], nt_ = tl_n[t_];                                                 \
      const float a0k = ag[i * KS + k_], a1k = ag[(32 + i) * KS + k_];                                          \
      const float* ztc = zt + (size_t)cur_ * PJ_T * NC + 32 * w + i;                                            \
      const uint64_t* mkc = mk + cur_ * PJ_T;                                                                   \
      for (int pr = 0; 2 * pr < nt_; ++pr) {                                                                    \
        const int j = 2 * pr + kk;                                                                              \
        const uint64_t m = mkc[j];                                                                              \
        const float bv = ztc[j * NC];                                                                           \
        const float a0 = ((m >> i) & 1ull) ? a0k : 0.f;                                                         \
        acc[0] = MFMA32(a0, bv, acc[0]);                                                                        \
        if (two) {                                                                                              \
          const float a1 = ((m >> (32 + i)) & 1ull) ? a1k : 0.f;                                                \
          acc[1] = MFMA32(a1, bv, acc[1]);                                                                      \
        }                                                                                                       \
      }                                                                                                         \
    } while (0)
#define SV_PJ_MUL16(t_)                                                                                         \
    do {                                                                                                        \
      const int cur_ = (t_) & 1, k_ = tl_k[t_], nt_ = tl_n[t_];                                                 \
      const float* ztc = zt + (size_t)cur_ * PJ_T * NC + 32 * w + i + (size_t)(8 * kk) * NC;                    \
      const uint4* afc = af + cur_ * 256 + l;                                                                   \
      f32x16 ta0, ta1;                                                                                          \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                        \
        if (ks == 1 && nt_ <= 16) break;                                                                        \
        f16x8 hh, ll;                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                         \
          const float v = ztc[(size_t)(16 * ks + e) * NC] * zscale;                                             \
          hh[e] = (_Float16)v;                                                                                  \
          ll[e] = (_Float16)(v - (float)hh[e]);                                                                 \
        }                                                                                                       \
        const uint4 a0u = afc[(2 * ks) * 64];                                                                   \
        const f16x8 a0 = __builtin_bit_cast(f16x8, a0u);                                                        \
        if (ks == 0) {                                                                                          \
          f32x16 zero_;                                                                                         \
          _Pragma("unroll") for (int r = 0; r < 16; ++r) zero_[r] = 0.f;                                        \
          ta0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, ll, zero_, 0, 0, 0);                                 \
        } else {                                                                                                \
          ta0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, ll, ta0, 0, 0, 0);                                   \
        }                                                                                                       \
        ta0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, hh, ta0, 0, 0, 0);                                     \
        if (two) {                                                                                              \
          const uint4 a1u = afc[(2 * ks + 1) * 64];                                                             \
          const f16x8 a1 = __builtin_bit_cast(f16x8, a1u);                                                      \
          if (ks == 0) {                                                                                        \
            f32x16 zero_;                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) zero_[r] = 0.f;                                      \
            ta1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, ll, zero_, 0, 0, 0);                               \
          } else {                                                                                              \
            ta1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, ll, ta1, 0, 0, 0);                                 \
          }                                                                                                     \
          ta1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, hh, ta1, 0, 0, 0);                                   \
        }                                                                                                       \
      }                                                                                                         \
      /* the weights of this lane's 16 rows (r = 4 q + r3 <-> segment 8 q + 4 kk + r3): four 16-byte reads per half */ \
      const float* agk = ag + k_ * 64 + 4 * kk;                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
        const float4 w0 = *reinterpret_cast<const float4*>(agk + 8 * q);                                        \
        acc[0][4 * q + 0] = fmaf(w0.x, ta0[4 * q + 0], acc[0][4 * q + 0]);                                      \
        acc[0][4 * q + 1] = fmaf(w0.y, ta0[4 * q + 1], acc[0][4 * q + 1]);                                      \
        acc[0][4 * q + 2] = fmaf(w0.z, ta0[4 * q + 2], acc[0][4 * q + 2]);                                      \
        acc[0][4 * q + 3] = fmaf(w0.w, ta0[4 * q + 3], acc[0][4 * q + 3]);                                      \
        if (two) {                                                                                              \
          const float4 w1 = *reinterpret_cast<const float4*>(agk + 32 + 8 * q);                                 \
          acc[1][4 * q + 0] = fmaf(w1.x, ta1[4 * q + 0], acc[1][4 * q + 0]);                                    \
          acc[1][4 * q + 1] = fmaf(w1.y, ta1[4 * q + 1], acc[1][4 * q + 1]);                                    \
          acc[1][4 * q + 2] = fmaf(w1.z, ta1[4 * q + 2], acc[1][4 * q + 2]);                                    \
          acc[1][4 * q + 3] = fmaf(w1.w, ta1[4 * q + 3], acc[1][4 * q + 3]);                                    \
        }                                                                                                       \
      }                                                                                                         \
    } while (0)
#define SV_PJ_MULX(t_)                                                                                          \
    do {                                                                                                        \
      if constexpr (F16) SV_PJ_MUL16(t_);                                                                       \
      else SV_PJ_MUL(t_);                                                                                       \
    } while (0)
    // set A carries the odd tiles' predecessors: tile 0 -> LDS directly, then A = tile 1, B = tile 2, A = tile 3, ...
    if (tiles > 0) {
      SV_PJ_FETCH(va, ma, 0);
      SV_PJ_FETCH_F(fa, 0);
      SV_PJ_STASH(va, ma, 0);
      SV_PJ_STASH_F(fa, 0);
    }
    if (tiles > 1) {
      SV_PJ_FETCH(va, ma, 1);
      SV_PJ_FETCH_F(fa, 1);
    }
    __syncthreads();   // ag and tile 0 visible
    for (int t = 0; t < tiles; t += 2) {
      // even tile t: tile t+1 waits in A, tile t+2 is requested into B
      if (t + 2 < tiles) {
        SV_PJ_FETCH(vb, mb, t + 2);
        SV_PJ_FETCH_F(fb, t + 2);
      }
      SV_PJ_MULX(t);
      if (t + 1 < tiles) {   // (the readers of that buffer -- tile t-1 -- passed the previous barrier)
        SV_PJ_STASH(va, ma, 1);
        SV_PJ_STASH_F(fa, 1);
      }
      __syncthreads();
      if (t + 1 >= tiles) break;
      // odd tile t+1: tile t+2 waits in B, tile t+3 is requested into A
      if (t + 3 < tiles) {
        SV_PJ_FETCH(va, ma, t + 3);
        SV_PJ_FETCH_F(fa, t + 3);
      }
      SV_PJ_MULX(t + 1);
      if (t + 2 < tiles) {
        SV_PJ_STASH(vb, mb, 0);
        SV_PJ_STASH_F(fb, 0);
      }
      __syncthreads();
    }
#undef SV_PJ_MULX
#undef SV_PJ_MUL16
#undef SV_PJ_STASH_F
#undef SV_PJ_FETCH_F
#undef SV_PJ_MUL
#undef SV_PJ_STASH
#undef SV_PJ_FETCH
    if (pcol < P) {
      const float cs = col_scale ? col_scale[pcol] : 1.f, mu = wmu[pcol];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int s = 32 * mt + pj_frag_row(r, kk);
          if (s < Sc) Y[(size_t)(s0 + 64 * sc + s) * P + pcol] = ((F16 ? acc[mt][r] * (1.f / zscale) : acc[mt][r]) - mu) * cs;
        }
      }
    }
  }
}

int sv_launch_project_aggregate(segvlad_ctx* ctx, const float* Z, const float* wmu, const float* block_norms, const float* gscale,
                                const uint64_t* colmask, const int32_t* lab_off, const int32_t* rowbase, const int32_t* seg_off_dev,
                                int B, int N, int K, int P, int SC, int S_max, const float* col_scale, float* Y, float zscale) {
  if (B <= 0 || S_max <= 0) return SEGVLAD_OK;
  if (P % 4) return ctx->fail(SEGVLAD_ERR_ARG, "project_aggregate: P=%d must be a multiple of 4", P);
  const int maxt = K + (N + PJ_T - 1) / PJ_T;   // every cluster may end in a partial tile
  const int nw = ctx->opt.pj_nw == 4 ? 4 : 8;
  if (K > 64 * nw) return ctx->fail(SEGVLAD_ERR_LIMIT, "project_aggregate: K=%d clusters need %d threads", K, K);
  // zscale > 0: the tile sums on the 16-bit pipe (8-wave workgroups; + the mask-fragment image: 2 x 4 KiB, 16-byte aligned)
  const bool f16 = zscale > 0.f && nw == 8 && ctx->opt.pj_f16 != 0;
  const size_t lds = (size_t)64 * (K + 1 + (K & 1)) * 4 + (size_t)2 * PJ_T * 32 * nw * 4 + (size_t)2 * PJ_T * 8 + (size_t)4 * maxt * 4 +
                     (size_t)(K + 1) * 4 + (f16 ? 16 + 2 * 256 * 16 : 0);
  if (lds > 160 * 1024)
    return ctx->fail(SEGVLAD_ERR_LIMIT, "project_aggregate: K=%d, N=%d need %zu B of LDS", K, N, lds);
  const void* fn = f16 ? reinterpret_cast<const void*>(project_aggregate_kernel<8, true>)
                       : nw == 8 ? reinterpret_cast<const void*>(project_aggregate_kernel<8>) : reinterpret_cast<const void*>(project_aggregate_kernel<4>);
  if (lds > 64 * 1024) SV_HIP(sv_max_dyn_lds(fn, (size_t)lds));
  const dim3 grid((P + 32 * nw - 1) / (32 * nw), B), block(64 * nw);
  if (f16)
    hipLaunchKernelGGL((project_aggregate_kernel<8, true>), grid, block, lds, ctx->stream, Z, wmu, block_norms, gscale, colmask, lab_off,
                       rowbase, seg_off_dev, N, K, P, SC, maxt, col_scale, Y, zscale);
  else if (nw == 8)
    hipLaunchKernelGGL(project_aggregate_kernel<8>, grid, block, lds, ctx->stream, Z, wmu, block_norms, gscale, colmask, lab_off, rowbase,
                       seg_off_dev, N, K, P, SC, maxt, col_scale, Y, 0.f);
  else
    hipLaunchKernelGGL(project_aggregate_kernel<4>, grid, block, lds, ctx->stream, Z, wmu, block_norms, gscale, colmask, lab_off, rowbase,
                       seg_off_dev, N, K, P, SC, maxt, col_scale, Y, 0.f);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
