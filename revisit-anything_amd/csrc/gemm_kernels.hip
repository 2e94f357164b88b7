// fp32 MFMA "NT" GEMM for gfx950: C[M][N] = epilogue(A[M][Kd] . B[N][Kd]^T), both operands row-major
// with the reduction dimension contiguous (descriptor rows), exact fp32 (v_mfma_f32_32x32x2_f32).
//
//   mode 0 (PCA apply, func_vpr.py:1434-1438): A' = A - a_sub[k];  C = acc * col_scale[n]
//   mode 1 (exact L2,  place_rec_main.py:53-56): C = row_add[m] + col_add[n] - 2 acc
//   mode 2 (exact L2, filtered): the same distance, but instead of storing the matrix every entry
//           with d2 <= thr[m] is appended to the per-query candidate list (atomic slot counter);
//           B rows may be a strided sample (row j of the operand = database row j * b_stride).
//
// Tile 128x128x32, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles.  Operands are staged k-major in
// LDS ([k][row], row stride 129) so that an MFMA operand read is 32 consecutive dwords per half-wave
// (conflict-free ds_read_b32) and the transposing stores are conflict-free too; LDS is double
// buffered (one barrier per k-tile), global loads run two k-tiles ahead in registers.
#include <algorithm>

#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int BM = 128, BN = 128, BK = 32, LDT = 129;  // LDT = 1 (mod 32): conflict-free k-major stores

__device__ __forceinline__ float4 ld4_guard(const float* base, int64_t row, int64_t nrows, int k, int Kd, int64_t ld,
                                            bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row < nrows) {
    const float* p = base + row * ld + k;
    if (vec && k + 3 < Kd) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (k < Kd) v.x = p[0];
      if (k + 1 < Kd) v.y = p[1];
      if (k + 2 < Kd) v.z = p[2];
      if (k + 3 < Kd) v.w = p[3];
    }
  }
  return v;
}

// One k-tile of global data per thread: 4 x float4 of A and of B (rows lrow + 32 j, k quad lk).
struct Stage {
  float4 a[4], b[4];
};

template <int MODE, bool FAST>
__device__ __forceinline__ void gload_tile(Stage& st, const float* __restrict__ A, const float* __restrict__ Bm,
                                           const float* __restrict__ a_sub, int64_t m0, int64_t n0, int M, int N, int Kd,
                                           int64_t ldb, int lrow, int lk, int k0, bool vec) {
  if (FAST) {  // interior tile: no guards, 16-B loads only
    const float* pa = A + (m0 + lrow) * (int64_t)Kd + k0 + lk;
    const float* pb = Bm + (n0 + lrow) * ldb + k0 + lk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st.a[j] = *reinterpret_cast<const float4*>(pa + (int64_t)32 * j * Kd);
      st.b[j] = *reinterpret_cast<const float4*>(pb + (int64_t)32 * j * ldb);
    }
    if (MODE == 0 && a_sub != nullptr) {
      const float4 s = *reinterpret_cast<const float4*>(a_sub + k0 + lk);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st.a[j].x -= s.x;
        st.a[j].y -= s.y;
        st.a[j].z -= s.z;
        st.a[j].w -= s.w;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      st.a[j] = ld4_guard(A, m0 + lrow + 32 * j, M, k0 + lk, Kd, Kd, vec);
      st.b[j] = ld4_guard(Bm, n0 + lrow + 32 * j, N, k0 + lk, Kd, ldb, vec);
    }
    if (MODE == 0 && a_sub != nullptr) {
      const float4 s = ld4_guard(a_sub, 0, 1, k0 + lk, Kd, 0, vec && ((reinterpret_cast<uintptr_t>(a_sub) & 15) == 0));
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // rows beyond M are never stored
        st.a[j].x -= s.x;
        st.a[j].y -= s.y;
        st.a[j].z -= s.z;
        st.a[j].w -= s.w;
      }
    }
  }
}

// write quarter q (= row group j) of the staged tile into the k-major LDS image
__device__ __forceinline__ void sstore_part(const Stage& st, float* As, float* Bs, int lrow, int lk, int j) {
  const int r = lrow + 32 * j;
  As[(lk + 0) * LDT + r] = st.a[j].x;
  As[(lk + 1) * LDT + r] = st.a[j].y;
  As[(lk + 2) * LDT + r] = st.a[j].z;
  As[(lk + 3) * LDT + r] = st.a[j].w;
  Bs[(lk + 0) * LDT + r] = st.b[j].x;
  Bs[(lk + 1) * LDT + r] = st.b[j].y;
  Bs[(lk + 2) * LDT + r] = st.b[j].z;
  Bs[(lk + 3) * LDT + r] = st.b[j].w;
}

// Main loop.  LDS is double buffered (one barrier per k-tile): while tile kt is multiplied out of
// buffer `cur`, the registers holding tile kt+1 are written into the other buffer during the second
// half of the MFMA sequence (their global loads were issued a full k-tile earlier), and the loads of
// tile kt+2 are issued right after the barrier.  MFMA operands for step ks+1 are read from LDS before
// the four MFMAs of step ks are issued.
template <int MODE, bool FAST>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[2][2], float* lds, const float* __restrict__ A,
                                              const float* __restrict__ Bm, const float* __restrict__ a_sub, int64_t m0,
                                              int64_t n0, int M, int N, int Kd, int64_t ldb, int tid, bool vec,
                                              int kbeg, int kend) {
  const int w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int wm = w >> 1, wn = w & 1;
  const int lrow = tid >> 3, lk = (tid & 7) << 2;
  const int ntiles = (kend - kbeg + BK - 1) / BK;
  // two register stages (ping-pong): tile kt+1 is being written to LDS while tile kt+2 is still in flight
  Stage s0, s1;
  gload_tile<MODE, FAST>(s0, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, lrow, lk, kbeg, vec);
#pragma unroll
  for (int j = 0; j < 4; ++j) sstore_part(s0, lds, lds + BK * LDT, lrow, lk, j);
  if (ntiles > 1) gload_tile<MODE, FAST>(s0, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, lrow, lk, kbeg + BK, vec);
  if (ntiles > 2) gload_tile<MODE, FAST>(s1, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, lrow, lk, kbeg + 2 * BK, vec);
  __syncthreads();
  int cur = 0;
  // one k-tile: multiply out of buffer `cur`, spill `st` (tile kt+1) into the other buffer during the last
  // four MFMA steps, barrier, then refill `st` with tile kt+3
  auto ktile = [&](Stage& st, int kt) {
    const float* As = lds + cur * (2 * BK * LDT);
    const float* Bs = As + BK * LDT;
    float* Asn = lds + (cur ^ 1) * (2 * BK * LDT);
    float* Bsn = Asn + BK * LDT;
    const bool have_next = kt + 1 < ntiles;
    const float* ap = As + kk * LDT + wm * 64 + i;
    const float* bp = Bs + kk * LDT + wn * 64 + i;
    float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
      if (ks + 1 < BK / 2) {
        na0 = ap[(2 * ks + 2) * LDT];
        na1 = ap[(2 * ks + 2) * LDT + 32];
        nb0 = bp[(2 * ks + 2) * LDT];
        nb1 = bp[(2 * ks + 2) * LDT + 32];
      }
      acc[0][0] = MFMA32(a0, b0, acc[0][0]);
      acc[0][1] = MFMA32(a0, b1, acc[0][1]);
      acc[1][0] = MFMA32(a1, b0, acc[1][0]);
      acc[1][1] = MFMA32(a1, b1, acc[1][1]);
      if (have_next && ks >= BK / 2 - 4) sstore_part(st, Asn, Bsn, lrow, lk, ks - (BK / 2 - 4));
      // keep the operand reads of step ks+1 AHEAD of the MFMAs of step ks (the scheduler otherwise sinks
      // them behind the MFMAs to reuse the operand registers, exposing the LDS latency every step)
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 8, 0);  // DS write (last four steps only)
      a0 = na0;
      a1 = na1;
      b0 = nb0;
      b1 = nb1;
    }
    __syncthreads();
    if (kt + 3 < ntiles) gload_tile<MODE, FAST>(st, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, lrow, lk, kbeg + (kt + 3) * BK, vec);
    cur ^= 1;
  };
  int kt = 0;
  for (; kt + 1 < ntiles; kt += 2) {
    ktile(s0, kt);      // s0 holds tile kt+1
    ktile(s1, kt + 1);  // s1 holds tile kt+2
  }
  if (kt < ntiles) ktile(s0, kt);
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                      float* __restrict__ C, int M, int N, int Kd, int64_t ldc,
                                                      const float* __restrict__ a_sub,
                                                      const float* __restrict__ col_scale,
                                                      const float* __restrict__ row_add,
                                                      const float* __restrict__ col_add, int tiles_m,
                                                      int b_stride, const float* __restrict__ thr, int64_t thr_ld,
                                                      uint32_t* __restrict__ cand_cnt, float* __restrict__ cand_d2,
                                                      uint32_t* __restrict__ cand_id, int cap, int k_per_split) {
  __shared__ float lds[2 * 2 * BK * LDT];  // [buffer][A|B][k][row]
  // tile order: m fastest so that the workgroups sharing a B panel (the big operand: database /
  // PCA components) are adjacent in dispatch order
  const int tile = blockIdx.x;
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int wm = w >> 1, wn = w & 1;
  const bool vec = ((Kd & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
  const int64_t ldb = (int64_t)Kd * b_stride;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // split-K (PCA projection only): blockIdx.y owns k in [kbeg, kend) and writes an un-scaled partial tile
  const int kbeg = (int)blockIdx.y * k_per_split;
  const int kend = (k_per_split > 0 && kbeg + k_per_split < Kd) ? kbeg + k_per_split : Kd;
  const bool fast = vec && (Kd % BK == 0) && (kbeg % BK == 0) && (m0 + BM <= M) && (n0 + BN <= N) &&
                    (MODE != 0 || a_sub == nullptr || (reinterpret_cast<uintptr_t>(a_sub) & 15) == 0);
  if (fast)
    gemm_mainloop<MODE, true>(acc, lds, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, tid, vec, kbeg, kend);
  else
    gemm_mainloop<MODE, false>(acc, lds, A, Bm, a_sub, m0, n0, M, N, Kd, ldb, tid, vec, kbeg, kend);
  if (MODE != 2 && gridDim.y > 1) C += (int64_t)blockIdx.y * M * ldc;   // split-K: un-scaled partial tiles

  // ---- epilogue ---------------------------------------------------------------------------------
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int64_t col = n0 + wn * 64 + nt * 32 + i;
    if (col >= N) continue;
    const float cs = (MODE == 0) ? (col_scale ? col_scale[col] : 1.f) : col_add[col * b_stride];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row < M) {
          if (MODE == 0) {
            C[row * ldc + col] = (gridDim.y > 1) ? acc[mt][nt][r] : acc[mt][nt][r] * cs;
          } else {
            const float v = sv_d2(row_add[row], cs, acc[mt][nt][r]);
            if (MODE == 1) {
              C[row * ldc + col] = (gridDim.y > 1) ? acc[mt][nt][r] : v;
            } else if (v <= thr[row * thr_ld]) {
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
}

// deterministic split-K reduction: out = (sum_s part[s]) * col_scale, slices added in index order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int64_t mn, int N,
                                                            int64_t ldc, const float* __restrict__ col_scale,
                                                            float* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= mn) return;
  float s = part[j];
  for (int t = 1; t < splits; ++t) s += part[(int64_t)t * mn + j];
  const int col = (int)(j % ldc);
  out[j] = (col < N && col_scale) ? s * col_scale[col] : s;
}

// split-K reduction of a distance block: out = ||q||^2 + ||r||^2 - 2 * (sum_s part[s]), slices added in index order
__global__ __launch_bounds__(256) void splitk_reduce_d2_kernel(const float* __restrict__ part, int splits, int M, int N, int64_t ldc,
                                                               const float* __restrict__ row_add,
                                                               const float* __restrict__ col_add, int b_stride,
                                                               float* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t mn = (int64_t)M * ldc;
  if (j >= mn) return;
  const int64_t row = j / ldc, col = j - row * ldc;
  if (col >= N) return;
  float s = part[j];
  for (int t = 1; t < splits; ++t) s += part[(int64_t)t * mn + j];
  out[j] = sv_d2(row_add[row], col_add[col * b_stride], s);
}

// parts_out != null (mode 1 with a K split only): the reduction of the partial sums is left to the caller, who gets the
// partial-sum block [splits][M][ldc] and the split count (sv_launch_l2_strided_parts: the single-image pass reduces and
// ranks in one kernel); *splits_out = 1 means nothing was split and C holds the finished distances.
static int gemm_launch(segvlad_ctx* ctx, int mode, const float* A, const float* Bm, float* C, int M, int N, int Kd,
                       int64_t ldc, const float* a_sub, const float* col_scale, const float* row_add,
                       const float* col_add, int b_stride, const float* thr, int64_t thr_ld, uint32_t* cand_cnt,
                       float* cand_d2, uint32_t* cand_id, int cap, bool split_d2 = false, const float** parts_out = nullptr,
                       int* splits_out = nullptr) {
  if (M <= 0 || N <= 0) return SEGVLAD_OK;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int64_t tiles = (int64_t)tiles_m * tiles_n;
  if (tiles > 0x7fffffffLL) return ctx->fail(SEGVLAD_ERR_LIMIT, "gemm: too many tiles");
  // split-K for the projection when the tile count cannot fill / balance the 512 resident workgroup slots
  int splits = 1, k_per_split = 0;
  if (mode == 0 && ldc == N && tiles < 4096 && Kd >= 8 * BK) {
    splits = (int)((2048 + tiles - 1) / tiles);
    if (splits > 16) splits = 16;
    k_per_split = (((Kd + splits - 1) / splits) + BK - 1) / BK * BK;
    splits = (Kd + k_per_split - 1) / k_per_split;
  }
  // a handful of distance tiles (the sampled exact level of a single-image pass: 50 queries x 244 rows = 2 tiles whose
  // 1024-deep fp32 k-loop is 90 us of pure latency): split K over 8-16 workgroups per tile.  Only where the caller allows
  // it -- the distances then feed thresholds; the matrix path, whose sums are the reference for bit-identity, never splits.
  // ... and a sample level over DEEP rows (raw K*D descriptors: 10 000 queries x 195 sample rows x 98 304 = 158 tiles with a
  // 3072-tile k-loop each, 0.6 of a wave of workgroups on the chip: 8.8 ms at 43 TF): K split so that ~3 workgroups per CU exist
  const bool deep_split = split_d2 && mode == 1 && tiles > 32 && tiles < 512 && Kd >= 8192;
  const bool d2_split = split_d2 && mode == 1 && (tiles <= 32 || deep_split) && Kd >= 16 * BK;
  if (deep_split) {
    splits = (int)std::min<int64_t>(16, std::max<int64_t>(2, 768 / tiles));
    k_per_split = (((Kd + splits - 1) / splits) + BK - 1) / BK * BK;
    splits = (Kd + k_per_split - 1) / k_per_split;
  } else if (d2_split) {
    // (four k-tiles of 32 per workgroup; two or one -- 16 / 32 slices -- measured slower: 46 vs 41 us for GEMM + reduce)
    splits = Kd / (4 * BK) < 16 ? Kd / (4 * BK) : 16;
    k_per_split = (((Kd + splits - 1) / splits) + BK - 1) / BK * BK;
    splits = (Kd + k_per_split - 1) / k_per_split;
  }
  float* Cout = C;
  if (splits > 1 && d2_split) {
    SV_HIP(ctx->s_l0part.reserve((size_t)splits * M * ldc * sizeof(float)));
    C = ctx->s_l0part.as<float>();
  } else if (splits > 1) {
    SV_HIP(ctx->s_dist.reserve((size_t)splits * M * ldc * sizeof(float)));
    C = ctx->s_dist.as<float>();
  } else {
    k_per_split = 0;
  }
#define SV_GEMM_ARGS                                                                                                    \
  dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, ctx->stream, A, Bm, C, M, N, Kd, ldc, a_sub, col_scale, row_add, \
      col_add, tiles_m, b_stride, thr, thr_ld, cand_cnt, cand_d2, cand_id, cap, k_per_split
  if (mode == 0)
    hipLaunchKernelGGL(gemm_nt_kernel<0>, SV_GEMM_ARGS);
  else if (mode == 1)
    hipLaunchKernelGGL(gemm_nt_kernel<1>, SV_GEMM_ARGS);
  else
    hipLaunchKernelGGL(gemm_nt_kernel<2>, SV_GEMM_ARGS);
#undef SV_GEMM_ARGS
  SV_HIP(hipGetLastError());
  if (splits_out) *splits_out = (splits > 1 && d2_split) ? splits : 1;
  if (parts_out && splits > 1 && d2_split) {
    *parts_out = C;
    return SEGVLAD_OK;
  }
  if (splits > 1 && d2_split) {
    const int64_t mn = (int64_t)M * ldc;
    hipLaunchKernelGGL(splitk_reduce_d2_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, ctx->stream, C, splits, M, N, ldc,
                       row_add, col_add, b_stride, Cout);
    SV_HIP(hipGetLastError());
  } else if (splits > 1) {
    const int64_t mn = (int64_t)M * ldc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, ctx->stream, C, splits, mn, N, ldc,
                       col_scale, Cout);
    SV_HIP(hipGetLastError());
  }
  return SEGVLAD_OK;
}

int sv_launch_gemm_nt(segvlad_ctx* ctx, int mode, const float* A, const float* Bm, float* C, int M, int N, int Kd,
                      int64_t ldc, const float* a_sub, const float* col_scale, const float* row_add,
                      const float* col_add) {
  return gemm_launch(ctx, mode, A, Bm, C, M, N, Kd, ldc, a_sub, col_scale, row_add, col_add, 1, nullptr, 0, nullptr, nullptr,
                     nullptr, 0);
}

int sv_launch_l2_strided(segvlad_ctx* ctx, const float* Q, const float* R, float* dist, int M, int n_sample, int Kd,
                         int64_t ldc, const float* qn, const float* rn, int b_stride, bool split_ok) {
  return gemm_launch(ctx, 1, Q, R, dist, M, n_sample, Kd, ldc, nullptr, nullptr, qn, rn, b_stride, nullptr, 0, nullptr, nullptr,
                     nullptr, 0, split_ok);
}

int sv_launch_l2_strided_parts(segvlad_ctx* ctx, const float* Q, const float* R, float* dist, int M, int n_sample, int Kd,
                               int64_t ldc, const float* qn, const float* rn, int b_stride, const float** parts, int* splits) {
  return gemm_launch(ctx, 1, Q, R, dist, M, n_sample, Kd, ldc, nullptr, nullptr, qn, rn, b_stride, nullptr, 0, nullptr, nullptr,
                     nullptr, 0, true, parts, splits);
}

int sv_launch_l2_filter(segvlad_ctx* ctx, const float* Q, const float* R, int M, int n_sample, int Kd, const float* qn,
                        const float* rn, int b_stride, const float* thr, int64_t thr_ld, uint32_t* cand_cnt,
                        float* cand_d2, uint32_t* cand_id, int cap) {
  return gemm_launch(ctx, 2, Q, R, nullptr, M, n_sample, Kd, 0, nullptr, nullptr, qn, rn, b_stride, thr, thr_ld, cand_cnt,
                     cand_d2, cand_id, cap);
}

// ------------------------------------------------------------------------------------------------
// row sum of squares / row normalisation: one wave per row, 16 B/lane streaming
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float row_sumsq(const float* x, int d, int lane) {
  float s = 0.f;
  if (((d & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int j = lane; j < (d >> 2); j += 64) {
      const float4 v = x4[j];
      s = fmaf(v.x, v.x, s);
      s = fmaf(v.y, v.y, s);
      s = fmaf(v.z, v.z, s);
      s = fmaf(v.w, v.w, s);
    }
  } else {
    for (int j = lane; j < d; j += 64) s = fmaf(x[j], x[j], s);
  }
  return wave_sum(s);
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ X, int64_t n, int d,
                                                        float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float s = row_sumsq(X + row * d, d, lane);
  if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ X, int64_t n, int d,
                                                             float* __restrict__ Y) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = X + row * d;
  float* y = Y + row * d;
  const float nrm = sqrtf(row_sumsq(x, d, lane));  // no epsilon: func_vpr.py:1675
  for (int j = lane; j < d; j += 64) y[j] = x[j] / nrm;
}

int sv_launch_row_sumsq(segvlad_ctx* ctx, const float* X, int64_t n, int d, float* out) {
  if (n <= 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, X, n, d, out);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

int sv_launch_normalize_rows(segvlad_ctx* ctx, const float* X, int64_t n, int d, float* Y) {
  if (n <= 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, X, n, d, Y);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
