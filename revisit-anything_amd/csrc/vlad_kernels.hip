// Segment-VLAD kernels for gfx950 (MI355X).  Reference semantics: func_vpr.py:1065-1210.
//
//   incidence_kernel   masks -> token-incidence bit rows            (func_vpr.py:1088-1092)
//   centroid_kernel    mask centroids for the Delaunay adjacency     (func_vpr.py:1314)
//   assign_kernel      tokens [D][N] -> labels, 1/||x||, top-2 gap, and the token-major copy
//                      Xt [N][D] that the aggregation streams          (func_vpr.py:1085,1145-1146)
//   prep_kernel        neighbour union (adj . inc) > 0, column masks, 1/sqrt(#non-empty blocks)
//                                                                      (func_vpr.py:1199-1200,1205)
//   aggregate_kernel   per (image, cluster): masked residual sums on the fp32 MFMA, intra-norm,
//                      global norm, packed store                      (func_vpr.py:1151,1195-1205)
//
// Wave = 64 lanes.  The two GEMM-shaped stages run on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).
#include <stdio.h>
#include <stdlib.h>

#include "ctx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// cross-lane move inside a 16-lane DPP row (quad_perm / row mirrors): plain VALU, no LDS crossbar
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// row of D-fragment register r for lane half kk (v_mfma_f32_32x32x2_f32 C/D layout)
__device__ __forceinline__ int frag_row(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

// ------------------------------------------------------------------------------------------------
// vocabulary: c^ = C / max(||C||, 1e-12), stored in MFMA B-operand order
//   bt[((dp * NT) + nt) * 64 + l] = c^[32 nt + (l & 31)][2 dp + (l >> 5)]   (0 outside K x D)
// ------------------------------------------------------------------------------------------------
__global__ void vocab_norm_kernel(const float* __restrict__ C, int K, int D, float* __restrict__ inv) {
  const int k = blockIdx.x;
  float s = 0.f;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float v = C[(size_t)k * D + d];
    s = fmaf(v, v, s);
  }
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) inv[k] = 1.0f / fmaxf(sqrtf(red[0]), 1e-12f);
}

__global__ void vocab_bt_kernel(const float* __restrict__ C, const float* __restrict__ inv, int K, int D, int NT,
                                int npairs, float* __restrict__ bt) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)npairs * NT * 64;
  if (idx >= total) return;
  const int l = idx & 63;
  const int nt = (idx >> 6) % NT;
  const int dp = (idx >> 6) / NT;
  const int k = 32 * nt + (l & 31), d = 2 * dp + (l >> 5);
  bt[idx] = (k < K && d < D) ? C[(size_t)k * D + d] * inv[k] : 0.f;
}

int sv_launch_vocab_prepare(segvlad_ctx* ctx) {
  const int K = ctx->K, D = ctx->D, NT = ctx->Kpad / 32;
  const int npairs = ((D + 63) / 64) * 32;
  SV_HIP(ctx->vocab_bt.reserve((size_t)npairs * NT * 64 * sizeof(float)));
  SV_HIP(ctx->s_misc.reserve((size_t)K * sizeof(float)));
  hipLaunchKernelGGL(vocab_norm_kernel, dim3(K), dim3(256), 0, ctx->stream, ctx->vocab.as<float>(), K, D,
                     ctx->s_misc.as<float>());
  const size_t total = (size_t)npairs * NT * 64;
  hipLaunchKernelGGL(vocab_bt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                     ctx->vocab.as<float>(), ctx->s_misc.as<float>(), K, D, NT, npairs, ctx->vocab_bt.as<float>());
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// incidence: one wave per (segment, 64-token word); lane = token.  The up-sampled pixel block of a
// token is walked through its distinct SOURCE rows/cols (nearest: src = min(floor(dst*scale), in-1)).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void incidence_kernel(const uint8_t* __restrict__ masks, int Hm, int Wm, int H, int W,
                                                        int patch, int dh, int dw, float sh, float sw,
                                                        uint64_t* __restrict__ inc, int nw) {
  const int s = blockIdx.x;
  const int word = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (word >= nw) return;
  const int lane = threadIdx.x & 63;
  const int N = dh * dw;
  const int t = word * 64 + lane;
  bool any = false;
  if (t < N) {
    const int ty = t / dw, tx = t - ty * dw;
    const int i0 = ty * patch, i1 = (ty == dh - 1) ? H - 1 : i0 + patch - 1;
    const int j0 = tx * patch, j1 = (tx == dw - 1) ? W - 1 : j0 + patch - 1;
    const uint8_t* m = masks + (size_t)s * Hm * Wm;
    int pr = -1;
    for (int i = i0; i <= i1 && !any; ++i) {
      int r = min((int)floorf((float)i * sh), Hm - 1);
      if (r == pr) continue;
      pr = r;
      const uint8_t* row = m + (size_t)r * Wm;
      int pc = -1;
      for (int j = j0; j <= j1; ++j) {
        int c = min((int)floorf((float)j * sw), Wm - 1);
        if (c == pc) continue;
        pc = c;
        if (row[c]) { any = true; break; }
      }
    }
  }
  const uint64_t b = __ballot(any);
  if (lane == 0) inc[(size_t)s * nw + word] = b;
}

// ------------------------------------------------------------------------------------------------
// fused incidence + centroid: one workgroup per mask, ONE coalesced 16-B/lane pass over the mask bytes.
//   phase 1: every 16-pixel chunk is reduced to a 16-bit "non-zero" word; non-empty chunks are OR-ed into an
//            LDS bitmap of the mask (Hm x Wm bits) and feed the exact integer centroid sums;
//   phase 2: lane = token: OR of the bitmap bits its up-sampled 14x14 cell samples (range test per source
//            row when the mask is not larger than the image, per-pixel walk otherwise); ballot -> u64 words.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t nz_nibble(uint32_t w) {  // bit j = (byte j of w != 0)
  uint32_t t = w | (w >> 4);
  t |= t >> 2;
  t |= t >> 1;
  t &= 0x01010101u;
  return ((t * 0x01020408u) >> 24) & 0xfu;
}

__global__ __launch_bounds__(256) void incidence_fused_kernel(const uint8_t* __restrict__ masks, int Hm, int Wm, int H, int W,
                                                              int patch, int dh, int dw, float sh, float sw,
                                                              uint64_t* __restrict__ inc, int nw,
                                                              double* __restrict__ cent) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* bits = reinterpret_cast<uint32_t*>(smem);  // [Hm][wpr]
  __shared__ unsigned long long red[3][4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int wpr = (Wm + 31) >> 5;  // u32 words per bitmap row
  const int cpr = Wm >> 4;         // 16-pixel chunks per row (Wm % 16 == 0)
  for (int j = tid; j < Hm * wpr; j += 256) bits[j] = 0;
  __syncthreads();
  const uint4* m16 = reinterpret_cast<const uint4*>(masks + (size_t)s * Hm * Wm);
  const int total = Hm * cpr;
  unsigned long long sr = 0, sc = 0, cnt = 0;
  for (int q = tid; q < total; q += 256) {
    const uint4 v = m16[q];
    if ((v.x | v.y | v.z | v.w) == 0) continue;
    const uint32_t nz = nz_nibble(v.x) | (nz_nibble(v.y) << 4) | (nz_nibble(v.z) << 8) | (nz_nibble(v.w) << 12);
    const int r = q / cpr, cc = q - r * cpr;
    atomicOr(&bits[r * wpr + (cc >> 1)], nz << (16 * (cc & 1)));
    const uint32_t pc = __popc(nz);
    const uint32_t pos = __popc(nz & 0xAAAAu) + 2 * __popc(nz & 0xCCCCu) + 4 * __popc(nz & 0xF0F0u) + 8 * __popc(nz & 0xFF00u);
    cnt += pc;
    sr += (unsigned long long)r * pc;
    sc += (unsigned long long)(16 * cc) * pc + pos;
  }
  if (cent) {  // exact integer sums -> bit-identical to np.nonzero(mask).mean(1)[::-1]
    for (int o = 32; o > 0; o >>= 1) {
      sr += __shfl_xor(sr, o);
      sc += __shfl_xor(sc, o);
      cnt += __shfl_xor(cnt, o);
    }
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = sr;
      red[1][tid >> 6] = sc;
      red[2][tid >> 6] = cnt;
    }
  }
  __syncthreads();
  if (cent && tid == 0) {
    const double c = (double)(red[2][0] + red[2][1] + red[2][2] + red[2][3]);
    cent[2 * (size_t)s + 0] = (double)(red[1][0] + red[1][1] + red[1][2] + red[1][3]) / c;
    cent[2 * (size_t)s + 1] = (double)(red[0][0] + red[0][1] + red[0][2] + red[0][3]) / c;
  }
  const int N = dh * dw;
  const bool ranges = (sh <= 1.0f) && (sw <= 1.0f);  // up-sampling: a cell samples contiguous source rows/cols
  for (int t0 = 0; t0 < nw * 64; t0 += 256) {
    const int t = t0 + tid;
    bool any = false;
    if (t < N) {
      const int ty = t / dw, tx = t - ty * dw;
      const int i0 = ty * patch, i1 = (ty == dh - 1) ? H - 1 : i0 + patch - 1;
      const int j0 = tx * patch, j1 = (tx == dw - 1) ? W - 1 : j0 + patch - 1;
      if (ranges) {
        const int rlo = min((int)floorf((float)i0 * sh), Hm - 1), rhi = min((int)floorf((float)i1 * sh), Hm - 1);
        const int clo = min((int)floorf((float)j0 * sw), Wm - 1), chi = min((int)floorf((float)j1 * sw), Wm - 1);
        const int w0 = clo >> 5, w1 = chi >> 5;
        for (int r = rlo; r <= rhi && !any; ++r) {
          for (int wv = w0; wv <= w1; ++wv) {
            const int lo = (wv == w0) ? (clo & 31) : 0, hi = (wv == w1) ? (chi & 31) : 31;
            const uint32_t msk = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
            if (bits[r * wpr + wv] & msk) { any = true; break; }
          }
        }
      } else {
        int pr = -1;
        for (int i = i0; i <= i1 && !any; ++i) {
          const int r = min((int)floorf((float)i * sh), Hm - 1);
          if (r == pr) continue;
          pr = r;
          int pc = -1;
          for (int j = j0; j <= j1; ++j) {
            const int c = min((int)floorf((float)j * sw), Wm - 1);
            if (c == pc) continue;
            pc = c;
            if ((bits[r * wpr + (c >> 5)] >> (c & 31)) & 1u) { any = true; break; }
          }
        }
      }
    }
    const uint64_t b = __ballot(any);
    const int word = t >> 6;
    if ((tid & 63) == 0 && word < nw) inc[(size_t)s * nw + word] = b;
  }
}

int sv_launch_incidence(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                        uint64_t* inc_bits, double* centroids) {
  const int dh = H / patch, dw = W / patch;
  const int N = dh * dw, nw = (N + 63) / 64;
  const float sh = (float)Hm / (float)H, sw = (float)Wm / (float)W;
  if (S == 0) return SEGVLAD_OK;
  const size_t lds = (size_t)Hm * ((Wm + 31) / 32) * 4;
  if ((Wm % 16) == 0 && (reinterpret_cast<uintptr_t>(masks) & 15) == 0 && lds <= 150 * 1024) {
    if (lds > 64 * 1024)
      SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(incidence_fused_kernel), (size_t)lds));
    hipLaunchKernelGGL(incidence_fused_kernel, dim3(S), dim3(256), lds, ctx->stream, masks, Hm, Wm, H, W, patch, dh, dw, sh, sw,
                       inc_bits, nw, centroids);
    SV_HIP(hipGetLastError());
    return SEGVLAD_OK;
  }
  if (centroids) SV_TRY(sv_launch_centroids(ctx, masks, S, Hm, Wm, centroids));
  hipLaunchKernelGGL(incidence_kernel, dim3(S, (nw + 3) / 4), dim3(256), 0, ctx->stream, masks, Hm, Wm, H, W, patch, dh,
                     dw, sh, sw, inc_bits, nw);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// centroids: exact integer sums of the non-zero (row, col), divided in fp64 -> bit-identical to
// np.nonzero(mask).mean(1)[::-1]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void centroid_kernel(const uint8_t* __restrict__ masks, int Hm, int Wm,
                                                       double* __restrict__ out) {
  const int s = blockIdx.x;
  const uint8_t* m = masks + (size_t)s * Hm * Wm;
  unsigned long long sr = 0, sc = 0, cnt = 0;
  const int total = Hm * Wm;
  for (int p = threadIdx.x; p < total; p += 256) {
    if (m[p]) {
      const int r = p / Wm;
      sr += r;
      sc += p - r * Wm;
      ++cnt;
    }
  }
  __shared__ unsigned long long red[3][256];
  red[0][threadIdx.x] = sr;
  red[1][threadIdx.x] = sc;
  red[2][threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      red[2][threadIdx.x] += red[2][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double c = (double)red[2][0];
    out[2 * s + 0] = (double)red[1][0] / c;  // x = mean col
    out[2 * s + 1] = (double)red[0][0] / c;  // y = mean row
  }
}

int sv_launch_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, double* out) {
  if (S == 0) return SEGVLAD_OK;
  hipLaunchKernelGGL(centroid_kernel, dim3(S), dim3(256), 0, ctx->stream, masks, Hm, Wm, out);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// adjacency: Delaunay neighbours of the mask centroids + self loop, raised to `order`
// (nbrMasksAGGFastSingle, func_vpr.py:1315-1345), one workgroup per image, no host round trip.
//
// Edge (u,v) belongs to the Delaunay triangulation iff some circle through u and v is empty.  The
// circles through u,v are a one-parameter family (centre on the bisector, parameter t); a point p
// on the left of u->v excludes t > tau_p, a point on the right excludes t < tau_p, with
//     tau_p = ((p-u).(p-v)) / cross(v-u, p-u)            (a cotangent of the angle u-p-v).
// So the edge exists iff max_{right} tau <= min_{left} tau, and a point strictly inside the segment
// blocks it.  O(S) per pair, fp64.  Generic inputs give exactly Qhull's triangulation.  Non-generic inputs -- exactly
// co-circular quadruples (max_right tau == min_left tau: the triangulation is not unique, both diagonals are kept here,
// Qhull picks one) and duplicate centroids (Qhull drops the coplanar duplicate) -- are REPORTED: every image that holds
// one adds 65536 to *n_bad and sets bit 1 of img_flags[b] (bit 0: an empty mask, i.e. a NaN centroid), so that the caller
// can route exactly those images through the reference's own Qhull path.  "Co-circular" includes quadruples whose two
// bounds agree to within a few ulps: there the sign of tmax - tmin is rounding noise, and either answer could differ from
// Qhull's (which decides such cases by its own perturbation rules).
// S <= 3 reproduces the reference's special case: every row = e0 (+ e1).
// ------------------------------------------------------------------------------------------------
// Threads per image (= blockDim.x, a launch parameter): up to 1024 -- S (S - 1) / 2 point pairs, ~1 per thread at S = 50 -- for
// small batches, where the launch is one image's latency (round 5: 256 threads walked ~10 pairs each, half of them skipped, one
// fp64 division chain after the other: 207 -> 39 us for 25 images); 256 for large batches, where 16-wave workgroups only wait for a
// whole CU beside the assignment pass they are overlapped with (717 us for 200 images, 1024 threads, under that pass).
__global__ __launch_bounds__(1024) void adjacency_kernel(const double* __restrict__ cent,
                                                        const int32_t* __restrict__ seg_off,
                                                        const int64_t* __restrict__ adj_off, int order, int S_max,
                                                        uint8_t* __restrict__ adj, uint32_t* __restrict__ n_bad,
                                                        uint8_t* __restrict__ img_flags) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  const int SW = (S_max + 63) >> 6;
  double* px = reinterpret_cast<double*>(smem);                  // [S_max]
  double* py = px + S_max;                                       // [S_max]
  uint64_t* A1 = reinterpret_cast<uint64_t*>(py + S_max);        // [S_max][SW]
  uint64_t* P = A1 + (size_t)S_max * SW;                         // [S_max][SW]
  uint64_t* Q = P + (size_t)S_max * SW;                          // [S_max][SW]
  const int tid = threadIdx.x;
  if (S == 0) return;
  uint8_t* out = adj + adj_off[b];
  for (int s = tid; s < S; s += (int)blockDim.x) {
    px[s] = cent[2 * (size_t)(s0 + s)];
    py[s] = cent[2 * (size_t)(s0 + s) + 1];
    if (px[s] != px[s]) {  // NaN centroid = empty mask (reference: ValueError)
      if (n_bad) atomicAdd(n_bad, 1u);
      if (img_flags) img_flags[b] = (uint8_t)(img_flags[b] | 1u);   // (same value from every writer; bit 1 is set later, by one thread)
    }
  }
  for (int j = tid; j < S * SW; j += (int)blockDim.x) A1[j] = 0;
  __shared__ int degenerate;
  if (tid == 0) degenerate = 0;
  __syncthreads();
  if (S <= 3) {
    for (int j = tid; j < S * S; j += (int)blockDim.x) {
      const int w = j % S;
      out[j] = (w == 0 || (w == 1 && S > 1)) ? 1 : 0;
    }
    return;
  }
  for (int u = tid; u < S; u += (int)blockDim.x) atomicOr(reinterpret_cast<unsigned long long*>(&A1[u * SW + (u >> 6)]), 1ull << (u & 63));
  const int npairs = S * (S - 1) / 2;
  for (int e = tid; e < npairs; e += (int)blockDim.x) {
    // pair e of the strict upper triangle, row-major: row u starts at u (2 S - u - 1) / 2
    int u = (int)(((double)(2 * S - 1) - sqrt((double)(2 * S - 1) * (double)(2 * S - 1) - 8.0 * (double)e)) * 0.5);
    while (u > 0 && u * (2 * S - u - 1) / 2 > e) --u;
    while ((u + 1) * (2 * S - u - 2) / 2 <= e) ++u;
    const int v = u + 1 + (e - u * (2 * S - u - 1) / 2);
    const double ux = px[u], uy = py[u], vx = px[v], vy = py[v];
    const double ex = vx - ux, ey = vy - uy;
    if (ex == 0.0 && ey == 0.0) degenerate = 1;   // duplicate centroid
    double tmin = INFINITY, tmax = -INFINITY, emin = 0.0, emax = 0.0;   // e*: rounding-error bounds of the two extremes
    bool blocked = false;
    for (int p = 0; p < S; ++p) {
      if (p == u || p == v) continue;
      const double ax = px[p] - ux, ay = py[p] - uy;
      const double bx = px[p] - vx, by = py[p] - vy;
      const double num = fma(ax, bx, ay * by);
      const double den = fma(ex, ay, -(ey * ax));
      if (den != 0.0) {
        const double t = num / den;
        // |computed tau - exact tau| for THESE coordinates: the differences, the two sums of products and the quotient
        // each carry a few 2^-53 relative to the magnitudes that entered them
        const double err = 9e-16 * ((fabs(ax * bx) + fabs(ay * by)) + fabs(t) * (fabs(ex * ay) + fabs(ey * ax))) / fabs(den);
        if (den > 0.0) {
          if (t < tmin) {
            tmin = t;
            emin = err;
          }
        } else if (t > tmax) {
          tmax = t;
          emax = err;
        }
      } else if (num < 0.0) {
        blocked = true;
      }
    }
    // an empty circle through four points (either diagonal is Delaunay): exactly (lattice centroids: every quantity above is
    // exact, equal taus compare equal), or so nearly that the sign of tmax - tmin is rounding noise
    if (!blocked && tmax > -INFINITY && tmin < INFINITY && fabs(tmax - tmin) <= emin + emax) degenerate = 1;
    if (!blocked && tmax <= tmin) {
      atomicOr(reinterpret_cast<unsigned long long*>(&A1[u * SW + (v >> 6)]), 1ull << (v & 63));
      atomicOr(reinterpret_cast<unsigned long long*>(&A1[v * SW + (u >> 6)]), 1ull << (u & 63));
    }
  }
  __syncthreads();
  if (tid == 0 && degenerate) {
    if (n_bad) atomicAdd(n_bad, 65536u);
    if (img_flags) img_flags[b] = (uint8_t)(img_flags[b] | 2u);
  }
  for (int j = tid; j < S * SW; j += (int)blockDim.x) P[j] = A1[j];
  __syncthreads();
  for (int it = 1; it < order; ++it) {  // P <- (P . A1) > 0
    for (int j = tid; j < S * SW; j += (int)blockDim.x) {
      const int v = j / SW, w = j - v * SW;
      uint64_t acc = 0;
      for (int uw = 0; uw < SW; ++uw) {
        uint64_t m = P[v * SW + uw];
        while (m) {
          const int u = (uw << 6) + __ffsll((unsigned long long)m) - 1;
          m &= m - 1;
          acc |= A1[u * SW + w];
        }
      }
      Q[j] = acc;
    }
    __syncthreads();
    for (int j = tid; j < S * SW; j += (int)blockDim.x) P[j] = Q[j];
    __syncthreads();
  }
  for (int j = tid; j < S * S; j += (int)blockDim.x) {
    const int v = j / S, w = j - v * S;
    out[j] = (uint8_t)((P[v * SW + (w >> 6)] >> (w & 63)) & 1ull);
  }
}

int sv_launch_adjacency(segvlad_ctx* ctx, const double* cent, const int32_t* seg_off_dev, const int64_t* adj_off_dev,
                        int B, int S_max, int order, uint8_t* adj, uint32_t* n_bad, uint8_t* img_flags) {
  if (B <= 0 || S_max <= 0) return SEGVLAD_OK;
  const int SW = (S_max + 63) / 64;
  const size_t lds = (size_t)S_max * 16 + (size_t)3 * S_max * SW * 8;
  if (lds > 160 * 1024)
    return ctx->fail(SEGVLAD_ERR_LIMIT, "adjacency: %d segments in one image exceed the LDS budget (%zu B)", S_max, lds);
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(adjacency_kernel), (size_t)lds));
  hipLaunchKernelGGL(adjacency_kernel, dim3(B), dim3(B <= 64 ? 1024 : 256), lds, ctx->stream, cent, seg_off_dev, adj_off_dev, order, S_max,
                     adj, n_bad, img_flags);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// assign: workgroup = (image, 64-token tile), 4 waves split D in 64-wide chunks.
//   A operand  (32 tokens x 2 d):  lane (i = l&31, kk = l>>5) loads T[d0+kk][t0+2i .. 2i+1]  (8 B/lane,
//              256 B contiguous per half-wave) -> M-tile 0 = even tokens, M-tile 1 = odd tokens
//   B operand  (2 d x 32 centres): pre-swizzled bt, 256 B contiguous per wave-load
// The raw values are also staged through a wave-private LDS tile and written token-major (Xt).
// ------------------------------------------------------------------------------------------------
template <int NT, int ABL = 0>   // ABL: timing ablations (SEGVLAD_ASSIGN_ABL; wrong results)
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ T, int N, int D, int K,
                                                     const float* __restrict__ bt, float* __restrict__ Xt,
                                                     uint8_t* __restrict__ labels, float* __restrict__ rnorm,
                                                     float* __restrict__ gap) {
  __shared__ float lds[4 * 64 * 65];
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const float* Tb = T + (size_t)b * D * N;
  float* tile = lds + w * (64 * 65);
  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.f;
  float ss0 = 0.f, ss1 = 0.f;
  const int ta = t0 + 2 * i;
  const bool v0 = ta < N, v1 = ta + 1 < N;
  const bool evenN = (N & 1) == 0;
  const int nchunks = (D + 63) >> 6;
  for (int ch = w; ch < nchunks; ch += 4) {
    const int dbase = ch << 6;
#pragma unroll 8
    for (int st = 0; st < 32; ++st) {
      const int d = dbase + 2 * st + kk;
      float x0 = 0.f, x1 = 0.f;
      if (d < D) {
        const float* p = Tb + (size_t)d * N + ta;
        if (v1 && evenN) {
          const float2 v = *reinterpret_cast<const float2*>(p);
          x0 = v.x;
          x1 = v.y;
        } else {
          if (v0) x0 = p[0];
          if (v1) x1 = p[1];
        }
      }
      ss0 = fmaf(x0, x0, ss0);
      ss1 = fmaf(x1, x1, ss1);
      const float* bp = bt + ((size_t)((dbase >> 1) + st) * NT) * 64 + l;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bv = bp[n * 64];
        if (ABL == 1) {
          acc[0][n][0] += x0 * bv;
          acc[1][n][0] += x1 * bv;
        } else {
          acc[0][n] = MFMA32(x0, bv, acc[0][n]);
          acc[1][n] = MFMA32(x1, bv, acc[1][n]);
        }
      }
      if (ABL != 2) {
        tile[(2 * i) * 65 + 2 * st + kk] = x0;
        tile[(2 * i + 1) * 65 + 2 * st + kk] = x1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int idx = it * 64 + l;
      const int tok = idx >> 4, c4 = (idx & 15) << 2;
      if (ABL != 2 && ABL != 3 && t0 + tok < N && dbase + c4 < D) {
        const float* q = tile + tok * 65 + c4;
        float4 v = make_float4(q[0], q[1], q[2], q[3]);
        *reinterpret_cast<float4*>(Xt + ((size_t)b * N + t0 + tok) * D + dbase + c4) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // ---- combine the four D-slices (fixed order w = 0..3: deterministic) ---------------------------
  ss0 += __shfl_xor(ss0, 32);
  ss1 += __shfl_xor(ss1, 32);
  __syncthreads();
  constexpr int KP = NT * 32;
  float* sc = lds;                    // [64][KP+1]
  float* ssum = lds + 64 * (KP + 1);  // [4][64]
  if (kk == 0) {
    ssum[w * 64 + 2 * i] = ss0;
    ssum[w * 64 + 2 * i + 1] = ss1;
  }
  for (int rnd = 0; rnd < 4; ++rnd) {
    if (w == rnd) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int tok = 2 * frag_row(r, kk) + mt;
            float* p = &sc[tok * (KP + 1) + n * 32 + i];
            *p = (rnd == 0) ? acc[mt][n][r] : (*p + acc[mt][n][r]);
          }
    }
    __syncthreads();
  }
  if (tid < 64 && t0 + tid < N) {
    const float s2 = ((ssum[tid] + ssum[64 + tid]) + ssum[128 + tid]) + ssum[192 + tid];
    const float rn = 1.0f / fmaxf(sqrtf(s2), 1e-12f);
    float best = -INFINITY, second = -INFINITY;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      const float v = sc[tid * (KP + 1) + k];
      if (v > best) {
        second = best;
        best = v;
        bi = k;
      } else if (v > second) {
        second = v;
      }
    }
    const size_t o = (size_t)b * N + t0 + tid;
    labels[o] = (uint8_t)bi;
    rnorm[o] = rn;
    if (gap) gap[o] = (K > 1) ? (best - second) * rn : INFINITY;
  }
}

// ------------------------------------------------------------------------------------------------
// assign, wide-load variant (D % 32 == 0, K <= 64): workgroup = (image, 128-token tile), wave w owns tokens
// 32w..32w+31 for ALL d (no cross-wave reduction: one sequential fp32 chain per score, d = 0..D-1).
//   * the [32 d][128 token] chunk is fetched with 16-B loads (512 contiguous bytes per d row; the old kernel's
//     8-B loads ran at 1.25 TB/s) one chunk ahead into registers, then parked in a double-buffered LDS tile
//     (row stride 130 floats: conflict-free for both access patterns below);
//   * MFMA A operand = tile[d][token] (lanes = consecutive tokens), B = pre-swizzled centres from L2;
//   * the token-major copy Xt is written from the same tile: a lane gathers 4 consecutive d of one token
//     (banks 8 c4 + token: all 64 distinct) and stores 16 B; 8 lanes cover 128 contiguous bytes of a token row.
// ------------------------------------------------------------------------------------------------
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// one [32 d][128 token] chunk of assign_wide_kernel out of its LDS tile: scores (fp32 MFMA against the pre-swizzled centres),
// squared norms, and the token-major copy Xt
template <int NT>
__device__ __forceinline__ void assign_wide_chunk(const float* __restrict__ tl, const float* __restrict__ bt, float* __restrict__ Xt,
                                                  int b, int N, int D, int t0, int tid, int w, int l, int i, int kk, int c,
                                                  f32x16 (&acc)[NT], float& ss) {
  constexpr int DC = 32, TS = 128;
  auto rot = [](int d) { return 8 * (d >> 2) + 32 * (d & 1); };
#pragma unroll 8
  for (int st = 0; st < DC / 2; ++st) {
    const float x = tl[(2 * st + kk) * TS + ((32 * w + i + rot(2 * st + kk)) & 127)];
    ss = fmaf(x, x, ss);
    const float* bp = bt + ((size_t)(c * (DC / 2) + st) * NT) * 64 + l;
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = MFMA32(x, bp[n * 64], acc[n]);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = it * 256 + tid;
    const int tok = idx >> 3, c4 = idx & 7;
    if (t0 + tok < N) {
      const float* q = tl + (4 * c4) * TS;   // rows 4 c4 + e: rotation 8 c4 + 32 (e & 1)
      const int t_a = (tok + 8 * c4) & 127, t_b = (tok + 8 * c4 + 32) & 127;
      const float4 o = make_float4(q[t_a], q[TS + t_b], q[2 * TS + t_a], q[3 * TS + t_b]);
      *reinterpret_cast<float4*>(Xt + ((size_t)b * N + t0 + tok) * D + c * DC + 4 * c4) = o;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256) void assign_wide_kernel(const float* __restrict__ T, int N, int D, int K,
                                                          const float* __restrict__ bt, float* __restrict__ Xt,
                                                          uint8_t* __restrict__ labels, float* __restrict__ rnorm,
                                                          float* __restrict__ gap, int B) {
  // LDS tile [d][token], row stride 128 floats, row d ROTATED by rot(d) = 8 (d >> 2) + 32 (d & 1) tokens: the MFMA A
  // operand reads two rows d, d+1 per instruction (their rotations differ by 32 banks: disjoint), the transposed gather
  // that writes Xt reads (4 c4 + e, tok) for 8 values of c4 and 8 consecutive tokens (banks tok + 8 c4: all 64 distinct),
  // and the loader stores whole 16-B quads.  (PMC, round 2: with a plain stride of 130 floats 39 % of the LDS cycles of
  // this kernel were bank conflicts -- two-way on the A reads and on the 8-B stores.)
  constexpr int DC = 32, TS = 128;
  auto rot = [](int d) { return 8 * (d >> 2) + 32 * (d & 1); };
  __shared__ __attribute__((aligned(16))) float tile[2 * DC * TS + 128];   // also [128][NT*32+1] scores at the end (NT <= 2)
  __shared__ float ssum[128];
  // XCD-aware order (round 6): the workgroups are dealt to the eight XCDs round-robin by their linear id, so with (tile, image) read
  // straight off the grid the 12 token tiles of an image ran on 8 different XCDs -- and a token tile's 512-byte piece of a D-row
  // shares its first and last 128-byte line with its neighbours (rows of N = 1530 floats are not line aligned): every boundary line
  // crossed the fabric into two L2s, 2.36 GB fetched for 1.88 GB of tokens (rocprofv3 FETCH_SIZE, profiles/r06_pmc_traffic.json).
  // Consecutive LOGICAL tiles now share an XCD: logical id = (linear id % 8) * per + linear id / 8.
  const int gx = (N + 127) / 128;
  const int per = (int)((gridDim.x + 7) / 8);
  const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (logical >= gx * B) return;
  const int b = logical / gx, t0 = (logical - b * gx) * 128;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const float* Tb = T + (size_t)b * D * N;
  const int lr = tid >> 5, lq = tid & 31;        // loader: rows lr + 8 j, tokens t0 + 4 lq .. + 3
  const int tl0 = t0 + 4 * lq;
  const int nvalid = N - tl0;                    // tokens of this lane's quad inside the image
  f32x16 acc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  float ss = 0.f;
  const int nch = D / DC;
  // Two named register sets: chunks c + 1 AND c + 2 are in flight while chunk c is multiplied (one 16-KiB chunk per
  // workgroup in flight, four workgroups per CU, left the kernel at 3.8 TB/s of its read + write traffic; the loop is
  // unrolled by two so that the sets are never indexed dynamically).
  float4 va[4], vb[4];
#define SV_LOAD_CHUNK(V, c)                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
    const float* p = Tb + (size_t)((c) * DC + lr + 8 * j) * N + tl0;                  \
    if (nvalid >= 4) {                                                                \
      const f32x4u u = *reinterpret_cast<const f32x4u*>(p);                           \
      V[j] = make_float4(u[0], u[1], u[2], u[3]);                                     \
    } else {                                                                          \
      V[j] = make_float4(nvalid > 0 ? p[0] : 0.f, nvalid > 1 ? p[1] : 0.f, nvalid > 2 ? p[2] : 0.f, 0.f); \
    }                                                                                 \
  }
#define SV_PARK_CHUNK(V, buf)                                                         \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
    const int d_ = lr + 8 * j;                                                        \
    float* q = tile + (buf) * (DC * TS) + d_ * TS + ((4 * lq + rot(d_)) & 127);       \
    *reinterpret_cast<float4*>(q) = V[j];                                             \
  }
  SV_LOAD_CHUNK(va, 0)
  SV_PARK_CHUNK(va, 0)
  if (1 < nch) { SV_LOAD_CHUNK(va, 1) }
  __syncthreads();
  // chunk c + 1 waits in set A when c is even, in set B when c is odd
#define SV_ASSIGN_STEP(c, VNEXT, VFREE)                                               \
  {                                                                                   \
    if ((c) + 2 < nch) { SV_LOAD_CHUNK(VFREE, (c) + 2) }                              \
    assign_wide_chunk<NT>(tile + ((c) & 1) * (DC * TS), bt, Xt, b, N, D, t0, tid, w, l, i, kk, (c), acc, ss);  \
    if ((c) + 1 < nch) { SV_PARK_CHUNK(VNEXT, ((c) + 1) & 1) }                        \
    __syncthreads();                                                                  \
  }
  for (int c = 0; c < nch; c += 2) {
    SV_ASSIGN_STEP(c, va, vb)
    if (c + 1 < nch) SV_ASSIGN_STEP(c + 1, vb, va)
  }
#undef SV_ASSIGN_STEP
#undef SV_LOAD_CHUNK
#undef SV_PARK_CHUNK
  // ---- scores -> LDS [token][cluster]; one thread per token takes the first maximum and the runner-up ----------
  constexpr int KP = NT * 32;
  float* sc = tile;   // [128][KP+1]
  ss += __shfl_xor(ss, 32);
  if (kk == 0) ssum[32 * w + i] = ss;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[(32 * w + frag_row(r, kk)) * (KP + 1) + n * 32 + i] = acc[n][r];
  __syncthreads();
  if (tid < 128 && t0 + tid < N) {
    const float rn = 1.0f / fmaxf(sqrtf(ssum[tid]), 1e-12f);
    float best = -INFINITY, second = -INFINITY;
    int bi = 0;
    for (int k = 0; k < K; ++k) {
      const float sv = sc[tid * (KP + 1) + k];
      if (sv > best) {
        second = best;
        best = sv;
        bi = k;
      } else if (sv > second) {
        second = sv;
      }
    }
    const size_t o = (size_t)b * N + t0 + tid;
    labels[o] = (uint8_t)bi;
    rnorm[o] = rn;
    if (gap) gap[o] = (K > 1) ? (best - second) * rn : INFINITY;
  }
}

int sv_launch_assign(segvlad_ctx* ctx, const float* tokens, int B, int N, float* xt, uint8_t* labels, float* rnorm,
                     float* gap) {
  const int NT = ctx->Kpad / 32;
  dim3 grid((N + 63) / 64, B), block(256);
  const float* bt = ctx->vocab_bt.as<float>();
  if (NT <= 2 && ctx->D % 32 == 0 && !ctx->opt.assign_narrow) {
    // 1-D grid of (tiles x images) rounded up to a multiple of 8 (the kernel maps its linear id to a (tile, image) pair itself:
    // XCD-aware order)
    const unsigned tiles_total = (unsigned)((N + 127) / 128) * (unsigned)B;
    dim3 gridw((tiles_total + 7) / 8 * 8, 1);
    if (NT == 1)
      hipLaunchKernelGGL(assign_wide_kernel<1>, gridw, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap, B);
    else
      hipLaunchKernelGGL(assign_wide_kernel<2>, gridw, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap, B);
    SV_HIP(hipGetLastError());
    return SEGVLAD_OK;
  }
  switch (NT) {
    case 1:
      hipLaunchKernelGGL(assign_kernel<1>, grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
      break;
    case 2: {
#ifdef SEGVLAD_ABLATIONS   // timing ablations (WRONG results): development builds only, never in the shipped library
      const char* ab = getenv("SEGVLAD_ASSIGN_ABL");
      const int a = ab ? atoi(ab) : 0;
      if (a == 1) {
        hipLaunchKernelGGL((assign_kernel<2, 1>), grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
        break;
      } else if (a == 2) {
        hipLaunchKernelGGL((assign_kernel<2, 2>), grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
        break;
      } else if (a == 3) {
        hipLaunchKernelGGL((assign_kernel<2, 3>), grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
        break;
      }
#endif
      hipLaunchKernelGGL(assign_kernel<2>, grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
      break;
    }
    case 4:
      hipLaunchKernelGGL(assign_kernel<4>, grid, block, 0, ctx->stream, tokens, N, ctx->D, ctx->K, bt, xt, labels, rnorm, gap);
      break;
    default:
      return ctx->fail(SEGVLAD_ERR_LIMIT, "K=%d: supported cluster counts are 1..64 and 97..128 (Kpad in {32,64,128})", ctx->K);
  }
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// prep: one workgroup per image.
//   inc2[s] = OR_{u : adj[s][u]} inc[u]                       (adj . inc) > 0
//   colmask[t][sc] bit (s - 64 sc) = inc2[s][t]               column form consumed by the aggregation
//   gscale[s] = 1/sqrt(#clusters k with a token t: label_t = k and inc2[s][t])
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_kernel(const uint8_t* __restrict__ labels, const uint64_t* __restrict__ inc,
                                                   const int32_t* __restrict__ seg_off,
                                                   const int64_t* __restrict__ adj_off, const uint8_t* __restrict__ adj,
                                                   int N, int K, int S_max, int SC, uint64_t* __restrict__ colmask,
                                                   float* __restrict__ gscale, int32_t* __restrict__ tok_order,
                                                   int32_t* __restrict__ lab_off, const float* __restrict__ rnorm,
                                                   float* __restrict__ rn_sorted, int stage, int ord_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int nw = (N + 63) >> 6;
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  uint64_t* inc2 = reinterpret_cast<uint64_t*>(smem);  // [S_max][nw]
  uint64_t* lbits = inc2 + (size_t)S_max * nw;         // [K][nw]
  const int tid = threadIdx.x;
  if (adj && stage) {
    // neighbour-union of the incidence rows, from LDS: the image's own rows and its adjacency (as bit rows) are staged
    // with coalesced loads first -- walking the S adjacency bytes of a row in global memory with a conditional 8-B load
    // behind each (S x nw x S dependent round trips spread over 256 threads) was 0.1 of this kernel's 0.16 ms
    const int SW = (S_max + 63) >> 6;
    uint64_t* inc0 = reinterpret_cast<uint64_t*>(reinterpret_cast<int*>(lbits + (size_t)K * nw) + ord_n);   // [S_max][nw]
    uint64_t* nbr = inc0 + (size_t)S_max * nw;                                                                       // [S_max][SW]
    for (int idx = tid; idx < S * nw; idx += 256) inc0[idx] = inc[(size_t)s0 * nw + idx];
    for (int idx = tid; idx < S * SW; idx += 256) nbr[idx] = 0;
    __syncthreads();
    const uint8_t* a = adj + adj_off[b];
    for (int e = tid; e < S * S; e += 256)
      if (a[e]) {
        const int s = e / S, u = e - s * S;
        atomicOr(reinterpret_cast<unsigned long long*>(&nbr[s * SW + (u >> 6)]), 1ull << (u & 63));
      }
    __syncthreads();
    for (int idx = tid; idx < S * nw; idx += 256) {
      const int s = idx / nw, w = idx - s * nw;
      uint64_t v = 0;
      for (int uw = 0; uw < SW; ++uw) {
        uint64_t m = nbr[s * SW + uw];
        while (m) {
          const int u = (uw << 6) + __builtin_ctzll(m);
          m &= m - 1;
          v |= inc0[u * nw + w];
        }
      }
      inc2[idx] = v;
    }
  } else {
    for (int idx = tid; idx < S * nw; idx += 256) {
      const int s = idx / nw, w = idx - s * nw;
      uint64_t v = 0;
      if (adj) {
        const uint8_t* a = adj + adj_off[b] + (size_t)s * S;
        for (int u = 0; u < S; ++u)
          if (a[u]) v |= inc[(size_t)(s0 + u) * nw + w];
      } else {
        v = inc[(size_t)(s0 + s) * nw + w];
      }
      inc2[idx] = v;
    }
  }
  for (int idx = tid; idx < K * nw; idx += 256) lbits[idx] = 0;
  __syncthreads();
  for (int t = tid; t < N; t += 256) {
    const int lab = labels[(size_t)b * N + t];
    atomicOr(reinterpret_cast<unsigned long long*>(&lbits[lab * nw + (t >> 6)]), 1ull << (t & 63));
  }
  __syncthreads();
  // token order grouped by label (ascending token id inside a label): the aggregation workgroup of (k, b) reads its
  // token list, the tokens' 1/||x|| and their segment masks from CONTIGUOUS ranges instead of re-scanning all N
  // labels and gathering per token (one dependent global round trip instead of four)
  __shared__ int kcnt[257];
  int* ord = reinterpret_cast<int*>(lbits + (size_t)K * nw);   // [ord_n >= max(N, S)]
  for (int k = tid; k < K; k += 256) {
    int c = 0;
    for (int w = 0; w < nw; ++w) c += __popcll(lbits[k * nw + w]);
    kcnt[k] = c;
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int k = 0; k < K; ++k) {
      const int c = kcnt[k];
      kcnt[k] = a;
      a += c;
    }
    kcnt[K] = a;
  }
  __syncthreads();
  for (int k = tid; k <= K; k += 256) lab_off[(size_t)b * (K + 1) + k] = kcnt[k];
  for (int k = tid; k < K; k += 256) {
    int pos = kcnt[k];
    for (int w = 0; w < nw; ++w) {
      uint64_t m = lbits[k * nw + w];
      while (m) {
        ord[pos++] = 64 * w + __builtin_ctzll(m);
        m &= m - 1;
      }
    }
  }
  __syncthreads();
  for (int p = tid; p < N; p += 256) {
    const int t = ord[p];
    const int w = t >> 6, bit = t & 63;
    tok_order[(size_t)b * N + p] = t;
    rn_sorted[(size_t)b * N + p] = rnorm[(size_t)b * N + t];
    for (int sc = 0; sc < SC; ++sc) {
      uint64_t m = 0;
      const int lo = sc * 64, hi = min(S, lo + 64);
      for (int s = lo; s < hi; ++s) m |= ((inc2[s * nw + w] >> bit) & 1ull) << (s - lo);
      colmask[((size_t)b * N + p) * SC + sc] = m;   // indexed by POSITION in the label-grouped order
    }
  }
  __syncthreads();
  // number of non-empty (segment, cluster) blocks per segment: one thread per PAIR (a thread per segment walked K * nw
  // word pairs alone while 200 threads idled), counted in the (now free) token-order array
  int* nnz = ord;
  for (int s = tid; s < S; s += 256) nnz[s] = 0;
  __syncthreads();
  for (int idx = tid; idx < S * K; idx += 256) {
    const int s = idx / K, k = idx - s * K;
    uint64_t any = 0;
    for (int w = 0; w < nw; ++w) any |= inc2[s * nw + w] & lbits[k * nw + w];
    if (any != 0) atomicAdd(&nnz[s], 1);
  }
  __syncthreads();
  for (int s = tid; s < S; s += 256) gscale[s0 + s] = nnz[s] > 0 ? (float)(1.0 / sqrt((double)nnz[s])) : 0.f;
}

int sv_launch_prep(segvlad_ctx* ctx, const uint8_t* labels, const uint64_t* inc_bits, const int32_t* seg_off_dev,
                   const int64_t* adj_off_dev, const uint8_t* adj, int B, int N, int K, int S_max, int SC,
                   uint64_t* colmask, float* gscale) {
  const int nw = (N + 63) / 64;
  if (K > 256) return ctx->fail(SEGVLAD_ERR_LIMIT, "prep: K=%d > 256", K);
  SV_HIP(ctx->s_tokorder.reserve((size_t)B * N * sizeof(int32_t)));
  SV_HIP(ctx->s_laboff.reserve((size_t)B * (K + 1) * sizeof(int32_t)));
  SV_HIP(ctx->s_rnsorted.reserve((size_t)B * N * sizeof(float)));
  const int ord_n = ((N > S_max ? N : S_max) + 1) & ~1;   // token order, later the per-segment block counts
  size_t lds = ((size_t)S_max + K) * nw * sizeof(uint64_t) + (size_t)ord_n * sizeof(int);
  if (lds > 160 * 1024)
    return ctx->fail(SEGVLAD_ERR_LIMIT, "prep: (S_max=%d + K=%d) x %d token words needs %zu B of LDS (limit 160 KiB)", S_max,
                     K, nw, lds);
  // staged neighbour union (the image's incidence rows + adjacency bit rows in LDS) when it fits beside the rest
  const size_t lds_staged = ((size_t)S_max + K) * nw * 8 + (size_t)ord_n * 4 + (size_t)S_max * nw * 8 +
                            (size_t)S_max * ((S_max + 63) / 64) * 8;
  const int stage = (adj != nullptr && lds_staged <= 150 * 1024) ? 1 : 0;
  if (stage) lds = lds_staged;
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(prep_kernel), (size_t)lds));
  hipLaunchKernelGGL(prep_kernel, dim3(B), dim3(256), lds, ctx->stream, labels, inc_bits, seg_off_dev, adj_off_dev, adj, N,
                     K, S_max, SC, colmask, gscale, ctx->s_tokorder.as<int32_t>(), ctx->s_laboff.as<int32_t>(),
                     ctx->s_rnorm.as<float>(), ctx->s_rnsorted.as<float>(), stage, ord_n);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}

// ------------------------------------------------------------------------------------------------
// aggregate: workgroup = (cluster k, image b); wave w owns d in [128 w, 128 w + 128).
//   V[s][k][d] = sum_{t in L_k} inc2[s][t] * (x_t[d] * rn_t - C[k][d])
// as an MFMA product  (32 segments x 2 tokens) . (2 tokens x 32 d):
//   A[i][kk] = bit (32 mt + i) of colmask[t_kk]   (0/1 exact)
//   B[kk][j] = fma(Xt[t_kk][dcol + 4 j + q], rn, -C[k][...])   q = N-tile; one 16-B load feeds 4 N-tiles
// then ||V[s,k,:]|| (LDS reduction across the waves), scale by gscale[s]/max(norm,1e-12), store packed.
// ------------------------------------------------------------------------------------------------
// PLANES: additionally (or instead: out may be NULL) emit the PCA input planes of the block,
//   (v - mean) * xscale = h1 + h2 (two fp16 terms, see gemm_f16x3_kernels.hip), so that the projection GEMM reads the
//   descriptor without a separate max-abs + split pass and the fp32 descriptor need not reach HBM at all
//   (|v| <= 1 by construction, which is what makes the scale known in advance).
typedef const __attribute__((address_space(1))) void* agg_gptr_t;
typedef __attribute__((address_space(3))) void* agg_lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Raw LDS reads for the loops that keep global->LDS DMAs in flight.  The compiler cannot tell the DMA'd queue slots from
// any other LDS address, so before every LDS load IT emits it waits for all outstanding DMAs (s_waitcnt vmcnt(0)) -- which
// turns a queue of QD loads in flight into one round trip per step.  Reads issued from inline asm are invisible to that
// pass; the counted s_waitcnt in the loop is then the only wait.
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(agg_lptr_t)p; }
// one step's operands: the queue slot (16 B), 1/||x|| and the column mask of the lane's token
__device__ __forceinline__ void lds_step_read(unsigned a_slot, unsigned a_rn, unsigned a_m, f32x4& x, float& rn, uint64_t& m) {
  asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x), "=&v"(rn), "=&v"(m)
               : "v"(a_slot), "v"(a_rn), "v"(a_m)
               : "memory");
}
__device__ __forceinline__ int lds_read_i32(unsigned a) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ f32x4 lds_read_f32x4(unsigned a) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
__device__ unsigned long long sv_agg_phase_cycles[8];   // ABL == 9: s_memtime phase sums of wave 0 (SEGVLAD_AGG_ABL=9 prints)
#define SV_APHASE(k)                                                              \
  if (ABL == 9) {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                 \
    if (threadIdx.x == 0) atomicAdd(&sv_agg_phase_cycles[k], now_ - phase_t0);    \
    phase_t0 = now_;                                                              \
  }

// PLANES: additionally / instead of the fp32 blocks, the two fp16 planes (v - mean) * xscale = h1 + h2 of the descriptor, in the
// blocked layout the projection GEMM reads (the "planes" form of segvlad_images_pca; its "project" form uses
// token_norms_kernel below instead of this kernel).
template <bool PLANES, int ABL = 0>   // ABL: timing ablations (SEGVLAD_AGG_ABL; wrong results)
__global__ __launch_bounds__(768) void aggregate_kernel(const float* __restrict__ Xt, const float* __restrict__ rnorm,
                                                        const int32_t* __restrict__ tok_order,
                                                        const int32_t* __restrict__ lab_off,
                                                        const uint64_t* __restrict__ colmask,
                                                        const float* __restrict__ C, const int32_t* __restrict__ seg_off,
                                                        const float* __restrict__ gscale, int N, int D, int K, int SC,
                                                        int Ncap, float* __restrict__ out,
                                                        float* __restrict__ block_norms, const float* __restrict__ mean,
                                                        float xscale, _Float16* __restrict__ h1, _Float16* __restrict__ h2,
                                                        int kpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, i = l & 31, kk = l >> 5;
  const int nwaves = blockDim.x >> 6;
  const int PW = nwaves * 2 + 1;
  uint64_t* mskl = reinterpret_cast<uint64_t*>(smem);         // [Ncap]
  int* tokl = reinterpret_cast<int*>(mskl + Ncap);            // [Ncap]
  float* rnl = reinterpret_cast<float*>(tokl + Ncap);         // [Ncap]
  float* part = rnl + Ncap;                                   // [64][PW]
  float* alpha = part + 64 * PW;                              // [64]
  int* wcount = reinterpret_cast<int*>(alpha + 64);           // [16]
  constexpr int QD = 6;                                       // token-pair rows in flight per wave
  unsigned char* xq = reinterpret_cast<unsigned char*>(wcount + 16);  // [nwaves][QD][1 KiB] DMA queue (16-B aligned)

  // a workgroup walks kpb consecutive clusters of its image: the (fire-and-forget) block stores of cluster k drain
  // while the lists and token rows of cluster k+1 are fetched -- with one workgroup per CU (registers) nothing else
  // would overlap the two
  const int k_end = min(K, ((int)blockIdx.x + 1) * kpb);
  unsigned long long phase_t0 = (ABL == 9) ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int k = (int)blockIdx.x * kpb; k < k_end; ++k) {
  // ---- L_k: the tokens assigned to cluster k, ascending (grouped by prep_kernel) ---------------------
  const int o0 = lab_off[(size_t)b * (K + 1) + k];
  const int n = lab_off[(size_t)b * (K + 1) + k + 1] - o0;
  for (int j = tid; j < n; j += blockDim.x) {
    tokl[j] = tok_order[(size_t)b * N + o0 + j];
    rnl[j] = rnorm[(size_t)b * N + o0 + j];   // 1/||x|| in the label-grouped order (prep_kernel)
  }
  if (tid == 0 && (n & 1)) {  // pad to an even count with a zero-weight entry
    tokl[n] = 0;
    rnl[n] = 0.f;
  }
  const int npairs = (n + 1) >> 1;
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  const int dcol = w * 128 + 4 * i;
  const bool dvalid = dcol < D;
  float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dvalid && C != nullptr) c4 = *reinterpret_cast<const float4*>(C + (size_t)k * D + dcol);
  float4 mu4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (PLANES && dvalid && mean != nullptr) mu4 = *reinterpret_cast<const float4*>(mean + (size_t)k * D + dcol);
  const float* Xb = Xt + (size_t)b * N * D + dcol;
  const size_t KD = (size_t)K * D;
  const int SCb = (S + 63) >> 6;

  for (int sc = 0; sc < SCb; ++sc) {
    for (int j = tid; j < n; j += blockDim.x) mskl[j] = colmask[((size_t)b * N + o0 + j) * SC + sc];
    if (tid == 0 && (n & 1)) mskl[n] = 0;
    __syncthreads();
    SV_APHASE(0)  // lists
    const int Sc = min(64, S - 64 * sc);
    const bool two = Sc > 32;
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][q][r] = 0.f;

    // token rows reach the MFMA through a wave-private LDS queue filled by global->LDS DMA (lane l's 16 B land at
    // slot + 16 l and are read back by the same lane): QD pairs in flight per wave without spending VGPRs on them --
    // with two register-staged loads in flight the loop was one HBM round trip per two token pairs
    {
      unsigned char* qbase = xq + (size_t)__builtin_amdgcn_readfirstlane(w) * (QD * 1024);   // wave-uniform: lives in M0
      auto issue = [&](int p) {
        const int t = lds_read_i32(lds_addr(tokl + 2 * p + kk));
        // lanes beyond D (a partial last wave) fetch a valid dummy address: their 16 B are never used
        const float* src = dvalid ? Xb + (size_t)t * D : Xt;
        __builtin_amdgcn_global_load_lds((agg_gptr_t)src, (agg_lptr_t)(qbase + (p % QD) * 1024), 16, 0, 0);
      };
      if (ABL != 1)
        for (int q = 0; q < QD && q < npairs; ++q) issue(q);
      for (int p = 0; p < npairs; ++p) {
        const int j = 2 * p + kk;
        const int rem = npairs - 1 - p;   // DMAs issued after pair p's
        if (rem >= QD - 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (rem == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (rem == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x4 x;
        float rn;
        uint64_t m;
        lds_step_read(lds_addr(qbase + (p % QD) * 1024 + l * 16), lds_addr(rnl + j), lds_addr(mskl + j), x, rn, m);
        // (the slot has been read -- lgkmcnt(0) inside -- before it is refilled)
        if (ABL != 1 && p + QD < npairs) issue(p + QD);
        if (!dvalid || ABL == 1) x = f32x4{0.f, 0.f, 0.f, 0.f};
        const float b0 = fmaf(x.x, rn, -c4.x), b1 = fmaf(x.y, rn, -c4.y);
        const float b2 = fmaf(x.z, rn, -c4.z), b3 = fmaf(x.w, rn, -c4.w);
        const float a0 = ((m >> i) & 1ull) ? 1.f : 0.f;
        acc[0][0] = MFMA32(a0, b0, acc[0][0]);
        acc[0][1] = MFMA32(a0, b1, acc[0][1]);
        acc[0][2] = MFMA32(a0, b2, acc[0][2]);
        acc[0][3] = MFMA32(a0, b3, acc[0][3]);
        if (two) {
          const float a1 = ((m >> (32 + i)) & 1ull) ? 1.f : 0.f;
          acc[1][0] = MFMA32(a1, b0, acc[1][0]);
          acc[1][1] = MFMA32(a1, b1, acc[1][1]);
          acc[1][2] = MFMA32(a1, b2, acc[1][2]);
          acc[1][3] = MFMA32(a1, b3, acc[1][3]);
        }
      }
    }
    SV_APHASE(1)  // main loop
    // ---- block norms: quad reduce in registers, then across lanes/waves through LDS ----------------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (mt == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = acc[mt][0][r] * acc[mt][0][r];
        p = fmaf(acc[mt][1][r], acc[mt][1][r], p);
        p = fmaf(acc[mt][2][r], acc[mt][2][r], p);
        p = fmaf(acc[mt][3][r], acc[mt][3][r], p);
        // sum over the 16 lanes of a DPP row (quad swaps, then the two mirrors): VALU only -- ds_bpermute shuffles and
        // 96 partials per row made this phase 40 % of the kernel
        p += dpp_f32<0xB1>(p);    // quad_perm [1,0,3,2]
        p += dpp_f32<0x4E>(p);    // quad_perm [2,3,0,1]
        p += dpp_f32<0x141>(p);   // row_half_mirror
        p += dpp_f32<0x140>(p);   // row_mirror
        if ((l & 15) == 0) part[(32 * mt + frag_row(r, kk)) * PW + w * 2 + ((l >> 4) & 1)] = p;
      }
    }
    SV_APHASE(5)  // partial norms computed
    __syncthreads();
    SV_APHASE(6)  // barrier after partial norms
    if (tid < Sc) {
      float sum = 0.f;
      const int cnt = nwaves * 2;
      for (int c = 0; c < cnt; ++c) sum += part[tid * PW + c];
      const float nrm = sqrtf(sum);
      const int sg = s0 + 64 * sc + tid;
      alpha[tid] = gscale[sg] / fmaxf(nrm, 1e-12f);
      if (block_norms) block_norms[(size_t)sg * K + k] = nrm;
    }
    SV_APHASE(7)  // row sums
    __syncthreads();
    SV_APHASE(2)  // barrier
    if (dvalid) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 32 * mt + frag_row(r, kk);
          if (row < Sc) {
            const float a = alpha[row];
            float4 v = make_float4(acc[mt][0][r] * a, acc[mt][1][r] * a, acc[mt][2][r] * a, acc[mt][3][r] * a);
            const size_t o = (size_t)(s0 + 64 * sc + row) * KD + (size_t)k * D + dcol;
            if ((!PLANES || out != nullptr) && ABL != 2) *reinterpret_cast<float4*>(out + o) = v;
            if (PLANES && ABL != 2) {
              typedef _Float16 h4 __attribute__((ext_vector_type(4)));
              const float f[4] = {(v.x - mu4.x) * xscale, (v.y - mu4.y) * xscale, (v.z - mu4.z) * xscale, (v.w - mu4.w) * xscale};
              h4 p1, p2;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                p1[e] = (_Float16)f[e];
                p2[e] = (_Float16)(f[e] - (float)p1[e]);
              }
              const size_t ob = sv_x3_off(s0 + 64 * sc + row, (int64_t)k * D + dcol, KD);   // blocked plane layout (ctx.h)
              *reinterpret_cast<h4*>(h1 + ob) = p1;
              *reinterpret_cast<h4*>(h2 + ob) = p2;
            }
          }
        }
      }
    }
    SV_APHASE(3)  // stores issued
    __syncthreads();
    SV_APHASE(4)  // barrier
  }
  }
}

// ---- block norms + grouped token planes ("project then aggregate", project_kernels.hip) -------------------------------------
// What segvlad_images_pca needs from the D-space when the projection runs per token: the block norms ||V_sk|| and the fp16
// planes of the token residuals x^_t - C_k(t), grouped by cluster.  No descriptor is stored, so nothing forces the workgroup-wide
// geometry of aggregate_kernel (12 waves side by side over D, two barriers per cluster): here ONE WAVE owns a whole
// (image, cluster) task and walks D in 128-column chunks, tokens streaming through a wave-private DMA queue that never
// drains between chunks; squared sums stay in registers until the task ends.  No barrier anywhere.
constexpr int TNK_QD = 8, TNK_LCAP = 256, TNK_WAVES = 4;

// s_waitcnt vmcnt(n), n in 0 .. 23.  vmcnt counts loads AND stores on gfx9 and both retire in issue order, so the plane
// stores between two DMAs are part of the count: "DMA f has landed" = "at most (DMAs + stores issued after it) outstanding".
__device__ __forceinline__ void tnk_wait_vm(int n) {
#define TNK_W(v) case v: asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); break;
  switch (n) {
    TNK_W(0) TNK_W(1) TNK_W(2) TNK_W(3) TNK_W(4) TNK_W(5) TNK_W(6) TNK_W(7) TNK_W(8) TNK_W(9) TNK_W(10) TNK_W(11)
    TNK_W(12) TNK_W(13) TNK_W(14) TNK_W(15) TNK_W(16) TNK_W(17) TNK_W(18) TNK_W(19) TNK_W(20) TNK_W(21) TNK_W(22)
    default: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
  }
#undef TNK_W
}

template <bool BIG>   // BIG: the (rare) clusters with >= TNK_LCAP tokens of one image, lists read from global memory step by step
__global__ __launch_bounds__(64 * TNK_WAVES, 2) void token_norms_kernel(const float* __restrict__ Xt, const float* __restrict__ rnorm,
                                                                    const int32_t* __restrict__ tok_order,
                                                                    const int32_t* __restrict__ lab_off,
                                                                    const uint64_t* __restrict__ colmask,
                                                                    const float* __restrict__ C,
                                                                    const int32_t* __restrict__ seg_off, int N, int D, int K, int SC,
                                                                    int Dpad, float* __restrict__ block_norms, float xscale,
                                                                    _Float16* __restrict__ h1, _Float16* __restrict__ h2,
                                                                    const int32_t* __restrict__ rowbase, int64_t dummy_row,
                                                                    int skip_le, const uint8_t* __restrict__ redo_task) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, i = l & 31, kk = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = (int)blockIdx.x * TNK_WAVES + w;
  if (k >= K) return;   // (no barriers in this kernel)
  const size_t wsz = (size_t)TNK_QD * 1024 + (size_t)TNK_LCAP * 16 + (size_t)Dpad * 4;
  unsigned char* qbase = smem + (size_t)w * wsz;                                  // [QD][1 KiB] DMA queue
  uint64_t* mskl = reinterpret_cast<uint64_t*>(qbase + TNK_QD * 1024);            // [LCAP]
  int* tokl = reinterpret_cast<int*>(mskl + TNK_LCAP);                            // [LCAP]
  float* rnl = reinterpret_cast<float*>(tokl + TNK_LCAP);                         // [LCAP]
  float* cl = rnl + TNK_LCAP;                                                     // [Dpad] centre k
  const int o0 = lab_off[(size_t)b * (K + 1) + k];
  const int n = lab_off[(size_t)b * (K + 1) + k + 1] - o0;
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  const int SCb = (S + 63) >> 6;
  const int npairs = (n + 1) >> 1;
  const int nd = Dpad >> 7, total = nd * npairs;
  const size_t tb = (size_t)b * N + o0;
  if ((n >= TNK_LCAP) != BIG) return;   // the other instantiation's task
  // gram_norms_kernel's task (skip_le = -1: none) -- unless that kernel found one of the task's segments CANCELLING (redo_task)
  if (n <= skip_le && !(redo_task && n > 0 && redo_task[(size_t)b * K + k])) return;
  if (n == 0) {   // an empty cluster: zero norms, no rows
    for (int s = l; s < S; s += 64) block_norms[(size_t)(s0 + s) * K + k] = 0.f;
    return;
  }
  for (int d = 4 * l; d < Dpad; d += 256)
    *reinterpret_cast<float4*>(cl + d) = d < D ? *reinterpret_cast<const float4*>(C + (size_t)k * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr bool single = !BIG;   // the token list fits the wave's LDS lists
  if (single) {
    for (int j = l; j < n; j += 64) {
      tokl[j] = tok_order[tb + j];
      rnl[j] = rnorm[tb + j];
    }
    if (l == 0 && (n & 1)) {
      tokl[n] = 0;
      rnl[n] = 0.f;
    }
  }
  const int64_t grow0 = rowbase[(size_t)b * K + k];
  const size_t nkb_d = (size_t)(D >> 5);
  const size_t ob_dummy = sv_x3_off(dummy_row < 0 ? -dummy_row : dummy_row, (4 * i) % D, D);
  const float* Xb = Xt + (size_t)b * N * D;

  for (int sc = 0; sc < SCb; ++sc) {
    if (single) {
      for (int j = l; j < n; j += 64) mskl[j] = colmask[(tb + j) * SC + sc];
      if (l == 0 && (n & 1)) mskl[n] = 0;
    }
    const int Sc = min(64, S - 64 * sc);
    const bool two = Sc > 32;
    f32x16 acc[2][4];
    float nsq[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        nsq[a][r] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q][r] = 0.f;
      }
    // step f = (chunk dc, pair p), dc-major; the DMA of step f + QD is issued when step f's slot has been read
    int idc = 0, ip = 0, fi = 0;   // issue cursor
    auto issue = [&]() {
      const int j = 2 * ip + kk;
      const int t = single ? lds_read_i32(lds_addr(tokl + j)) : (j < n ? tok_order[tb + j] : 0);
      const int dcol = idc * 128 + 4 * i;
      const float* src = dcol < D ? Xb + (size_t)t * D + dcol : Xt;   // lanes beyond D fetch a valid dummy address
      __builtin_amdgcn_global_load_lds((agg_gptr_t)src, (agg_lptr_t)(qbase + (fi % TNK_QD) * 1024), 16, 0, 0);
      ++fi;
      if (++ip == npairs) {
        ip = 0;
        ++idc;
      }
    };
    for (int q = 0; q < TNK_QD && q < total; ++q) issue();
    int f = 0;
    for (int dc = 0; dc < nd; ++dc) {
      const int dcol = dc * 128 + 4 * i;
      const bool dvalid = dcol < D;
      const f32x4 c4 = lds_read_f32x4(lds_addr(cl + dcol));
      const size_t kb_dc = (size_t)(dcol >> 5);                       // k-block of this lane's four columns
      const unsigned lane_c = (unsigned)((dcol & 31) >> 3), lane_o = (unsigned)(dcol & 7);
      for (int p = 0; p < npairs; ++p, ++f) {
        const int j = 2 * p + kk;
        float rn = 0.f;
        uint64_t m = 0ull;
        f32x4 x;
        if (!single) {
          rn = j < n ? rnorm[tb + j] : 0.f;
          m = j < n ? colmask[(tb + j) * SC + sc] : 0ull;
        }
        // ops issued after step f's DMA: min(QD - 1, rem) later DMAs and, on the first segment chunk, two plane stores for
        // each of the last min(QD, f) steps (always issued: lanes without a valid element write the dummy row)
        const int rem = total - 1 - f;
        if (single && dummy_row >= 0) {
          // steady state (a full queue behind and ahead): the count is the constant QD - 1 (+ 2 QD plane stores)
          if (f >= TNK_QD && rem >= TNK_QD - 1) {
            if (sc == 0) asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
          } else {
            tnk_wait_vm((rem < TNK_QD - 1 ? rem : TNK_QD - 1) + (sc == 0 ? 2 * (f < TNK_QD ? f : TNK_QD) : 0));
          }
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // list entries come from global memory on this path
        }
        if (single)
          lds_step_read(lds_addr(qbase + (f % TNK_QD) * 1024 + l * 16), lds_addr(rnl + j), lds_addr(mskl + j), x, rn, m);
        else
          x = lds_read_f32x4(lds_addr(qbase + (f % TNK_QD) * 1024 + l * 16));
        // (the slot has been read -- lgkmcnt(0) inside -- before it is refilled)
        if (fi < total) issue();
        if (!dvalid) x = f32x4{0.f, 0.f, 0.f, 0.f};
        const float b0 = fmaf(x.x, rn, -c4.x), b1 = fmaf(x.y, rn, -c4.y);
        const float b2 = fmaf(x.z, rn, -c4.z), b3 = fmaf(x.w, rn, -c4.w);
        if (sc == 0) {   // the residual's fp16 planes, grouped row grow0 + j
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          const float fv[4] = {b0 * xscale, b1 * xscale, b2 * xscale, b3 * xscale};
          h4 p1, p2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p1[e] = (_Float16)fv[e];
            p2[e] = (_Float16)(fv[e] - (float)p1[e]);
          }
          const bool real = dvalid && j < n;
          // sv_x3_off(grow0 + j, dcol, D) with the lane / chunk constants of this 128-column chunk taken out of the loop
          const unsigned rr = (unsigned)(grow0 + j);
          const size_t ob = real ? ((size_t)(rr >> 7) * nkb_d + kb_dc) * 4096 + (size_t)(((rr & 127u) << 5) + (((lane_c ^ (unsigned)sv_x3_swz(rr)) << 3) | lane_o))
                                 : ob_dummy;
          *reinterpret_cast<h4*>(h1 + ob) = p1;
          *reinterpret_cast<h4*>(h2 + ob) = p2;
        }
        const float a0 = ((m >> i) & 1ull) ? 1.f : 0.f;
        acc[0][0] = MFMA32(a0, b0, acc[0][0]);
        acc[0][1] = MFMA32(a0, b1, acc[0][1]);
        acc[0][2] = MFMA32(a0, b2, acc[0][2]);
        acc[0][3] = MFMA32(a0, b3, acc[0][3]);
        if (two) {
          const float a1 = ((m >> (32 + i)) & 1ull) ? 1.f : 0.f;
          acc[1][0] = MFMA32(a1, b0, acc[1][0]);
          acc[1][1] = MFMA32(a1, b1, acc[1][1]);
          acc[1][2] = MFMA32(a1, b2, acc[1][2]);
          acc[1][3] = MFMA32(a1, b3, acc[1][3]);
        }
      }
      // this chunk's 128 columns of every block row: square, add, restart the accumulators
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float q2 = nsq[mt][r];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            q2 = fmaf(acc[mt][q][r], acc[mt][q][r], q2);
            acc[mt][q][r] = 0.f;
          }
          nsq[mt][r] = q2;
        }
      }
    }
    // row sums over the 32 lanes of each half-wave (DPP inside the 16-lane rows, one cross-row exchange)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if (mt == 1 && !two) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = nsq[mt][r];
        p += dpp_f32<0xB1>(p);
        p += dpp_f32<0x4E>(p);
        p += dpp_f32<0x141>(p);
        p += dpp_f32<0x140>(p);
        p += __shfl_xor(p, 16);
        const int row = 32 * mt + frag_row(r, kk);
        if (i == 0 && row < Sc) block_norms[(size_t)(s0 + 64 * sc + row) * K + k] = sqrtf(p);
      }
    }
  }
}

// ---- block norms from the Gram matrix of a task's residuals (round 4) ---------------------------------------------------------
// ||sum_t m_t r_t||^2 = m^T (R R^T) m: a task (image, cluster) with n <= 32 tokens needs the n x n Gram matrix G of its token
// residuals -- ONE 32 x 32 accumulator tile instead of the 64 x 128 block sums of token_norms_kernel (128 accumulator registers,
// eight fp32 MFMAs per pair of tokens: that kernel is bound by the fp32 matrix pipe, 8 B/clk/CU of tokens) -- and, per segment,
// a bit-masked quadratic form over G.  G runs on the 16-bit pipe: the residuals are split r * xscale = h + l + e exactly as the
// planes want them (|e| <= 2^-22 |r|), G += l h^T + h l^T + h h^T with the A and B fragments being the SAME registers
// (v_mfma_f32_32x32x16_f16: lane (i, kk) holds token i, columns 8 kk .. 8 kk + 7 of a 16-column k-step), and those fragments
// are, byte for byte, the planes' 16-byte chunks.  One wave per task; tokens stream through a two-slot wave-private DMA queue,
// 32 tokens x 32 columns (4 KiB, four pieces of 8 tokens x 128 B, source-side swizzled: chunk ^ ((token >> 1) & 7)) per step.
// Tasks with more than 64 tokens of one image in one cluster stay with token_norms_kernel (skip_le).
// T = 2: tasks of 33 .. 64 tokens -- two token tiles, the three Gram tiles G00, G01, G11 (G10 = G01^T is not formed), 8 KiB per
// step and queue slot, two waves per workgroup.
constexpr int GNK_SLOT = 4096;   // one token tile of a step: 32 tokens x 128 B

template <int T, int QS>   // token tiles of a task, slots of the DMA queue (steps in flight + the one being read)
__global__ __launch_bounds__(T == 1 ? 256 : 128, 2) void gram_norms_kernel(const float* __restrict__ Xt, const float* __restrict__ rnorm,
                                                                         const int32_t* __restrict__ tok_order,
                                                                         const int32_t* __restrict__ lab_off,
                                                                         const uint64_t* __restrict__ colmask,
                                                                         const float* __restrict__ C,
                                                                         const int32_t* __restrict__ seg_off, int N, int D, int K, int SC,
                                                                         float* __restrict__ block_norms, float xscale,
                                                                         _Float16* __restrict__ h1, _Float16* __restrict__ h2,
                                                                         const int32_t* __restrict__ rowbase, int safe_waits,
                                                                         uint8_t* __restrict__ redo_task) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  constexpr int WAVES = T == 1 ? 4 : 2;
  constexpr int NG = T * (T + 1) / 2;   // Gram tiles held: (0,0) | (0,0), (0,1), (1,1)
  const int b = blockIdx.y;
  const int tid = threadIdx.x, l = tid & 63, i = l & 31, kk = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = (int)blockIdx.x * WAVES + w;
  if (k >= K) return;   // (no barriers in this kernel)
  const size_t wsz = (size_t)QS * T * GNK_SLOT + (size_t)D * 4;
  unsigned char* qbase = smem + (size_t)w * wsz;                      // [QS][T x 4 KiB] DMA queue
  float* cl = reinterpret_cast<float*>(qbase + QS * T * GNK_SLOT);    // [D] centre k
  const int o0 = lab_off[(size_t)b * (K + 1) + k];
  const int n = lab_off[(size_t)b * (K + 1) + k + 1] - o0;
  if (n > 32 * T || (T == 2 && n <= 32)) return;   // the other instantiation's / token_norms_kernel's task
  const int s0 = seg_off[b], S = seg_off[b + 1] - s0;
  if (n == 0) {   // an empty cluster: zero norms, no rows
    for (int s = l; s < S; s += 64) block_norms[(size_t)(s0 + s) * K + k] = 0.f;
    return;
  }
  const size_t tb = (size_t)b * N + o0;
  for (int d = 4 * l; d < D; d += 256) *reinterpret_cast<float4*>(cl + d) = *reinterpret_cast<const float4*>(C + (size_t)k * D + d);
  bool real[T];
  float rn[T];
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    real[tt] = 32 * tt + i < n;
    rn[tt] = real[tt] ? rnorm[tb + 32 * tt + i] : 0.f;
  }
  const float* Xb = Xt + (size_t)b * N * D;
  // DMA sources of this lane: piece pz = tokens 8 pz .. 8 pz + 7, lane -> token 8 pz + (l >> 3), LDS position l & 7
  const float* src[4 * T];
#pragma unroll
  for (int pz = 0; pz < 4 * T; ++pz) {
    const int jj = 8 * pz + (l >> 3);
    const int t = jj < n ? tok_order[tb + jj] : tok_order[tb];   // (rows beyond n: a valid address, never used)
    src[pz] = Xb + (size_t)t * D + 4 * ((l & 7) ^ ((jj >> 1) & 7));
  }
  const int steps = D >> 5;
  // only the pieces that hold a token are requested (NP of them, wave-uniform), and of the last one only its tokens' lanes: a task
  // holds ~24 of its tile's 32 rows on average
  const int NP = (n + 7) >> 3;
  auto issue = [&](int f) {
#pragma unroll
    for (int pz = 0; pz < 4 * T; ++pz)
      if (pz < NP) {   // (wave-uniform: the instruction is issued, and counted, exactly NP times per step)
        if (8 * pz + (l >> 3) < n)
          __builtin_amdgcn_global_load_lds((agg_gptr_t)(src[pz] + 32 * f), (agg_lptr_t)(qbase + (f % QS) * (T * GNK_SLOT) + pz * 1024), 16, 0, 0);
      }
  };
#pragma unroll
  for (int q = 0; q < QS; ++q)
    if (q < steps) issue(q);
  f32x16 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  size_t prow[T];
  unsigned pswz[T];
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    const int64_t rr = (int64_t)rowbase[(size_t)b * K + k] + 32 * tt + i;   // this lane's row of the grouped planes
    prow[tt] = (size_t)(rr >> 7) * (size_t)(D >> 5) * 4096 + (size_t)(rr & 127) * 32;
    pswz[tt] = (unsigned)sv_x3_swz(rr);
  }
  const unsigned fsw = (unsigned)((i >> 1) & 7);
  const unsigned a_tok = lds_addr(qbase) + (unsigned)i * 128u;
  for (int f = 0; f < steps; ++f) {
    // the DMAs of step f have landed: behind them were issued, for each of the QS - 1 steps in between, 4 T plane stores and NP
    // DMAs, and the 4 T stores of step f - 1 (loads and stores retire in issue order)
    // (safe_waits: the development switch debug_search = 7 -- everything, at every step; the test compares the two bit for bit)
    if (f >= QS && f + QS - 1 < steps && !safe_waits) tnk_wait_vm(QS * 4 * T + (QS - 1) * NP);   // (<= 23 by the helper: stricter beyond)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 x[T][4], c[4];
    {
      const unsigned ac = lds_addr(cl) + (unsigned)(32 * f + 8 * kk) * 4u;
      // chunks 2 kk, 2 kk + 1 (k-step 0) and 4 + 2 kk, 5 + 2 kk (k-step 1) of token i, at their swizzled positions
      const unsigned p0 = ((unsigned)(2 * kk) ^ fsw) << 4, p1 = ((unsigned)(2 * kk + 1) ^ fsw) << 4;
      const unsigned p2 = ((unsigned)(4 + 2 * kk) ^ fsw) << 4, p3 = ((unsigned)(5 + 2 * kk) ^ fsw) << 4;
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        const unsigned a0 = a_tok + (unsigned)(f % QS) * (T * GNK_SLOT) + (unsigned)tt * GNK_SLOT;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                     : "=&v"(x[tt][0]), "=&v"(x[tt][1]), "=&v"(x[tt][2]), "=&v"(x[tt][3])
                     : "v"(a0 + p0), "v"(a0 + p1), "v"(a0 + p2), "v"(a0 + p3)
                     : "memory");
      }
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\tds_read_b128 %3, %4 offset:80\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                   : "v"(ac)
                   : "memory");
#pragma unroll
      for (int tt = 0; tt < T; ++tt) asm volatile("" : "+v"(x[tt][0]), "+v"(x[tt][1]), "+v"(x[tt][2]), "+v"(x[tt][3]));   // (final behind the wait)
    }
    if (f + QS < steps) issue(f + QS);   // the slot has been read
    asm volatile("" ::: "memory");     // (the plane stores below stay behind these DMAs: the counted wait relies on the order)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 hh[T], ll[T];
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xv = x[tt][2 * ks + (e >> 2)][e & 3], cv = c[2 * ks + (e >> 2)][e & 3];
          const float fv = real[tt] ? fmaf(xv, rn[tt], -cv) * xscale : 0.f;
          hh[tt][e] = (_Float16)fv;
          ll[tt][e] = (_Float16)(fv - (float)hh[tt][e]);
        }
        // the planes' 16-byte chunk (row, columns 32 f + 16 ks + 8 kk ..): issued by every step (the counted waits above;
        // lane 0 of tile 0 is always a real token, tile 1 of a T = 2 task always holds one)
        const size_t ob = prow[tt] + (size_t)f * 4096 + (size_t)((((unsigned)(2 * ks + kk)) ^ pswz[tt]) << 3);
        if (real[tt]) {
          *reinterpret_cast<f16x8*>(h1 + ob) = hh[tt];
          *reinterpret_cast<f16x8*>(h2 + ob) = ll[tt];
        }
      }
      // G(a, b) += l_a h_b^T + h_a l_b^T + h_a h_b^T, small terms first
#pragma unroll
      for (int ta = 0, g = 0; ta < T; ++ta)
#pragma unroll
        for (int tb2 = ta; tb2 < T; ++tb2, ++g) {
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ll[ta], hh[tb2], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh[ta], ll[tb2], acc[g], 0, 0, 0);
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh[ta], hh[tb2], acc[g], 0, 0, 0);
        }
    }
  }
  // ---- per segment: m^T G m over this lane's 16 entries G(a, b)[frag_row(r, kk)][i] of every tile, then across the wave ------
  // CANCELLATION: the quadratic form carries ~n 2^-22 of the COVERED DIAGONAL sum_t m_t G_tt, not of its own value;
  // when a segment's residuals nearly cancel (m^T G m << the diagonal sum) that is no longer small against the norm it divides
  // by.  Such a task is flagged (redo_task) and token_norms_kernel recomputes it with the fp32 block sums, which have no such
  // term (1/16: relative error of the norm <= ~1e-4 at 64 tokens on this side of the threshold).
  float dgl[T];   // this lane's diagonal entry G_tt of token 32 tt + i, if the lane holds it (one of the two half-waves does)
#pragma unroll
  for (int tt = 0; tt < T; ++tt) {
    dgl[tt] = 0.f;
    const int gd = tt == 0 ? 0 : NG - 1;   // tiles (0,0) and (T-1,T-1)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (frag_row(r, kk) == i) dgl[tt] = acc[gd][r];
  }
  bool cancels = false;
  const float inv_x2 = 1.f / (xscale * xscale);
  const int SCb = (S + 63) >> 6;
  for (int sc = 0; sc < SCb; ++sc) {
    const int Sc = min(64, S - 64 * sc);
    uint64_t mt[T];
#pragma unroll
    for (int tt = 0; tt < T; ++tt) mt[tt] = real[tt] ? colmask[(tb + 32 * tt + i) * SC + sc] : 0ull;   // (lanes 32..63 repeat lanes 0..31)
    for (int s = 0; s < Sc; ++s) {
      uint32_t W[T];   // bit t: token 32 tt + t covered by segment s
      bool any = false;
#pragma unroll
      for (int tt = 0; tt < T; ++tt) {
        W[tt] = (uint32_t)__builtin_amdgcn_ballot_w64(((mt[tt] >> s) & 1ull) != 0ull);
        any |= W[tt] != 0u;
      }
      float p = 0.f;
      if (any) {   // (wave-uniform)
#pragma unroll
        for (int ta = 0, g = 0; ta < T; ++ta)
#pragma unroll
          for (int tb2 = ta; tb2 < T; ++tb2, ++g) {
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) q += ((W[ta] >> frag_row(r, kk)) & 1u) ? acc[g][r] : 0.f;
            q = ((W[tb2] >> i) & 1u) ? q : 0.f;
            p += (ta == tb2) ? q : 2.f * q;   // the off-diagonal tile stands for its transpose as well
          }
        p += dpp_f32<0xB1>(p);
        p += dpp_f32<0x4E>(p);
        p += dpp_f32<0x141>(p);
        p += dpp_f32<0x140>(p);
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        float dg = 0.f;   // the covered diagonal
#pragma unroll
        for (int tt = 0; tt < T; ++tt) dg += ((W[tt] >> i) & 1u) ? dgl[tt] : 0.f;
        dg += dpp_f32<0xB1>(dg);
        dg += dpp_f32<0x4E>(dg);
        dg += dpp_f32<0x141>(dg);
        dg += dpp_f32<0x140>(dg);
        dg += __shfl_xor(dg, 16);
        dg += __shfl_xor(dg, 32);
        cancels = cancels || (p < 0.0625f * dg);
      }
      if (l == 0) block_norms[(size_t)(s0 + 64 * sc + s) * K + k] = sqrtf(fmaxf(p, 0.f) * inv_x2);
    }
  }
  if (cancels && l == 0 && redo_task) redo_task[(size_t)b * K + k] = 1;   // (wave-uniform: p and dg are wave sums)
}

int sv_launch_token_norms(segvlad_ctx* ctx, const float* xt, const uint64_t* colmask, const float* centres, int K, int D,
                          const int32_t* seg_off_dev, int B, int N, int SC, float* block_norms, float xscale, uint16_t* h1,
                          uint16_t* h2, const int32_t* rowbase, int64_t dummy_row) {
  if (D % 4) return ctx->fail(SEGVLAD_ERR_ARG, "token_norms: D=%d must be a multiple of 4", D);
  const int Dpad = (D + 127) & ~127;
  const size_t lds = (size_t)TNK_WAVES * ((size_t)TNK_QD * 1024 + (size_t)TNK_LCAP * 16 + (size_t)Dpad * 4);
  if (lds > 160 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "token_norms: D=%d needs %zu B of LDS", D, lds);
  // tasks of <= 64 tokens: the Gram kernels (option tnk_gram, default on; D a multiple of 32); the rest: the block-sum kernels
  const bool gram = ctx->opt.tnk_gram != 0 && (D % 32) == 0;
  const int skip_le = gram ? 64 : -1;
  uint8_t* redo_task = nullptr;   // [B][K]: Gram tasks with a cancelling segment, recomputed by the block-sum kernel below
  if (gram) {
    SV_HIP(ctx->s_tnk_redo.reserve((size_t)B * K));
    redo_task = ctx->s_tnk_redo.as<uint8_t>();
    SV_HIP(hipMemsetAsync(redo_task, 0, (size_t)B * K, ctx->stream));
    // the two-tile tasks are few (6 % of the tokens at 24 per cluster): their launch is one task's latency with most of the chip
    // idle -- on the context's side stream it runs beside the one-tile launch instead of in front of it (tnk_fork)
    const bool fork = N > 32 && ctx->opt.tnk_fork != 0;
    if (fork) SV_TRY(sv_fork_side(ctx));
    for (int T = 2; T >= 1; --T) {
      if (T == 2 && N <= 32) continue;
      const int waves = T == 1 ? 4 : 2;
      const int qs = T == 1 ? 3 : 2;   // (three slots of 4 KiB, or two of 8)
      const size_t glds = (size_t)waves * ((size_t)qs * T * GNK_SLOT + (size_t)D * 4);
      auto gk = T == 1 ? gram_norms_kernel<1, 3> : gram_norms_kernel<2, 2>;
      if (glds > 64 * 1024) SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(gk), glds));
      hipLaunchKernelGGL(gk, dim3((K + waves - 1) / waves, B), dim3(64 * waves), glds, (T == 2 && fork) ? ctx->side : ctx->stream, xt,
                         ctx->s_rnsorted.as<float>(), ctx->s_tokorder.as<int32_t>(), ctx->s_laboff.as<int32_t>(), colmask, centres,
                         seg_off_dev, N, D, K, SC, block_norms, xscale, reinterpret_cast<_Float16*>(h1),
                         reinterpret_cast<_Float16*>(h2), rowbase, dummy_row < 0 ? 1 : 0, redo_task);
      SV_HIP(hipGetLastError());
    }
    if (fork) SV_TRY(sv_join_side(ctx));
  }
  for (int big = 0; big < 2; ++big) {
    if (big && N < TNK_LCAP) break;
    auto kern = big ? token_norms_kernel<true> : token_norms_kernel<false>;
    if (lds > 64 * 1024)
      SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
    hipLaunchKernelGGL(kern, dim3((K + TNK_WAVES - 1) / TNK_WAVES, B), dim3(64 * TNK_WAVES), lds, ctx->stream, xt,
                       ctx->s_rnsorted.as<float>(), ctx->s_tokorder.as<int32_t>(), ctx->s_laboff.as<int32_t>(), colmask, centres,
                       seg_off_dev, N, D, K, SC, Dpad, block_norms, xscale, reinterpret_cast<_Float16*>(h1),
                       reinterpret_cast<_Float16*>(h2), rowbase, dummy_row, skip_le, redo_task);
    SV_HIP(hipGetLastError());
  }
  return SEGVLAD_OK;
}

int sv_launch_aggregate(segvlad_ctx* ctx, const float* xt, const float* /*rnorm: label-grouped copy from prep*/,
                        const uint8_t* /*labels: grouped by prep*/,
                        const uint64_t* colmask, const float* centres, int K, int D, const int32_t* seg_off_dev,
                        const float* gscale, int B, int N, int SC, float* out, float* block_norms, const float* mean,
                        float xscale, uint16_t* h1, uint16_t* h2) {
  const int nwaves = (D + 127) / 128;
  if (nwaves > 12)
    return ctx->fail(SEGVLAD_ERR_LIMIT, "aggregate: D=%d exceeds the 1536-wide workgroup of this build", D);
  const int Ncap = (N + 2) & ~1;
  const int PW = nwaves * 2 + 1;
  size_t lds = (size_t)Ncap * 16 + (size_t)(64 * PW + 64) * sizeof(float) + 16 * sizeof(int);
  lds = (lds + 15) & ~(size_t)15;
  lds += (size_t)nwaves * 6 * 1024;   // DMA queue
  if (lds > 160 * 1024) return ctx->fail(SEGVLAD_ERR_LIMIT, "aggregate: N=%d tokens need %zu B of LDS (limit 160 KiB)", N, lds);
  auto kern = (h1 != nullptr) ? aggregate_kernel<true> : aggregate_kernel<false>;
  bool phases = false;
#ifdef SEGVLAD_ABLATIONS   // timing ablations / phase timing: development builds only
  if (const char* ab = getenv("SEGVLAD_AGG_ABL")) {
    if (atoi(ab) == 1) kern = aggregate_kernel<false, 1>;
    if (atoi(ab) == 2) kern = aggregate_kernel<false, 2>;
    if (atoi(ab) == 9) {
      kern = aggregate_kernel<false, 9>;
      phases = true;
      unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      SV_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sv_agg_phase_cycles), z, sizeof(z)));
    }
  }
#endif
  if (lds > 64 * 1024)
    SV_HIP(sv_max_dyn_lds(reinterpret_cast<const void*>(kern), (size_t)lds));
  int kpb = ctx->opt.agg_kpb;   // clusters per workgroup
  if (kpb < 1) kpb = 1;
  hipLaunchKernelGGL(kern, dim3((K + kpb - 1) / kpb, B), dim3(nwaves * 64), lds, ctx->stream, xt, ctx->s_rnsorted.as<float>(),
                     ctx->s_tokorder.as<int32_t>(),
                     ctx->s_laboff.as<int32_t>(), colmask, centres, seg_off_dev,
                     gscale, N, D, K, SC, Ncap, out, block_norms, mean, xscale, reinterpret_cast<_Float16*>(h1),
                     reinterpret_cast<_Float16*>(h2), kpb);
  SV_HIP(hipGetLastError());
#ifdef SEGVLAD_ABLATIONS
  if (phases) {
    unsigned long long c8[8];
    SV_HIP(hipStreamSynchronize(ctx->stream));
    SV_HIP(hipMemcpyFromSymbol(c8, HIP_SYMBOL(sv_agg_phase_cycles), sizeof(c8)));
    double tot = 0;
    for (int q = 0; q < 8; ++q) tot += (double)c8[q];
    fprintf(stderr, "[aggregate phases] lists %.1f%% main %.1f%% | partial norms %.1f%% barrier %.1f%% row sums %.1f%% barrier %.1f%% | stores %.1f%% barrier %.1f%% (%.0f cycles per (k, image))\n",
            100 * c8[0] / tot, 100 * c8[1] / tot, 100 * c8[5] / tot, 100 * c8[6] / tot, 100 * c8[7] / tot, 100 * c8[2] / tot,
            100 * c8[3] / tot, 100 * c8[4] / tot, tot / ((double)K * B));
  }
#endif
  (void)phases;
  return SEGVLAD_OK;
}
