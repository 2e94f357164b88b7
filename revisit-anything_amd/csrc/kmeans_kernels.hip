// Vocabulary k-means, the centroid update of one Lloyd iteration (reference: vlad_c_centers_pt_gen.py:86-158,
// utilities.py:749-791 -> fast_pytorch_kmeans.KMeans(mode='cosine'): centres = plain means of the assigned unit tokens).
//
// Round 6: a kernel of its own.  Rounds 3-5 recovered the per-cluster sums from the NORMALISED output of the
// segment-VLAD kernels (S_k = out_k * ||V_k|| * sqrt(#blocks) + n_k C_k): exact algebra, but a cluster whose residual sum is tiny
// against n_k C_k (tokens tightly around their centre -- the converged state) gets its sum back through a cancellation, and the
// K * D-wide descriptor of every image is written and read for nothing.
//
//   centroid_sums_kernel    one workgroup per (image, chunk of CH columns): walks the image's tokens IN ORDER out of the token-major
//                           copy the assignment pass leaves behind (Xt [B][N][D], 1 / ||x|| beside it), thread c adds
//                           x[t][c] / ||x_t|| to its own LDS slot [label_t][c] -- no atomics, one fixed order: deterministic --
//                           and writes the image's partial sums [K][CH] (fp32: at most N terms of magnitude <= 1 each) and, from
//                           the workgroup of chunk 0, the image's label histogram.  Reads every token once: HBM-bound.
//   centroid_reduce_kernel  sums the images' partials in image order in fp64 -> sums [K][D] fp64, counts [K] int64.
#include "ctx.h"

__global__ __launch_bounds__(256) void centroid_sums_kernel(const float* __restrict__ xt, const float* __restrict__ rnorm,
                                                            const uint8_t* __restrict__ labels, int N, int D, int K, int CH,
                                                            float* __restrict__ part, int32_t* __restrict__ part_cnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* acc = reinterpret_cast<float*>(smem);                       // [K][CH]
  __shared__ int32_t s_cnt[256];
  const int b = blockIdx.y, c0 = blockIdx.x * CH, tid = threadIdx.x;
  const int cw = min(CH, D - c0);
  for (int j = tid; j < K * CH; j += 256) acc[j] = 0.f;
  s_cnt[tid] = 0;
  __syncthreads();
  const float* xb = xt + (size_t)b * N * D + c0;
  const float* rb = rnorm + (size_t)b * N;
  const uint8_t* lb = labels + (size_t)b * N;
  // thread `tid` owns column c0 + tid (CH <= 256): its LDS slots are touched by nobody else -> plain read-modify-write, tokens in order.
  // Eight tokens' loads are in flight per round trip.
  if (tid < cw) {
    for (int t0 = 0; t0 < N; t0 += 8) {
      float v[8], r[8];
      int l[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = min(t0 + u, N - 1);
        v[u] = xb[(size_t)t * D + tid];
        r[u] = rb[t];
        l[u] = lb[t];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t0 + u < N) acc[l[u] * CH + tid] += v[u] * r[u];   // x^_t = x_t * (1 / ||x_t||): the assignment pass's own normalisation
    }
  }
  if (blockIdx.x == 0) {   // the image's label histogram (every thread a strided share of the tokens)
    for (int t = tid; t < N; t += 256) atomicAdd(&s_cnt[lb[t]], 1);
  }
  __syncthreads();
  float* pb = part + ((size_t)b * K) * D + c0;
  for (int j = tid; j < K * CH; j += 256) {
    const int k = j / CH, c = j - k * CH;
    if (c < cw) pb[(size_t)k * D + c] = acc[j];
  }
  if (blockIdx.x == 0 && tid < K) part_cnt[(size_t)b * K + tid] = s_cnt[tid];
}

__global__ __launch_bounds__(256) void centroid_reduce_kernel(const float* __restrict__ part, const int32_t* __restrict__ part_cnt, int B,
                                                              int K, int D, double* __restrict__ sums, int64_t* __restrict__ counts) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kd = (int64_t)K * D;
  if (j < kd) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += (double)part[(size_t)b * kd + j];   // image order: deterministic
    sums[j] += s;
  }
  if (j < K) {
    int64_t c = 0;
    for (int b = 0; b < B; ++b) c += part_cnt[(size_t)b * K + j];
    counts[j] += c;
  }
}

// sums [K][D] fp64 and counts [K] int64 are ACCUMULATED into (the caller zeroes them in front of the first batch of an iteration)
int sv_launch_centroid_sums(segvlad_ctx* ctx, const float* xt, const float* rnorm, const uint8_t* labels, int B, int N, int D, int K,
                            double* sums, int64_t* counts) {
  if (B <= 0) return SEGVLAD_OK;
  if (K < 1 || K > 256) return ctx->fail(SEGVLAD_ERR_LIMIT, "kmeans: K=%d (labels are bytes)", K);
  int CH = 256;
  while ((size_t)K * CH * 4 > 64 * 1024) CH >>= 1;   // [K][CH] fp32 in <= 64 KiB of LDS
  SV_HIP(ctx->s_km_part.reserve((size_t)B * K * D * sizeof(float)));
  SV_HIP(ctx->s_km_cnt.reserve((size_t)B * K * sizeof(int32_t)));
  hipLaunchKernelGGL(centroid_sums_kernel, dim3((D + CH - 1) / CH, B), dim3(256), (size_t)K * CH * 4, ctx->stream, xt, rnorm, labels, N, D,
                     K, CH, ctx->s_km_part.as<float>(), ctx->s_km_cnt.as<int32_t>());
  SV_HIP(hipGetLastError());
  const int64_t kd = (int64_t)K * D;
  hipLaunchKernelGGL(centroid_reduce_kernel, dim3((unsigned)((kd + 255) / 256)), dim3(256), 0, ctx->stream, ctx->s_km_part.as<float>(),
                     ctx->s_km_cnt.as<int32_t>(), B, K, D, sums, counts);
  SV_HIP(hipGetLastError());
  return SEGVLAD_OK;
}
