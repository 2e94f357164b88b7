// Micro-benchmark for the deep-row exact refinement (refine_exact_wide_kernel): 64 candidate rows per workgroup, every step
// fetches PIECE bytes of each row (coalesced: PIECE / 16 lanes per row), DEPTH steps in flight in registers.
//   rows are `pitch` bytes apart times a random row id (pitch = 393216 = a raw K*D = 98 304-d fp32 row);
//   skew = 1: row r is r pieces behind row 0 (its address differs in the low bits as well).
// hipcc --offload-arch=gfx950 -O3 -o gather_bw gather_bw.hip ; ./gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int PIECE, int DEPTH>
__global__ __launch_bounds__(256) void k(const unsigned char* base, const unsigned* ids, size_t pitch, int nsteps, int skew, float* out) {
  constexpr int LPR = PIECE / 16;          // lanes per row piece
  constexpr int NP = 64 * LPR / 256;       // pieces per thread and step
  const int tid = threadIdx.x, seg = tid % LPR, r0 = tid / LPR;
  const unsigned char* src[NP];
#pragma unroll
  for (int u = 0; u < NP; ++u) src[u] = base + (size_t)ids[blockIdx.x * 64 + r0 + (256 / LPR) * u] * pitch + seg * 16;
  float4 g[DEPTH][NP];
  float acc = 0.f;
  auto off = [&](int s, int u) -> size_t {
    const int r = r0 + (256 / LPR) * u;
    int c = skew ? s - r : s;
    if (c < 0) c += nsteps;
    return (size_t)c * PIECE;
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int u = 0; u < NP; ++u) g[d][u] = *reinterpret_cast<const float4*>(src[u] + off(d, u));
  for (int s = 0; s + DEPTH <= nsteps; s += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int u = 0; u < NP; ++u) acc += g[d][u].x + g[d][u].w;
      if (s + DEPTH + d < nsteps) {
#pragma unroll
        for (int u = 0; u < NP; ++u) g[d][u] = *reinterpret_cast<const float4*>(src[u] + off(s + DEPTH + d, u));
      }
      __syncthreads();   // one barrier per step, as in the real kernel
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <int PIECE, int DEPTH>
void run(const unsigned char* base, const unsigned* ids, size_t pitch, int nsteps, int skew, float* out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PIECE, DEPTH>), dim3(blocks), dim3(256), 0, 0, base, ids, pitch, nsteps, skew, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<PIECE, DEPTH>), dim3(blocks), dim3(256), 0, 0, base, ids, pitch, nsteps, skew, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * 64 * nsteps * PIECE;
  printf("piece %4d B depth %d pitch %zu skew %d blocks %d: %.2f ms -> %.2f TB/s\n", PIECE, DEPTH, pitch, skew, blocks, ms, bytes / ms / 1e9);
}

int main() {
  const size_t nrows = 40000, maxpitch = 393216 + 4096;
  unsigned char* base; hipMalloc(&base, nrows * maxpitch); hipMemset(base, 1, nrows * maxpitch);
  const int blocks = 2048;
  std::vector<unsigned> h(blocks * 64);
  srand(7);
  for (auto& x : h) x = (unsigned)(rand() % nrows);
  unsigned* ids; hipMalloc(&ids, h.size() * 4); hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  float* out; hipMalloc(&out, blocks * 256 * 4);
  for (size_t pitch : {(size_t)393216, (size_t)393216 + 512, (size_t)393216 + 4096}) {
    for (int skew : {0, 1}) {
      run<512, 1>(base, ids, pitch, 768 / 4, skew, out, blocks);
      run<512, 2>(base, ids, pitch, 768 / 4, skew, out, blocks);
      run<512, 4>(base, ids, pitch, 768 / 4, skew, out, blocks);
      run<1024, 2>(base, ids, pitch, 384 / 4, skew, out, blocks);
    }
  }
  return 0;
}
