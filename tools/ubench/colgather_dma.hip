// Micro-benchmark behind DESIGN 7 "Describe" (round 6): could the Gram kernels read the ORIGINAL token layout [B][D][N] (N contiguous)
// instead of the token-major copy the assignment pass writes -- i.e. gather a task's ~24 token COLUMNS, 4 bytes per (row d, token),
// through global_load_lds_dword -- so that the copy (1.88 GB written + 1.78 GB read per 200 images) disappears?
// One wave per (image, cluster) task, 32 token slots, per step 32 D-rows x 32 tokens = 4 KiB landed in LDS by 16 DMA instructions of
// 64 lanes x 4 B whose 64 addresses lie in 64 different lines (the tokens of a cluster are scattered over the image's 1530).
//   xcd = 1: all 16 workgroups (4 waves each) of an image on ONE XCD (logical id = (bid % 8) * per + bid / 8), else round robin.
// hipcc --offload-arch=gfx950 -O3 -o colgather_dma colgather_dma.hip ; ./colgather_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(256) void k(const float* __restrict__ T, const int* __restrict__ tok, int B, int N, int D, int K, int xcd, int depth,
                                         float* __restrict__ out) {
  extern __shared__ unsigned char smem[];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  int bid = blockIdx.x;
  if (xcd) {
    const int per = (gridDim.x + 7) / 8;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= (int)gridDim.x) return;
  }
  const int wg_per_img = K / 4;
  const int b = bid / wg_per_img, kcl = (bid % wg_per_img) * 4 + w;
  unsigned char* q = smem + w * (4 * 4096);
  const float* Tb = T + (size_t)b * D * N;
  // lane l of DMA instruction p: token slot 2 p + (l >> 5), D-row l & 31 of the step
  const float* src[16];
#pragma unroll
  for (int p = 0; p < 16; ++p) src[p] = Tb + (size_t)(l & 31) * N + tok[((size_t)b * K + kcl) * 32 + 2 * p + (l >> 5)];
  const int steps = D / 32;
  float acc = 0.f;
  for (int f = 0; f < steps; ++f) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
      __builtin_amdgcn_global_load_lds((gptr_t)(src[p] + (size_t)32 * f * N), (lptr_t)(q + (f % depth) * 4096 + p * 256), 4, 0, 0);
    if (f % depth == depth - 1 || f == steps - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += reinterpret_cast<const float*>(q)[l] + reinterpret_cast<const float*>(q + 4096 * ((f) % depth))[l * 13 % 1024];
    }
  }
  out[(size_t)bid * 256 + tid] = acc;
}

int main() {
  const int B = 200, N = 1530, D = 1536, K = 64;
  float* T;
  hipMalloc(&T, (size_t)B * D * N * 4);
  hipMemset(T, 0, (size_t)B * D * N * 4);
  std::vector<int> tok((size_t)B * K * 32);
  srand(1);
  for (int b = 0; b < B; ++b) {   // a random partition of the tokens into K clusters, first 32 (or fewer: repeated) per cluster
    std::vector<int> perm(N);
    for (int i = 0; i < N; ++i) perm[i] = i;
    for (int i = N - 1; i > 0; --i) std::swap(perm[i], perm[rand() % (i + 1)]);
    for (int kk = 0; kk < K; ++kk)
      for (int j = 0; j < 32; ++j) tok[((size_t)b * K + kk) * 32 + j] = perm[(kk * 24 + j % 24) % N];
  }
  int* dtok;
  hipMalloc(&dtok, tok.size() * 4);
  hipMemcpy(dtok, tok.data(), tok.size() * 4, hipMemcpyHostToDevice);
  float* out;
  hipMalloc(&out, (size_t)B * 16 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int xcd = 0; xcd < 2; ++xcd)
    for (int depth : {1, 2, 4}) {
      const int grid = B * (K / 4);
      hipLaunchKernelGGL(k, dim3((grid + 7) / 8 * 8), dim3(256), 4 * 4 * 4096, 0, T, dtok, B, N, D, K, xcd, depth, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3((grid + 7) / 8 * 8), dim3(256), 4 * 4 * 4096, 0, T, dtok, B, N, D, K, xcd, depth, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("column gather by dword DMA, %d images x %d tasks x 24 tokens x %d rows: xcd-aware %d, %d steps in flight: %.3f ms (%.2f TB/s of useful token bytes)\n",
             B, K, D, xcd, depth, ms, (double)B * K * 24 * D * 4 / ms / 1e9);
    }
  return 0;
}
