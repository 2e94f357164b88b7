// Micro-benchmark: global -> LDS DMA (global_load_lds_dwordx4) throughput per CU, to find the operand-delivery ceiling of
// the 256x256 GEMM structures (kNN filter, PCA):   hipcc --offload-arch=gfx950 -O3 -o lds_dma_bw lds_dma_bw.hip
//   one 512-thread workgroup per CU; every wave issues `pieces` 1-KiB pieces per round into its own LDS slot, waits
//   until at most `keep` of its pieces are outstanding, optionally joins a workgroup barrier, and loops.
//   FOOT 0: all workgroups of an XCD read the same 2 MiB (L2 hits); FOOT 1: every workgroup streams its own region
//   (HBM / MALL misses); FOOT 2: half and half (the filter's A / B mix).
//   FOOT 3 / 4 / 5: the single-image filter's database stream -- a workgroup owns 128 consecutive 2-KiB rows (256 KiB) at a
//   time and walks them in k-tiles: each 1-KiB piece = 8 rows x 128 B (FOOT 3), 4 rows x 256 B (4) or 2 rows x 512 B (5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int KEEP>
__device__ __forceinline__ void wait_keep() {
  if (KEEP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (KEEP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (KEEP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (KEEP == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
}

template <int PIECES, int KEEP, int BAR, int FOOT>
__global__ __launch_bounds__(512) void k(const unsigned char* src, size_t src_bytes, int rounds, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7;
  // slots: each wave owns (PIECES + KEEP) KiB of LDS, pieces rotate through it
  constexpr int SLOTS = PIECES + KEEP;
  unsigned char* mine = lds + (size_t)w * SLOTS * 1024;
  size_t pos = 0;
  int slot = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      size_t off;
      const bool shared = FOOT == 0 || (FOOT == 2 && (p & 1) == 0);
      if (shared) off = (size_t)xcd * (2u << 20) + ((pos + (size_t)w * 65536) & ((2u << 20) - 1));
      else off = (16u << 20) + (((size_t)blockIdx.x * 8 + w) * (8u << 20) + pos) % (src_bytes - (32u << 20));
      size_t lane_off = l * 16;
      if (FOOT >= 3) {
        // this workgroup's piece counter (all waves in step): 256 pieces per 256-KiB tile
        constexpr int RB = FOOT == 3 ? 128 : FOOT == 4 ? 256 : 512;     // bytes of one row in a piece
        constexpr int RPP = 1024 / RB;                                   // rows per piece
        constexpr int LPR = RB / 16;                                     // lanes per row
        const size_t pc = pos / 1024 * 8 + w;                            // piece index of the workgroup's stream
        const size_t tile = pc / 256, in = pc % 256;                     // tile = 128 rows x 2 KiB
        const size_t kt = in / (128 / RPP), rp = in % (128 / RPP);       // k-tile, row group
        const size_t tbase = (16u << 20) + ((tile * 256 + blockIdx.x) * (256u << 10)) % (src_bytes - (32u << 20));
        off = tbase + (rp * RPP + l / LPR) * 2048 + kt * RB;
        lane_off = (l % LPR) * 16;
      }
      if (FOOT == 6) {
        // k-split stream: the workgroup's tile = 16 consecutive 2-KiB rows (32 KiB, contiguous); wave w takes bytes
        // [256 w, 256 w + 256) of every row: 4 pieces of 4 rows x 256 B; tiles dealt round-robin over the workgroups
        const size_t pc = pos / 1024;                    // this wave's piece counter
        const size_t tile = pc / 4, rp = pc % 4;
        off = (16u << 20) + ((tile * gridDim.x + blockIdx.x) * (32u << 10)) % (src_bytes - (32u << 20)) + (rp * 4 + l / 16) * 2048 + w * 256;
        lane_off = (l % 16) * 16;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)(src + off + lane_off), (lptr_t)(mine + slot * 1024), 16, 0, 0);
      pos += 1024;
      slot = slot + 1 == SLOTS ? 0 : slot + 1;
    }
    wait_keep<KEEP>();
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned*>(lds + 64);
}

template <int PIECES, int KEEP, int BAR, int FOOT>
static void run(const unsigned char* src, size_t bytes, unsigned* sink, const char* what) {
  const int rounds = 4096 / PIECES * 4;
  const size_t lds = (size_t)8 * (PIECES + KEEP) * 1024;
  auto kern = k<PIECES, KEEP, BAR, FOOT>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, bytes, rounds / 8, sink);
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("%-44s launch failed (LDS %zu)\n", what, lds); return; }
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, src, bytes, rounds, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double tot = 256.0 * 8 * rounds * PIECES * 1024.0;
  printf("%-44s pieces/round %2d keep %2d barrier %d: %7.3f ms  %6.2f TB/s  %5.1f GB/s per CU  (~%4.1f B/clk/CU at 2.1 GHz)\n", what, PIECES, KEEP,
         BAR, ms, tot / ms / 1e9, tot / ms / 1e6 / 256, tot / ms / 1e6 / 256 / 2.1);
}

int main() {
  const size_t bytes = (size_t)20 << 30;
  unsigned char* src;
  unsigned* sink;
  if (hipMalloc(&src, bytes) != hipSuccess) return 1;
  hipMalloc(&sink, 4096);
  hipMemset(src, 1, bytes);
  run<8, 0, 1, 0>(src, bytes, sink, "L2-shared, drain each round");
  run<8, 8, 1, 0>(src, bytes, sink, "L2-shared, one round in flight");
  run<8, 16, 1, 0>(src, bytes, sink, "L2-shared, two rounds in flight");
  run<8, 16, 0, 0>(src, bytes, sink, "L2-shared, two rounds in flight, no barrier");
  run<4, 8, 1, 0>(src, bytes, sink, "L2-shared (A only shape)");
  run<8, 0, 1, 1>(src, bytes, sink, "streaming, drain each round");
  run<8, 8, 1, 1>(src, bytes, sink, "streaming, one round in flight");
  run<8, 16, 1, 1>(src, bytes, sink, "streaming, two rounds in flight");
  run<8, 16, 0, 1>(src, bytes, sink, "streaming, two rounds in flight, no barrier");
  run<8, 8, 1, 2>(src, bytes, sink, "half shared / half streaming, one round in flight");
  run<8, 16, 1, 2>(src, bytes, sink, "half shared / half streaming, two rounds");
  run<16, 16, 1, 2>(src, bytes, sink, "half/half, 16 pieces per round, one round");
  run<8, 8, 1, 3>(src, bytes, sink, "row tiles, 128-B row pieces, one round in flight");
  run<8, 8, 1, 4>(src, bytes, sink, "row tiles, 256-B row pieces, one round in flight");
  run<8, 8, 1, 5>(src, bytes, sink, "row tiles, 512-B row pieces, one round in flight");
  run<4, 16, 0, 3>(src, bytes, sink, "row tiles, 128-B row pieces, 16 in flight, no barrier");
  run<4, 16, 0, 5>(src, bytes, sink, "row tiles, 512-B row pieces, 16 in flight, no barrier");
  run<4, 16, 0, 1>(src, bytes, sink, "streaming (1-KiB contiguous), 16 in flight, no barrier");
  run<4, 4, 0, 6>(src, bytes, sink, "k-split tiles (4 rows x 256 B), 4-8 in flight, no barrier");
  run<4, 8, 0, 6>(src, bytes, sink, "k-split tiles (4 rows x 256 B), 8-12 in flight, no barrier");
  run<4, 16, 0, 6>(src, bytes, sink, "k-split tiles (4 rows x 256 B), 16-20 in flight, no barrier");
  run<4, 16, 1, 6>(src, bytes, sink, "k-split tiles (4 rows x 256 B), 16-20 in flight, barrier");
  return 0;
}
