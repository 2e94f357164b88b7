// Micro-benchmark: sustained v_mfma_f32_32x32x16_f16 rate (registers only; optional LDS fragment reads / barriers),
// to locate the ceiling the kNN filter / PCA kernels can reach.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// MODE 0: MFMA only (2x4 accumulators, operands fixed in registers)
// MODE 1: + 6 ds_read_b128 per 8 MFMAs (fragments re-read from LDS, software prefetch of the next step)
// MODE 2: MODE 1 + s_barrier every 4 steps
// MODE 3: MODE 2 + 8 global->LDS DMA pieces (1 KiB each) per wave and 32 MFMAs, counted vmcnt before the barrier
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int MODE, int RND>
__global__ __launch_bounds__(512) void k(float* out, int iters, const unsigned char* src, size_t src_bytes, int foot) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int j = tid; j < 65536 / 2; j += 512) {
    unsigned h = (j * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    reinterpret_cast<_Float16*>(lds)[j] = RND ? (_Float16)(((int)(h & 2047) - 1024) * (1.f / 1024.f)) : (_Float16)(0.001f * (j & 15));
  }
  __syncthreads();
  f32x16 acc[2][4];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f16x8 fa[2], fb[4];
  const unsigned char* base = lds + (tid >> 6) * 4096 + l * 16;
  for (int t = 0; t < 2; ++t) fa[t] = *reinterpret_cast<const f16x8*>(base + t * 1024);
  for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const f16x8*>(base + 2048 + t * 1024 - (t == 3 ? 2048 : 0));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = st0;
  for (int it = 0; it < iters; ++it) {
    f16x8 na[2], nb[4];
    if (MODE >= 1) {
      const unsigned char* p = lds + ((it * 6144 + (tid >> 6) * 4096) & 32767) + l * 16;
      for (int t = 0; t < 2; ++t) na[t] = *reinterpret_cast<const f16x8*>(p + t * 1024);
      for (int t = 0; t < 4; ++t) nb[t] = *reinterpret_cast<const f16x8*>(p + 8192 + t * 1024);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = MFMA(fa[a], fb[b], acc[a][b]);
    if (MODE >= 1) {
      for (int t = 0; t < 2; ++t) fa[t] = na[t];
      for (int t = 0; t < 4; ++t) fb[t] = nb[t];
    }
    if (MODE == 3) {   // two pieces per step: 8 per 4 steps; lands in the upper 64 KiB (never read: pure traffic)
      const size_t off = foot == 0 ? (size_t)w * 32768 + ((size_t)it * 2048) % 32768 + l * 16   /* 256 KiB shared: L2 hits */
                                   : (((size_t)blockIdx.x * 8 + w) * 131072 + (size_t)it * 2048 + l * 16) % (src_bytes - 4096);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + off), (lptr_t)(lds + 65536 + w * 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + off + 1024), (lptr_t)(lds + 65536 + w * 2048 + 1024), 16, 0, 0);
    }
    if (MODE == 6) {   // MODE 3's two pieces per step as buffer_load ... lds: SGPR resource + one 32-bit VGPR offset (no 64-bit pointer math)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
      const unsigned off = foot == 0 ? (unsigned)(w * 32768 + (it * 2048) % 32768 + l * 16)
                                     : (unsigned)((((size_t)blockIdx.x * 8 + w) * 131072 + (size_t)it * 2048 + l * 16) % (src_bytes - 4096));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + 65536 + w * 2048), 16, off, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + 65536 + w * 2048 + 1024), 16, off, 0, 1024, 0);
    }
    if (MODE == 4) {   // same bytes through registers: 2 global_load_dwordx4 per step, ds_write_b128 one step later
      const size_t off = foot == 0 ? (size_t)w * 32768 + ((size_t)it * 2048) % 32768 + l * 16   /* 256 KiB shared: L2 hits */
                                   : (((size_t)blockIdx.x * 8 + w) * 131072 + (size_t)it * 2048 + l * 16) % (src_bytes - 4096);
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<f32x4*>(lds + 65536 + w * 2048 + l * 16) = st0;
      *reinterpret_cast<f32x4*>(lds + 65536 + w * 2048 + 1024 + l * 16) = st1;
      st0 = *reinterpret_cast<const f32x4*>(src + off);
      st1 = *reinterpret_cast<const f32x4*>(src + off + 1024);
    }
    if (MODE >= 2 && (it & 3) == 3) {
      if (MODE == 3 || MODE == 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

// MODE 5: the same flops as MODE 0 through v_mfma_f32_16x16x32_f16 (4 x 4 accumulator tiles of 16 x 16; deeper k per
// instruction: a quarter of the accumulator traffic per flop) -- does the power-limited rate depend on the MFMA shape?
template <int RND>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, l = tid & 63;
  for (int j = tid; j < 65536 / 2; j += 512) {
    unsigned h = (j * 2654435761u) ^ (blockIdx.x * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    // RND 2: random values with the low 3 mantissa bits zero (8 significant bits): does the power-limited rate depend on the
    // operands' mantissa width?
    reinterpret_cast<_Float16*>(lds)[j] = RND == 2 ? (_Float16)(((int)(h & 2040) - 1024) * (1.f / 1024.f))
                                        : RND ? (_Float16)(((int)(h & 2047) - 1024) * (1.f / 1024.f)) : (_Float16)(0.001f * (j & 15));
  }
  __syncthreads();
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  f16x8 fa[4], fb[4];
  const unsigned char* base = lds + (tid >> 6) * 8192 + l * 16;
  for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const f16x8*>(base + t * 1024);
  for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const f16x8*>(base + 4096 + t * 1024);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
  }
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}
// fp32 MFMA, the two shapes (assignment scores, masked aggregation, P-space aggregation, the fp32 filter): same flops per
// iteration (8 x 32x32x2 = 16 x 16x16x4 = 32768 flop per wave-iteration... x 2)
template <int SHAPE>
__global__ __launch_bounds__(512) void kf32(float* out, int iters) {
  const int tid = threadIdx.x;
  unsigned h = (tid * 2654435761u) ^ (blockIdx.x * 40503u);
  h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
  float fa[4], fb[4];
  for (int t = 0; t < 4; ++t) { fa[t] = ((int)((h >> (t * 3)) & 2047) - 1024) * (1.f / 1024.f); fb[t] = ((int)((h >> (t * 5 + 1)) & 2047) - 1024) * (1.f / 1024.f); }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  float s = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  } else {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  }
  out[blockIdx.x * 512 + tid] = s;
}
template <int SHAPE>
void runf32(const char* name, float* out, int iters, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((kf32<SHAPE>), dim3(blocks), dim3(512), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((kf32<SHAPE>), dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 8 * iters * (SHAPE == 0 ? 8 * 4096.0 : 16 * 2048.0);
  printf("%-44s blocks=%d iters=%d: %.3f ms -> %.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
}

template <int RND>
void run16(const char* name, float* out, int iters, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k16<RND>), dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k16<RND>), dim3(blocks), dim3(512), 65536, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 8 * iters * 16 * 16384.0;
  printf("%-44s blocks=%d iters=%d: %.3f ms -> %.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
}

template <int MODE, int RND>
void run(const char* name, float* out, int iters, int blocks, const unsigned char* src, size_t src_bytes, int foot = 0) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, RND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k16<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t ldsb = MODE >= 3 ? 65536 + 16384 : 65536;   // (modes 3, 4, 6: a landing zone for the pieces)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, RND>), dim3(blocks), dim3(512), ldsb, 0, out, iters, src, src_bytes, foot);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, RND>), dim3(blocks), dim3(512), ldsb, 0, out, iters, src, src_bytes, foot);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 8 * iters * 8 * 32768.0;
  printf("%-44s blocks=%d iters=%d: %.3f ms -> %.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  const size_t src_bytes = (size_t)1 << 30;
  unsigned char* src; hipMalloc(&src, src_bytes); hipMemset(src, 0x3c, src_bytes);
  for (int blocks : {256, 1024}) {
    run<0, 0>("mfma only (smooth data)", out, iters, blocks, src, src_bytes);
    run<0, 1>("mfma only (random data)", out, iters, blocks, src, src_bytes);
    run16<1>("mfma 16x16x32 only (random data)", out, iters, blocks);
    run16<2>("mfma 16x16x32 only (random, 8-bit mantissas)", out, iters, blocks);
    runf32<0>("fp32 mfma 32x32x2 only (random data)", out, iters / 8, blocks);
    runf32<1>("fp32 mfma 16x16x4 only (random data)", out, iters / 8, blocks);
    run<1, 1>("mfma + 6 ds_read_b128 / 8 mfma (random)", out, iters, blocks, src, src_bytes);
    run<2, 1>("  + s_barrier / 32 mfma", out, iters, blocks, src, src_bytes);
    run<3, 1>("  + 8 DMA pieces / 32 mfma + vmcnt", out, iters, blocks, src, src_bytes);
    run<6, 1>("  + 8 DMA pieces as buffer_load lds", out, iters, blocks, src, src_bytes);
    run<4, 1>("  + 8 (global_load x4 -> ds_write_b128)", out, iters, blocks, src, src_bytes);
    run<3, 1>("  + 8 DMA pieces, streaming 1 GiB", out, iters, blocks, src, src_bytes, 1);
  }
  return 0;
}
