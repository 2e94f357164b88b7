// Device helpers shared by the kNN translation units (knn_filter_kernels.hip, small_pass_kernels.hip): the order-preserving
// float <-> key map, the workgroup bitonic sort of (key << 32 | id) words, DPP wave reductions.  Moved out of
// knn_filter_kernels.hip in round 6, unchanged.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

__device__ __forceinline__ uint32_t f2key_(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f_(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// ascending bitonic sort of n (a power of two) 64-bit words in LDS by a 256-thread workgroup
__device__ __forceinline__ void bitonic64(uint64_t* a, int n, int tid) {
  for (int size = 2; size <= n; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (n >> 1); t += 256) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const uint64_t x = a[lo], y = a[hi];
        if ((y < x) == up) {
          a[lo] = y;
          a[hi] = x;
        }
      }
    }
  }
  __syncthreads();
}

// sum over the 64 lanes, returned wave-uniform: quad swaps, half-row and row mirrors (DPP: no LDS crossbar), then the four
// row sums through readlane
__device__ __forceinline__ uint32_t wave_sum_u32_(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);   // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);   // row_mirror
  return (uint32_t)(__builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
                    __builtin_amdgcn_readlane(x, 48));
}

__device__ __forceinline__ uint32_t wave_min_u32_(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
  return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_max_u32_(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false));
  return max(max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             max((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

