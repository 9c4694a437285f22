// Internal helpers shared by the gfx950 kernels (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#define OBMAN_ABI_VERSION 1
#define OBMAN_WAVE 64

#define OBMAN_LAUNCH_CHECK()                       \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int obman_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Squared distance with a pinned evaluation order (no contraction differences between the
// streaming pass and the index-resolution pass): fma(dz,dz, fma(dy,dy, dx*dx)).
__device__ __forceinline__ float obman_dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// Wave-level sum over 64 lanes (all lanes get the result).
__device__ __forceinline__ float obman_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float obman_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
