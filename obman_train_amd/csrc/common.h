// Internal helpers shared by the gfx950 kernels (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#define OBMAN_ABI_VERSION 7
#define OBMAN_WAVE 64

#define OBMAN_LAUNCH_CHECK()                       \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int obman_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Per-device caches of one-time function attributes (hipFuncSetAttribute is per device): index by the calling thread's device.
constexpr int MAX_DEVICES = 16;
static inline int current_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev >= 0 && dev < MAX_DEVICES ? dev : 0;
}

// Squared distance with a pinned evaluation order (no contraction differences between the
// streaming pass and the index-resolution pass): fma(dz,dz, fma(dy,dy, dx*dx)).
__device__ __forceinline__ float obman_dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
}

// Lane exchange inside a 16-lane DPP row: the operand modifier of a VALU op (no LDS-crossbar round trip like ds_bpermute,
// whose ~80-cycle dependent latency dominated kernels that end in many reductions).
template <int CTRL>
__device__ __forceinline__ float obman_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// Wave-level sum over 64 lanes (all lanes get the result).  Butterfly inside each 16-lane row with DPP
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row sums are read as scalars and added in a fixed
// order: ((r0 + r1) + (r2 + r3)).  Deterministic; every lane sees the same bits.
__device__ __forceinline__ float obman_wave_sum(float v) {
  v += obman_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += obman_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += obman_dpp<0x141>(v);  // row_half_mirror
  v += obman_dpp<0x140>(v);  // row_mirror
  const int vi = __builtin_bit_cast(int, v);  // readlane is an integer builtin: pass the bits, not the value
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float obman_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Stream-ordered fill of `n_words` 32-bit words - instead of hipMemsetAsync.  A memset issued inside a stream capture becomes a
// hipGraph MEMSET node; replays of the captured training step (trainer.GraphedTrainStep) at configs[2] died with GPU memory
// faults a few replays in (profiles/r04_graph_fault.md), a graph whose only nodes are kernels does not.  16 bytes per lane.
static __global__ void obman_fill_u32_kernel(unsigned* __restrict__ p, unsigned v, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n) {
    if ((reinterpret_cast<size_t>(p + i) & 15) == 0) {
      *reinterpret_cast<uint4*>(p + i) = make_uint4(v, v, v, v);
    } else {
      p[i] = v; p[i + 1] = v; p[i + 2] = v; p[i + 3] = v;
    }
  } else {
    for (size_t k = i; k < n; ++k) p[k] = v;
  }
}
static inline hipError_t obman_fill_u32(void* p, unsigned v, size_t n_words, hipStream_t st) {
  if (n_words == 0) return hipSuccess;
  const size_t blocks = (n_words + 1023) / 1024;
  obman_fill_u32_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(static_cast<unsigned*>(p), v, n_words);
  return hipGetLastError();
}
