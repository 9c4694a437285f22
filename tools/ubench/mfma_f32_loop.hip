// fp32 counterpart of mfma_loop.hip: what does one k-step (8 k) of the rows2f kernel cost when its pieces are added one by one?
// 512 threads (2 waves / SIMD), one block per CU, 8 x v_mfma_f32_32x32x2_f32 per iteration on two accumulator tiles, plus by variant
//   1: + 2 ds_read_b128 weight fragments (double-buffered: consumed in the next iteration)
//   2: + the h2 transform (4 add + 4 max) on values held in registers
//   3: + 2 sixteen-byte global loads per iteration (every lane its own row), consumed 4 iterations later
//   5: variant 0 with 256 threads per block (1 wave / SIMD)      6: variant 0 with FOUR accumulator tiles (16 MFMAs per iteration)
// Prints wall-clock ns per iteration and the fp32 matrix rate it corresponds to (peak 157.3 TFLOP/s at 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_loop mfma_f32_loop.hip && ./mfma_f32_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ float rnd(unsigned x) {  // hash -> [-1, 1): operands with realistic bit activity (the matrix pipe's power depends on it)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (float)(int)x * (1.0f / 2147483648.0f);
}
template <int V, int NT, bool RANDOM = false>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  for (int i = tid; i < 136 * 1024 / 16; i += blockDim.x)
    reinterpret_cast<float4*>(smem)[i] = RANDOM ? make_float4(rnd(4 * i), rnd(4 * i + 1), rnd(4 * i + 2), rnd(4 * i + 3)) : make_float4(1.f, 1.f, 1.f, 1.f);
  __syncthreads();
  const int KP2 = 524;
  const float* wl = reinterpret_cast<const float*>(smem) + li * KP2 + 4 * h;
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* gp = g + ((size_t)blockIdx.x * 512 + tid) % 2048 * ld + 4 * h;
  float4 q[4][2];
  if (V >= 3) for (int u = 0; u < 4; ++u) { q[u][0] = *reinterpret_cast<const float4*>(gp + u * 8); q[u][1] = *reinterpret_cast<const float4*>(gp + 2048 * ld + u * 8); }
  float4 wa[NT], wb[NT];
  for (int j = 0; j < NT; ++j) { wa[j] = make_float4(1.f, 1.f, 1.f, 1.f); wb[j] = wa[j]; }
  float a[4] = {1.f, 1.f, 1.f, 1.f};
  if (RANDOM) for (int t = 0; t < 4; ++t) a[t] = rnd(tid * 4 + t + 12345u) * 1e-3f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < iters; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = (s + u) & 63;
      float4 (&cur)[NT] = (u & 1) ? wb : wa;
      float4 (&nxt)[NT] = (u & 1) ? wa : wb;
      if (V >= 1 || RANDOM) {
#pragma unroll
        for (int j = 0; j < NT; ++j) nxt[j] = *reinterpret_cast<const float4*>(wl + j * 32 * KP2 + ks * 8);
      }
      if (V >= 2) {
        float4 g0, g1;
        if (V >= 3) { g0 = q[u][0]; g1 = q[u][1]; }
        else { g0 = make_float4(a[0], a[1], a[2], a[3]); g1 = g0; }
        a[0] = fmaxf(g0.x + g1.x, 0.f); a[1] = fmaxf(g0.y + g1.y, 0.f); a[2] = fmaxf(g0.z + g1.z, 0.f); a[3] = fmaxf(g0.w + g1.w, 0.f);
        if (V >= 3) {
          const int nk = (s + u + 4) & 63;
          q[u][0] = *reinterpret_cast<const float4*>(gp + nk * 8);
          q[u][1] = *reinterpret_cast<const float4*>(gp + 2048 * ld + nk * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], cur[j].x, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], cur[j].y, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], cur[j].z, acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], cur[j].w, acc[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int j = 0; j < NT; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
  out[(size_t)blockIdx.x * 512 + tid] = r;
  if (lane == 0) ticks[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int V, int NT, bool RANDOM = false>
void run(const char* what, const float* g, float* out, unsigned long long* ticks, int blocks, int threads, int iters, int ld) {
  const int lds = 140 * 1024;
  hipFuncSetAttribute((const void*)k<V, NT, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V, NT, RANDOM><<<blocks, threads, lds>>>(g, out, ticks, iters, ld);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V, NT, RANDOM><<<blocks, threads, lds>>>(g, out, ticks, iters, ld);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * (threads / 64) * iters * 4.0 * NT * 4096.0;
  printf("{\"variant\": %d, \"what\": \"%s\", \"waves_per_simd\": %d, \"mfma_per_iter\": %d, \"wall_ns_per_iter\": %.2f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f}\n",
         V, what, threads / 256, 4 * NT, ms * 1e6 / iters, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
}
int main() {
  const int blocks = 256, iters = 8192, ld = 528;
  float *g, *out; unsigned long long* ticks;
  hipMalloc(&g, (size_t)4096 * ld * 4 + 65536); hipMemset(g, 0, (size_t)4096 * ld * 4 + 65536);
  hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&ticks, blocks * 8 * 8);
  run<0, 2>("8 MFMAs only", g, out, ticks, blocks, 512, iters, ld);
  run<1, 2>("+ 2 ds_read_b128 (next step's fragments)", g, out, ticks, blocks, 512, iters, ld);
  run<2, 2>("+ 4 add + 4 max", g, out, ticks, blocks, 512, iters, ld);
  run<3, 2>("+ 2 global 16-byte loads, own row per lane", g, out, ticks, blocks, 512, iters, ld);
  run<0, 2>("8 MFMAs only, 1 wave per SIMD", g, out, ticks, blocks, 256, iters, ld);
  run<0, 4>("16 MFMAs only (4 accumulator tiles)", g, out, ticks, blocks, 512, iters, ld);
  run<3, 4>("4 tiles + everything", g, out, ticks, blocks, 512, iters, ld);
  // sustained: the same MFMA-only loop for ~0.5 s per launch, five launches back to back (does the clock hold under power?)
  for (int rep = 0; rep < 5; ++rep) run<0, 2>("8 MFMAs only, sustained 0.5 s", g, out, ticks, blocks, 512, 1 << 20, ld);
  run<2, 2>("+ LDS + VALU, sustained", g, out, ticks, blocks, 512, 1 << 20, ld);
  // the same with RANDOM operand values (weights in [-1, 1) from LDS, per-lane A values): realistic bit activity in the matrix pipe
  for (int rep = 0; rep < 3; ++rep) run<1, 2, true>("8 MFMAs + fragment reads, RANDOM operands, sustained 0.5 s", g, out, ticks, blocks, 512, 1 << 20, ld);
  run<1, 2, true>("8 MFMAs + fragment reads, RANDOM operands, short", g, out, ticks, blocks, 512, 8192, ld);
  return 0;
}
