// What does one rows2 k-step cost when its pieces are added one by one?  512 threads (2 waves / SIMD), one block per CU,
// 4 independent v_mfma_f32_32x32x16_bf16 per iteration (accumulators in arch VGPRs) plus, by variant:
//   1: + 6 ds_read_b128 (conflict-free) waited for before the MFMAs   2: + 12 VALU on the loaded values (pk_add, cvt_pk, pk_max_i16)
//   3: + 2 buffer-style 16-byte global loads per iteration, consumed 4 iterations later
// Prints s_memtime ticks per iteration (average over waves) and wall-clock ns per iteration.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_loop mfma_loop.hip && ./mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters, int ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
  for (int i = tid; i < 140 * 1024 / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  const int KP2 = 536;
  const unsigned short* wl = reinterpret_cast<const unsigned short*>(smem) + li * KP2 + h * 8;
  const float* fy = reinterpret_cast<const float*>(smem + 131 * KP2 * 2) + (lane >> 2 & 7) * 532 + h * 8;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // variant 3: every lane its own row (64 cache lines per load); variant 4: the rows2 mode-2 pattern (8 lanes share a row, 4 rows per wave)
  const float* gp = V == 4 ? g + ((size_t)(blockIdx.x * 32 + (tid >> 6) * 4 + (li & 3)) % 4096) * ld + h * 8
                           : g + ((size_t)blockIdx.x * 512 + tid) % 4096 * ld + h * 8;
  u32x4 q[4][2];
  if (V >= 3) for (int u = 0; u < 4; ++u) { q[u][0] = *reinterpret_cast<const u32x4*>(gp + u * 16); q[u][1] = *reinterpret_cast<const u32x4*>(gp + u * 16 + 4); }
  u32x4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int s = 0; s < iters; s += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ks = (s + u) & 31;
      bf16x8 fb[4];
      if (V >= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(wl + j * 32 * KP2 + ks * 16);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = __builtin_bit_cast(bf16x8, a0);
      }
      if (V >= 1) {
        const float4 f0 = *reinterpret_cast<const float4*>(fy + ks * 16), f1 = *reinterpret_cast<const float4*>(fy + ks * 16 + 4);
        __builtin_amdgcn_sched_barrier(0);
        if (V >= 2) {
          float4 g0, g1;
          if (V >= 3) { g0 = __builtin_bit_cast(float4, q[u][0]); g1 = __builtin_bit_cast(float4, q[u][1]); }
          else { g0 = __builtin_bit_cast(float4, a0); g1 = g0; }
          const f32x2 s0 = f32x2{g0.x, g0.y} + f32x2{f0.x, f0.y}, s1 = f32x2{g0.z, g0.w} + f32x2{f0.z, f0.w};
          const f32x2 s2 = f32x2{g1.x, g1.y} + f32x2{f1.x, f1.y}, s3 = f32x2{g1.z, g1.w} + f32x2{f1.z, f1.w};
          const s16x2 z = {0, 0};
          a0.x = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(s0, bf16x2)), z));
          a0.y = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(s1, bf16x2)), z));
          a0.z = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(s2, bf16x2)), z));
          a0.w = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(s3, bf16x2)), z));
          if (V >= 3) {
            const int nk = (s + u + 4) & 31;
            q[u][0] = *reinterpret_cast<const u32x4*>(gp + nk * 16);
            q[u][1] = *reinterpret_cast<const u32x4*>(gp + nk * 16 + 4);
          }
        } else {
          a0.x ^= __builtin_bit_cast(unsigned, f0.x) & 1u; a0.y ^= __builtin_bit_cast(unsigned, f1.x) & 1u;
        }
      }
      const bf16x8 fa = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
  out[(size_t)blockIdx.x * 512 + tid] = r;
  if (lane == 0) ticks[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int V>
void run(const float* g, float* out, unsigned long long* ticks, int blocks, int iters, int ld) {
  const int lds = 160 * 1024 - 512;
  hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<V><<<blocks, 512, lds>>>(g, out, ticks, iters, ld);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<V><<<blocks, 512, lds>>>(g, out, ticks, iters, ld);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  printf("{\"variant\": %d, \"ticks_per_iter_per_wave\": %.1f, \"wall_ns_per_iter\": %.2f, \"mfma_per_iter\": 4, \"ideal_ticks_2waves\": 256}\n", V,
         s / h.size() / iters, ms * 1e6 / iters);
}
int main() {
  const int blocks = 256, iters = 4096, ld = 528;
  float *g, *out; unsigned long long* ticks;
  hipMalloc(&g, (size_t)4096 * ld * 4 + 65536); hipMemset(g, 0, (size_t)4096 * ld * 4 + 65536);
  hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&ticks, blocks * 8 * 8);
  run<0>(g, out, ticks, blocks, iters, ld);
  run<1>(g, out, ticks, blocks, iters, ld);
  run<2>(g, out, ticks, blocks, iters, ld);
  run<3>(g, out, ticks, blocks, iters, ld);
  run<4>(g, out, ticks, blocks, iters, ld);
  return 0;
}
