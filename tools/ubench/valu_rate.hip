// Micro-benchmark (gfx950): issue rate of the VALU instructions the pair-min kernel is made of, per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
// Every wave runs ITER x 32 independent instructions of one kind (8 accumulators x 4); with 8 waves per SIMD the issue rate is the
// pipe's, not the dependency latency's.  Reports instructions per ns per SIMD and, from s_memtime around the loop, cycles per
// wave-instruction on one SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters, float seed) {
  float a[8];
  f2 p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = seed + k + threadIdx.x; p[k] = f2{a[k], a[k] + 1.f}; }
  const float b = seed * 0.5f, c = seed * 0.25f;
  const f2 pb = {b, b}, pc = {c, c};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
        if (MODE == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (MODE == 3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        if (MODE == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b) : "vcc");
        if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
        if (MODE == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
        if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k] + p[k][0] + p[k][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_instr_ops) {
  const int blocks = 256 * 8, iters = 2000;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  rate_kernel<MODE><<<blocks, 256>>>(out, cyc, 10, 1.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  rate_kernel<MODE><<<blocks, 256>>>(out, cyc, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[256 * 8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0;
  for (int i = 0; i < blocks; ++i) avg += (double)h[i];
  avg /= blocks;
  const double instr_per_wave = (double)iters * 32 * (MODE == 4 ? 2 : 1);
  const double waves_per_simd = 8.0;
  // s_memtime counts at a fixed 100 MHz on gfx9; report the wall-clock figure and the implied cycles at 2.4 GHz
  const double ns = ms * 1e6;
  const double instr_per_simd = instr_per_wave * waves_per_simd;
  printf("{\"instr\": \"%s\", \"ns_per_wave_instr_per_simd\": %.4f, \"cycles_at_2.4GHz\": %.3f, \"lane_ops_per_instr\": %d, \"kernel_ms\": %.4f, "
         "\"memtime_ticks\": %.0f}\n",
         name, ns / instr_per_simd, ns / instr_per_simd * 2.4, per_instr_ops, ms, avg);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_pk_fma_f32", 2);
  run<2>("v_min_f32", 1);
  run<3>("v_min3_f32", 1);
  run<4>("v_cmp_lt_f32 + v_cndmask_b32 (per instruction of the pair)", 1);
  run<5>("v_pk_add_f32", 2);
  run<6>("v_sub_f32", 1);
  run<7>("v_pk_mul_f32", 2);
  return 0;
}
