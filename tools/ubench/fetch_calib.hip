// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access widths the kernels of this repository
// use (the guide calibrates the x2 correction for 16-byte-per-lane streams only).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./fetch_calib      (and a second pass with --pmc WRITE_SIZE)
// Each kernel streams a KNOWN number of bytes once, from a buffer far larger than the 256 MB Infinity Cache, and writes 4 bytes
// per block (read kernels) or streams the same number of bytes out (write kernels):
//   read_b32 / read_b64 / read_b96 / read_b128: every lane loads one record of 4 / 8 / 12 / 16 bytes, consecutive lanes consecutive
//   records (global_load_dword / dwordx2 / dwordx3 / dwordx4) - read_b96 is the [B,N,3] fp32 point layout of pairmin / contains;
//   write_b32 / write_b128: the store counterparts.
// The program prints the bytes each kernel moved; tools/pmc_fetch_calib.sh divides the counters by them.
#include <hip/hip_runtime.h>

#include <cstdio>

struct R3 { float x, y, z; };

template <class T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ p, size_t n, float* __restrict__ sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = p[i];
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += f[k];
  }
  if (acc == 12345.678f) sink[blockIdx.x] = acc;  // never true for zero-filled input: the loads stay, the store does not happen
}
template <class T>
__global__ __launch_bounds__(256) void write_kernel(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}

int main() {
  const size_t bytes = (size_t)3 << 30;  // 3 GiB per pass
  void* buf;
  float* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 1 << 20) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_kernel<float>, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_kernel<float2>, dim3(grid), dim3(256), 0, 0, (const float2*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(read_kernel<R3>, dim3(grid), dim3(256), 0, 0, (const R3*)buf, bytes / 12, sink);
    hipLaunchKernelGGL(read_kernel<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(write_kernel<float>, dim3(grid), dim3(256), 0, 0, (float*)buf, bytes / 4, 0.f);
    hipLaunchKernelGGL(write_kernel<R3>, dim3(grid), dim3(256), 0, 0, (R3*)buf, bytes / 12, R3{0.f, 0.f, 0.f});
    hipLaunchKernelGGL(write_kernel<float4>, dim3(grid), dim3(256), 0, 0, (float4*)buf, bytes / 16, make_float4(0.f, 0.f, 0.f, 0.f));
    hipDeviceSynchronize();
  }
  printf("bytes_per_kernel %zu\n", bytes);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
