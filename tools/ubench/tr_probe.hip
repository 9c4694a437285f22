// Probe (gfx950): lane mapping of ds_read_b64_tr_b16.  LDS is filled with u16 values equal to their own element index; every lane
// reads 8 bytes at byte address base + lane * stride (two layouts: stride 8 = lane-linear, and a [16 rows][pitch] image where a
// lane addresses (row = lane & 15, 4-element column block = lane >> 4)).  Prints, per lane, the four u16 it received.
//   hipcc --offload-arch=gfx950 -O2 -o tr_probe tr_probe.hip && ./tr_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                                   // lane-linear: lane l -> elements 4l .. 4l+3
  else if (mode == 1) addr = (lane & 15) * 64 + (lane >> 4) * 8;     // [16 rows][32 el pitch]: row = l&15, col block = l>>4
  else addr = (lane & 15) * 32 + (lane >> 4) * 8;                    // [16 rows][16 el pitch]
  addr += (unsigned)(size_t)lds;  // LDS aperture offset is 0-based for ds ops: take the low bits of the generic pointer
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[lane * 4 + 0] = (unsigned short)(v.x & 0xffff);
  out[lane * 4 + 1] = (unsigned short)(v.x >> 16);
  out[lane * 4 + 2] = (unsigned short)(v.y & 0xffff);
  out[lane * 4 + 3] = (unsigned short)(v.y >> 16);
}

int main() {
  unsigned short* d;
  hipMalloc(&d, 64 * 4 * sizeof(unsigned short));
  for (int mode = 0; mode < 3; ++mode) {
    probe<<<1, 64>>>(d, mode);
    unsigned short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (element indices received by each lane)\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
