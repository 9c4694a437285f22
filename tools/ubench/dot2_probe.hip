// v_dot2c_f32_bf16 semantics probe (gfx950): compares the instruction with an fp32 FMA chain on random bf16 pairs.
//   hipcc --offload-arch=gfx950 -O2 -o dot2_probe dot2_probe.hip && ./dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* o, float* o4, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), c[i], false);
  // chained like the decoder's side column
  float s = c[i];
  for (int e = 0; e < 4; ++e) {
    int j = (i + e) % n;
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[j]), __builtin_bit_cast(bf16x2, b[j]), s, false);
  }
  o4[i] = s;
}
static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const int n = 4096;
  unsigned *a, *b; float *c, *o, *o4;
  hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4); hipMallocManaged(&c, n * 4); hipMallocManaged(&o, n * 4); hipMallocManaged(&o4, n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    float x[4];
    for (int e = 0; e < 4; ++e) x[e] = (rand() / (float)RAND_MAX - 0.5f) * 8.f;
    unsigned u[4]; for (int e = 0; e < 4; ++e) { memcpy(&u[e], &x[e], 4); u[e] >>= 16; }
    a[i] = u[0] | (u[1] << 16); b[i] = u[2] | (u[3] << 16);
    c[i] = (rand() / (float)RAND_MAX - 0.5f) * 100.f;
  }
  k<<<n / 256, 256>>>(a, b, c, o, o4, n);
  hipDeviceSynchronize();
  double worst = 0, worst4 = 0; int bad = 0;
  for (int i = 0; i < n; ++i) {
    float r = fmaf(bf(a[i] >> 16), bf(b[i] >> 16), fmaf(bf(a[i] & 0xffff), bf(b[i] & 0xffff), c[i]));
    double e = fabs((double)r - o[i]); if (e > worst) worst = e;
    if (e > 1e-3 && bad < 5) { printf("i %d a %g %g b %g %g c %g  fma %g dot2 %g\n", i, bf(a[i] & 0xffff), bf(a[i] >> 16), bf(b[i] & 0xffff), bf(b[i] >> 16), c[i], r, o[i]); ++bad; }
    float s = c[i];
    for (int e2 = 0; e2 < 4; ++e2) { int j = (i + e2) % n; s = fmaf(bf(a[j] >> 16), bf(b[j] >> 16), fmaf(bf(a[j] & 0xffff), bf(b[j] & 0xffff), s)); }
    double e4 = fabs((double)s - o4[i]); if (e4 > worst4) worst4 = e4;
  }
  printf("single worst abs diff %g, chain of 4 worst %g\n", worst, worst4);
  return 0;
}
