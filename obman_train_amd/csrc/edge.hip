// K8 - edge-length regulariser of the predicted object mesh, forward + backward (gfx950).
//
// Replaces edge_loss (atlasbranch.py:153-167): three [B,F,3] gathers, squared edge lengths, concat [B,3F], per-sample
// mean, mean absolute deviation over the batch - ~15 torch kernels and 6 temporaries - by two small kernels per
// direction.  loss = mean_{b,e} | l_e(b) - mean_e l_e(b) |,  l = squared edge length, e over the 3F face edges
// (interior edges counted twice, as in the reference).
// Backward: g_e = sign(l_e - m)/(3F B) ; since sum_e d|l_e - m|/dm = -sum_e sign, dl_e = g_e - mean_e(g_e); every edge
// pushes +-2 (a - b) dl onto its two vertices; the scatter is an owner scan over the faces (deterministic).
// Bound: latency / L2 (12 KB of vertices + 15 KB of faces per sample).
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

__device__ __forceinline__ float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }
__device__ __forceinline__ float sq(float3 a, float3 b) {
  const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z;
  return x * x + y * y + z * z;
}

// pass 1: per-sample sums of the 3F squared edge lengths -> mean[b]; pass 2 (same kernel, second phase): sum |l - m|
__global__ __launch_bounds__(256) void edge_fwd_kernel(const float* __restrict__ V, const int* __restrict__ faces, int N, int F,
                                                       float* __restrict__ mean, float* __restrict__ absdev,
                                                       float* __restrict__ nsign) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* vb = V + (size_t)b * N * 3;
  __shared__ float red[4];
  __shared__ float s_mean;
  float s = 0.f;
  for (int f = tid; f < F; f += 256) {
    const float3 a = ld3(vb + faces[f * 3] * 3), bb = ld3(vb + faces[f * 3 + 1] * 3), c = ld3(vb + faces[f * 3 + 2] * 3);
    s += sq(bb, a) + sq(c, bb) + sq(a, c);
  }
  s = obman_wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) s_mean = ((red[0] + red[1]) + (red[2] + red[3])) / (3.f * F);
  __syncthreads();
  const float m = s_mean;
  float d = 0.f, sg = 0.f;
  for (int f = tid; f < F; f += 256) {
    const float3 a = ld3(vb + faces[f * 3] * 3), bb = ld3(vb + faces[f * 3 + 1] * 3), c = ld3(vb + faces[f * 3 + 2] * 3);
    const float l0 = sq(bb, a) - m, l1 = sq(c, bb) - m, l2 = sq(a, c) - m;
    d += fabsf(l0) + fabsf(l1) + fabsf(l2);
    sg += (l0 > 0.f ? 1.f : (l0 < 0.f ? -1.f : 0.f)) + (l1 > 0.f ? 1.f : (l1 < 0.f ? -1.f : 0.f)) + (l2 > 0.f ? 1.f : (l2 < 0.f ? -1.f : 0.f));
  }
  d = obman_wave_sum(d);
  sg = obman_wave_sum(sg);
  __syncthreads();
  if (lane == 0) red[wave] = d;
  __syncthreads();
  if (tid == 0) { absdev[b] = (red[0] + red[1]) + (red[2] + red[3]); mean[b] = m; }
  __syncthreads();
  if (lane == 0) red[wave] = sg;
  __syncthreads();
  if (tid == 0) nsign[b] = ((red[0] + red[1]) + (red[2] + red[3])) / (3.f * F);  // mean sign: the d/dm term
}

__global__ __launch_bounds__(64) void edge_finalize_kernel(const float* __restrict__ absdev, int B, int F, float* __restrict__ loss) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += absdev[b];
  s = obman_wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (3.f * F * B);
}

// owner scan: vertex v collects the contributions of every face edge that touches it (ascending face order)
__global__ __launch_bounds__(256) void edge_bwd_kernel(const float* __restrict__ V, const int* __restrict__ faces, int N, int F, int B,
                                                       const float* __restrict__ mean, const float* __restrict__ nsign,
                                                       const float* __restrict__ g_loss, float* __restrict__ grad) {
  const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
  const float* vb = V + (size_t)b * N * 3;
  const float m = mean[b], ms = nsign[b];
  const float scale = 2.f * g_loss[0] / (3.f * F * B);
  extern __shared__ int s_faces[];  // [chunk][3]
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const float3 p = v < N ? ld3(vb + v * 3) : make_float3(0, 0, 0);
  constexpr int CH = 1024;
  for (int base = 0; base < F; base += CH) {
    const int cnt = min(CH, F - base);
    for (int i = threadIdx.x; i < cnt * 3; i += 256) s_faces[i] = faces[(size_t)base * 3 + i];
    __syncthreads();
    for (int f = 0; f < cnt; ++f) {
      const int i0 = s_faces[f * 3], i1 = s_faces[f * 3 + 1], i2 = s_faces[f * 3 + 2];
      if (i0 != v && i1 != v && i2 != v) continue;
      const float3 a = ld3(vb + i0 * 3), bb = ld3(vb + i1 * 3), c = ld3(vb + i2 * 3);
      // edges: (a,b) (b,c) (c,a); dl = sign(l - m) - mean_sign
      const float l0 = sq(bb, a) - m, l1 = sq(c, bb) - m, l2 = sq(a, c) - m;
      const float d0 = (l0 > 0.f ? 1.f : (l0 < 0.f ? -1.f : 0.f)) - ms, d1 = (l1 > 0.f ? 1.f : (l1 < 0.f ? -1.f : 0.f)) - ms,
                  d2 = (l2 > 0.f ? 1.f : (l2 < 0.f ? -1.f : 0.f)) - ms;
      if (i0 == v) {  // a: edges (a,b) and (c,a)
        gx += d0 * (a.x - bb.x) + d2 * (a.x - c.x); gy += d0 * (a.y - bb.y) + d2 * (a.y - c.y); gz += d0 * (a.z - bb.z) + d2 * (a.z - c.z);
      }
      if (i1 == v) {  // b: edges (a,b) and (b,c)
        gx += d0 * (bb.x - a.x) + d1 * (bb.x - c.x); gy += d0 * (bb.y - a.y) + d1 * (bb.y - c.y); gz += d0 * (bb.z - a.z) + d1 * (bb.z - c.z);
      }
      if (i2 == v) {  // c: edges (b,c) and (c,a)
        gx += d1 * (c.x - bb.x) + d2 * (c.x - a.x); gy += d1 * (c.y - bb.y) + d2 * (c.y - a.y); gz += d1 * (c.z - bb.z) + d2 * (c.z - a.z);
      }
    }
    __syncthreads();
  }
  if (v < N) {
    float* g = grad + ((size_t)b * N + v) * 3;
    g[0] = scale * gx; g[1] = scale * gy; g[2] = scale * gz;
  }
  (void)p;
}

}  // namespace

extern "C" {

/* stats [3*B] = mean | absdev | mean sign (fwd -> bwd) */
int obman_edge_loss_fwd(const float* verts, const int* faces, int B, int N, int F, float* loss, float* stats, obman_stream_t stream) {
  if (!verts || !faces || !loss || !stats || B <= 0 || N <= 0 || F <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  edge_fwd_kernel<<<B, 256, 0, st>>>(verts, faces, N, F, stats, stats + B, stats + 2 * B);
  OBMAN_LAUNCH_CHECK();
  edge_finalize_kernel<<<1, 64, 0, st>>>(stats + B, B, F, loss);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_edge_loss_bwd(const float* verts, const int* faces, int B, int N, int F, const float* stats, const float* g_loss,
                        float* grad, obman_stream_t stream) {
  if (!verts || !faces || !stats || !g_loss || !grad || B <= 0 || N <= 0 || F <= 0) return -1;
  dim3 grid(obman_cdiv(N, 256), B);
  edge_bwd_kernel<<<grid, 256, 1024 * 3 * sizeof(int), (hipStream_t)stream>>>(verts, faces, N, F, B, stats, stats + 2 * B, g_loss, grad);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
