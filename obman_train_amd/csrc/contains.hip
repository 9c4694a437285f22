// K4 - ray-parity inside test: Moeller-Trumbore for every (point, triangle) pair, gfx950.
//
// Replaces batch_mesh_contains_points (contactutils.py:62-159): the reference materialises ~16
// tensors of B x P x T elements (~8 GB live at bs 64).  Here each lane keeps PPT query points in
// VGPRs; per-triangle constants (v0, e1, e2, pvec = dir x e2, 1/(det+1e-8) - NaN when |det| < tol so
// every comparison fails exactly like the reference's `not parallel` factor) are computed ONCE per
// block while staging the triangle chunk into LDS (4 x float4 per triangle, broadcast ds_read_b128),
// then the hot loop is pure VALU (round 3: the per-pair cross product folded into per-triangle
// vectors, ~20 lane-ops/pair instead of ~31).  Hit counts are integers: chunks of the
// triangle list handled by different blocks merge with integer atomicAdd => deterministic.
// Same constants/inequalities as the reference (App. C #8): fixed direction, tol 1e-7, strict
// u>0,u<1,v>0,u+v<1, t>=tol, exterior <=> even hit count.
// Bound: fp32 VALU; algorithmic bytes B*(P*12 + Nv*12 + F*12 + P*4).
//
// Grouped mode (multi-patch templates, an extension of this build: BASELINE.json configs 3/5): the face list is G
// consecutive groups of `group_faces` triangles, each a closed patch surface.  "Inside the union of the patches" is the OR
// of the per-patch parities, not the parity of the total count (a point inside two overlapping patches must stay
// interior), so the kernel folds each group's count into bit g of a per-point word; blocks that share a group merge with
// atomicXor (commutative => deterministic).  Triangle tiles never straddle a group boundary.
#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace {

constexpr float RAY_X = 0.4395064455f, RAY_Y = 0.617598629942f, RAY_Z = 0.652231566745f;
constexpr float TOL = 0.0000001f;
constexpr int MC_THREADS = 256;
constexpr int MC_TRI_TILE = 512;  // 4 float4 per triangle -> 32 KiB LDS

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return __fmaf_rn(az, bz, __fmaf_rn(ay, by, ax * bx));
}

template <int PPT>
__global__ __launch_bounds__(MC_THREADS) void contains_kernel(const float* __restrict__ points,
                                                              const float* __restrict__ verts,
                                                              const int* __restrict__ faces, int P, int Nv, int F,
                                                              int ptiles, int tchunk, int tsplit, int group_faces,
                                                              int* __restrict__ hits) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int pt = blockIdx.x % ptiles, ts = blockIdx.x / ptiles;
  const float* __restrict__ pb = points + (size_t)b * P * 3;
  const float* __restrict__ vb = verts + (size_t)b * Nv * 3;
  __shared__ float4 stri[MC_TRI_TILE * 4];

  float ox[PPT], oy[PPT], oz[PPT];
  int cnt[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int pi = pt * (MC_THREADS * PPT) + k * MC_THREADS + tid;
    const int pc = pi < P ? pi : P - 1;
    ox[k] = pb[(size_t)pc * 3];
    oy[k] = pb[(size_t)pc * 3 + 1];
    oz[k] = pb[(size_t)pc * 3 + 2];
    cnt[k] = 0;
  }
  const int tbeg = ts * tchunk, tend = min(F, tbeg + tchunk);
  int bits[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) bits[k] = 0;
  int n = 0;
  for (int base = tbeg; base < tend; base += n) {
    n = min(MC_TRI_TILE, tend - base);
    const int grp = group_faces ? base / group_faces : 0;
    if (group_faces) n = min(n, (grp + 1) * group_faces - base);  // stop at the patch boundary
    for (int t = tid; t < n; t += MC_THREADS) {
      const int* f = faces + (size_t)(base + t) * 3;
      const float* a = vb + (size_t)f[0] * 3;
      const float* bb = vb + (size_t)f[1] * 3;
      const float* c = vb + (size_t)f[2] * 3;
      const float ax = a[0], ay = a[1], az = a[2];
      const float e1x = bb[0] - ax, e1y = bb[1] - ay, e1z = bb[2] - az;
      const float e2x = c[0] - ax, e2y = c[1] - ay, e2z = c[2] - az;
      // pvec = dir x e2
      const float px = RAY_Y * e2z - RAY_Z * e2y;
      const float py = RAY_Z * e2x - RAY_X * e2z;
      const float pz = RAY_X * e2y - RAY_Y * e2x;
      const float det = e1x * px + e1y * py + e1z * pz;
      float inv = 1.f / (det + 0.1f * TOL);
      if (fabsf(det) < TOL) inv = __builtin_nanf("");  // parallel: every test below becomes false
      // The reference's v = dir . (tvec x e1) and t = e2 . (tvec x e1) are triple products: tvec . (e1 x dir) and tvec . (e1 x e2).
      // Both cross products and the 1/det factor belong to the triangle, so the per-pair work is three 3-term dot products of
      // tvec (was: a dot, a cross product, two dots and three multiplies - 21 -> 12 arithmetic lane-ops per pair).
      const float wx = e1y * RAY_Z - e1z * RAY_Y, wy = e1z * RAY_X - e1x * RAY_Z, wz = e1x * RAY_Y - e1y * RAY_X;
      const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
      stri[t * 4 + 0] = make_float4(ax, ay, az, 0.f);
      stri[t * 4 + 1] = make_float4(px * inv, py * inv, pz * inv, 0.f);
      stri[t * 4 + 2] = make_float4(wx * inv, wy * inv, wz * inv, 0.f);
      stri[t * 4 + 3] = make_float4(nx * inv, ny * inv, nz * inv, 0.f);
    }
    __syncthreads();
#pragma unroll 2
    for (int t = 0; t < n; ++t) {
      const float4 A = stri[t * 4], PU = stri[t * 4 + 1], PW = stri[t * 4 + 2], PN = stri[t * 4 + 3];
#pragma unroll
      for (int k = 0; k < PPT; ++k) {
        const float tx = ox[k] - A.x, ty = oy[k] - A.y, tz = oz[k] - A.z;
        const float u = dot3(tx, ty, tz, PU.x, PU.y, PU.z);
        const float v = dot3(tx, ty, tz, PW.x, PW.y, PW.z);
        const float tt = dot3(tx, ty, tz, PN.x, PN.y, PN.z);
        const bool hit = (u > 0.f) & (u < 1.f) & (v > 0.f) & (u + v < 1.f) & (tt >= TOL);
        cnt[k] += hit ? 1 : 0;
      }
    }
    __syncthreads();
    if (group_faces) {  // fold this tile's parity into the patch's bit
#pragma unroll
      for (int k = 0; k < PPT; ++k) { bits[k] ^= (cnt[k] & 1) << grp; cnt[k] = 0; }
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int pi = pt * (MC_THREADS * PPT) + k * MC_THREADS + tid;
    if (pi >= P) continue;
    int* dst = hits + (size_t)b * P + pi;
    if (group_faces) {
      if (tsplit == 1) *dst = bits[k];
      else if (bits[k]) atomicXor(dst, bits[k]);
    } else if (tsplit == 1) *dst = cnt[k];
    else if (cnt[k]) atomicAdd(dst, cnt[k]);
  }
}

}  // namespace

namespace {
int contains_launch(const float* points, const float* verts, const int* faces, int B, int P, int Nv, int F, int group_faces,
                    int* hits, hipStream_t st) {
  if (B < 0 || P < 0 || Nv <= 0 || F < 0 || !hits) return -1;
  if (group_faces < 0 || (group_faces > 0 && (F % group_faces != 0 || F / group_faces > 32))) return -1;
  if (B == 0 || P == 0) return 0;
  if (F == 0) return (int)obman_fill_u32(hits, 0u, (size_t)B * P, st);
  int ppt = 4;
  while (ppt > 1 && MC_THREADS * (ppt / 2) >= P) ppt >>= 1;  // smallest PPT whose tile still covers P
  if (P > MC_THREADS * 4) ppt = 4;
  const int ptiles = obman_cdiv(P, MC_THREADS * ppt);
  // split the triangle list until the grid covers the chip ~4x (chunks are multiples of 64 triangles)
  int tsplit = 1;
  const long base_blocks = (long)B * ptiles;
  if (base_blocks < 1024) {
    tsplit = (int)((1024 + base_blocks - 1) / base_blocks);
    const int maxsplit = obman_cdiv(F, 64);
    if (tsplit > maxsplit) tsplit = maxsplit;
  }
  int tchunk = obman_cdiv(obman_cdiv(F, tsplit), 64) * 64;
  tsplit = obman_cdiv(F, tchunk);
  if (tsplit > 1) {
    hipError_t e = obman_fill_u32(hits, 0u, (size_t)B * P, st);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(ptiles * tsplit, B);
  ObmanProfScope prof(OBMAN_K_CONTAINS, st);
  switch (ppt) {
    case 4: contains_kernel<4><<<grid, MC_THREADS, 0, st>>>(points, verts, faces, P, Nv, F, ptiles, tchunk, tsplit, group_faces, hits); break;
    case 2: contains_kernel<2><<<grid, MC_THREADS, 0, st>>>(points, verts, faces, P, Nv, F, ptiles, tchunk, tsplit, group_faces, hits); break;
    default: contains_kernel<1><<<grid, MC_THREADS, 0, st>>>(points, verts, faces, P, Nv, F, ptiles, tchunk, tsplit, group_faces, hits); break;
  }
  OBMAN_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int obman_mesh_contains_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                       int F, int* hits, obman_stream_t stream) {
  return contains_launch(points, verts, faces, B, P, Nv, F, 0, hits, (hipStream_t)stream);
}

extern "C" int obman_mesh_contains_groups_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                              int F, int group_faces, int* parity_bits, obman_stream_t stream) {
  if (group_faces <= 0) return -1;
  return contains_launch(points, verts, faces, B, P, Nv, F, group_faces, parity_bits, (hipStream_t)stream);
}
