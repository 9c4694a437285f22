// K4 - ray-parity inside test: Moeller-Trumbore along the reference's fixed ray direction, gfx950.
//
// Replaces batch_mesh_contains_points (contactutils.py:62-159): the reference materialises ~16
// tensors of B x P x T elements (~8 GB live at bs 64) because tensors force the all-pairs form.
// Per-triangle constants (v0, pvec = dir x e2, (e1 x dir), (e1 x e2), each scaled by 1/(det+1e-8) - NaN when
// |det| < tol so every comparison fails exactly like the reference's `not parallel` factor) come from ONE
// device function (tri_setup), the per-pair arithmetic from ONE device function (ray_hit): three 3-term dot products
// of tvec = origin - v0 and the reference's inequalities (App. C #8): tol 1e-7, strict u>0, u<1, v>0, u+v<1, t>=tol,
// exterior <=> even hit count.  The library is built with -ffp-contract=off, so both kernels below evaluate a pair
// with the same IEEE operations in the same order.
//
// Two kernels share that arithmetic:
//
//  * contains_kernel (round 1-4): ALL pairs.  PPT query points per lane in VGPRs, triangle chunks staged in LDS.
//    fp32-VALU bound (~20 lane-ops per pair; 904 us at 64 x 778 x 32 000).  Kept as the CHECKER
//    (obman_mesh_contains_bruteforce_fwd) and as the OBMAN_MC_BINNED=0 A/B.
//
//  * contains_binned_kernel (round 5, the product path): the ray direction is one constant (contactutils.py:65), so a
//    pair can only hit when the point's projection on the plane normal to the ray lies in the triangle's projection.
//    Each block sorts its tile of <= 1024 query points into a uniform 2-D grid over their projected bounding box (LDS
//    counting sort), then every lane takes triangles of the block's chunk, computes the triangle's projected bounding box
//    INFLATED by a round-off bound (below), and runs ray_hit only against the points of the grid cells that box touches
//    (a contiguous range of the sorted array per grid row).  Hits are integer LDS atomics (add, or xor of the patch bit
//    in grouped mode), merged across triangle chunks by integer global atomics as before => deterministic.
//    Pairs skipped are pairs whose fp32 evaluation provably fails the u / v / u+v tests, so the hit words are
//    BIT-IDENTICAL to contains_kernel's (tests/test_contains_binned_gpu.py: every existing case + > 1e5 random scenes with
//    slivers, grazing rays, degenerate and non-finite input).
//
// Round-off bound (eps = 2^-24, all quantities of one triangle; "*" = exact real arithmetic on the fp32 inputs):
//   stored PU_j = fl(p_j inv), inv = fl(1 / fl(det + 1e-8)), p = fl(dir x e2), det = fl(e1 . p).  Write D* = det* + 1e-8.
//   (i)  |p_j - p*_j| <= 3 eps |e2|, |det - det*| <= 9.2 eps |e1||e2|  =>  inv = (1 + rho) / D*, |rho| <= (9.2 k + 4) eps,
//        k = |e1||e2| / |D*|.
//   (ii) u_c = fl(tvec . PU) = (1 + rho) U* + a_u with U* = (o - v0) . p* / D* (the reference's u in exact arithmetic) and an
//        ADDITIVE part |a_u| <= 11.4 eps |o - v0| |e2| / |D*| (component errors of p, of tvec, of the products, the 3-term dot);
//        likewise v_c = (1 + rho) V* + a_v with |e1| in place of |e2| (same rho: one inv scales all three vectors).
//   (iii) U*, V* are the barycentric coordinates of the point's projection along the ray w.r.t. the triangle
//        T_f = {v0 + a e1 + b e2 : a, b > 0, a + b < f}, f = D* / det* = 1 + 1e-8 / det* (the reference's "+ 0.1 tol" scales
//        the triangle about v0; |det| >= 1e-7 for a non-parallel triangle, so 0.9 <= f <= 1.1).
//   A pair passes the fp32 tests u_c > 0, v_c > 0, fl(u_c + v_c) < 1 only if U* > -eta, V* > -eta, U* + V* < 1 + eta with
//   eta = (|rho| + 2 max|a|) / (1 - |rho|).  A point whose projection is outside the bounding box of the projected T_f by
//   more than 2 eta ext (ext <= f L3, L3 = longest edge) has a barycentric coordinate below -eta, so it cannot pass.  With
//   Lm = max(|e1|,|e2|), Q = Lm L3 / |D|:  2 eta ext <= eps [L3 (21 Q + 9) + 52 Q |o - v0|] ; the kernel uses
//       m = 128 eps Q (Tmax + L3) + 64 eps Cmax      (>= 2x the bound; Tmax >= |o - v0| over the tile, Cmax >= |coordinates|:
//   the second term covers the rounding of the projections themselves and the ~1e-8 non-orthogonality of the fp32 basis),
//   and gives up on the box (the triangle then visits EVERY cell = the all-pairs test for that triangle) when Q > 1e4
//   (|rho| <= 6e-3 there), or anything is non-finite.  Triangles with |det| < tol are skipped: their inv is NaN in both
//   kernels.  The grid-cell function is monotone in the projected coordinate and shared by points and boxes, so no
//   further rounding enters.
//
// Bound: latency / LDS for the binned kernel (pair tests drop ~100x); algorithmic bytes B*(P*12 + Nv*12 + F*12 + P*4).
//
// Grouped mode (multi-patch templates, an extension of this build: BASELINE.json configs 3/5): the face list is G
// consecutive groups of `group_faces` triangles, each a closed patch surface.  "Inside the union of the patches" is the OR
// of the per-patch parities, not the parity of the total count (a point inside two overlapping patches must stay
// interior), so the kernels fold each group's count into bit g of a per-point word; blocks that share a group merge with
// atomicXor (commutative => deterministic).  In contains_kernel triangle tiles never straddle a group boundary.
#include <cfloat>
#include <cstdlib>

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

// Everything a pair test needs from one triangle (and, for the binned kernel, its edges and D = det + 0.1 tol).
struct TriSetup {
  float4 A, PU, PW, PN;
  float e1x, e1y, e1z, e2x, e2y, e2z, det, D;
};

__device__ __forceinline__ TriSetup tri_setup(const float* __restrict__ vb, const int* __restrict__ f) {
  TriSetup s;
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
  const float D = det + 0.1f * TOL;
  float inv = 1.f / D;
  if (fabsf(det) < TOL) inv = __builtin_nanf("");  // parallel: every test below becomes false
  // The reference's v = dir . (tvec x e1) and t = e2 . (tvec x e1) are triple products: tvec . (e1 x dir) and tvec . (e1 x e2).
  // Both cross products and the 1/det factor belong to the triangle, so the per-pair work is three 3-term dot products of
  // tvec (was: a dot, a cross product, two dots and three multiplies - 21 -> 12 arithmetic lane-ops per pair).
  const float wx = e1y * RAY_Z - e1z * RAY_Y, wy = e1z * RAY_X - e1x * RAY_Z, wz = e1x * RAY_Y - e1y * RAY_X;
  const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
  s.A = make_float4(ax, ay, az, 0.f);
  s.PU = make_float4(px * inv, py * inv, pz * inv, 0.f);
  s.PW = make_float4(wx * inv, wy * inv, wz * inv, 0.f);
  s.PN = make_float4(nx * inv, ny * inv, nz * inv, 0.f);
  s.e1x = e1x; s.e1y = e1y; s.e1z = e1z; s.e2x = e2x; s.e2y = e2y; s.e2z = e2z;
  s.det = det; s.D = D;
  return s;
}

__device__ __forceinline__ bool ray_hit(float ox, float oy, float oz, const float4& A, const float4& PU, const float4& PW,
                                        const float4& PN) {
  const float tx = ox - A.x, ty = oy - A.y, tz = oz - A.z;
  const float u = dot3(tx, ty, tz, PU.x, PU.y, PU.z);
  const float v = dot3(tx, ty, tz, PW.x, PW.y, PW.z);
  const float tt = dot3(tx, ty, tz, PN.x, PN.y, PN.z);
  return (u > 0.f) & (u < 1.f) & (v > 0.f) & (u + v < 1.f) & (tt >= TOL);
}

// ---------------------------------------------------------------------------------------------------------------------
// All pairs (the checker).
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
      const TriSetup s = tri_setup(vb, faces + (size_t)(base + t) * 3);
      stri[t * 4 + 0] = s.A;
      stri[t * 4 + 1] = s.PU;
      stri[t * 4 + 2] = s.PW;
      stri[t * 4 + 3] = s.PN;
    }
    __syncthreads();
#pragma unroll 2
    for (int t = 0; t < n; ++t) {
      const float4 A = stri[t * 4], PU = stri[t * 4 + 1], PW = stri[t * 4 + 2], PN = stri[t * 4 + 3];
#pragma unroll
      for (int k = 0; k < PPT; ++k) cnt[k] += ray_hit(ox[k], oy[k], oz[k], A, PU, PW, PN) ? 1 : 0;
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

// ---------------------------------------------------------------------------------------------------------------------
// Binned (the product path).
constexpr int MB_PT = 1024;       // query points per block tile (4 per lane while binning)
constexpr int MB_PPT = MB_PT / MC_THREADS;
constexpr int MB_MAXCELL = 1024;  // grid cells per tile (Gx * Gy <= MB_MAXCELL, Gx, Gy <= 64)
constexpr float MB_EPS = 5.9604645e-8f;  // 2^-24
constexpr float MB_QMAX = 1.0e4f;
constexpr int MB_DEFER = 512;       // capacity of the per-block list of wide triangles
constexpr int MB_DEFER_POINTS = 48; // a lane keeps a triangle whose cells hold at most this many points
// Orthonormal basis of the plane normal to the ray (fp32 roundings of the exact vectors: |dir . ax|, |dir . ay| < 1e-8).
constexpr float AX_X = -0.814752659671f, AX_Y = 0.579808678409f, AX_Z = 0.f;
constexpr float AY_X = -0.378169522731f, AY_Y = -0.531407403727f, AY_Z = 0.758019777672f;

__device__ __forceinline__ float proj_x(float x, float y, float z) { return dot3(x, y, z, AX_X, AX_Y, AX_Z); }
__device__ __forceinline__ float proj_y(float x, float y, float z) { return dot3(x, y, z, AY_X, AY_Y, AY_Z); }

// Monotone non-decreasing in g (fl(-), fl(*) by a non-negative factor, max, min, truncation of a non-negative float);
// NaN -> cell 0.  Shared by the points and by the triangles' boxes.
__device__ __forceinline__ int grid_cell(float g, float g0, float inv_cell, float last) {
  const float t = (g - g0) * inv_cell;
  return (int)fminf(fmaxf(t, 0.f), last);
}

struct MbShared {
  float4 pts[MB_PT];         // tile points sorted by cell; .w = bits of the point's index in the tile
  int hit[MB_PT];            // per sorted slot: hit count / patch parity bits
  int start[MB_MAXCELL + 1]; // cell -> first sorted slot
  int fill[MB_MAXCELL];      // counting-sort cursors
  float red[4][16];          // block reductions (one row per wave)
  int wsum[4];
  float grid[14];            // g0x, g0y, inv_cell_x, inv_cell_y, Gx-1, Gy-1, Gx, centre xyz, radius, Cmax, g1x, g1y
  int ndefer;                // triangles whose box covers many points: done by the whole block after the lane loop
  int defer[MB_DEFER];       // (triangle index, packed cell box)
  int defer_box[MB_DEFER];
};

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

template <bool GROUPED>
__global__ __launch_bounds__(MC_THREADS) void contains_binned_kernel(const float* __restrict__ points,
                                                                     const float* __restrict__ verts,
                                                                     const int* __restrict__ faces, int P, int Nv, int F,
                                                                     int ptiles, int tchunk, int tsplit, int group_faces,
                                                                     int dbg_stop, int* __restrict__ hits) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int pt = blockIdx.x % ptiles, ts = blockIdx.x / ptiles;
  const float* __restrict__ pb = points + (size_t)b * P * 3;
  const float* __restrict__ vb = verts + (size_t)b * Nv * 3;
  __shared__ MbShared sh;

  const int p0 = pt * MB_PT;
  const int npts = min(MB_PT, P - p0);

  // ---- phase A: load the tile, project, bounding boxes -------------------------------------------------------------------
  float ox[MB_PPT], oy[MB_PPT], oz[MB_PPT], gx[MB_PPT], gy[MB_PPT];
  float mn[5] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};       // gx, gy, x, y, z
  float mx[5] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  bool finite = true;
#pragma unroll
  for (int k = 0; k < MB_PPT; ++k) {
    const int li = k * MC_THREADS + tid;
    if (li < npts) {
      const float* q = pb + (size_t)(p0 + li) * 3;
      ox[k] = q[0]; oy[k] = q[1]; oz[k] = q[2];
      gx[k] = proj_x(ox[k], oy[k], oz[k]);
      gy[k] = proj_y(ox[k], oy[k], oz[k]);
      const float v5[5] = {gx[k], gy[k], ox[k], oy[k], oz[k]};
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        mn[j] = fminf(mn[j], v5[j]);
        mx[j] = fmaxf(mx[j], v5[j]);
        finite = finite && (fabsf(v5[j]) <= FLT_MAX);  // false for NaN and inf
      }
    } else {
      ox[k] = oy[k] = oz[k] = gx[k] = gy[k] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    mn[j] = wave_min(mn[j]);
    mx[j] = obman_wave_max(mx[j]);
  }
  const bool wave_finite = __all(finite);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) { sh.red[wid][j] = mn[j]; sh.red[wid][5 + j] = mx[j]; }
    sh.red[wid][10] = wave_finite ? 1.f : 0.f;
  }
  for (int c = tid; c < MB_MAXCELL; c += MC_THREADS) sh.fill[c] = 0;
  for (int i = tid; i < MB_PT; i += MC_THREADS) sh.hit[i] = 0;
  if (tid == 0) sh.ndefer = 0;
  __syncthreads();
#ifdef OBMAN_ABLATION
  if (dbg_stop == 1) return;  // measurement only (-DOBMAN_ABLATION build + OBMAN_MC_DBG): wrong results; not in the product library
#endif
  if (tid == 0) {
    float lo[5], hi[5];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      lo[j] = fminf(fminf(sh.red[0][j], sh.red[1][j]), fminf(sh.red[2][j], sh.red[3][j]));
      hi[j] = fmaxf(fmaxf(sh.red[0][5 + j], sh.red[1][5 + j]), fmaxf(sh.red[2][5 + j], sh.red[3][5 + j]));
    }
    for (int w = 0; w < 4; ++w) ok = ok && sh.red[w][10] != 0.f;
    // grid: ~1 point per cell, square cells, Gx * Gy <= MB_MAXCELL.  Any choice is valid (the cell function is monotone);
    // a non-finite tile gets one cell = the all-pairs test.
    int Gx = 1, Gy = 1;
    float icx = 0.f, icy = 0.f;
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1];
    if (ok && npts > 1 && (ex > 0.f || ey > 0.f) && ex <= FLT_MAX && ey <= FLT_MAX) {
      const float target = (float)min(npts, MB_MAXCELL);
      const float exs = fmaxf(ex, 1e-3f * ey), eys = fmaxf(ey, 1e-3f * ex);
      const float cell = sqrtf(exs) * sqrtf(eys) / sqrtf(target);
      Gx = max(1, min(64, (int)ceilf(exs / cell)));
      Gy = max(1, min(64, (int)ceilf(eys / cell)));
      while (Gx * Gy > MB_MAXCELL) { if (Gx >= Gy) --Gx; else --Gy; }
      icx = ex > 0.f ? (float)Gx / ex : 0.f;
      icy = ey > 0.f ? (float)Gy / ey : 0.f;
      if (!(icx <= FLT_MAX)) { icx = 0.f; }
      if (!(icy <= FLT_MAX)) { icy = 0.f; }
      if (icx == 0.f) Gx = 1;
      if (icy == 0.f) Gy = 1;
    }
    sh.grid[0] = ok ? lo[0] : 0.f;
    sh.grid[1] = ok ? lo[1] : 0.f;
    sh.grid[2] = icx;
    sh.grid[3] = icy;
    sh.grid[4] = (float)(Gx - 1);
    sh.grid[5] = (float)(Gy - 1);
    sh.grid[6] = (float)Gx;
    // centre / radius of the tile (bounds |o - v0| per triangle) and the largest coordinate magnitude
    const float cx = 0.5f * (lo[2] + hi[2]), cy = 0.5f * (lo[3] + hi[3]), cz = 0.5f * (lo[4] + hi[4]);
    const float hx = hi[2] - cx, hy = hi[3] - cy, hz = hi[4] - cz;
    const float hx2 = cx - lo[2], hy2 = cy - lo[3], hz2 = cz - lo[4];
    const float rx = fmaxf(hx, hx2), ry = fmaxf(hy, hy2), rz = fmaxf(hz, hz2);
    sh.grid[7] = cx; sh.grid[8] = cy; sh.grid[9] = cz;
    sh.grid[10] = ok ? 1.001f * sqrtf(rx * rx + ry * ry + rz * rz) : __builtin_inff();
    float cm = 0.f;
#pragma unroll
    for (int j = 2; j < 5; ++j) cm = fmaxf(cm, fmaxf(fabsf(lo[j]), fabsf(hi[j])));
    sh.grid[11] = ok ? 1.75f * cm : __builtin_inff();  // >= |o| for every point of the tile (sqrt(3) * max coordinate)
    sh.grid[12] = ok ? hi[0] : 0.f;
    sh.grid[13] = ok ? hi[1] : 0.f;
  }
  __syncthreads();
  const float g0x = sh.grid[0], g0y = sh.grid[1], icx = sh.grid[2], icy = sh.grid[3];
  const float lastx = sh.grid[4], lasty = sh.grid[5];
  const int Gx = (int)sh.grid[6], Gy = (int)lasty + 1;
  const int ncell = Gx * Gy;

  // ---- phase B: counting sort of the points by cell --------------------------------------------------------------------
  int cell[MB_PPT];
#pragma unroll
  for (int k = 0; k < MB_PPT; ++k) {
    const int li = k * MC_THREADS + tid;
    cell[k] = grid_cell(gy[k], g0y, icy, lasty) * Gx + grid_cell(gx[k], g0x, icx, lastx);
    if (li < npts) atomicAdd(&sh.fill[cell[k]], 1);
  }
  __syncthreads();
  {  // exclusive scan of the cell counts: 4 consecutive cells per lane, wave scan, wave offsets
    int c4[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { c4[j] = sh.fill[tid * 4 + j]; s += c4[j]; }
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) sh.wsum[wid] = incl;
    __syncthreads();
    int base = incl - s;
    for (int w = 0; w < wid; ++w) base += sh.wsum[w];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh.start[tid * 4 + j] = base; sh.fill[tid * 4 + j] = base; base += c4[j]; }
    if (tid == MC_THREADS - 1) sh.start[MB_MAXCELL] = base;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < MB_PPT; ++k) {
    const int li = k * MC_THREADS + tid;
    if (li < npts) {
      const int slot = atomicAdd(&sh.fill[cell[k]], 1);
      sh.pts[slot] = make_float4(ox[k], oy[k], oz[k], __int_as_float(li));
    }
  }
  __syncthreads();
#ifdef OBMAN_ABLATION
  if (dbg_stop == 2) return;
#endif

  // ---- phase C: one triangle per lane against the cells its inflated box touches -----------------------------------------
  const float tcx = sh.grid[7], tcy = sh.grid[8], tcz = sh.grid[9], trad = sh.grid[10], cmax = sh.grid[11];
  const float g1x_pts = sh.grid[12], g1y_pts = sh.grid[13];
  const int tbeg = ts * tchunk, tend = min(F, tbeg + tchunk);
  for (int t = tbeg + tid; t < tend; t += MC_THREADS) {
    const TriSetup s = tri_setup(vb, faces + (size_t)t * 3);
    if (!(fabsf(s.det) >= TOL)) continue;  // parallel (or NaN): inv is NaN, no pair can hit
    const int bit = GROUPED ? 1 << (t / group_faces) : 1;
    const float e3x = s.e2x - s.e1x, e3y = s.e2y - s.e1y, e3z = s.e2z - s.e1z;
    const float l1 = dot3(s.e1x, s.e1y, s.e1z, s.e1x, s.e1y, s.e1z);
    const float l2 = dot3(s.e2x, s.e2y, s.e2z, s.e2x, s.e2y, s.e2z);
    const float l3 = dot3(e3x, e3y, e3z, e3x, e3y, e3z);
    const float Lm = sqrtf(fmaxf(l1, l2)), L3 = sqrtf(fmaxf(fmaxf(l1, l2), l3));
    const float absD = fabsf(s.D);
    const float Q = Lm * L3 / absD;
    const float dx = s.A.x - tcx, dy = s.A.y - tcy, dz = s.A.z - tcz;
    const float tmax = 1.001f * sqrtf(dot3(dx, dy, dz, dx, dy, dz)) + trad;
    const float call = cmax + fabsf(s.A.x) + fabsf(s.A.y) + fabsf(s.A.z) + L3;
    const float m = 128.f * MB_EPS * 1.01f * Q * (tmax + L3) + 64.f * MB_EPS * call;
    const float ff = 1.f + 1.2e-8f / absD;  // >= f = D / det
    const float gax = proj_x(s.A.x, s.A.y, s.A.z), gay = proj_y(s.A.x, s.A.y, s.A.z);
    const float g1x = ff * proj_x(s.e1x, s.e1y, s.e1z), g1y = ff * proj_y(s.e1x, s.e1y, s.e1z);
    const float g2x = ff * proj_x(s.e2x, s.e2y, s.e2z), g2y = ff * proj_y(s.e2x, s.e2y, s.e2z);
    const float xlo = gax + fminf(0.f, fminf(g1x, g2x)) - m, xhi = gax + fmaxf(0.f, fmaxf(g1x, g2x)) + m;
    const float ylo = gay + fminf(0.f, fminf(g1y, g2y)) - m, yhi = gay + fmaxf(0.f, fmaxf(g1y, g2y)) + m;
    int cx0 = 0, cx1 = Gx - 1, cy0 = 0, cy1 = Gy - 1;
    const bool boxed = (Q <= MB_QMAX) && (fabsf(xlo) <= FLT_MAX) && (fabsf(xhi) <= FLT_MAX) && (fabsf(ylo) <= FLT_MAX) &&
                       (fabsf(yhi) <= FLT_MAX);
    if (boxed) {
      // The tile's projected coordinates span [g0, g1] (the very fp32 values the cells are computed from): a box that misses
      // that range holds no point.  Otherwise a box partly beyond the grid clamps onto border cells (conservative).
      if (xhi < g0x || xlo > g1x_pts || yhi < g0y || ylo > g1y_pts) continue;
      cx0 = grid_cell(xlo, g0x, icx, lastx); cx1 = grid_cell(xhi, g0x, icx, lastx);
      cy0 = grid_cell(ylo, g0y, icy, lasty); cy1 = grid_cell(yhi, g0y, icy, lasty);
    }
#ifdef OBMAN_ABLATION
    if (dbg_stop == 3) continue;
#endif
    // A triangle whose cells hold many points (a wide box: large triangle, edge-on triangle with a large margin, or no box at
    // all) would keep its lane - and so its wave - busy for hundreds of tests: the block does those together afterwards.
    if (cy1 > cy0 || cx1 - cx0 > 3) {
      int cand = 0;
      for (int cy = cy0; cy <= cy1; ++cy) cand += sh.start[cy * Gx + cx1 + 1] - sh.start[cy * Gx + cx0];
      if (cand == 0) continue;
      if (cand > MB_DEFER_POINTS) {
        const int slot = atomicAdd(&sh.ndefer, 1);
        if (slot < MB_DEFER) {
          sh.defer[slot] = t;
          sh.defer_box[slot] = cx0 | (cx1 << 8) | (cy0 << 16) | (cy1 << 24);
          continue;
        }
      }
    }
    for (int cy = cy0; cy <= cy1; ++cy) {
      const int i0 = sh.start[cy * Gx + cx0], i1 = sh.start[cy * Gx + cx1 + 1];
      for (int i = i0; i < i1; ++i) {
        const float4 q = sh.pts[i];
        if (ray_hit(q.x, q.y, q.z, s.A, s.PU, s.PW, s.PN)) {
          if (GROUPED) atomicXor(&sh.hit[i], bit);
          else atomicAdd(&sh.hit[i], 1);
        }
      }
    }
  }
  __syncthreads();
#ifdef OBMAN_ABLATION
  if (dbg_stop == 4) return;
#endif
  {  // the wide triangles: one wave per triangle, the lanes stride over the candidate points of each grid row
    const int nd = min(sh.ndefer, MB_DEFER);
    for (int k = wid; k < nd; k += MC_THREADS / 64) {
      const int t = sh.defer[k], box = sh.defer_box[k];
      const int cx0 = box & 255, cx1 = (box >> 8) & 255, cy0 = (box >> 16) & 255, cy1 = (box >> 24) & 255;
      const TriSetup s = tri_setup(vb, faces + (size_t)t * 3);
      const int bit = GROUPED ? 1 << (t / group_faces) : 1;
      for (int cy = cy0; cy <= cy1; ++cy) {
        const int i0 = sh.start[cy * Gx + cx0], i1 = sh.start[cy * Gx + cx1 + 1];
        for (int i = i0 + lane; i < i1; i += 64) {
          const float4 q = sh.pts[i];
          if (ray_hit(q.x, q.y, q.z, s.A, s.PU, s.PW, s.PN)) {
            if (GROUPED) atomicXor(&sh.hit[i], bit);
            else atomicAdd(&sh.hit[i], 1);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- phase D: merge ---------------------------------------------------------------------------------------------------
  for (int i = tid; i < npts; i += MC_THREADS) {
    const int li = __float_as_int(sh.pts[i].w);
    const int h = sh.hit[i];
    int* dst = hits + (size_t)b * P + p0 + li;
    if (tsplit == 1) *dst = h;
    else if (h) { if (GROUPED) atomicXor(dst, h); else atomicAdd(dst, h); }
  }
  (void)ncell;
}

// OBMAN_MC_BINNED=0: the product entry points run the all-pairs kernel (A/B measurements).
bool binned_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("OBMAN_MC_BINNED");
    return !(e && e[0] == '0');
  }();
  return on;
}

int contains_args_ok(int B, int P, int Nv, int F, int group_faces, const int* hits) {
  if (B < 0 || P < 0 || Nv <= 0 || F < 0 || !hits) return -1;
  if (group_faces < 0 || (group_faces > 0 && (F % group_faces != 0 || F / group_faces > 32))) return -1;
  return 0;
}

int contains_launch(const float* points, const float* verts, const int* faces, int B, int P, int Nv, int F, int group_faces,
                    int* hits, hipStream_t st) {
  if (contains_args_ok(B, P, Nv, F, group_faces, hits)) return -1;
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

int contains_binned_launch(const float* points, const float* verts, const int* faces, int B, int P, int Nv, int F,
                           int group_faces, int* hits, hipStream_t st) {
  if (contains_args_ok(B, P, Nv, F, group_faces, hits)) return -1;
  if (B == 0 || P == 0) return 0;
  if (F == 0) return (int)obman_fill_u32(hits, 0u, (size_t)B * P, st);
  const int ptiles = obman_cdiv(P, MB_PT);
  // Every block re-sorts its point tile (a few us), then takes tchunk / 256 triangles per lane: split the triangle list until
  // the grid covers the chip ~8x, but keep >= 2 triangles per lane.
  int tsplit = 1;
  const long base_blocks = (long)B * ptiles;
  if (base_blocks < 2048) {
    tsplit = (int)((2048 + base_blocks - 1) / base_blocks);
    const int maxsplit = obman_cdiv(F, 2 * MC_THREADS);
    if (tsplit > maxsplit) tsplit = maxsplit;
    if (tsplit < 1) tsplit = 1;
  }
  int tchunk = obman_cdiv(obman_cdiv(F, tsplit), MC_THREADS) * MC_THREADS;
  tsplit = obman_cdiv(F, tchunk);
  if (tsplit > 1) {
    hipError_t e = obman_fill_u32(hits, 0u, (size_t)B * P, st);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(ptiles * tsplit, B);
#ifdef OBMAN_ABLATION
  static const int dbg = [] { const char* e = std::getenv("OBMAN_MC_DBG"); return e ? atoi(e) : 0; }();  // measurement only
#else
  const int dbg = 0;  // the product library carries no wrong-result switch
#endif
  ObmanProfScope prof(OBMAN_K_CONTAINS, st);
  if (group_faces)
    contains_binned_kernel<true><<<grid, MC_THREADS, 0, st>>>(points, verts, faces, P, Nv, F, ptiles, tchunk, tsplit, group_faces, dbg, hits);
  else
    contains_binned_kernel<false><<<grid, MC_THREADS, 0, st>>>(points, verts, faces, P, Nv, F, ptiles, tchunk, tsplit, 0, dbg, hits);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int obman_mesh_contains_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                       int F, int* hits, obman_stream_t stream) {
  if (binned_enabled()) return contains_binned_launch(points, verts, faces, B, P, Nv, F, 0, hits, (hipStream_t)stream);
  return contains_launch(points, verts, faces, B, P, Nv, F, 0, hits, (hipStream_t)stream);
}

extern "C" int obman_mesh_contains_groups_fwd(const float* points, const float* verts, const int* faces, int B, int P, int Nv,
                                              int F, int group_faces, int* parity_bits, obman_stream_t stream) {
  if (group_faces <= 0) return -1;
  if (binned_enabled())
    return contains_binned_launch(points, verts, faces, B, P, Nv, F, group_faces, parity_bits, (hipStream_t)stream);
  return contains_launch(points, verts, faces, B, P, Nv, F, group_faces, parity_bits, (hipStream_t)stream);
}

extern "C" int obman_mesh_contains_bruteforce_fwd(const float* points, const float* verts, const int* faces, int B, int P,
                                                  int Nv, int F, int group_faces, int* hits, obman_stream_t stream) {
  return contains_launch(points, verts, faces, B, P, Nv, F, group_faces, hits, (hipStream_t)stream);
}
