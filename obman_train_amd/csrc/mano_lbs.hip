// K7 - MANO linear-blend skinning, forward + hand-written backward, one workgroup per sample (gfx950).
//
// Replaces the external manopth.ManoLayer the reference calls at manobranch.py:92-105,170-182
// (a chain of ~60 small torch ops: launch-latency bound on a GPU) with ONE fused kernel per
// direction: PCA -> axis-angle -> quaternion Rodrigues -> shape/pose blend shapes -> joint
// regression -> 16-joint kinematic chain -> rest-pose removal -> skinning -> 21 joints -> centring
// -> x1000.  The model (1.4 MB, shared by every sample) is one packed fp32 blob with
// blend-shape bases stored k-major so that lane e reads element e of every basis: fully coalesced;
// it stays L2 / Infinity-Cache resident.  Per-sample state (local/global rotations, joints, posed
// rest shape: 2766 floats) is saved for the backward, which therefore never re-reads posedirs for
// the forward product.  Bound: L2 bandwidth / latency, not HBM, not MFMA.
#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace {

constexpr int NV = 778, NE = 2334, NJ = 16, NPM = 135;
constexpr int MT = 1024;            // threads per block (16 waves): one sample per block, 4x the loads in flight of a 256-thread block
constexpr int MW = MT / 64;
// model blob layout (float offsets) - mirrored by obman_train_amd/mano_model.py
constexpr int OFF_COMPS = 0;                  // [45][45] PCA basis rows
constexpr int OFF_MEAN = OFF_COMPS + 2025;    // [45]
constexpr int OFF_VT = OFF_MEAN + 45;         // [2334] template, e = v*3+c
constexpr int OFF_SD = OFF_VT + NE;           // [10][2334] shape basis, k-major
constexpr int OFF_PD = OFF_SD + 10 * NE;      // [135][2334] pose basis, k-major
constexpr int OFF_JT = OFF_PD + NPM * NE;     // [48] J_regressor . template
constexpr int OFF_JS = OFF_JT + 48;           // [10][48] J_regressor . shapedirs, k-major
constexpr int OFF_W = OFF_JS + 480;           // [16][778] skinning weights, joint-major
constexpr int OFF_TIPS = OFF_W + NJ * NV;     // [5] fingertip vertex ids (as floats)
constexpr int OFF_PALM = OFF_TIPS + 5;        // [2]
// per-sample saved state (float offsets)
constexpr int S_R = 0, S_J = 144, S_GR = 192, S_GT = 336, S_AA = 384, S_VP = 432;

__constant__ int c_reorder[21] = {0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20};

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {  // C = A B
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

__device__ __forceinline__ void rodrigues(const float* a, float* R) {
  const float sx = a[0] + 1e-8f, sy = a[1] + 1e-8f, sz = a[2] + 1e-8f;
  const float theta = sqrtf(sx * sx + sy * sy + sz * sz);
  const float inv = 1.f / theta, h = 0.5f * theta;
  float sh, ch;
  sincosf(h, &sh, &ch);
  float w = ch, x = sh * a[0] * inv, y = sh * a[1] * inv, z = sh * a[2] * inv;
  const float qn = 1.f / sqrtf(w * w + x * x + y * y + z * z);
  w *= qn; x *= qn; y *= qn; z *= qn;
  const float w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const float wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  R[0] = w2 + x2 - y2 - z2; R[1] = 2 * xy - 2 * wz;     R[2] = 2 * wy + 2 * xz;
  R[3] = 2 * wz + 2 * xy;     R[4] = w2 - x2 + y2 - z2; R[5] = 2 * yz - 2 * wx;
  R[6] = 2 * xz - 2 * wy;     R[7] = 2 * wx + 2 * yz;     R[8] = w2 - x2 - y2 + z2;
}

// d(loss)/d(axis-angle) from d(loss)/dR, same quaternion route as rodrigues().
__device__ __forceinline__ void rodrigues_bwd(const float* a, const float* G, float* ga) {
  const float sx = a[0] + 1e-8f, sy = a[1] + 1e-8f, sz = a[2] + 1e-8f;
  const float theta = sqrtf(sx * sx + sy * sy + sz * sz);
  const float inv = 1.f / theta, h = 0.5f * theta;
  float sh, ch;
  sincosf(h, &sh, &ch);
  const float nx = a[0] * inv, ny = a[1] * inv, nz = a[2] * inv;
  const float w0 = ch, x0 = sh * nx, y0 = sh * ny, z0 = sh * nz;
  const float qn = 1.f / sqrtf(w0 * w0 + x0 * x0 + y0 * y0 + z0 * z0);
  const float w = w0 * qn, x = x0 * qn, y = y0 * qn, z = z0 * qn;
  const float gw = 2.f * (w * (G[0] + G[4] + G[8]) + z * (G[3] - G[1]) + y * (G[2] - G[6]) + x * (G[7] - G[5]));
  const float gx = 2.f * (x * (G[0] - G[4] - G[8]) + y * (G[1] + G[3]) + z * (G[2] + G[6]) + w * (G[7] - G[5]));
  const float gy = 2.f * (y * (-G[0] + G[4] - G[8]) + x * (G[1] + G[3]) + w * (G[2] - G[6]) + z * (G[5] + G[7]));
  const float gz = 2.f * (z * (-G[0] - G[4] + G[8]) + w * (G[3] - G[1]) + x * (G[2] + G[6]) + y * (G[5] + G[7]));
  // q = q0 / |q0|
  const float dot = w * gw + x * gx + y * gy + z * gz;
  const float g0w = (gw - w * dot) * qn, g0x = (gx - x * dot) * qn, g0y = (gy - y * dot) * qn, g0z = (gz - z * dot) * qn;
  // q0 = (cos h, sin h * n)
  const float gh = -sh * g0w + ch * (nx * g0x + ny * g0y + nz * g0z);
  const float gnx = sh * g0x, gny = sh * g0y, gnz = sh * g0z;
  // n = a / theta ; h = theta / 2 ; theta = |a + 1e-8|
  const float gtheta = 0.5f * gh - (gnx * a[0] + gny * a[1] + gnz * a[2]) * inv * inv;
  ga[0] = gnx * inv + gtheta * sx * inv;
  ga[1] = gny * inv + gtheta * sy * inv;
  ga[2] = gnz * inv + gtheta * sz * inv;
}

__global__ __launch_bounds__(MT) void mano_fwd_kernel(const float* __restrict__ m_right, const float* __restrict__ m_left,
                                                       const int* __restrict__ side, const float* __restrict__ pose,
                                                       const float* __restrict__ betas, int npose, int ncomps, int use_pca,
                                                       int center_idx, int root_palm, float* __restrict__ verts,
                                                       float* __restrict__ joints, float* __restrict__ state) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ M = (side && side[b]) ? m_left : m_right;
  __shared__ float s_aa[48], s_beta[10], s_R[144], s_pm[NPM + 1], s_J[48], s_GR[144], s_Gt[48], s_trel[48];
  __shared__ float s_vp[NE];   // posed rest shape, overwritten in place by the skinned vertices
  __shared__ float s_jc[63];   // 16 chain joints + 5 tips (un-reordered), then centre
  const float* p = pose + (size_t)b * npose;

  if (tid < 10) s_beta[tid] = betas ? betas[(size_t)b * 10 + tid] : 0.f;
  if (tid >= 64 && tid < 67) s_aa[tid - 64] = p[tid - 64];
  if (tid >= 128 && tid < 173) {
    const int m = tid - 128;
    float h = M[OFF_MEAN + m];
    if (use_pca) {
      for (int k = 0; k < ncomps; ++k) h = __fmaf_rn(p[3 + k], M[OFF_COMPS + k * 45 + m], h);
    } else {
      h += p[3 + m];
    }
    s_aa[3 + m] = h;
  }
  __syncthreads();
  if (tid < NJ) {
    float R[9];
    rodrigues(&s_aa[tid * 3], R);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      s_R[tid * 9 + k] = R[k];
      if (tid > 0) s_pm[(tid - 1) * 9 + k] = R[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
    }
  }
  if (tid >= 64 && tid < 112) {
    const int e = tid - 64;
    float j = M[OFF_JT + e];
#pragma unroll
    for (int k = 0; k < 10; ++k) j = __fmaf_rn(M[OFF_JS + k * 48 + e], s_beta[k], j);
    s_J[e] = j;
  }
  __syncthreads();
  // kinematic chain: lanes 0..4 walk one finger each (3 joints), lane 5 writes the root
  if (tid < 5) {
    float Rp[9], tp[3], Jp[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rp[k] = s_R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) { tp[k] = s_J[k]; Jp[k] = s_J[k]; }
    for (int l = 0; l < 3; ++l) {
      const int i = 1 + 3 * tid + l;
      float Rl[9], Ri[9], ti[3], d[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) Rl[k] = s_R[i * 9 + k];
      mat3_mul(Rp, Rl, Ri);
#pragma unroll
      for (int k = 0; k < 3; ++k) d[k] = s_J[i * 3 + k] - Jp[k];
#pragma unroll
      for (int r = 0; r < 3; ++r) ti[r] = Rp[r * 3] * d[0] + Rp[r * 3 + 1] * d[1] + Rp[r * 3 + 2] * d[2] + tp[r];
#pragma unroll
      for (int k = 0; k < 9; ++k) { s_GR[i * 9 + k] = Ri[k]; Rp[k] = Ri[k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { s_Gt[i * 3 + k] = ti[k]; tp[k] = ti[k]; Jp[k] = s_J[i * 3 + k]; }
    }
  } else if (tid == 5) {
    for (int k = 0; k < 9; ++k) s_GR[k] = s_R[k];
    for (int k = 0; k < 3; ++k) s_Gt[k] = s_J[k];
  }
  // blend shapes: v_posed[e] = T[e] + sum_k S[k][e] beta[k] + sum_k P[k][e] pose_map[k]
  for (int e = tid; e < NE; e += MT) {
    float acc = M[OFF_VT + e];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc = __fmaf_rn(M[OFF_SD + k * NE + e], s_beta[k], acc);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* pd = M + OFF_PD + e;
#pragma unroll 9
    for (int k = 0; k < NPM; k += 3) {
      a0 = __fmaf_rn(pd[(size_t)k * NE], s_pm[k], a0);
      a1 = __fmaf_rn(pd[(size_t)(k + 1) * NE], s_pm[k + 1], a1);
      a2 = __fmaf_rn(pd[(size_t)(k + 2) * NE], s_pm[k + 2], a2);
    }
    s_vp[e] = acc + (a0 + a1 + a2);
  }
  __syncthreads();
  if (tid < NJ) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      s_trel[tid * 3 + r] = s_Gt[tid * 3 + r] - (s_GR[tid * 9 + r * 3] * s_J[tid * 3] + s_GR[tid * 9 + r * 3 + 1] * s_J[tid * 3 + 1] +
                                                 s_GR[tid * 9 + r * 3 + 2] * s_J[tid * 3 + 2]);
  }
  if (state) {
    float* st = state + (size_t)b * OBMAN_MANO_STATE_FLOATS;
    for (int k = tid; k < 144; k += MT) { st[S_R + k] = s_R[k]; st[S_GR + k] = s_GR[k]; }
    if (tid < 48) { st[S_J + tid] = s_J[tid]; st[S_GT + tid] = s_Gt[tid]; st[S_AA + tid] = s_aa[tid]; }
    for (int e = tid; e < NE; e += MT) st[S_VP + e] = s_vp[e];
  }
  __syncthreads();
  // skinning: one vertex per lane
  for (int v = tid; v < NV; v += MT) {
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const float w = M[OFF_W + i * NV + v];
#pragma unroll
      for (int k = 0; k < 9; ++k) T[k] = __fmaf_rn(w, s_GR[i * 9 + k], T[k]);
#pragma unroll
      for (int k = 0; k < 3; ++k) T[9 + k] = __fmaf_rn(w, s_trel[i * 3 + k], T[9 + k]);
    }
    const float x = s_vp[v * 3], y = s_vp[v * 3 + 1], z = s_vp[v * 3 + 2];
    const float ox = T[0] * x + T[1] * y + T[2] * z + T[9];
    const float oy = T[3] * x + T[4] * y + T[5] * z + T[10];
    const float oz = T[6] * x + T[7] * y + T[8] * z + T[11];
    s_vp[v * 3] = ox; s_vp[v * 3 + 1] = oy; s_vp[v * 3 + 2] = oz;
  }
  __syncthreads();
  if (tid < 63) {
    const int j = tid / 3, c = tid % 3;
    float val;
    if (j < 16) {
      val = s_Gt[j * 3 + c];
      if (j == 0 && root_palm) {
        const int p0 = (int)M[OFF_PALM], p1 = (int)M[OFF_PALM + 1];
        val = (s_vp[p0 * 3 + c] + s_vp[p1 * 3 + c]) * 0.5f;
      }
    } else {
      val = s_vp[(int)M[OFF_TIPS + j - 16] * 3 + c];
    }
    s_jc[tid] = val;
  }
  __syncthreads();
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (center_idx >= 0) {
    const int src = c_reorder[center_idx];
    cx = s_jc[src * 3]; cy = s_jc[src * 3 + 1]; cz = s_jc[src * 3 + 2];
  }
  float* vo = verts + (size_t)b * NE;
  for (int e = tid; e < NE; e += MT) {
    const int c = e % 3;
    vo[e] = (s_vp[e] - (c == 0 ? cx : (c == 1 ? cy : cz))) * 1000.f;
  }
  if (tid < 63) {
    const int j = tid / 3, c = tid % 3;
    joints[(size_t)b * 63 + tid] = (s_jc[c_reorder[j] * 3 + c] - (c == 0 ? cx : (c == 1 ? cy : cz))) * 1000.f;
  }
}

__global__ __launch_bounds__(MT) void mano_bwd_kernel(const float* __restrict__ m_right, const float* __restrict__ m_left,
                                                       const int* __restrict__ side, const float* __restrict__ state,
                                                       const float* __restrict__ g_verts, const float* __restrict__ g_joints,
                                                       int npose, int ncomps, int use_pca, int center_idx, int root_palm,
                                                       float* __restrict__ g_pose, float* __restrict__ g_betas) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ M = (side && side[b]) ? m_left : m_right;
  const float* __restrict__ st = state + (size_t)b * OBMAN_MANO_STATE_FLOATS;
  __shared__ float s_R[144], s_J[48], s_GR[144], s_Gt[48], s_aa[48];
  __shared__ float s_vp[NE], s_gv[NE];          // posed rest shape; grads of skinned verts, later of v_posed
  __shared__ float s_gj[63], s_cat[63];         // joint grads (21-order) and un-reordered
  __shared__ float s_red[MW][4];                 // centre reduction
  __shared__ float s_part[MW][NJ][12];           // per-wave partial joint-transform grads
  __shared__ float s_gGR[144], s_gGt[48], s_gtrel[48], s_gJ[48], s_gR[144];
  __shared__ float s_root[5][16];               // per-finger contributions to the root (9 + 3 + 3)
  __shared__ float s_gpm[NPM + 10], s_gaa[48];

  for (int k = tid; k < 144; k += MT) { s_R[k] = st[S_R + k]; s_GR[k] = st[S_GR + k]; }
  if (tid < 48) { s_J[tid] = st[S_J + tid]; s_Gt[tid] = st[S_GT + tid]; s_aa[tid] = st[S_AA + tid]; }
  for (int e = tid; e < NE; e += MT) s_vp[e] = st[S_VP + e];
  // phase 0: scale, centre
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int v = tid; v < NV; v += MT) {
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (g_verts) {
      const float* g = g_verts + (size_t)b * NE + v * 3;
      gx = g[0] * 1000.f; gy = g[1] * 1000.f; gz = g[2] * 1000.f;
    }
    s_gv[v * 3] = gx; s_gv[v * 3 + 1] = gy; s_gv[v * 3 + 2] = gz;
    sx += gx; sy += gy; sz += gz;
  }
  sx = obman_wave_sum(sx); sy = obman_wave_sum(sy); sz = obman_wave_sum(sz);
  if (lane == 0) { s_red[wave][0] = sx; s_red[wave][1] = sy; s_red[wave][2] = sz; }
  if (tid < 63) s_gj[tid] = g_joints ? g_joints[(size_t)b * 63 + tid] * 1000.f : 0.f;
  __syncthreads();
  if (tid == 0) {
    if (center_idx >= 0) {
      for (int c = 0; c < 3; ++c) {
        float tot = 0.f;
        for (int w = 0; w < MW; ++w) tot += s_red[w][c];
        for (int j = 0; j < 21; ++j) tot += s_gj[j * 3 + c];
        s_gj[center_idx * 3 + c] -= tot;
      }
    }
    for (int j = 0; j < 21; ++j)
      for (int c = 0; c < 3; ++c) s_cat[c_reorder[j] * 3 + c] = s_gj[j * 3 + c];
    if (root_palm) {
      const int p0 = (int)M[OFF_PALM], p1 = (int)M[OFF_PALM + 1];
      for (int c = 0; c < 3; ++c) {
        const float g = 0.5f * s_cat[c];
        s_gv[p0 * 3 + c] += g;
        s_gv[p1 * 3 + c] += g;
        s_cat[c] = 0.f;
      }
    }
    for (int t = 0; t < 5; ++t) {
      const int v = (int)M[OFF_TIPS + t];
      for (int c = 0; c < 3; ++c) s_gv[v * 3 + c] += s_cat[(16 + t) * 3 + c];
    }
    for (int k = 0; k < 48; ++k) s_gGt[k] = s_cat[k];
  }
  __syncthreads();
  // phase 2: skinning backward.  verts[v] = sum_i w_vi (GR_i vp_v + trel_i)
  static_assert(MT >= NV, "one vertex per lane");
  const bool vok = tid < NV;
  const float gvx = vok ? s_gv[tid * 3] : 0.f, gvy = vok ? s_gv[tid * 3 + 1] : 0.f, gvz = vok ? s_gv[tid * 3 + 2] : 0.f;
  const float px = vok ? s_vp[tid * 3] : 0.f, py = vok ? s_vp[tid * 3 + 1] : 0.f, pz = vok ? s_vp[tid * 3 + 2] : 0.f;
  __syncthreads();  // every lane has its s_gv in registers: s_gv can now receive d(loss)/d(v_posed)
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
#pragma unroll 4
  for (int i = 0; i < NJ; ++i) {
    const float* G = &s_GR[i * 9];
    const float wji = vok ? M[OFF_W + i * NV + tid] : 0.f;
    const float wx = wji * gvx, wy = wji * gvy, wz = wji * gvz;
    float acc[12] = {wx * px, wx * py, wx * pz, wy * px, wy * py, wy * pz, wz * px, wz * py, wz * pz, wx, wy, wz};
    gpx += G[0] * wx + G[3] * wy + G[6] * wz;  // d/d(v_posed) = sum_i w GR_i^T gv
    gpy += G[1] * wx + G[4] * wy + G[7] * wz;
    gpz += G[2] * wx + G[5] * wy + G[8] * wz;
    if (wave < (NV + 63) / 64) {  // waves beyond the last vertex contribute zeros
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        const float r = obman_wave_sum(acc[k]);
        if (lane == 0) s_part[wave][i][k] = r;
      }
    }
  }
  if (vok) { s_gv[tid * 3] = gpx; s_gv[tid * 3 + 1] = gpy; s_gv[tid * 3 + 2] = gpz; }
  __syncthreads();
  if (tid < NJ * 12) {
    const int i = tid / 12, k = tid % 12;
    float r = 0.f;
    for (int w = 0; w < (NV + 63) / 64; ++w) r += s_part[w][i][k];
    if (k < 9) s_gGR[i * 9 + k] = r; else s_gtrel[i * 3 + k - 9] = r;
  }
  __syncthreads();
  // phase 4: trel_i = Gt_i - GR_i J_i
  if (tid < NJ) {
    const int i = tid;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g = s_gtrel[i * 3 + r];
      s_gGt[i * 3 + r] += g;
#pragma unroll
      for (int c = 0; c < 3; ++c) s_gGR[i * 9 + r * 3 + c] -= g * s_J[i * 3 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      s_gJ[i * 3 + c] = -(s_GR[i * 9 + c] * s_gtrel[i * 3] + s_GR[i * 9 + 3 + c] * s_gtrel[i * 3 + 1] + s_GR[i * 9 + 6 + c] * s_gtrel[i * 3 + 2]);
  }
  __syncthreads();
  // phase 5: chain backward, one finger per lane, tips -> root
  if (tid < 5) {
    float rootc[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) rootc[k] = 0.f;
    for (int l = 2; l >= 0; --l) {
      const int i = 1 + 3 * tid + l, pidx = l == 0 ? 0 : i - 1;
      const float* GRp = &s_GR[pidx * 9];
      float gGRi[9], gGti[3], d[3], dj[3];
#pragma unroll
      for (int k = 0; k < 9; ++k) gGRi[k] = s_gGR[i * 9 + k];
#pragma unroll
      for (int k = 0; k < 3; ++k) { gGti[k] = s_gGt[i * 3 + k]; d[k] = s_J[i * 3 + k] - s_J[pidx * 3 + k]; }
      // gR_i = GRp^T gGR_i
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          s_gR[i * 9 + r * 3 + c] = GRp[r] * gGRi[c] + GRp[3 + r] * gGRi[3 + c] + GRp[6 + r] * gGRi[6 + c];
      // dj = GRp^T gGt_i
#pragma unroll
      for (int c = 0; c < 3; ++c) dj[c] = GRp[c] * gGti[0] + GRp[3 + c] * gGti[1] + GRp[6 + c] * gGti[2];
      // parent accumulations: gGR_p += gGR_i R_i^T + gGt_i (x) d ; gGt_p += gGt_i ; gJ_i += dj ; gJ_p -= dj
      float up[9];
      const float* Ri = &s_R[i * 9];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
          up[r * 3 + c] = gGRi[r * 3] * Ri[c * 3] + gGRi[r * 3 + 1] * Ri[c * 3 + 1] + gGRi[r * 3 + 2] * Ri[c * 3 + 2] + gGti[r] * d[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) s_gJ[i * 3 + c] += dj[c];
      if (l > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) s_gGR[pidx * 9 + k] += up[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_gGt[pidx * 3 + k] += gGti[k]; s_gJ[pidx * 3 + k] -= dj[k]; }
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) rootc[k] = up[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) { rootc[9 + k] = gGti[k]; rootc[12 + k] = -dj[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 15; ++k) s_root[tid][k] = rootc[k];
  }
  __syncthreads();
  if (tid < 15) {  // root: GR_0 = R_0, Gt_0 = J_0 ; fixed finger order
    float add = 0.f;
    for (int f = 0; f < 5; ++f) add += s_root[f][tid];
    if (tid < 9) s_gR[tid] = s_gGR[tid] + add;
    else if (tid < 12) s_gGt[tid - 9] += add;   // becomes d/dJ_0 below
    else s_gJ[tid - 12] += add;
  }
  __syncthreads();
  if (tid < 3) s_gJ[tid] += s_gGt[tid];
  // phase 6: d/d(pose_map[k]) = <P[k], g_vp>, d/d(beta[k]) = <S[k], g_vp> (+ joint path): one row per wave pass
  for (int row = wave; row < NPM + 10; row += MW) {
    const float* basis = row < NPM ? M + OFF_PD + (size_t)row * NE : M + OFF_SD + (size_t)(row - NPM) * NE;
    float a0 = 0.f, a1 = 0.f;
    int e = lane;
    for (; e + 64 < NE; e += 128) {
      a0 = __fmaf_rn(basis[e], s_gv[e], a0);
      a1 = __fmaf_rn(basis[e + 64], s_gv[e + 64], a1);
    }
    if (e < NE) a0 = __fmaf_rn(basis[e], s_gv[e], a0);
    const float r = obman_wave_sum(a0 + a1);
    if (lane == 0) s_gpm[row] = r;
  }
  __syncthreads();
  if (g_betas && tid < 10) {
    float g = s_gpm[NPM + tid];
    for (int e = 0; e < 48; ++e) g = __fmaf_rn(M[OFF_JS + tid * 48 + e], s_gJ[e], g);
    g_betas[(size_t)b * 10 + tid] = g;
  }
  // phase 7: Rodrigues backward
  if (tid >= 64 && tid < 64 + NJ) {
    const int i = tid - 64;
    float G[9], ga[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = s_gR[i * 9 + k] + (i > 0 ? s_gpm[(i - 1) * 9 + k] : 0.f);
    rodrigues_bwd(&s_aa[i * 3], G, ga);
    s_gaa[i * 3] = ga[0]; s_gaa[i * 3 + 1] = ga[1]; s_gaa[i * 3 + 2] = ga[2];
  }
  __syncthreads();
  float* gp = g_pose + (size_t)b * npose;
  if (tid < 3) gp[tid] = s_gaa[tid];
  if (use_pca) {
    if (tid >= 64 && tid < 64 + ncomps) {
      const int k = tid - 64;
      float g = 0.f;
      for (int m = 0; m < 45; ++m) g = __fmaf_rn(M[OFF_COMPS + k * 45 + m], s_gaa[3 + m], g);
      gp[3 + k] = g;
    }
  } else if (tid >= 64 && tid < 64 + 45) {
    gp[3 + tid - 64] = s_gaa[3 + tid - 64];
  }
}

}  // namespace

extern "C" {

int obman_mano_model_floats(void) { return OFF_PALM + 2; }
int obman_mano_state_floats(void) { return OBMAN_MANO_STATE_FLOATS; }

int obman_mano_lbs_fwd(const float* model_right, const float* model_left, const int* side, const float* pose,
                       const float* betas, int B, int ncomps, int use_pca, int center_idx, int root_palm,
                       float* verts, float* joints, float* state, obman_stream_t stream) {
  if (B < 0 || !model_right || !pose || !verts || !joints) return -1;
  if (use_pca ? (ncomps < 0 || ncomps > 45) : 0) return -2;
  if (center_idx < -1 || center_idx > 20) return -3;
  if (side && !model_left) return -4;
  if (B == 0) return 0;
  const int npose = 3 + (use_pca ? ncomps : 45);
  ObmanProfScope prof(OBMAN_K_MANO_FWD, (hipStream_t)stream);
  mano_fwd_kernel<<<B, MT, 0, (hipStream_t)stream>>>(model_right, model_left, side, pose, betas, npose, ncomps, use_pca,
                                                       center_idx, root_palm, verts, joints, state);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_mano_lbs_bwd(const float* model_right, const float* model_left, const int* side, const float* state,
                       const float* g_verts, const float* g_joints, int B, int ncomps, int use_pca, int center_idx,
                       int root_palm, float* g_pose, float* g_betas, obman_stream_t stream) {
  if (B < 0 || !model_right || !state || !g_pose) return -1;
  if (use_pca ? (ncomps < 0 || ncomps > 45) : 0) return -2;
  if (side && !model_left) return -4;
  if (B == 0) return 0;
  const int npose = 3 + (use_pca ? ncomps : 45);
  ObmanProfScope prof(OBMAN_K_MANO_BWD, (hipStream_t)stream);
  mano_bwd_kernel<<<B, MT, 0, (hipStream_t)stream>>>(model_right, model_left, side, state, g_verts, g_joints, npose, ncomps,
                                                       use_pca, center_idx, root_palm, g_pose, g_betas);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
