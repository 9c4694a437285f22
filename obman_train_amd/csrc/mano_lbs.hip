// K7 - MANO linear-blend skinning, forward + hand-written backward, four vertex-tile workgroups per sample (gfx950).
//
// Replaces the external manopth.ManoLayer the reference calls at manobranch.py:92-105,170-182
// (a chain of ~60 small torch ops: launch-latency bound on a GPU) with ONE fused kernel per
// direction: PCA -> axis-angle -> quaternion Rodrigues -> shape/pose blend shapes -> joint
// regression -> 16-joint kinematic chain -> rest-pose removal -> skinning -> 21 joints -> centring
// -> x1000.  Pose input modes (`use_pca` argument of the launchers): 1 = root axis-angle + ncomps PCA
// coefficients, 0 = root + 45 axis-angle values, 2 = sixteen 3x3 rotation matrices used as given (manobranch.py:52-54,
// 126-128: ManoLayer(use_pca=False) fed [b,16,3,3]).  The model (1.4 MB, shared by every sample) is one packed fp32 blob with
// blend-shape bases stored k-major so that lane e reads element e of every basis: fully coalesced;
// it stays L2 / Infinity-Cache resident.  Per-sample state (local/global rotations, joints, posed
// rest shape: 2766 floats) is saved for the backward, which therefore never re-reads posedirs for
// the forward product.  Bound: L2 bandwidth / latency, not HBM, not MFMA.
#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace {

constexpr int NV = 778, NE = 2334, NJ = 16, NPM = 135;
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

constexpr int NTILE = 4, TILE_V = 195, FT = 256;   // 4 vertex tiles x 195 vertices (last one 193), 256 threads per block
constexpr int LIST_V = TILE_V + 7;                   // tile + 5 fingertips + 2 palm vertices
constexpr int PART = 400;                            // per (sample, tile) backward partials: 192 joint sums | 145 rows | 48 joint grads

__device__ __forceinline__ bool center_needs_verts(int center_idx, int root_palm) {
  if (center_idx < 0) return false;
  const int src = c_reorder[center_idx];
  return src >= 16 || (root_palm && src == 0);
}

// Pose phase shared by every block of a sample: PCA -> axis-angle -> rotations, joints, kinematic chain, rest-pose removal.
// (A few hundred flops; recomputed per vertex tile instead of a second launch.)  Ends with a barrier.
__device__ __forceinline__ void mano_pose_phase(const float* __restrict__ M, const float* __restrict__ p, const float* __restrict__ betas_b,
                                                int ncomps, int use_pca, int tid, int nthreads, float* s_aa, float* s_beta, float* s_R,
                                                float* s_pm, float* s_J, float* s_GR, float* s_Gt, float* s_trel, float* s_tab) {
  // Small model tables go through LDS first (one coalesced pass): a serial loop of dependent L2 loads (30 PCA steps) costs
  // ~0.7 us per step otherwise.  s_tab: [ncomps*45 PCA rows | 48 pose coefficients]
  const int npose = use_pca == 2 ? 0 : 3 + (use_pca ? ncomps : 45);
  if (use_pca == 1)
    for (int i = tid; i < ncomps * 45; i += nthreads) s_tab[i] = M[OFF_COMPS + i];
  for (int i = tid; i < npose; i += nthreads) s_tab[45 * 45 + i] = p[i];
  if (tid < 10) s_beta[tid] = betas_b ? betas_b[tid] : 0.f;
  if (use_pca == 2) {  // rotation matrices given: no PCA, no Rodrigues
    for (int i = tid; i < 144; i += nthreads) {
      const float r = p[i];
      s_R[i] = r;
      if (i >= 9) { const int k = i % 9; s_pm[i - 9] = r - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f); }
    }
    if (tid < 48) s_aa[tid] = 0.f;
  }
  __syncthreads();
  const float* sp = s_tab + 45 * 45;
  if (use_pca != 2 && tid < 3) s_aa[tid] = sp[tid];
  if (use_pca != 2 && tid >= 64 && tid < 109) {
    const int m = tid - 64;
    float h = M[OFF_MEAN + m];
    if (use_pca) {
      float h1 = 0.f;
      int k = 0;
      for (; k + 1 < ncomps; k += 2) {
        h = __fmaf_rn(sp[3 + k], s_tab[k * 45 + m], h);
        h1 = __fmaf_rn(sp[4 + k], s_tab[(k + 1) * 45 + m], h1);
      }
      if (k < ncomps) h = __fmaf_rn(sp[3 + k], s_tab[k * 45 + m], h);
      h += h1;
    } else {
      h += sp[3 + m];
    }
    s_aa[3 + m] = h;
  }
  __syncthreads();
  if (use_pca != 2 && tid < NJ) {
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
    float js[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) js[k] = M[OFF_JS + k * 48 + e];  // 10 independent loads
    float j = M[OFF_JT + e];
#pragma unroll
    for (int k = 0; k < 10; ++k) j = __fmaf_rn(js[k], s_beta[k], j);
    s_J[e] = j;
  }
  __syncthreads();
  if (tid < 5) {  // one finger per lane, 3 joints each
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
  __syncthreads();
  if (tid < NJ) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
      s_trel[tid * 3 + r] = s_Gt[tid * 3 + r] - (s_GR[tid * 9 + r * 3] * s_J[tid * 3] + s_GR[tid * 9 + r * 3 + 1] * s_J[tid * 3 + 1] +
                                                 s_GR[tid * 9 + r * 3 + 2] * s_J[tid * 3 + 2]);
  }
  __syncthreads();
}

// Forward: grid (NTILE, B).  Each block blends + skins its 195 vertices (the 1.26 MB pose basis of a sample is read by 4 CUs
// instead of one); block 0 of a sample additionally evaluates the 7 special vertices and writes the joints.
__global__ __launch_bounds__(FT) void mano_fwd_kernel(const float* __restrict__ m_right, const float* __restrict__ m_left,
                                                       const int* __restrict__ side, const float* __restrict__ pose,
                                                       const float* __restrict__ betas, int npose, int ncomps, int use_pca,
                                                       int center_idx, int root_palm, float* __restrict__ verts,
                                                       float* __restrict__ joints, float* __restrict__ state) {
  const int vt = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* __restrict__ M = (side && side[b]) ? m_left : m_right;
  __shared__ float s_aa[48], s_beta[10], s_R[144], s_pm[NPM + 1], s_J[48], s_GR[144], s_Gt[48], s_trel[48];
  __shared__ float s_vp[LIST_V * 3];  // posed rest shape of the block's vertex list, overwritten by the skinned vertices
  __shared__ int s_vl[LIST_V];
  __shared__ float s_jc[63];
  __shared__ float s_tab[45 * 45 + 48];
  const int v0 = vt * TILE_V, nv = min(NV, v0 + TILE_V) - v0;
  const bool spec = vt == 0 || center_needs_verts(center_idx, root_palm);
  const int nl = nv + (spec ? 7 : 0);
  if (tid < nv) s_vl[tid] = v0 + tid;
  if (spec && tid < 7) s_vl[nv + tid] = tid < 5 ? (int)M[OFF_TIPS + tid] : (int)M[OFF_PALM + tid - 5];
  mano_pose_phase(M, pose + (size_t)b * npose, betas ? betas + (size_t)b * 10 : nullptr, ncomps, use_pca, tid, FT, s_aa, s_beta, s_R,
                  s_pm, s_J, s_GR, s_Gt, s_trel, s_tab);
  // blend shapes: v_posed[e] = T[e] + sum_k S[k][e] beta[k] + sum_k P[k][e] pose_map[k]
  for (int i = tid; i < nl * 3; i += FT) {
    const int e = s_vl[i / 3] * 3 + i % 3;
    float acc = M[OFF_VT + e];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc = __fmaf_rn(M[OFF_SD + k * NE + e], s_beta[k], acc);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* pd = M + OFF_PD + e;
#pragma unroll 15
    for (int k = 0; k < NPM; k += 3) {
      a0 = __fmaf_rn(pd[(size_t)k * NE], s_pm[k], a0);
      a1 = __fmaf_rn(pd[(size_t)(k + 1) * NE], s_pm[k + 1], a1);
      a2 = __fmaf_rn(pd[(size_t)(k + 2) * NE], s_pm[k + 2], a2);
    }
    s_vp[i] = acc + (a0 + a1 + a2);
  }
  __syncthreads();
  if (state) {
    float* st = state + (size_t)b * OBMAN_MANO_STATE_FLOATS;
    if (vt == 0) {
      for (int k = tid; k < 144; k += FT) { st[S_R + k] = s_R[k]; st[S_GR + k] = s_GR[k]; }
      if (tid < 48) { st[S_J + tid] = s_J[tid]; st[S_GT + tid] = s_Gt[tid]; st[S_AA + tid] = s_aa[tid]; }
    }
    for (int i = tid; i < nv * 3; i += FT) st[S_VP + v0 * 3 + i] = s_vp[i];
  }
  __syncthreads();
  if (tid < nl) {  // skinning: one vertex per lane
    const int v = s_vl[tid];
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
    const float x = s_vp[tid * 3], y = s_vp[tid * 3 + 1], z = s_vp[tid * 3 + 2];
    s_vp[tid * 3] = T[0] * x + T[1] * y + T[2] * z + T[9];
    s_vp[tid * 3 + 1] = T[3] * x + T[4] * y + T[5] * z + T[10];
    s_vp[tid * 3 + 2] = T[6] * x + T[7] * y + T[8] * z + T[11];
  }
  __syncthreads();
  if (tid < 63) {  // un-reordered joints: 16 chain joints (or the palm point) + 5 fingertip vertices
    const int j = tid / 3, c = tid % 3;
    float val = 0.f;
    if (j < 16) {
      val = s_Gt[j * 3 + c];
      if (j == 0 && root_palm && spec) val = (s_vp[(nv + 5) * 3 + c] + s_vp[(nv + 6) * 3 + c]) * 0.5f;
    } else if (spec) {
      val = s_vp[(nv + j - 16) * 3 + c];
    }
    s_jc[tid] = val;
  }
  __syncthreads();
  float cx = 0.f, cy = 0.f, cz = 0.f;
  if (center_idx >= 0) {
    const int src = c_reorder[center_idx];
    cx = s_jc[src * 3]; cy = s_jc[src * 3 + 1]; cz = s_jc[src * 3 + 2];
  }
  float* vo = verts + (size_t)b * NE + v0 * 3;
  for (int i = tid; i < nv * 3; i += FT) {
    const int c = i % 3;
    vo[i] = (s_vp[i] - (c == 0 ? cx : (c == 1 ? cy : cz))) * 1000.f;
  }
  if (vt == 0 && tid < 63) {
    const int j = tid / 3, c = tid % 3;
    joints[(size_t)b * 63 + tid] = (s_jc[c_reorder[j] * 3 + c] - (c == 0 ? cx : (c == 1 ? cy : cz))) * 1000.f;
  }
}

// Backward, part 1: grid (NTILE, B).  Per vertex tile: skinning backward (gradient of the posed rest shape + partial sums of the
// 16 joint-transform gradients) and the tile's share of d/d(pose_map), d/d(beta).  Partials -> part[b][tile][PART].
__global__ __launch_bounds__(FT) void mano_bwd_tile_kernel(const float* __restrict__ m_right, const float* __restrict__ m_left,
                                                           const int* __restrict__ side, const float* __restrict__ state,
                                                           const float* __restrict__ g_verts, const float* __restrict__ g_joints,
                                                           int center_idx, int root_palm, float* __restrict__ part) {
  const int vt = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ M = (side && side[b]) ? m_left : m_right;
  const float* __restrict__ st = state + (size_t)b * OBMAN_MANO_STATE_FLOATS;
  float* __restrict__ out = part + ((size_t)b * NTILE + vt) * PART;
  __shared__ float s_GR[144];
  __shared__ float s_vp[TILE_V * 3], s_gv[TILE_V * 3];
  __shared__ float s_gj[63], s_cat[63], s_red[4][4];
  __shared__ float s_w[TILE_V * NJ];   // skinning weights of the tile, vertex-major
  __shared__ float s_mm[4][256];       // per-wave 16x16 MFMA results
  const int v0 = vt * TILE_V, nv = min(NV, v0 + TILE_V) - v0;
  for (int k = tid; k < 144; k += FT) s_GR[k] = st[S_GR + k];
  for (int i = tid; i < nv * 3; i += FT) {
    s_vp[i] = st[S_VP + v0 * 3 + i];
    s_gv[i] = g_verts ? g_verts[(size_t)b * NE + v0 * 3 + i] * 1000.f : 0.f;
  }
  // the centring term needs the gradient sum over ALL vertices of the sample (every tile recomputes it: 9 loads / lane)
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (g_verts) {
    for (int v = tid; v < NV; v += FT) {
      const float* g = g_verts + (size_t)b * NE + v * 3;
      sx += g[0]; sy += g[1]; sz += g[2];
    }
  }
  sx = obman_wave_sum(sx) * 1000.f; sy = obman_wave_sum(sy) * 1000.f; sz = obman_wave_sum(sz) * 1000.f;
  if (lane == 0) { s_red[wave][0] = sx; s_red[wave][1] = sy; s_red[wave][2] = sz; }
  if (tid < 63) s_gj[tid] = g_joints ? g_joints[(size_t)b * 63 + tid] * 1000.f : 0.f;
  __syncthreads();
  if (tid == 0) {
    if (center_idx >= 0) {
      for (int c = 0; c < 3; ++c) {
        float tot = (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
        for (int j = 0; j < 21; ++j) tot += s_gj[j * 3 + c];
        s_gj[center_idx * 3 + c] -= tot;
      }
    }
    for (int j = 0; j < 21; ++j)
      for (int c = 0; c < 3; ++c) s_cat[c_reorder[j] * 3 + c] = s_gj[j * 3 + c];
    if (root_palm) {
      for (int q = 0; q < 2; ++q) {
        const int v = (int)M[OFF_PALM + q] - v0;
        if (v >= 0 && v < nv)
          for (int c = 0; c < 3; ++c) s_gv[v * 3 + c] += 0.5f * s_cat[c];
      }
      for (int c = 0; c < 3; ++c) s_cat[c] = 0.f;
    }
    for (int t = 0; t < 5; ++t) {
      const int v = (int)M[OFF_TIPS + t] - v0;
      if (v >= 0 && v < nv)
        for (int c = 0; c < 3; ++c) s_gv[v * 3 + c] += s_cat[(16 + t) * 3 + c];
    }
  }
  __syncthreads();
  if (vt == 0 && tid < 48) out[337 + tid] = s_cat[tid];  // d/d(chain joint translations), consumed by the chain kernel
  // skinning backward.  verts[v] = sum_i w_vi (GR_i vp_v + trel_i)
  //   d/d(v_posed_v) = sum_i w_vi GR_i^T gv_v                                  (per lane, VALU)
  //   d/d(GR_i | trel_i) = sum_v w_vi [gv_v (x) vp_v | gv_v]  = W^T X,  X[v] = (gv (x) vp, gv) in R^12
  // The second line is a [16 x nv] x [nv x 12] product: done on the matrix core (v_mfma_f32_16x16x4_f32, exact fp32 fma
  // chains) instead of 192 wave reductions of 6 dependent ds_bpermute steps each (51 of the kernel's 76 us before).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const bool vok = tid < nv;
  float wv[NJ];
#pragma unroll
  for (int i = 0; i < NJ; ++i) wv[i] = vok ? M[OFF_W + i * NV + v0 + tid] : 0.f;  // 16 independent coalesced loads
  if (vok) {
#pragma unroll
    for (int i = 0; i < NJ; ++i) s_w[tid * NJ + i] = wv[i];
  }
  __syncthreads();
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int j = lane & 15, g = lane >> 4;
    for (int t = wave; t * 4 < nv; t += 4) {
      const int v = 4 * t + g;
      float a = 0.f, x = 0.f;
      if (v < nv) {
        a = s_w[v * NJ + j];                                   // A[i = lane&15][k = lane>>4] = w[v][i]
        if (j < 9) x = s_gv[v * 3 + j / 3] * s_vp[v * 3 + j % 3];  // B[k][j] = X[v][j]
        else if (j < 12) x = s_gv[v * 3 + j - 9];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_mm[wave][(g * 4 + r) * 16 + j] = acc[r];  // D: row = 4*(lane>>4)+r (joint), col = lane&15
  }
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  if (vok) {
    const float gvx = s_gv[tid * 3], gvy = s_gv[tid * 3 + 1], gvz = s_gv[tid * 3 + 2];
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const float* G = &s_GR[i * 9];
      const float wx = wv[i] * gvx, wy = wv[i] * gvy, wz = wv[i] * gvz;
      gpx += G[0] * wx + G[3] * wy + G[6] * wz;
      gpy += G[1] * wx + G[4] * wy + G[7] * wz;
      gpz += G[2] * wx + G[5] * wy + G[8] * wz;
    }
  }
  __syncthreads();  // MFMA operands have been read: s_gv can now receive d(loss)/d(v_posed)
  if (vok) { s_gv[tid * 3] = gpx; s_gv[tid * 3 + 1] = gpy; s_gv[tid * 3 + 2] = gpz; }
  if (tid < NJ * 12) {
    const int i = tid / 12, k = tid % 12;
    out[tid] = (s_mm[0][i * 16 + k] + s_mm[1][i * 16 + k]) + (s_mm[2][i * 16 + k] + s_mm[3][i * 16 + k]);
  }
  __syncthreads();
  // the tile's share of <P[k], g_vp> (k < 135) and <S[k], g_vp> (10 shape rows): one row per wave pass, lanes over the tile
  const int ne = nv * 3;
  constexpr int RG = 6;  // rows per pass: 6 x ~10 independent loads in flight per lane (one row at a time exposes a full L2
                         // round trip per row: 36 rows x ~2 us)
  for (int row0 = wave * RG; row0 < NPM + 10; row0 += 4 * RG) {
    float acc[RG];
    const float* basis[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const int row = row0 + r < NPM + 10 ? row0 + r : NPM + 9;
      basis[r] = (row < NPM ? M + OFF_PD + (size_t)row * NE : M + OFF_SD + (size_t)(row - NPM) * NE) + v0 * 3;
      acc[r] = 0.f;
    }
    constexpr int NIT = (TILE_V * 3 + 63) / 64;  // 10 lane-strided steps cover the tile; fully unrolled so that all
#pragma unroll                                    // RG x NIT loads are issued before the first one is consumed
    for (int it = 0; it < NIT; ++it) {
      const int e = lane + 64 * it;
      const bool ok = e < ne;
      const int ec = ok ? e : 0;
      const float g = ok ? s_gv[ec] : 0.f;
#pragma unroll
      for (int r = 0; r < RG; ++r) acc[r] = __fmaf_rn(basis[r][ec], g, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const float v = obman_wave_sum(acc[r]);
      if (lane == 0 && row0 + r < NPM + 10) out[192 + row0 + r] = v;
    }
  }
}

// Backward, part 2: grid (B), 64 threads.  Sums the tile partials, then the serial part: rest-pose removal, kinematic chain,
// Rodrigues and PCA backward.
__global__ __launch_bounds__(64) void mano_bwd_chain_kernel(const float* __restrict__ m_right, const float* __restrict__ m_left,
                                                            const int* __restrict__ side, const float* __restrict__ state,
                                                            const float* __restrict__ part, int npose, int ncomps, int use_pca,
                                                            float* __restrict__ g_pose, float* __restrict__ g_betas) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* __restrict__ M = (side && side[b]) ? m_left : m_right;
  const float* __restrict__ st = state + (size_t)b * OBMAN_MANO_STATE_FLOATS;
  const float* __restrict__ pb = part + (size_t)b * NTILE * PART;
  __shared__ float s_R[144], s_J[48], s_GR[144], s_aa[48];
  __shared__ float s_gGR[144], s_gGt[48], s_gtrel[48], s_gJ[48], s_gR[144];
  __shared__ float s_root[5][16];
  __shared__ float s_gpm[NPM + 10], s_gaa[48];
  __shared__ float s_comps[45 * 45], s_js[480];  // small model tables: staged once, coalesced (serial L2 loads are ~0.7 us each)
  if (use_pca == 1)
    for (int i = tid; i < ncomps * 45; i += 64) s_comps[i] = M[OFF_COMPS + i];
  if (g_betas)
    for (int i = tid; i < 480; i += 64) s_js[i] = M[OFF_JS + i];
  for (int k = tid; k < 144; k += 64) { s_R[k] = st[S_R + k]; s_GR[k] = st[S_GR + k]; }
  if (tid < 48) { s_J[tid] = st[S_J + tid]; s_aa[tid] = st[S_AA + tid]; s_gGt[tid] = pb[337 + tid]; }
  for (int k = tid; k < 192 + NPM + 10; k += 64) {
    const float r = (pb[k] + pb[PART + k]) + (pb[2 * PART + k] + pb[3 * PART + k]);
    if (k < 192) {
      const int i = k / 12, q = k % 12;
      if (q < 9) s_gGR[i * 9 + q] = r; else s_gtrel[i * 3 + q - 9] = r;
    } else {
      s_gpm[k - 192] = r;
    }
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
  __syncthreads();
  if (g_betas && tid < 10) {
    float g = s_gpm[NPM + tid];
    for (int e = 0; e < 48; ++e) g = __fmaf_rn(s_js[tid * 48 + e], s_gJ[e], g);
    g_betas[(size_t)b * 10 + tid] = g;
  }
  float* gp = g_pose + (size_t)b * npose;
  if (use_pca == 2) {  // rotation-matrix input: d/dR_i = chain gradient + pose-blend-shape gradient (R_i - I enters pose_map)
    for (int k = tid; k < 144; k += 64) gp[k] = s_gR[k] + (k >= 9 ? s_gpm[k - 9] : 0.f);
    return;
  }
  // phase 7: Rodrigues backward
  if (tid >= 32 && tid < 32 + NJ) {
    const int i = tid - 32;
    float G[9], ga[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = s_gR[i * 9 + k] + (i > 0 ? s_gpm[(i - 1) * 9 + k] : 0.f);
    rodrigues_bwd(&s_aa[i * 3], G, ga);
    s_gaa[i * 3] = ga[0]; s_gaa[i * 3 + 1] = ga[1]; s_gaa[i * 3 + 2] = ga[2];
  }
  __syncthreads();
  if (tid < 3) gp[tid] = s_gaa[tid];
  if (use_pca) {
    for (int k = tid; k < ncomps; k += 64) {
      float g = 0.f;
      for (int m = 0; m < 45; ++m) g = __fmaf_rn(s_comps[k * 45 + m], s_gaa[3 + m], g);
      gp[3 + k] = g;
    }
  } else if (tid < 45) {
    gp[3 + tid] = s_gaa[3 + tid];
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
  if (use_pca < 0 || use_pca > 2 || (use_pca == 1 && (ncomps < 0 || ncomps > 45))) return -2;
  if (center_idx < -1 || center_idx > 20) return -3;
  if (side && !model_left) return -4;
  if (B == 0) return 0;
  const int npose = use_pca == 2 ? 144 : 3 + (use_pca ? ncomps : 45);
  ObmanProfScope prof(OBMAN_K_MANO_FWD, (hipStream_t)stream);
  mano_fwd_kernel<<<dim3(NTILE, B), FT, 0, (hipStream_t)stream>>>(model_right, model_left, side, pose, betas, npose, ncomps, use_pca,
                                                       center_idx, root_palm, verts, joints, state);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_mano_lbs_bwd(const float* model_right, const float* model_left, const int* side, const float* state,
                       const float* g_verts, const float* g_joints, int B, int ncomps, int use_pca, int center_idx,
                       int root_palm, float* g_pose, float* g_betas, float* scratch, obman_stream_t stream) {
  if (B < 0 || !model_right || !state || !g_pose || !scratch) return -1;
  if (use_pca < 0 || use_pca > 2 || (use_pca == 1 && (ncomps < 0 || ncomps > 45))) return -2;
  if (side && !model_left) return -4;
  if (B == 0) return 0;
  const int npose = use_pca == 2 ? 144 : 3 + (use_pca ? ncomps : 45);
  hipStream_t st = (hipStream_t)stream;
  ObmanProfScope prof(OBMAN_K_MANO_BWD, st);
  mano_bwd_tile_kernel<<<dim3(NTILE, B), FT, 0, st>>>(model_right, model_left, side, state, g_verts, g_joints, center_idx, root_palm,
                                                       scratch);
  OBMAN_LAUNCH_CHECK();
  mano_bwd_chain_kernel<<<B, 64, 0, st>>>(model_right, model_left, side, state, scratch, npose, ncomps, use_pca, g_pose, g_betas);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_mano_bwd_scratch_floats(int B) { return B * NTILE * PART; }

}  // extern "C"
