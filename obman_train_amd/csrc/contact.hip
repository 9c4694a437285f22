// K5 - contact / penetration loss tail, forward + backward (gfx950).
//
// Replaces contactloss.py:173-308 after the pair-min (K2) and inside test (K4): gather of the closest
// object vertex, norms / tanh penalties, attraction & repulsion masks (all | tips | zones with per-zone
// arg-min), the two batch-global masked means and the penetration-depth metrics - ~40 small torch
// kernels and two host syncs (`if valid_vals > 0`) in the reference, here one per-sample kernel plus
// a one-block finalize that keeps every scalar on the device.
// Backward: d/d(hand) directly, d/d(obj) by in-order sums over sorted closest-point groups (deterministic,
// no float atomics).  contact_target (all|obj|hand) only selects which side receives gradient.
// Quirks kept: `dist` mode thresholds SQUARED distances with the unsquared threshold,
// `dist_tanh` attracts everything, empty mask => loss 0.
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

enum { MODE_DIST_SQ = 0, MODE_DIST = 1, MODE_DIST_TANH = 2 };
enum { ZONES_ALL = 0, ZONES_LIST = 1, ZONES_ARGMIN = 2 };
enum { TARGET_ALL = 0, TARGET_OBJ = 1, TARGET_HAND = 2 };

__device__ __forceinline__ float penalty(int mode, float thresh, float sq, float anchor) {
  if (mode == MODE_DIST_SQ) return sq;
  if (mode == MODE_DIST) return anchor;
  return thresh * tanhf(anchor / thresh);
}

// d(penalty)/d(delta) = coef * delta
__device__ __forceinline__ float penalty_coef(int mode, float thresh, float anchor) {
  if (mode == MODE_DIST_SQ) return 2.f;
  if (anchor == 0.f) return 0.f;  // torch.norm backward masks the 0/0 case to 0
  if (mode == MODE_DIST) return 1.f / anchor;
  const float th = tanhf(anchor / thresh);
  return (1.f - th * th) / anchor;
}

struct ContactCfg {
  int V, N, zone_mode, n_zones, contact_mode, collision_mode;
  float contact_thresh, collision_thresh;
};

constexpr int CT_MAXV = 1024;

__global__ __launch_bounds__(256) void contact_fwd_kernel(const float* __restrict__ hand, const float* __restrict__ obj,
                                                          const int* __restrict__ idx21, const float* __restrict__ mins21,
                                                          const int* __restrict__ hits, const int* __restrict__ zone_ids,
                                                          const int* __restrict__ zone_off, ContactCfg cfg,
                                                          unsigned char* __restrict__ attr_mask,
                                                          unsigned char* __restrict__ rep_mask,
                                                          float* __restrict__ contact_points, float* __restrict__ partials) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = cfg.V;
  __shared__ unsigned char s_allow[CT_MAXV];
  __shared__ float s_red[4][6];
  const float* m21 = mins21 + (size_t)b * V;
  for (int v = tid; v < V; v += 256) s_allow[v] = cfg.zone_mode == ZONES_ALL ? 1 : 0;
  __syncthreads();
  if (cfg.zone_mode == ZONES_LIST) {
    const int n = zone_off[cfg.n_zones];
    for (int k = tid; k < n; k += 256) s_allow[zone_ids[k]] = 1;
  } else if (cfg.zone_mode == ZONES_ARGMIN) {
    // per zone keep only the vertex closest to the object (first in list order on ties)
    for (int z = wave; z < cfg.n_zones; z += 4) {
      const int beg = zone_off[z], end = zone_off[z + 1];
      float best = __builtin_inff();
      int bpos = 0x7fffffff;
      for (int k = beg + lane; k < end; k += 64) {
        const float d = m21[zone_ids[k]];
        if (d < best) { best = d; bpos = k; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int op = __shfl_xor(bpos, off, 64);
        if (ob < best || (ob == best && op < bpos)) { best = ob; bpos = op; }
      }
      if (lane == 0 && bpos != 0x7fffffff) s_allow[zone_ids[bpos]] = 1;
    }
  }
  __syncthreads();
  float s_mv = 0.f, s_mc = 0.f, s_pv = 0.f, s_pc = 0.f, s_dmax = 0.f, s_dsum = 0.f;
  const float* hb = hand + (size_t)b * V * 3;
  const float* ob = obj + (size_t)b * cfg.N * 3;
  for (int v = tid; v < V; v += 256) {
    const int j = idx21[(size_t)b * V + v];
    const float cx = ob[(size_t)j * 3], cy = ob[(size_t)j * 3 + 1], cz = ob[(size_t)j * 3 + 2];
    const float dx = cx - hb[v * 3], dy = cy - hb[v * 3 + 1], dz = cz - hb[v * 3 + 2];
    const float sq = dx * dx + dy * dy + dz * dz;
    const float anchor = sqrtf(sq);
    const bool exterior = (hits[(size_t)b * V + v] & 1) == 0;
    const float d21 = m21[v];
    bool below = true;
    if (cfg.contact_mode == MODE_DIST_SQ) below = d21 < cfg.contact_thresh * cfg.contact_thresh;
    else if (cfg.contact_mode == MODE_DIST) below = d21 < cfg.contact_thresh;
    const bool missed = below && exterior && s_allow[v];
    const bool penetr = !exterior;
    if (missed) { s_mv += penalty(cfg.contact_mode, cfg.contact_thresh, sq, anchor); s_mc += 1.f; }
    if (penetr) {
      s_pv += penalty(cfg.collision_mode, cfg.collision_thresh, sq, anchor);
      s_pc += 1.f;
      s_dmax = fmaxf(s_dmax, anchor);
      s_dsum += anchor;
    }
    attr_mask[(size_t)b * V + v] = missed ? 1 : 0;
    rep_mask[(size_t)b * V + v] = penetr ? 1 : 0;
    float* cp = contact_points + ((size_t)b * V + v) * 3;
    cp[0] = cx; cp[1] = cy; cp[2] = cz;
  }
  s_mv = obman_wave_sum(s_mv); s_mc = obman_wave_sum(s_mc); s_pv = obman_wave_sum(s_pv);
  s_pc = obman_wave_sum(s_pc); s_dsum = obman_wave_sum(s_dsum); s_dmax = obman_wave_max(s_dmax);
  if (lane == 0) {
    s_red[wave][0] = s_mv; s_red[wave][1] = s_mc; s_red[wave][2] = s_pv;
    s_red[wave][3] = s_pc; s_red[wave][4] = s_dmax; s_red[wave][5] = s_dsum;
  }
  __syncthreads();
  if (tid < 6) {
    float r;
    if (tid == 4) r = fmaxf(fmaxf(s_red[0][4], s_red[1][4]), fmaxf(s_red[2][4], s_red[3][4]));
    else r = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
    if (tid == 5) r /= (float)V;
    partials[(size_t)b * 8 + tid] = r;
  }
}

// out[0]=missed_loss out[1]=penetr_loss out[2]=max_penetr out[3]=mean_penetr out[4]=n_missed out[5]=n_penetr
__global__ __launch_bounds__(64) void contact_finalize_kernel(const float* __restrict__ partials, int B, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  for (int b = lane; b < B; b += 64)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] += partials[(size_t)b * 8 + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) acc[k] = obman_wave_sum(acc[k]);
  if (lane == 0) {
    out[0] = acc[1] > 0.f ? acc[0] / acc[1] : 0.f;
    out[1] = acc[3] > 0.f ? acc[2] / acc[3] : 0.f;
    out[2] = acc[4] / (float)B;
    out[3] = acc[5] / (float)B;
    out[4] = acc[1];
    out[5] = acc[3];
    out[6] = 0.f;
    out[7] = 0.f;
  }
}

// Per-vertex d(loss)/d(delta), delta = closest_obj_vertex - hand_vertex.
__device__ __forceinline__ void contact_gdelta(const float* hb, const float* ob, int v, int j, bool missed, bool penetr,
                                               float wm, float wp, const ContactCfg& cfg, float& gx, float& gy, float& gz) {
  const float dx = ob[(size_t)j * 3] - hb[v * 3], dy = ob[(size_t)j * 3 + 1] - hb[v * 3 + 1], dz = ob[(size_t)j * 3 + 2] - hb[v * 3 + 2];
  const float anchor = sqrtf(dx * dx + dy * dy + dz * dz);
  float coef = 0.f;
  if (missed) coef += wm * penalty_coef(cfg.contact_mode, cfg.contact_thresh, anchor);
  if (penetr) coef += wp * penalty_coef(cfg.collision_mode, cfg.collision_thresh, anchor);
  gx = coef * dx; gy = coef * dy; gz = coef * dz;
}

// Backward, one block per (slice of <= CB_SLICE object points, sample).  A hand vertex belongs to the slice that holds its closest
// object point; the block evaluates the per-vertex gradient g_delta of ITS vertices only, writes their hand side, and builds the
// object side of its slice: the gradient of point n is the sum, in ascending hand-vertex order, of the g_delta of the vertices whose
// closest point is n.
//   1. the slice's vertices are compacted (ballot + popcount per wave, wave totals through LDS) as keys (point << 10 | vertex);
//   2. a bitonic sort of the keys in LDS (a hand close to one patch puts most of its 778 vertices into one slice: 55 compare-exchange
//      stages at worst) makes every point's group contiguous, ascending in the vertex number;
//   3. one lane per sorted vertex: gather, square root, tanh -> g_delta record in LDS, hand-side gradient to memory;
//   4. the first lane of a group is the point's OWNER and adds the group's records in order - the additions, and their order, of the
//      point-major scan of rounds 1-4 (every point looked at all V vertices: 114 us at 16 050 points, ~4 x that at 64 050);
//   5. the slice's LDS image goes out as one coalesced write (points nobody maps to: zeros).
// Bit-identical results: tests/test_contact_gpu.py compares with a sequential fp32 scatter of the hand-side gradient.
constexpr int CB_SLICE = 2048;
constexpr int CB_VBITS = 10;

__global__ __launch_bounds__(256) void contact_bwd_kernel(const float* __restrict__ hand, const float* __restrict__ obj,
                                                          const int* __restrict__ idx21,
                                                          const unsigned char* __restrict__ attr_mask,
                                                          const unsigned char* __restrict__ rep_mask,
                                                          const float* __restrict__ out, const float* __restrict__ g_missed,
                                                          const float* __restrict__ g_penetr, ContactCfg cfg, int target, int slice,
                                                          float* __restrict__ grad_hand, float* __restrict__ grad_obj) {
  static_assert(CT_MAXV == 4 * 256 && CT_MAXV == 1 << CB_VBITS, "four vertices per thread, vertex number in the key's low bits");
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, V = cfg.V, N = cfg.N;
  const float* hb = hand + (size_t)b * V * 3;
  const float* ob = obj + (size_t)b * N * 3;
  const float wm = (g_missed && out[4] > 0.f) ? g_missed[0] / out[4] : 0.f;
  const float wp = (g_penetr && out[5] > 0.f) ? g_penetr[0] / out[5] : 0.f;
  const int n0 = blockIdx.x * slice, cnt = min(N - n0, slice);
  __shared__ float s_gx[CT_MAXV], s_gy[CT_MAXV], s_gz[CT_MAXV];  // g_delta of the sorted vertices
  __shared__ int s_key[CT_MAXV], s_cnt[16];
  __shared__ float s_acc[CB_SLICE * 3];
  if (grad_obj)
    for (int i = tid; i < cnt * 3; i += 256) s_acc[i] = 0.f;
  bool in[4];
  int pre[4], key[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int v = it * 256 + tid;
    const int j = v < V ? idx21[(size_t)b * V + v] : -1;
    in[it] = j >= n0 && j < n0 + cnt;
    // an index outside [0, N) lies in no slice (the forward never produces one; a caller-supplied idx21 might): the first slice's
    // block gives such a vertex a zero gradient instead of leaving its rows of grad_hand unwritten
    if (blockIdx.x == 0 && grad_hand && v < V && (j < 0 || j >= N)) {
      float* g = grad_hand + ((size_t)b * V + v) * 3;
      g[0] = 0.f; g[1] = 0.f; g[2] = 0.f;
    }
    key[it] = ((j - n0) << CB_VBITS) | v;
    const unsigned long long bal = __ballot(in[it]);
    pre[it] = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[it * 4 + wave] = __popcll(bal);
  }
  __syncthreads();
  int n_in = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if ((q & 3) == wave && in[q >> 2]) s_key[n_in + pre[q >> 2]] = key[q >> 2];
    n_in += s_cnt[q];
  }
  if (grad_obj) {  // block-uniform.  (Without an object side nothing depends on the order.)
    int n_pad = 1;
    while (n_pad < n_in) n_pad <<= 1;
    for (int i = n_in + tid; i < n_pad; i += 256) s_key[i] = 0x7fffffff;
    for (int size = 2; size <= n_pad; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        __syncthreads();
        for (int t = tid; t < (n_pad >> 1); t += 256) {
          const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
          const int a = s_key[lo], c = s_key[hi];
          if ((a > c) == ((lo & size) == 0)) { s_key[lo] = c; s_key[hi] = a; }
        }
      }
  }
  __syncthreads();
  for (int k = tid; k < n_in; k += 256) {
    const int v = s_key[k] & (CT_MAXV - 1), j = n0 + (s_key[k] >> CB_VBITS);
    float gx, gy, gz;
    contact_gdelta(hb, ob, v, j, attr_mask[(size_t)b * V + v], rep_mask[(size_t)b * V + v], wm, wp, cfg, gx, gy, gz);
    if (grad_hand) {
      float* g = grad_hand + ((size_t)b * V + v) * 3;
      const bool on = target != TARGET_OBJ;
      g[0] = on ? -gx : -0.f; g[1] = on ? -gy : -0.f; g[2] = on ? -gz : -0.f;
    }
    const bool on = target != TARGET_HAND;
    s_gx[k] = on ? gx : 0.f; s_gy[k] = on ? gy : 0.f; s_gz[k] = on ? gz : 0.f;
  }
  if (!grad_obj) return;  // block-uniform
  __syncthreads();
  for (int k = tid; k < n_in; k += 256) {
    const int p = s_key[k] >> CB_VBITS;
    if (k > 0 && (s_key[k - 1] >> CB_VBITS) == p) continue;  // not the first vertex of its point
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int u = k; u < n_in && (s_key[u] >> CB_VBITS) == p; ++u) { ax += s_gx[u]; ay += s_gy[u]; az += s_gz[u]; }
    float* a = s_acc + p * 3;
    a[0] = ax; a[1] = ay; a[2] = az;
  }
  __syncthreads();
  float* g = grad_obj + ((size_t)b * N + n0) * 3;
  for (int i = tid; i < cnt * 3; i += 256) g[i] = s_acc[i];
}

bool cfg_ok(const ContactCfg& c) {
  return c.V > 0 && c.V <= CT_MAXV && c.N > 0 && c.zone_mode >= 0 && c.zone_mode <= 2 && c.contact_mode >= 0 &&
         c.contact_mode <= 2 && c.collision_mode >= 0 && c.collision_mode <= 2;
}

}  // namespace

extern "C" {

int obman_contact_fwd(const float* hand, const float* obj, const int* idx21, const float* mins21, const int* hits,
                      int B, int V, int N, const int* zone_ids, const int* zone_offsets, int n_zones, int zone_mode,
                      int contact_mode, float contact_thresh, int collision_mode, float collision_thresh,
                      unsigned char* attr_mask, unsigned char* rep_mask, float* contact_points, float* partials,
                      float* out, obman_stream_t stream) {
  ContactCfg cfg{V, N, zone_mode, n_zones, contact_mode, collision_mode, contact_thresh, collision_thresh};
  if (B <= 0 || !cfg_ok(cfg)) return -1;
  if (zone_mode != ZONES_ALL && (!zone_ids || !zone_offsets || n_zones <= 0)) return -2;
  hipStream_t st = (hipStream_t)stream;
  contact_fwd_kernel<<<B, 256, 0, st>>>(hand, obj, idx21, mins21, hits, zone_ids, zone_offsets, cfg, attr_mask, rep_mask,
                                        contact_points, partials);
  OBMAN_LAUNCH_CHECK();
  contact_finalize_kernel<<<1, 64, 0, st>>>(partials, B, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_contact_bwd(const float* hand, const float* obj, const int* idx21, const unsigned char* attr_mask,
                      const unsigned char* rep_mask, const float* out, const float* g_missed, const float* g_penetr,
                      int B, int V, int N, int contact_mode, float contact_thresh, int collision_mode,
                      float collision_thresh, int target, float* grad_hand, float* grad_obj, obman_stream_t stream) {
  ContactCfg cfg{V, N, 0, 0, contact_mode, collision_mode, contact_thresh, collision_thresh};
  if (B <= 0 || !cfg_ok(cfg) || target < 0 || target > 2) return -1;
  // slices: at most CB_SLICE points (the LDS image), and at least ~12 per sample when there are points for them (parallelism: a
  // block's own work is a few dozen vertices).  Without an object-side gradient one block per sample owns every vertex.
  int nslices = obman_cdiv(N, CB_SLICE);
  const int want = N / 32 < 12 ? (N / 32 > 0 ? N / 32 : 1) : 12;
  if (nslices < want) nslices = want;
  const int slice = grad_obj ? obman_cdiv(N, nslices) : N;
  dim3 grid(obman_cdiv(N, slice), B);
  contact_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(hand, obj, idx21, attr_mask, rep_mask, out, g_missed, g_penetr,
                                                             cfg, target, slice, grad_hand, grad_obj);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
