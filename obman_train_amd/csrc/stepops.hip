// Round 6: the small operators of the training step that used to be stock PyTorch launches (1.64 of the
// 10.06 ms of kernel time of a configs[2] step were ~150 "other PyTorch kernels", each costing its full 4 - 13 us because the GPU
// is back-to-back busy).  Everything here is bandwidth- or latency-trivial; the point is ONE launch where there were 4 - 40.
//
//   K11 obman_adam_step          torch.optim.Adam over every parameter (traineval.py:112-127) as a multi-tensor kernel that
//                                also writes the bf16 SHADOW copy of a filter the autocast encoder reads (no per-step cast kernels)
//   K12 obman_affine_points_*    objpoints3d = scale * verts + trans (atlasbranch.py:133-138) and its backward (two full
//                                [B,N,3] -> [B,1,3] reductions in the stock form: 51 us each at 16 050 points)
//   K13 obman_mse_terms_*        the MSE heads of ManoLoss / AtlasLoss (manobranch.py:251-318, atlasbranch.py:211-228): k
//                                mean-squared errors over k differently sized tensors, forward and backward, one launch each
//   K14 obman_gt_object_stats    centroid, centred cloud and max point norm of the ground-truth object points
//                                (atlasbranch.py:211-222: gt.mean(1), gt - centroid, norm(.,2,2).max(1))
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------ K11 Adam
constexpr int ADAM_MAX = 64;        // tensors per launch: 64 x 52 bytes of kernel arguments (limit 4 KB)
constexpr int ADAM_CHUNK = 4096;    // elements per block: 256 threads x 4 float4
struct AdamBatch {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  unsigned short* shadow[ADAM_MAX];
  float* step[ADAM_MAX];
  long n[ADAM_MAX];
  int first_block[ADAM_MAX + 1];  // prefix of the per-tensor block counts
  int count;
};
static_assert(sizeof(AdamBatch) <= 4096, "kernel argument block");

__global__ void adam_tick_kernel(AdamBatch b) {
  const int t = threadIdx.x;
  if (t < b.count) *b.step[t] += 1.f;
}

__device__ __forceinline__ unsigned short bf16_rne(float x) {  // round to nearest even, NaN kept quiet (torch's float -> bfloat16)
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

struct AdamK { float lr_over_bc1, bc2_sqrt, beta1w, beta2, beta2w, eps, wd; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamK& k) {
  // torch/aten/src/ATen/native/cuda/fused_adam_utils.cuh adam_math (ADAM_MODE::ORIGINAL, no amsgrad, no maximize), same operation order
  if (k.wd != 0.f) g += p * k.wd;
  m = m + k.beta1w * (g - m);                       // lerp(exp_avg, grad, 1 - beta1)
  v = k.beta2 * v + k.beta2w * g * g;
  const float denom = sqrtf(v) / k.bc2_sqrt + k.eps;
  p -= k.lr_over_bc1 * m / denom;
}

// lr, beta1, beta2 arrive as DOUBLES (what torch.optim.Adam holds): 1 - beta2 formed in fp32 from the rounded 0.999f is 1.3e-5 off
// the 0.001 every torch implementation multiplies by (first version: exp_avg_sq drifted from torch's by exactly that factor)
__global__ __launch_bounds__(256) void adam_kernel(AdamBatch b, double lr, double beta1, double beta2, float eps, float wd) {
  // which tensor: binary search of the block id in the (scalar) prefix table
  int lo = 0, hi = b.count;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= b.first_block[mid]) lo = mid; else hi = mid;
  }
  const int t = lo;
  const long n = b.n[t], base = (long)(blockIdx.x - b.first_block[t]) * ADAM_CHUNK;
  float* __restrict__ p = b.p[t];
  const float* __restrict__ g = b.g[t];
  float* __restrict__ m = b.m[t];
  float* __restrict__ v = b.v[t];
  unsigned short* __restrict__ sh = b.shadow[t];
  const double step = (double)*b.step[t];  // already incremented by adam_tick_kernel
  AdamK k;
  k.lr_over_bc1 = (float)(lr / (1.0 - pow(beta1, step)));
  k.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, step));
  k.beta1w = (float)(1.0 - beta1); k.beta2 = (float)beta2; k.beta2w = (float)(1.0 - beta2); k.eps = eps; k.wd = wd;
  const bool vec = ((reinterpret_cast<size_t>(p) | reinterpret_cast<size_t>(g) | reinterpret_cast<size_t>(m) | reinterpret_cast<size_t>(v)) & 15) == 0 &&
                   (reinterpret_cast<size_t>(sh) & 7) == 0;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const long i = base + ((long)it * 256 + threadIdx.x) * 4;
    if (i >= n) break;
    if (vec && i + 4 <= n) {
      float4 P = *reinterpret_cast<const float4*>(p + i), M = *reinterpret_cast<const float4*>(m + i), V = *reinterpret_cast<const float4*>(v + i);
      const float4 G = *reinterpret_cast<const float4*>(g + i);
      adam_one(P.x, G.x, M.x, V.x, k); adam_one(P.y, G.y, M.y, V.y, k); adam_one(P.z, G.z, M.z, V.z, k); adam_one(P.w, G.w, M.w, V.w, k);
      *reinterpret_cast<float4*>(p + i) = P; *reinterpret_cast<float4*>(m + i) = M; *reinterpret_cast<float4*>(v + i) = V;
      if (sh) {
        uint2 s;
        s.x = bf16_rne(P.x) | ((unsigned)bf16_rne(P.y) << 16);
        s.y = bf16_rne(P.z) | ((unsigned)bf16_rne(P.w) << 16);
        *reinterpret_cast<uint2*>(sh + i) = s;
      }
    } else {
      for (long j = i; j < n && j < i + 4; ++j) {
        float P = p[j], M = m[j], V = v[j];
        adam_one(P, g[j], M, V, k);
        p[j] = P; m[j] = M; v[j] = V;
        if (sh) sh[j] = bf16_rne(P);
      }
    }
  }
}

__global__ __launch_bounds__(256) void bf16_shadow_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n && ((reinterpret_cast<size_t>(src) & 15) | (reinterpret_cast<size_t>(dst) & 7)) == 0) {
    const float4 P = *reinterpret_cast<const float4*>(src + i);
    uint2 s;
    s.x = bf16_rne(P.x) | ((unsigned)bf16_rne(P.y) << 16);
    s.y = bf16_rne(P.z) | ((unsigned)bf16_rne(P.w) << 16);
    *reinterpret_cast<uint2*>(dst + i) = s;
  } else {
    for (long j = i; j < n && j < i + 4; ++j) dst[j] = bf16_rne(src[j]);
  }
}

// --------------------------------------------------------------------------------------------------- K12 scale * verts + trans
constexpr int AFF_SLICES = 16;
__global__ __launch_bounds__(256) void affine_fwd_kernel(const float* __restrict__ verts, const float* __restrict__ scale,
                                                        const float* __restrict__ trans, long per_sample, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float s = scale ? scale[b] : 1.f;
  float t[3] = {0.f, 0.f, 0.f};
  if (trans) { t[0] = trans[b * 3]; t[1] = trans[b * 3 + 1]; t[2] = trans[b * 3 + 2]; }
  const float* vb = verts + (size_t)b * per_sample;
  float* ob = out + (size_t)b * per_sample;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    // the reference's two roundings: fl(fl(s * v) + t)
    ob[i] = __fadd_rn(__fmul_rn(s, vb[i]), c == 0 ? t[0] : (c == 1 ? t[1] : t[2]));
  }
}

// gverts = g * s;  partial sums of g (-> gtrans) and g . v (-> gscale) per (sample, slice): [B][AFF_SLICES][4]
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ verts,
                                                        const float* __restrict__ scale, int N, float* __restrict__ gverts,
                                                        float* __restrict__ part) {
  const int b = blockIdx.y, sl = blockIdx.x;
  const float s = scale ? scale[b] : 1.f;
  const int per = (N + AFF_SLICES - 1) / AFF_SLICES, n0 = sl * per, n1 = min(N, n0 + per);
  const float* gb = g + (size_t)b * N * 3;
  const float* vb = verts + (size_t)b * N * 3;
  float* ob = gverts ? gverts + (size_t)b * N * 3 : nullptr;
  float ax = 0.f, ay = 0.f, az = 0.f, ad = 0.f;
  for (int n = n0 + threadIdx.x; n < n1; n += 256) {
    const float gx = gb[n * 3], gy = gb[n * 3 + 1], gz = gb[n * 3 + 2];
    const float vx = vb[n * 3], vy = vb[n * 3 + 1], vz = vb[n * 3 + 2];
    ax += gx; ay += gy; az += gz;
    ad += gx * vx + gy * vy + gz * vz;
    if (ob) { ob[n * 3] = gx * s; ob[n * 3 + 1] = gy * s; ob[n * 3 + 2] = gz * s; }
  }
  __shared__ float red[4][4];
  ax = obman_wave_sum(ax); ay = obman_wave_sum(ay); az = obman_wave_sum(az); ad = obman_wave_sum(ad);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = ax; red[w][1] = ay; red[w][2] = az; red[w][3] = ad; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    part[((size_t)b * AFF_SLICES + sl) * 4 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  }
}
__global__ void affine_bwd_finalize_kernel(const float* __restrict__ part, int B, float* __restrict__ gscale, float* __restrict__ gtrans) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (sample, component)
  if (i >= B * 4) return;
  const int b = i >> 2, c = i & 3;
  float a = 0.f;
#pragma unroll
  for (int s = 0; s < AFF_SLICES; ++s) a += part[((size_t)b * AFF_SLICES + s) * 4 + c];
  if (c < 3) { if (gtrans) gtrans[b * 3 + c] = a; }
  else if (gscale) gscale[b] = a;
}

// ------------------------------------------------------------------------------------------------------------ K13 MSE terms
constexpr int MSE_MAX = 8, MSE_SLICES = 32;
struct MseBatch {
  const float* pred[MSE_MAX];
  const float* target[MSE_MAX];
  float* grad[MSE_MAX];
  long n[MSE_MAX];
  int count;
};
__global__ __launch_bounds__(256) void mse_fwd_kernel(MseBatch b, float* __restrict__ part) {
  const int t = blockIdx.y, sl = blockIdx.x;
  const long n = b.n[t], per = (n + MSE_SLICES - 1) / MSE_SLICES, i0 = sl * per, i1 = min(n, i0 + per);
  const float* __restrict__ p = b.pred[t];
  const float* __restrict__ q = b.target[t];
  float a = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) {
    const float d = p[i] - (q ? q[i] : 0.f);
    a = __fmaf_rn(d, d, a);
  }
  a = obman_wave_sum(a);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) part[t * MSE_SLICES + sl] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void mse_finalize_kernel(MseBatch b, const float* __restrict__ part, float* __restrict__ out) {
  const int t = threadIdx.x;
  if (t >= b.count) return;
  float a = 0.f;
  for (int s = 0; s < MSE_SLICES; ++s) a += part[t * MSE_SLICES + s];
  out[t] = a / (float)b.n[t];
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(MseBatch b, const float* __restrict__ g_out) {
  const int t = blockIdx.y;
  const long n = b.n[t];
  float* __restrict__ gr = b.grad[t];
  if (!gr) return;
  const float* __restrict__ p = b.pred[t];
  const float* __restrict__ q = b.target[t];
  const float k = 2.f / (float)n, go = g_out[t];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    gr[i] = k * (p[i] - (q ? q[i] : 0.f)) * go;  // aten mse_loss_backward: (2 / N) * (input - target) * grad_output
}

// ------------------------------------------------------------------------------------------------------------- K14 GT stats
// one block per sample: centroid (mean over the points), centred cloud, max over the points of the centred point's norm
__global__ __launch_bounds__(256) void gt_stats_kernel(const float* __restrict__ gt, int N, float* __restrict__ centroid,
                                                      float* __restrict__ centred, float* __restrict__ maxnorm) {
  const int b = blockIdx.x;
  const float* gb = gt + (size_t)b * N * 3;
  __shared__ float red[4][3];
  __shared__ float cen[3];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) { ax += gb[n * 3]; ay += gb[n * 3 + 1]; az += gb[n * 3 + 2]; }
  ax = obman_wave_sum(ax); ay = obman_wave_sum(ay); az = obman_wave_sum(az);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = ax; red[threadIdx.x >> 6][1] = ay; red[threadIdx.x >> 6][2] = az; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    const float m = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) / (float)N;
    cen[c] = m;
    centroid[b * 3 + c] = m;
  }
  __syncthreads();
  const float cx = cen[0], cy = cen[1], cz = cen[2];
  float mx = 0.f;
  for (int n = threadIdx.x; n < N; n += 256) {
    const float dx = gb[n * 3] - cx, dy = gb[n * 3 + 1] - cy, dz = gb[n * 3 + 2] - cz;
    float* o = centred + ((size_t)b * N + n) * 3;
    o[0] = dx; o[1] = dy; o[2] = dz;
    mx = fmaxf(mx, sqrtf(dx * dx + dy * dy + dz * dz));
  }
  mx = obman_wave_max(mx);
  __shared__ float rmx[4];
  if ((threadIdx.x & 63) == 0) rmx[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) maxnorm[b] = fmaxf(fmaxf(rmx[0], rmx[1]), fmaxf(rmx[2], rmx[3]));
}

}  // namespace

extern "C" {

int obman_adam_step(const obman_adam_tensor* tensors, int count, double lr, double beta1, double beta2, double eps,
                    double weight_decay, obman_stream_t stream) {
  if (count < 0 || (count > 0 && !tensors)) return -1;
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < count; base += ADAM_MAX) {
    AdamBatch b;
    b.count = count - base < ADAM_MAX ? count - base : ADAM_MAX;
    int blocks = 0;
    for (int i = 0; i < b.count; ++i) {
      const obman_adam_tensor& t = tensors[base + i];
      if (!t.p || !t.g || !t.m || !t.v || !t.step || t.n < 0) return -2;
      b.p[i] = t.p; b.g[i] = t.g; b.m[i] = t.m; b.v[i] = t.v; b.shadow[i] = t.shadow_bf16; b.step[i] = t.step; b.n[i] = t.n;
      b.first_block[i] = blocks;
      blocks += obman_cdiv(t.n, ADAM_CHUNK);
    }
    for (int i = b.count; i <= ADAM_MAX; ++i) b.first_block[i] = blocks;
    for (int i = b.count; i < ADAM_MAX; ++i) { b.p[i] = nullptr; b.g[i] = nullptr; b.m[i] = nullptr; b.v[i] = nullptr; b.shadow[i] = nullptr; b.step[i] = nullptr; b.n[i] = 0; }
    adam_tick_kernel<<<1, ADAM_MAX, 0, st>>>(b);
    OBMAN_LAUNCH_CHECK();
    if (blocks > 0) {
      adam_kernel<<<blocks, 256, 0, st>>>(b, lr, beta1, beta2, (float)eps, (float)weight_decay);
      OBMAN_LAUNCH_CHECK();
    }
  }
  return 0;
}

int obman_bf16_shadow(const float* src, uint16_t* dst, long n, obman_stream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst))) return -1;
  if (n == 0) return 0;
  bf16_shadow_kernel<<<obman_cdiv(n, 1024), 256, 0, (hipStream_t)stream>>>(src, dst, n);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_affine_points_fwd(const float* verts, const float* scale, const float* trans, int B, int N, float* out,
                            obman_stream_t stream) {
  if (B <= 0 || N <= 0 || !verts || !out) return -1;
  const long per = (long)N * 3;
  int bx = obman_cdiv(per, 256 * 8);
  if (bx > 64) bx = 64;
  affine_fwd_kernel<<<dim3(bx, B), 256, 0, (hipStream_t)stream>>>(verts, scale, trans, per, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

long obman_affine_points_ws_floats(int B) { return (long)B * AFF_SLICES * 4; }

int obman_affine_points_bwd(const float* g, const float* verts, const float* scale, int B, int N, float* gverts, float* gscale,
                            float* gtrans, float* ws, obman_stream_t stream) {
  if (B <= 0 || N <= 0 || !g || !verts || !ws) return -1;
  hipStream_t st = (hipStream_t)stream;
  affine_bwd_kernel<<<dim3(AFF_SLICES, B), 256, 0, st>>>(g, verts, scale, N, gverts, ws);
  OBMAN_LAUNCH_CHECK();
  if (gscale || gtrans) {
    affine_bwd_finalize_kernel<<<obman_cdiv(B * 4, 256), 256, 0, st>>>(ws, B, gscale, gtrans);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}

long obman_mse_terms_ws_floats(void) { return (long)MSE_MAX * MSE_SLICES; }

static int mse_batch(const obman_mse_term* terms, int count, MseBatch& b) {
  if (count <= 0 || count > MSE_MAX || !terms) return -1;
  b.count = count;
  for (int i = 0; i < MSE_MAX; ++i) {
    if (i < count) {
      if (!terms[i].pred || terms[i].n <= 0) return -2;
      b.pred[i] = terms[i].pred; b.target[i] = terms[i].target; b.grad[i] = terms[i].grad; b.n[i] = terms[i].n;
    } else {
      b.pred[i] = nullptr; b.target[i] = nullptr; b.grad[i] = nullptr; b.n[i] = 0;
    }
  }
  return 0;
}

int obman_mse_terms_fwd(const obman_mse_term* terms, int count, float* ws, float* out, obman_stream_t stream) {
  MseBatch b;
  const int e = mse_batch(terms, count, b);
  if (e) return e;
  if (!ws || !out) return -1;
  hipStream_t st = (hipStream_t)stream;
  mse_fwd_kernel<<<dim3(MSE_SLICES, count), 256, 0, st>>>(b, ws);
  OBMAN_LAUNCH_CHECK();
  mse_finalize_kernel<<<1, 64, 0, st>>>(b, ws, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_mse_terms_bwd(const obman_mse_term* terms, int count, const float* g_out, obman_stream_t stream) {
  MseBatch b;
  const int e = mse_batch(terms, count, b);
  if (e) return e;
  if (!g_out) return -1;
  long nmax = 0;
  for (int i = 0; i < count; ++i) nmax = terms[i].n > nmax ? terms[i].n : nmax;
  int bx = obman_cdiv(nmax, 256 * 4);
  if (bx > 256) bx = 256;
  mse_bwd_kernel<<<dim3(bx, count), 256, 0, (hipStream_t)stream>>>(b, g_out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_gt_object_stats(const float* gt, int B, int N, float* centroid, float* centred, float* maxnorm, obman_stream_t stream) {
  if (B <= 0 || N <= 0 || !gt || !centroid || !centred || !maxnorm) return -1;
  gt_stats_kernel<<<B, 256, 0, (hipStream_t)stream>>>(gt, N, centroid, centred, maxnorm);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
