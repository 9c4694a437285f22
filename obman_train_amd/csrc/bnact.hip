// Fused BatchNorm2d (+ residual add) (+ ReLU), NHWC fp32, forward and backward, for the ResNet encoder
// (resnet.py:25-52 BasicBlock: conv -> bn -> relu -> conv -> bn -> (+skip) -> relu).
//
// Eager PyTorch runs this as separate memory-bound passes (MIOpen BN statistics + normalise, clamp, add; in the
// backward threshold_backward + two BN kernels): ~3.4 ms of a 13 ms step (profiles/r01b).  Here:
//   fwd : stats pass (read x)            + apply pass (read x [, skip], write y)
//   bwd : sums pass (read dy, x [, y])   + apply pass (read dy|dz, x, write dx)   [dz materialised only when a skip
//         connection needs it as its own gradient]
// The ReLU mask of the non-residual case is recomputed from x (s*x+t > 0), so y is never read back.
// Layout: rows = B*H*W, C channels contiguous (torch channels_last).  Block = 16 row-lanes x 16 float4 columns
// (64 channels); a wave reads 4 rows x 256 contiguous bytes.  Per-block partial moments are fp64; one wave per
// channel merges them in fixed order (deterministic).  Bound: HBM bandwidth.
#include <cstdlib>

#include "common.h"
#include "../../include/obman_hip.h"

namespace bnact {

constexpr int CT = 64;  // channels per block column

// Activation element type: fp32, or bf16 (raw bits) for the bf16-autocast encoder of BASELINE configs[2].  A lane always moves
// four consecutive channels (16 bytes of fp32, 8 bytes of bf16); the arithmetic and every statistic are fp32 / fp64 in both
// flavours, the bf16 flavour rounds once (to nearest even) when it stores.
typedef unsigned short bfraw;
template <class T> struct Act;
template <> struct Act<float> {
  static __device__ __forceinline__ float4 ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void st(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
  static __device__ __forceinline__ float4 round(const float4 v) { return v; }
};
__device__ __forceinline__ unsigned bf_bits(float f) {  // round to nearest even; NaN stays NaN
  const unsigned u = __float_as_uint(f);
  return (u & 0x7fffffffu) > 0x7f800000u ? ((u >> 16) | 0x40u) : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <> struct Act<bfraw> {
  static __device__ __forceinline__ float4 ld(const bfraw* p) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
  }
  static __device__ __forceinline__ void st(bfraw* p, const float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(bf_bits(v.x) | (bf_bits(v.y) << 16), bf_bits(v.z) | (bf_bits(v.w) << 16));
  }
  static __device__ __forceinline__ float4 round(const float4 v) {
    return make_float4(__uint_as_float(bf_bits(v.x) << 16), __uint_as_float(bf_bits(v.y) << 16), __uint_as_float(bf_bits(v.z) << 16),
                       __uint_as_float(bf_bits(v.w) << 16));
  }
};

struct Geo { long R; int C, rows_per_blk, nblk; };

__device__ __forceinline__ double wsum64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// accumulate two per-channel quantities over a chunk of rows; partial[blk][C][2]
template <int MODE, class T>  // 0: (x, x^2)   1: (dz, dz*xhat) mask from s*x+t   2: same, mask from y, optional dz store   3: no relu
                              // DY2 (mode 2 only, may be null): a second incoming gradient, added to DY before the mask
__global__ __launch_bounds__(256) void sums_kernel(const T* __restrict__ X, const T* __restrict__ DY, const T* __restrict__ DY2, const T* __restrict__ Y,
                                                   const float* __restrict__ sc, const float* __restrict__ sh,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd, Geo g,
                                                   T* __restrict__ DZ, double* __restrict__ partial) {
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * CT + cl * 4;
  const long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = min(g.R, r0 + g.rows_per_blk);
  float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
  float4 vs = make_float4(0, 0, 0, 0), vt = vs, vm = vs, vr = vs;
  if (MODE != 0) {
    vm = *reinterpret_cast<const float4*>(mean + c);
    vr = *reinterpret_cast<const float4*>(rstd + c);
    if (MODE == 1) { vs = *reinterpret_cast<const float4*>(sc + c); vt = *reinterpret_cast<const float4*>(sh + c); }
  }
  // one row of the thread's stripe; `d` = dy (modes 1-3), `y` and `d2` only in mode 2
  auto row = [&](size_t o, const float4 x, float4 d, const float4 d2, const float4 y) {
    if (MODE == 2) { d.x += d2.x; d.y += d2.y; d.z += d2.z; d.w += d2.w; }  // (+ 0 without a second gradient)
    if (MODE == 0) {
      a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
      b.x = __fmaf_rn(x.x, x.x, b.x); b.y = __fmaf_rn(x.y, x.y, b.y); b.z = __fmaf_rn(x.z, x.z, b.z); b.w = __fmaf_rn(x.w, x.w, b.w);
    } else {
      if (MODE == 1) {
        d.x = __fmaf_rn(vs.x, x.x, vt.x) > 0.f ? d.x : 0.f; d.y = __fmaf_rn(vs.y, x.y, vt.y) > 0.f ? d.y : 0.f;
        d.z = __fmaf_rn(vs.z, x.z, vt.z) > 0.f ? d.z : 0.f; d.w = __fmaf_rn(vs.w, x.w, vt.w) > 0.f ? d.w : 0.f;
      } else if (MODE == 2) {
        d.x = y.x > 0.f ? d.x : 0.f; d.y = y.y > 0.f ? d.y : 0.f; d.z = y.z > 0.f ? d.z : 0.f; d.w = y.w > 0.f ? d.w : 0.f;
        if (DZ) Act<T>::st(DZ + o, d);
      }
      a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
      b.x = __fmaf_rn(d.x, (x.x - vm.x) * vr.x, b.x); b.y = __fmaf_rn(d.y, (x.y - vm.y) * vr.y, b.y);
      b.z = __fmaf_rn(d.z, (x.z - vm.z) * vr.z, b.z); b.w = __fmaf_rn(d.w, (x.w - vm.w) * vr.w, b.w);
    }
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  long r = r0 + rl;
  // four rows per step, every load issued before the first use: a single 16-byte load in flight per thread left the read-only
  // statistics pass at 2.8 TB/s (the 268 MB stem activation in 97 us); rows are still accumulated in their original order
  for (; r + 48 < r1; r += 64) {
    size_t o[4];
    float4 x[4], d[4], d2[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (size_t)(r + 16 * j) * g.C + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = Act<T>::ld(X + o[j]);
      d[j] = MODE != 0 ? Act<T>::ld(DY + o[j]) : z4;
      d2[j] = (MODE == 2 && DY2) ? Act<T>::ld(DY2 + o[j]) : z4;
      y[j] = MODE == 2 ? Act<T>::ld(Y + o[j]) : z4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) row(o[j], x[j], d[j], d2[j], y[j]);
  }
  for (; r < r1; r += 16) {
    const size_t o = (size_t)r * g.C + c;
    row(o, Act<T>::ld(X + o), MODE != 0 ? Act<T>::ld(DY + o) : z4, (MODE == 2 && DY2) ? Act<T>::ld(DY2 + o) : z4, MODE == 2 ? Act<T>::ld(Y + o) : z4);
  }
  __shared__ float red[16][CT][2];
  red[rl][cl * 4 + 0][0] = a.x; red[rl][cl * 4 + 1][0] = a.y; red[rl][cl * 4 + 2][0] = a.z; red[rl][cl * 4 + 3][0] = a.w;
  red[rl][cl * 4 + 0][1] = b.x; red[rl][cl * 4 + 1][1] = b.y; red[rl][cl * 4 + 2][1] = b.z; red[rl][cl * 4 + 3][1] = b.w;
  __syncthreads();
  if (tid < CT * 2) {
    const int ch = tid >> 1, q = tid & 1;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (double)red[k][ch][q];
    partial[((size_t)(blockIdx.y * CT + ch) * g.nblk + blockIdx.x) * 2 + q] = s;  // channel-major: the finalize wave reads contiguously
  }
}

// forward statistics -> mean, rstd, affine (s, t), running-stat update.  One wave per channel.
__global__ __launch_bounds__(256) void fwd_finalize_kernel(const double* __restrict__ partial, int nblk, long R, int C, int training,
                                                           float eps, float momentum, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ sc, float* __restrict__ sh) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float m, v;
  if (training) {
    double s1 = 0, s2 = 0;
    for (int b = lane; b < nblk; b += 64) { s1 += partial[((size_t)c * nblk + b) * 2]; s2 += partial[((size_t)c * nblk + b) * 2 + 1]; }
    s1 = wsum64(s1); s2 = wsum64(s2);
    const double mu = s1 / R;
    double var = s2 / R - mu * mu;
    if (var < 0) var = 0;
    m = (float)mu; v = (float)var;
    if (rmean && lane == 0) {
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * ((double)R / (R > 1 ? R - 1 : 1)));
    }
  } else {
    m = rmean[c]; v = rvar[c];
  }
  if (lane != 0) return;
  const float rs = 1.f / sqrtf(v + eps);
  mean[c] = m; rstd[c] = rs;
  sc[c] = gamma[c] * rs;
  sh[c] = beta[c] - m * gamma[c] * rs;
}

// y = act(s*x + t [+ skip])
template <class T>
__global__ __launch_bounds__(256) void fwd_apply_kernel(const T* __restrict__ X, const T* __restrict__ S, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, Geo g, int relu, T* __restrict__ Y) {
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * CT + cl * 4;
  const long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = min(g.R, r0 + g.rows_per_blk);
  const float4 vs = *reinterpret_cast<const float4*>(sc + c), vt = *reinterpret_cast<const float4*>(sh + c);
  auto row = [&](size_t o, const float4 x, const float4 k) {
    float4 y = make_float4(__fmaf_rn(vs.x, x.x, vt.x), __fmaf_rn(vs.y, x.y, vt.y), __fmaf_rn(vs.z, x.z, vt.z), __fmaf_rn(vs.w, x.w, vt.w));
    if (S) { y.x += k.x; y.y += k.y; y.z += k.z; y.w += k.w; }
    if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
    Act<T>::st(Y + o, y);
  };
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  long r = r0 + rl;
  for (; r + 48 < r1; r += 64) {  // four rows per step, loads first (see sums_kernel)
    size_t o[4];
    float4 x[4], k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (size_t)(r + 16 * j) * g.C + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = Act<T>::ld(X + o[j]); k[j] = S ? Act<T>::ld(S + o[j]) : z4; }
#pragma unroll
    for (int j = 0; j < 4; ++j) row(o[j], x[j], k[j]);
  }
  for (; r < r1; r += 16) {
    const size_t o = (size_t)r * g.C + c;
    row(o, Act<T>::ld(X + o), S ? Act<T>::ld(S + o) : z4);
  }
}

// backward sums -> d_gamma, d_beta and the coefficients of dx = k1 * (dz - k2 - xhat * k3)
__global__ __launch_bounds__(256) void bwd_finalize_kernel(const double* __restrict__ partial, int nblk, long R, int C, int training,
                                                           const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ k) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int b = lane; b < nblk; b += 64) { s1 += partial[((size_t)c * nblk + b) * 2]; s2 += partial[((size_t)c * nblk + b) * 2 + 1]; }
  s1 = wsum64(s1); s2 = wsum64(s2);
  if (lane != 0) return;
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
  k[c] = gamma[c] * rstd[c];
  k[C + c] = training ? (float)(s1 / R) : 0.f;
  k[2 * C + c] = training ? (float)(s2 / R) : 0.f;
}

// dx = k1 * (dz - k2 - xhat*k3); dz = DZ (materialised) or dy masked by s*x+t > 0 (relu) or dy (no relu)
template <int MODE, class T>  // 1: mask from x   2: dz given   3: no relu
__global__ __launch_bounds__(256) void bwd_apply_kernel(const T* __restrict__ X, const T* __restrict__ D, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ k, Geo g,
                                                        T* __restrict__ DX) {
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * CT + cl * 4;
  const long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = min(g.R, r0 + g.rows_per_blk);
  const float4 vm = *reinterpret_cast<const float4*>(mean + c), vr = *reinterpret_cast<const float4*>(rstd + c);
  const float4 k1 = *reinterpret_cast<const float4*>(k + c), k2 = *reinterpret_cast<const float4*>(k + g.C + c),
               k3 = *reinterpret_cast<const float4*>(k + 2 * g.C + c);
  float4 vs = make_float4(0, 0, 0, 0), vt = vs;
  if (MODE == 1) { vs = *reinterpret_cast<const float4*>(sc + c); vt = *reinterpret_cast<const float4*>(sh + c); }
  auto row = [&](size_t o, const float4 x, float4 d) {
    if (MODE == 1) {
      d.x = __fmaf_rn(vs.x, x.x, vt.x) > 0.f ? d.x : 0.f; d.y = __fmaf_rn(vs.y, x.y, vt.y) > 0.f ? d.y : 0.f;
      d.z = __fmaf_rn(vs.z, x.z, vt.z) > 0.f ? d.z : 0.f; d.w = __fmaf_rn(vs.w, x.w, vt.w) > 0.f ? d.w : 0.f;
    }
    float4 o4;
    o4.x = k1.x * (d.x - k2.x - (x.x - vm.x) * vr.x * k3.x);
    o4.y = k1.y * (d.y - k2.y - (x.y - vm.y) * vr.y * k3.y);
    o4.z = k1.z * (d.z - k2.z - (x.z - vm.z) * vr.z * k3.z);
    o4.w = k1.w * (d.w - k2.w - (x.w - vm.w) * vr.w * k3.w);
    Act<T>::st(DX + o, o4);
  };
  long r = r0 + rl;
  for (; r + 48 < r1; r += 64) {  // four rows per step, loads first (see sums_kernel)
    size_t o[4];
    float4 x[4], d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (size_t)(r + 16 * j) * g.C + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = Act<T>::ld(X + o[j]); d[j] = Act<T>::ld(D + o[j]); }
#pragma unroll
    for (int j = 0; j < 4; ++j) row(o[j], x[j], d[j]);
  }
  for (; r < r1; r += 16) {
    const size_t o = (size_t)r * g.C + c;
    row(o, Act<T>::ld(X + o), Act<T>::ld(D + o));
  }
}

// ---------------------------------------------------------------------------------------------------- stem: + MaxPool(3,2,1)
// y_pool[b,ho,wo,c] = max over the 3x3 / stride-2 / pad-1 window of relu(s*x + t).  The full-resolution activation is
// never written (268 MB at bs 64): forward reads x once and writes the pooled map; backward recomputes y from x and
// routes d(pool) to every pixel with y == y_pool of a covering window and y > 0 (identical to torch's arg-max routing
// except on exact ties of positive values; ties at 0 carry no gradient through the ReLU either way).
struct PoolGeo { int H, W, Ho, Wo; };

// (sample, row, column) of a flat pixel index, advanced incrementally: the 64-bit div/mod per 16-byte access this
// replaces cost more than the access (pool kernels ran at 2-3 TB/s).
struct Pix { int b, h, w; };
__device__ __forceinline__ Pix pix_of(long r, int H, int W) {
  const int rr = (int)r, t = rr / W;
  return Pix{t / H, t % H, rr - t * W};
}
__device__ __forceinline__ void pix_advance(Pix& p, int step, int H, int W) {
  p.w += step;
  while (p.w >= W) {
    p.w -= W;
    if (++p.h == H) { p.h = 0; ++p.b; }
  }
}

template <class T>  // Ia (bf16 flavour): window tap index (0..8, row-major; 255 = none) of the FIRST maximum, for the backward's routing
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* __restrict__ X, const float* __restrict__ sc, const float* __restrict__ sh,
                                                       Geo g, PoolGeo pg, T* __restrict__ Yp, unsigned char* __restrict__ Ia) {
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * CT + cl * 4;
  const long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = min(g.R, r0 + g.rows_per_blk);  // rows = pooled pixels
  const float4 vs = *reinterpret_cast<const float4*>(sc + c), vt = *reinterpret_cast<const float4*>(sh + c);
  Pix px = pix_of(r0 + rl, pg.Ho, pg.Wo);
  for (long r = r0 + rl; r < r1; r += 16, pix_advance(px, 16, pg.Ho, pg.Wo)) {
    const int wo = px.w, ho = px.h;
    const long b = px.b;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // relu >= 0 and the window centre is always in bounds
    // branch-free: out-of-image taps are clamped onto a pixel of the same window (a duplicate cannot change a max), so all
    // nine 16-byte loads are issued back to back
    float4 xs[9];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int h = min(max(2 * ho + dy, 0), pg.H - 1);
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int w = min(max(2 * wo + dx, 0), pg.W - 1);
        xs[(dy + 1) * 3 + dx + 1] = Act<T>::ld(X + ((size_t)(b * pg.H + h) * pg.W + w) * g.C + c);
      }
    }
    if (Ia) {  // strict > over the in-image taps in scan order: torch's max_pool2d arg-max (first maximum); relu floor 0 = no tap
      unsigned i0 = 255, i1 = 255, i2 = 255, i3 = 255;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = 2 * ho + t / 3 - 1, ww = 2 * wo + t % 3 - 1;
        const bool in = hh >= 0 && hh < pg.H && ww >= 0 && ww < pg.W;
        const float v0 = __fmaf_rn(vs.x, xs[t].x, vt.x), v1 = __fmaf_rn(vs.y, xs[t].y, vt.y), v2 = __fmaf_rn(vs.z, xs[t].z, vt.z),
                    v3 = __fmaf_rn(vs.w, xs[t].w, vt.w);
        if (in && v0 > m.x) { m.x = v0; i0 = t; }
        if (in && v1 > m.y) { m.y = v1; i1 = t; }
        if (in && v2 > m.z) { m.z = v2; i2 = t; }
        if (in && v3 > m.w) { m.w = v3; i3 = t; }
      }
      *reinterpret_cast<unsigned*>(Ia + (size_t)r * g.C + c) = i0 | (i1 << 8) | (i2 << 16) | (i3 << 24);
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        m.x = fmaxf(m.x, __fmaf_rn(vs.x, xs[t].x, vt.x)); m.y = fmaxf(m.y, __fmaf_rn(vs.y, xs[t].y, vt.y));
        m.z = fmaxf(m.z, __fmaf_rn(vs.z, xs[t].z, vt.z)); m.w = fmaxf(m.w, __fmaf_rn(vs.w, xs[t].w, vt.w));
      }
    }
    Act<T>::st(Yp + (size_t)r * g.C + c, m);
  }
}

// d(loss)/d(relu output) of pixel (b,h,w): sum of the pooled gradients of the covering windows whose max this pixel is
// Backward over 2 x 2 pixel quads.  The quad (2i..2i+1, 2j..2j+1) is covered by exactly the pooled windows
// (i..i+1, j..j+1): pixel (2i,2j) only by (i,j), the two edge pixels by two windows, (2i+1,2j+1) by all four.  One thread
// owns a quad x 4 channels: 4 x-loads + 4 (y_pool, d_pool) tap pairs, all issued together (12 independent 16-byte loads),
// i.e. one tap pair per pixel instead of 2.25 in a per-pixel formulation.
__device__ __forceinline__ float4 relu_bn(const float4 x, const float4 s, const float4 t) {
  return make_float4(fmaxf(__fmaf_rn(s.x, x.x, t.x), 0.f), fmaxf(__fmaf_rn(s.y, x.y, t.y), 0.f), fmaxf(__fmaf_rn(s.z, x.z, t.z), 0.f),
                     fmaxf(__fmaf_rn(s.w, x.w, t.w), 0.f));
}
__device__ __forceinline__ void route(float4& dz, const float4 y, const float4 yp, const float4 dp, bool ok) {
  dz.x += (ok && y.x > 0.f && y.x == yp.x) ? dp.x : 0.f; dz.y += (ok && y.y > 0.f && y.y == yp.y) ? dp.y : 0.f;
  dz.z += (ok && y.z > 0.f && y.z == yp.z) ? dp.z : 0.f; dz.w += (ok && y.w > 0.f && y.w == yp.w) ? dp.w : 0.f;
}

// index flavour of route(): the window's recorded arg-max tap (byte per channel) against this pixel's tap index in that window
__device__ __forceinline__ void route_idx(float4& dz, const float4 y, unsigned ia, const float4 dp, bool ok, unsigned tap) {
  dz.x += (ok && y.x > 0.f && (ia & 255u) == tap) ? dp.x : 0.f; dz.y += (ok && y.y > 0.f && ((ia >> 8) & 255u) == tap) ? dp.y : 0.f;
  dz.z += (ok && y.z > 0.f && ((ia >> 16) & 255u) == tap) ? dp.z : 0.f; dz.w += (ok && y.w > 0.f && (ia >> 24) == tap) ? dp.w : 0.f;
}

template <bool APPLY, class T>  // false: per-block sums (dz, dz*xhat); true: dx = k1*(dz - k2 - xhat*k3).  g describes the QUAD rows.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ X, const float* __restrict__ Yp, const unsigned char* __restrict__ Ia,
                                                       const T* __restrict__ Dp,
                                                       const float* __restrict__ sc, const float* __restrict__ sh,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ k, Geo g, PoolGeo pg, double* __restrict__ partial,
                                                       T* __restrict__ DX) {
  const int tid = threadIdx.x, cl = tid & 15, rl = tid >> 4;
  const int c = blockIdx.y * CT + cl * 4;
  const int Hq = (pg.H + 1) / 2, Wq = (pg.W + 1) / 2;
  const long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = min(g.R, r0 + g.rows_per_blk);  // rows = quads
  const float4 vs = *reinterpret_cast<const float4*>(sc + c), vt = *reinterpret_cast<const float4*>(sh + c);
  const float4 vm = *reinterpret_cast<const float4*>(mean + c), vr = *reinterpret_cast<const float4*>(rstd + c);
  float4 k1 = make_float4(0, 0, 0, 0), k2 = k1, k3 = k1;
  if (APPLY) {
    k1 = *reinterpret_cast<const float4*>(k + c); k2 = *reinterpret_cast<const float4*>(k + g.C + c);
    k3 = *reinterpret_cast<const float4*>(k + 2 * g.C + c);
  }
  float4 a = make_float4(0, 0, 0, 0), bb = a;
  Pix q = pix_of(r0 + rl, Hq, Wq);
  for (long r = r0 + rl; r < r1; r += 16, pix_advance(q, 16, Hq, Wq)) {
    const int h0 = 2 * q.h, w0 = 2 * q.w;
    const bool ph = h0 + 1 < pg.H, pw = w0 + 1 < pg.W;          // does the quad's second row / column exist
    const bool wh = q.h + 1 < pg.Ho, ww = q.w + 1 < pg.Wo;      // does the second window row / column exist
    const int h1 = ph ? h0 + 1 : h0, w1 = pw ? w0 + 1 : w0, i1 = wh ? q.h + 1 : q.h, j1 = ww ? q.w + 1 : q.w;
    const size_t xb = (size_t)q.b * pg.H, pb = (size_t)q.b * pg.Ho;
    const size_t ox[4] = {((xb + h0) * pg.W + w0) * g.C + c, ((xb + h0) * pg.W + w1) * g.C + c, ((xb + h1) * pg.W + w0) * g.C + c,
                          ((xb + h1) * pg.W + w1) * g.C + c};
    const size_t op[4] = {((pb + q.h) * pg.Wo + q.w) * g.C + c, ((pb + q.h) * pg.Wo + j1) * g.C + c, ((pb + i1) * pg.Wo + q.w) * g.C + c,
                          ((pb + i1) * pg.Wo + j1) * g.C + c};
    float4 x[4], yp[4], dp[4];
    unsigned ia[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = Act<T>::ld(X + ox[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (Ia) { ia[t] = *reinterpret_cast<const unsigned*>(Ia + op[t]); yp[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
      else { yp[t] = *reinterpret_cast<const float4*>(Yp + op[t]); ia[t] = 0; }
      dp[t] = Act<T>::ld(Dp + op[t]);
    }
    const bool pix_ok[4] = {true, pw, ph, ph && pw};
    const bool win_ok[4] = {true, ww, wh, wh && ww};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (!pix_ok[t]) continue;  // uniform per quad position except at odd image borders
      const float4 y = relu_bn(x[t], vs, vt);
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (Ia) {  // bf16 flavour: bf16 inputs tie often and equality would route to every tie; tap of pixel (a, b) of the quad in
                 // window (i + wi, j + wj) is (a - 2 wi + 1) * 3 + (b - 2 wj + 1)
        const unsigned a = t >> 1, b = t & 1;
        route_idx(d, y, ia[0], dp[0], true, (a + 1) * 3 + b + 1);
        if (t & 1) route_idx(d, y, ia[1], dp[1], win_ok[1], (a + 1) * 3);
        if (t & 2) route_idx(d, y, ia[2], dp[2], win_ok[2], b + 1);
        if (t == 3) route_idx(d, y, ia[3], dp[3], win_ok[3], 0);
      } else {
        route(d, y, yp[0], dp[0], true);                       // window (i, j) covers every pixel of the quad
        if (t & 1) route(d, y, yp[1], dp[1], win_ok[1]);       // (i, j+1) covers the odd column
        if (t & 2) route(d, y, yp[2], dp[2], win_ok[2]);       // (i+1, j) covers the odd row
        if (t == 3) route(d, y, yp[3], dp[3], win_ok[3]);
      }
      const float4 xh = make_float4((x[t].x - vm.x) * vr.x, (x[t].y - vm.y) * vr.y, (x[t].z - vm.z) * vr.z, (x[t].w - vm.w) * vr.w);
      if (APPLY) {
        Act<T>::st(DX + ox[t], make_float4(k1.x * (d.x - k2.x - xh.x * k3.x), k1.y * (d.y - k2.y - xh.y * k3.y),
                                           k1.z * (d.z - k2.z - xh.z * k3.z), k1.w * (d.w - k2.w - xh.w * k3.w)));
      } else {
        a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
        bb.x = __fmaf_rn(d.x, xh.x, bb.x); bb.y = __fmaf_rn(d.y, xh.y, bb.y); bb.z = __fmaf_rn(d.z, xh.z, bb.z); bb.w = __fmaf_rn(d.w, xh.w, bb.w);
      }
    }
  }
  if (APPLY) return;
  __shared__ float red[16][CT][2];
  red[rl][cl * 4 + 0][0] = a.x; red[rl][cl * 4 + 1][0] = a.y; red[rl][cl * 4 + 2][0] = a.z; red[rl][cl * 4 + 3][0] = a.w;
  red[rl][cl * 4 + 0][1] = bb.x; red[rl][cl * 4 + 1][1] = bb.y; red[rl][cl * 4 + 2][1] = bb.z; red[rl][cl * 4 + 3][1] = bb.w;
  __syncthreads();
  if (tid < CT * 2) {
    const int ch = tid >> 1, q2 = tid & 1;
    double s2 = 0;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) s2 += (double)red[kk][ch][q2];
    partial[((size_t)(blockIdx.y * CT + ch) * g.nblk + blockIdx.x) * 2 + q2] = s2;
  }
}

Geo geo(long R, int C) {
  Geo g; g.R = R; g.C = C;
  const long col_blocks = C / CT;
  static const long target = [] { const char* e = getenv("OBMAN_BN_BLOCKS"); return e ? atol(e) : 1024L; }();  // tuning knob
  long want = target / col_blocks;  // ~1024 blocks in total (A/B on one box: 1024 >= 2048 >= 4096 within 1%)
  if (want < 1) want = 1;
  long rows = (R + want - 1) / want;
  rows = (rows + 15) / 16 * 16;
  if (rows < 64) rows = 64;
  g.rows_per_blk = (int)rows;
  g.nblk = (int)((R + rows - 1) / rows);
  return g;
}

}  // namespace bnact

extern "C" {

long obman_bnact_ws_floats(long R, int C) {
  if (R <= 0 || C <= 0 || C % bnact::CT) return -1;
  const bnact::Geo g = bnact::geo(R, C);
  return (long)g.nblk * C * 2 * 2 + 3L * C + 16;  // fp64 partials + backward coefficients
}

}  // extern "C"

namespace {
using bnact::Act;

/* stats: [mean C | rstd C | scale C | shift C] written by fwd and read by bwd */
template <class T>
int bnact_fwd(const T* x, const T* skip, const float* gamma, const float* beta, float* rmean, float* rvar, long R, int C,
              int training, float eps, float momentum, int relu, T* y, float* stats, float* ws, obman_stream_t stream) {
  if (!x || !y || !stats || !ws || !gamma || !beta || R <= 0 || C <= 0 || C % bnact::CT) return -1;
  if (!training && (!rmean || !rvar)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const bnact::Geo g = bnact::geo(R, C);
  dim3 grid(g.nblk, C / bnact::CT);
  double* partial = reinterpret_cast<double*>(ws);
  float *mean = stats, *rstd = stats + C, *sc = stats + 2 * C, *sh = stats + 3 * C;
  if (training) {
    bnact::sums_kernel<0, T><<<grid, 256, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, g, nullptr, partial);
    OBMAN_LAUNCH_CHECK();
  }
  bnact::fwd_finalize_kernel<<<obman_cdiv(C, 4), 256, 0, st>>>(partial, g.nblk, R, C, training, eps, momentum, gamma, beta, rmean, rvar, mean,
                                                                rstd, sc, sh);
  OBMAN_LAUNCH_CHECK();
  bnact::fwd_apply_kernel<T><<<grid, 256, 0, st>>>(x, skip, sc, sh, g, relu, y);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

/* dy -> dx, dgamma, dbeta [, dskip].  y is needed only with a skip connection (mask of the post-add ReLU).  dskip (the
 * gradient flowing into the skip branch = masked dy) is written when non-NULL; it doubles as the dz scratch, so it is
 * required whenever relu && has_skip.  dy2 (relu && has_skip only, may be NULL): the output had two consumers and their gradients
 * arrive separately; the sums pass adds them while it reads (instead of a separate add kernel over the activation). */
template <class T>
int bnact_bwd(const T* x, const T* y, const T* dy, const T* dy2, const float* gamma, const float* stats, long R, int C, int training,
              int relu, int has_skip, T* dx, float* dgamma, float* dbeta, T* dskip, float* ws, obman_stream_t stream) {
  if (!x || !dy || !stats || !ws || !dx || !dgamma || !dbeta || R <= 0 || C % bnact::CT) return -1;
  if (relu && has_skip && (!y || !dskip)) return -2;
  if (dy2 && !(relu && has_skip)) return -3;
  hipStream_t st = (hipStream_t)stream;
  const bnact::Geo g = bnact::geo(R, C);
  dim3 grid(g.nblk, C / bnact::CT);
  double* partial = reinterpret_cast<double*>(ws);
  float* k = ws + (size_t)g.nblk * C * 2 * 2;  // 3*C coefficients live behind the partials
  const float *mean = stats, *rstd = stats + C, *sc = stats + 2 * C, *sh = stats + 3 * C;
  const int mode = !relu ? 3 : (has_skip ? 2 : 1);
  if (mode == 1) bnact::sums_kernel<1, T><<<grid, 256, 0, st>>>(x, dy, nullptr, nullptr, sc, sh, mean, rstd, g, nullptr, partial);
  else if (mode == 2) bnact::sums_kernel<2, T><<<grid, 256, 0, st>>>(x, dy, dy2, y, sc, sh, mean, rstd, g, dskip, partial);
  else bnact::sums_kernel<3, T><<<grid, 256, 0, st>>>(x, dy, nullptr, nullptr, sc, sh, mean, rstd, g, nullptr, partial);
  OBMAN_LAUNCH_CHECK();
  bnact::bwd_finalize_kernel<<<obman_cdiv(C, 4), 256, 0, st>>>(partial, g.nblk, R, C, training, gamma, rstd, dgamma, dbeta, k);
  OBMAN_LAUNCH_CHECK();
  if (mode == 1) bnact::bwd_apply_kernel<1, T><<<grid, 256, 0, st>>>(x, dy, sc, sh, mean, rstd, k, g, dx);
  else if (mode == 2) bnact::bwd_apply_kernel<2, T><<<grid, 256, 0, st>>>(x, dskip, sc, sh, mean, rstd, k, g, dx);
  else bnact::bwd_apply_kernel<3, T><<<grid, 256, 0, st>>>(x, dy, sc, sh, mean, rstd, k, g, dx);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

/* Stem: y_pool = maxpool3x3/s2/p1(relu(bn(x))).  x [B,H,W,C] NHWC, y_pool [B,Ho,Wo,C], Ho = (H-1)/2+1.
 * ws: obman_bnact_ws_floats(B*H*W, C) floats. */
template <class T>
int bnpool_fwd(const T* x, const float* gamma, const float* beta, float* rmean, float* rvar, int B, int H, int W, int C,
               int training, float eps, float momentum, T* y_pool, unsigned char* amax, float* stats, float* ws, obman_stream_t stream) {
  if (!x || !y_pool || !stats || !ws || !gamma || !beta || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % bnact::CT) return -1;
  if ((long)B * H * W >= (1L << 31)) return -1;  // pixel indices are 32-bit
  if (!training && (!rmean || !rvar)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)B * H * W;
  const bnact::Geo g = bnact::geo(R, C);
  double* partial = reinterpret_cast<double*>(ws);
  float *mean = stats, *rstd = stats + C, *sc = stats + 2 * C, *sh = stats + 3 * C;
  if (training) {
    bnact::sums_kernel<0, T><<<dim3(g.nblk, C / bnact::CT), 256, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, g, nullptr, partial);
    OBMAN_LAUNCH_CHECK();
  }
  bnact::fwd_finalize_kernel<<<obman_cdiv(C, 4), 256, 0, st>>>(partial, g.nblk, R, C, training, eps, momentum, gamma, beta, rmean, rvar, mean,
                                                                rstd, sc, sh);
  OBMAN_LAUNCH_CHECK();
  const bnact::PoolGeo pg{H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
  const bnact::Geo gp = bnact::geo((long)B * pg.Ho * pg.Wo, C);
  bnact::pool_fwd_kernel<T><<<dim3(gp.nblk, C / bnact::CT), 256, 0, st>>>(x, sc, sh, gp, pg, y_pool, amax);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

template <class T>
int bnpool_bwd(const T* x, const float* y_pool, const unsigned char* amax, const T* d_pool, const float* gamma, const float* stats, int B, int H, int W,
               int C, int training, T* dx, float* dgamma, float* dbeta, float* ws, obman_stream_t stream) {
  if (!x || (!y_pool && !amax) || !d_pool || !stats || !ws || !dx || !dgamma || !dbeta || B <= 0 || C % bnact::CT) return -1;
  if ((long)B * H * W >= (1L << 31)) return -1;  // pixel indices are 32-bit
  hipStream_t st = (hipStream_t)stream;
  const long R = (long)B * H * W;
  const bnact::PoolGeo pg{H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
  const bnact::Geo g = bnact::geo((long)B * ((H + 1) / 2) * ((W + 1) / 2), C);  // rows of the backward = 2 x 2 pixel quads
  double* partial = reinterpret_cast<double*>(ws);
  float* k = ws + (size_t)bnact::geo(R, C).nblk * C * 2 * 2;  // workspace is sized for the pixel geometry (>= quad geometry)
  const float *mean = stats, *rstd = stats + C, *sc = stats + 2 * C, *sh = stats + 3 * C;
  dim3 grid(g.nblk, C / bnact::CT);
  bnact::pool_bwd_kernel<false, T><<<grid, 256, 0, st>>>(x, y_pool, amax, d_pool, sc, sh, mean, rstd, nullptr, g, pg, partial, nullptr);
  OBMAN_LAUNCH_CHECK();
  bnact::bwd_finalize_kernel<<<obman_cdiv(C, 4), 256, 0, st>>>(partial, g.nblk, R, C, training, gamma, rstd, dgamma, dbeta, k);
  OBMAN_LAUNCH_CHECK();
  bnact::pool_bwd_kernel<true, T><<<grid, 256, 0, st>>>(x, y_pool, amax, d_pool, sc, sh, mean, rstd, k, g, pg, nullptr, dx);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int obman_bnact_fwd(const float* x, const float* skip, const float* gamma, const float* beta, float* rmean, float* rvar, long R, int C,
                    int training, float eps, float momentum, int relu, float* y, float* stats, float* ws, obman_stream_t stream) {
  return bnact_fwd<float>(x, skip, gamma, beta, rmean, rvar, R, C, training, eps, momentum, relu, y, stats, ws, stream);
}
int obman_bnact_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* stats, long R, int C, int training,
                    int relu, int has_skip, float* dx, float* dgamma, float* dbeta, float* dskip, float* ws, obman_stream_t stream) {
  return bnact_bwd<float>(x, y, dy, nullptr, gamma, stats, R, C, training, relu, has_skip, dx, dgamma, dbeta, dskip, ws, stream);
}
int obman_bnact_bwd2(const float* x, const float* y, const float* dy, const float* dy2, const float* gamma, const float* stats, long R, int C,
                     int training, int relu, int has_skip, float* dx, float* dgamma, float* dbeta, float* dskip, float* ws, obman_stream_t stream) {
  return bnact_bwd<float>(x, y, dy, dy2, gamma, stats, R, C, training, relu, has_skip, dx, dgamma, dbeta, dskip, ws, stream);
}
int obman_bnpool_fwd(const float* x, const float* gamma, const float* beta, float* rmean, float* rvar, int B, int H, int W, int C,
                     int training, float eps, float momentum, float* y_pool, float* stats, float* ws, obman_stream_t stream) {
  return bnpool_fwd<float>(x, gamma, beta, rmean, rvar, B, H, W, C, training, eps, momentum, y_pool, nullptr, stats, ws, stream);
}
int obman_bnpool_bwd(const float* x, const float* y_pool, const float* d_pool, const float* gamma, const float* stats, int B, int H, int W,
                     int C, int training, float* dx, float* dgamma, float* dbeta, float* ws, obman_stream_t stream) {
  return bnpool_bwd<float>(x, y_pool, nullptr, d_pool, gamma, stats, B, H, W, C, training, dx, dgamma, dbeta, ws, stream);
}
/* bf16 activations (raw bits), everything else as above */
int obman_bnact_fwd_bf16(const uint16_t* x, const uint16_t* skip, const float* gamma, const float* beta, float* rmean, float* rvar, long R,
                         int C, int training, float eps, float momentum, int relu, uint16_t* y, float* stats, float* ws,
                         obman_stream_t stream) {
  return bnact_fwd<bnact::bfraw>(x, skip, gamma, beta, rmean, rvar, R, C, training, eps, momentum, relu, y, stats, ws, stream);
}
int obman_bnact_bwd_bf16(const uint16_t* x, const uint16_t* y, const uint16_t* dy, const float* gamma, const float* stats, long R, int C,
                         int training, int relu, int has_skip, uint16_t* dx, float* dgamma, float* dbeta, uint16_t* dskip, float* ws,
                         obman_stream_t stream) {
  return bnact_bwd<bnact::bfraw>(x, y, dy, nullptr, gamma, stats, R, C, training, relu, has_skip, dx, dgamma, dbeta, dskip, ws, stream);
}
int obman_bnact_bwd2_bf16(const uint16_t* x, const uint16_t* y, const uint16_t* dy, const uint16_t* dy2, const float* gamma, const float* stats,
                          long R, int C, int training, int relu, int has_skip, uint16_t* dx, float* dgamma, float* dbeta, uint16_t* dskip,
                          float* ws, obman_stream_t stream) {
  return bnact_bwd<bnact::bfraw>(x, y, dy, dy2, gamma, stats, R, C, training, relu, has_skip, dx, dgamma, dbeta, dskip, ws, stream);
}
int obman_bnpool_fwd_bf16(const uint16_t* x, const float* gamma, const float* beta, float* rmean, float* rvar, int B, int H, int W, int C,
                          int training, float eps, float momentum, uint16_t* y_pool, uint8_t* amax, float* stats, float* ws,
                          obman_stream_t stream) {
  if (!amax) return -1;
  return bnpool_fwd<bnact::bfraw>(x, gamma, beta, rmean, rvar, B, H, W, C, training, eps, momentum, y_pool, amax, stats, ws, stream);
}
int obman_bnpool_bwd_bf16(const uint16_t* x, const uint8_t* amax, const uint16_t* d_pool, const float* gamma, const float* stats, int B,
                          int H, int W, int C, int training, uint16_t* dx, float* dgamma, float* dbeta, float* ws, obman_stream_t stream) {
  if (!amax) return -1;
  return bnpool_bwd<bnact::bfraw>(x, nullptr, amax, d_pool, gamma, stats, B, H, W, C, training, dx, dgamma, dbeta, ws, stream);
}

}  // extern "C"
