// K10: GPU-side image input stream.
//
// Replaces the per-sample CPU pixel pipeline of HandDataset.get_sample (handataset.py:373-405): Gaussian blur and colour
// jitter of the full source image (Pillow / torchvision-PIL-backend uint8 arithmetic), the nearest-neighbour affine crop
// (handutils.py:48-60 -> Pillow Geometry.c affine_fixed), to_tensor and normalize.  Byte / integer work: results are
// bit-identical to the CPU path (oracle/inputstream.py, pinned against Pillow and the reference's own outputs).
//
// Three kernels per batch, all HBM/L2-bound byte streams:
//   blur_kernel    source RGB888 -> blurred RGBX copy: 3 horizontal + 3 vertical extended-box passes on an LDS tile with
//                  halo (only launched when some sample is blurred)
//   mean_kernel    per-sample integer sum of the luma of (ops preceding the contrast op)(pixel) over the WHOLE source
//                  image: ImageEnhance.Contrast blends against the image's mean grey level
//   warp_kernel    one lane per output pixel: 16.16 fixed-point source coordinate, pointwise colour ops (they commute
//                  with nearest-neighbour sampling), /255, black frame, normalise, store NCHW or NHWC
// Pointwise ops are applied to the 65 536 sampled pixels instead of the whole source image; only blur (a neighbourhood
// op) and the contrast mean (a global reduction) touch every source pixel.
#include <stdint.h>

#include "../../include/obman_hip.h"
#include "common.h"

#pragma clang fp contract(off)

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned char u8;

// Source pixels arrive exactly as the decoder left them (3 bytes / pixel: the host stages a plain memcpy); a wave reads
// 192 consecutive bytes per row segment.  The blurred intermediate is RGBX (one dword / pixel).
__device__ __forceinline__ u32 load_rgb(const u8* __restrict__ img, size_t pixel) {
  const u8* p = img + pixel * 3;
  return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16);
}

enum { OP_BRIGHTNESS = 1, OP_SATURATION = 2, OP_HUE = 3, OP_CONTRAST = 4 };
constexpr int BLUR_TILE = 32;

// Pillow Blend.c: out = in1 + alpha * (in2 - in1) in C float; truncation inside [0,1], clipping outside.
__device__ __forceinline__ int blend8(int a, int b, float alpha, bool interp) {
  const float t = __fadd_rn((float)a, __fmul_rn(alpha, (float)(b - a)));
  if (interp) return (int)t & 0xff;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// Pillow Convert.c L24: ITU-R 601-2 luma, 16.16 fixed point
__device__ __forceinline__ int luma8(int r, int g, int b) { return (int)(((u32)r * 19595u + (u32)g * 38470u + (u32)b * 7471u + 0x8000u) >> 16); }

// Pillow Convert.c rgb2hsv_row / hsv2rgb_row with a uint8-wrapping hue shift in between (torchvision adjust_hue).
// float variables, double-typed literals promote: mirrored operation by operation.
__device__ __forceinline__ void hue_rotate(int& r, int& g, int& b, int shift) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
    else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
    double hd = __dadd_rn(__ddiv_rn((double)h, 6.0), 1.0);
    hd = hd - floor(hd);  // fmod(x, 1.0) for x > 0 (exact)
    h = (float)hd;
    uh = min(255, max(0, (int)__dmul_rn((double)h, 255.0)));
    us = min(255, max(0, (int)__dmul_rn((double)s, 255.0)));
  }
  uh = (uh + shift) & 0xff;
  if (us == 0) { r = g = b = uv; return; }
  const double hf = __ddiv_rn(__dmul_rn((double)(float)uh, 6.0), 255.0);
  const float fi = (float)floor(hf);
  const double f = (double)(float)__dsub_rn(hf, (double)fi);
  const double fs = (double)(float)__ddiv_rn((double)(float)us, 255.0);
  const double vf = (double)uv;
  const int p = min(255, max(0, (int)floor(__dadd_rn(__dmul_rn(vf, __dsub_rn(1.0, fs)), 0.5))));
  const int q = min(255, max(0, (int)floor(__dadd_rn(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, f))), 0.5))));
  const int t = min(255, max(0, (int)floor(__dadd_rn(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn(fs, __dsub_rn(1.0, f)))), 0.5))));
  switch (((int)fi) % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// ops [first, last) of the sample's list; `mean` is the contrast grey level (only read by OP_CONTRAST)
__device__ __forceinline__ void apply_ops(const obman_img_params& P, int first, int last, int mean, int& r, int& g, int& b) {
  for (int k = first; k < last; ++k) {
    const int op = P.op[k];
    const float f = P.factor[k];
    const bool interp = f >= 0.f && f <= 1.f;
    if (op == OP_BRIGHTNESS) {
      r = blend8(0, r, f, interp); g = blend8(0, g, f, interp); b = blend8(0, b, f, interp);
    } else if (op == OP_SATURATION) {
      const int L = luma8(r, g, b);
      r = blend8(L, r, f, interp); g = blend8(L, g, f, interp); b = blend8(L, b, f, interp);
    } else if (op == OP_CONTRAST) {
      r = blend8(mean, r, f, interp); g = blend8(mean, g, f, interp); b = blend8(mean, b, f, interp);
    } else if (op == OP_HUE) {
      hue_rotate(r, g, b, P.hue_shift);
    }
  }
}

__device__ __forceinline__ int contrast_index(const obman_img_params& P) {
  for (int k = 0; k < P.n_ops; ++k)
    if (P.op[k] == OP_CONTRAST) return k;
  return -1;
}

// One extended-box tap set on a packed RGBX pixel (Pillow BoxBlur.c ImagingLineBoxBlur32, UINT32 arithmetic):
// out_c = (sum_{|d|<=r} in_c[x+d] * ww + (in_c[x-r-1] + in_c[x+r+1]) * fw + 2^23) >> 24
struct Acc3 { u32 r, g, b; };
__device__ __forceinline__ void acc_add(Acc3& a, u32 px) { a.r += px & 0xff; a.g += (px >> 8) & 0xff; a.b += (px >> 16) & 0xff; }
__device__ __forceinline__ u32 box_out(const Acc3& in, const Acc3& far, u32 ww, u32 fw) {
  const u32 r = (in.r * ww + far.r * fw + (1u << 23)) >> 24;
  const u32 g = (in.g * ww + far.g * fw + (1u << 23)) >> 24;
  const u32 b = (in.b * ww + far.b * fw + (1u << 23)) >> 24;
  return r | (g << 8) | (b << 16);
}

// grid (tiles_x, tiles_y, B), 256 lanes; dynamic LDS = 2 * T * T * 4 bytes, T = BLUR_TILE + 2 * halo.
// Every pass reads neighbours at image coordinates clamped to the image (Pillow replicates the edge pixel of the CURRENT
// intermediate image), then clamped to the tile (only garbage in the outer halo, which shrinks by r+1 per pass and never
// reaches the inner 32x32 outputs because halo = 3 (r_max + 1)).
__global__ __launch_bounds__(256) void blur_kernel(const u8* __restrict__ src, const obman_img_params* __restrict__ params, int pitch_h,
                                                   int pitch_w, int halo, u32* __restrict__ dst) {
  extern __shared__ u32 lds[];
  const obman_img_params P = params[blockIdx.z];
  if (P.blur_r < 0) return;
  const int W = P.src_w, H = P.src_h;
  const int x0 = blockIdx.x * BLUR_TILE, y0 = blockIdx.y * BLUR_TILE;
  if (x0 >= W || y0 >= H) return;
  const int T = BLUR_TILE + 2 * halo;
  u32* bufA = lds;
  u32* bufB = lds + T * T;
  const u8* img = src + (size_t)blockIdx.z * pitch_h * pitch_w * 3;
  const int ox = x0 - halo, oy = y0 - halo;
  const int step_y = 256 / T, step_x = 256 - step_y * T;  // (ty, tx) advance of a 256-lane stride: no division per position
  const int ty0 = (int)threadIdx.x / T, tx0 = (int)threadIdx.x - ty0 * T;
  for (int i = threadIdx.x, ty = ty0, tx = tx0; i < T * T; i += 256) {
    const int gx = min(max(ox + tx, 0), W - 1), gy = min(max(oy + ty, 0), H - 1);
    bufA[i] = load_rgb(img, (size_t)gy * pitch_w + gx);
    tx += step_x; ty += step_y;
    if (tx >= T) { tx -= T; ++ty; }
  }
  __syncthreads();
  const int r = P.blur_r;
  const u32 ww = P.blur_ww, fw = P.blur_fw;
  for (int pass = 0; pass < 6; ++pass) {
    const bool horiz = pass < 3;
    for (int i = threadIdx.x, ty = ty0, tx = tx0; i < T * T; i += 256) {
      const int g = horiz ? ox + tx : oy + ty;     // image coordinate along the filtered axis
      const int lim = horiz ? W - 1 : H - 1;
      const int org = horiz ? ox : oy;
      const int stride = horiz ? 1 : T;
      const int base = horiz ? ty * T : tx;
      Acc3 in = {0, 0, 0}, far = {0, 0, 0};
      for (int d = -r; d <= r; ++d) {
        const int t = min(max(min(max(g + d, 0), lim) - org, 0), T - 1);
        acc_add(in, bufA[base + t * stride]);
      }
      const int tl = min(max(min(max(g - r - 1, 0), lim) - org, 0), T - 1);
      const int tr = min(max(min(max(g + r + 1, 0), lim) - org, 0), T - 1);
      acc_add(far, bufA[base + tl * stride]);
      acc_add(far, bufA[base + tr * stride]);
      bufB[i] = box_out(in, far, ww, fw);
      tx += step_x; ty += step_y;
      if (tx >= T) { tx -= T; ++ty; }
    }
    __syncthreads();
    u32* t = bufA; bufA = bufB; bufB = t;
  }
  u32* out = dst + (size_t)blockIdx.z * pitch_h * pitch_w;
  for (int i = threadIdx.x; i < BLUR_TILE * BLUR_TILE; i += 256) {
    const int ty = i / BLUR_TILE, tx = i - ty * BLUR_TILE;
    const int gx = x0 + tx, gy = y0 + ty;
    if (gx < W && gy < H) out[(size_t)gy * pitch_w + gx] = bufA[(ty + halo) * T + tx + halo];
  }
}

// grid (blocks, B): integer luma sum of the whole source image after the ops that precede the contrast op.
__global__ __launch_bounds__(256) void mean_kernel(const u8* __restrict__ src, const u32* __restrict__ blurred,
                                                   const obman_img_params* __restrict__ params, int pitch_h, int pitch_w,
                                                   u64* __restrict__ sums) {
  const obman_img_params P = params[blockIdx.y];
  const int ci = contrast_index(P);
  if (ci < 0) return;
  const size_t slot = (size_t)blockIdx.y * pitch_h * pitch_w;
  const bool use_blurred = P.blur_r >= 0;
  const int W = P.src_w, H = P.src_h;
  u32 acc = 0;  // <= 255 * pixels per lane (rows / gridDim.x * ceil(W / 256)): far below 2^32 for any image that fits the grid
  for (int y = blockIdx.x; y < H; y += gridDim.x)
  for (int x = threadIdx.x; x < W; x += 256) {
    const size_t at = slot + (size_t)y * pitch_w + x;
    const u32 px = use_blurred ? blurred[at] : load_rgb(src, at);
    int r = px & 0xff, g = (px >> 8) & 0xff, b = (px >> 16) & 0xff;
    apply_ops(P, 0, ci, 0, r, g, b);
    acc += (u32)luma8(r, g, b);
  }
  __shared__ u32 part[256];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(&sums[blockIdx.y], (u64)part[0]);  // integer: order-independent
}

struct WarpCfg {
  int res, channels_last, pad;  // pad = black frame width (0 = none)
  float mean[3], std[3];
};

// grid (ceil(res*res/256), B): one lane per output pixel
__global__ __launch_bounds__(256) void warp_kernel(const u8* __restrict__ src, const u32* __restrict__ blurred,
                                                   const obman_img_params* __restrict__ params, int pitch_h, int pitch_w,
                                                   const u64* __restrict__ sums, WarpCfg cfg, float* __restrict__ out) {
  const obman_img_params P = params[blockIdx.y];
  const int R = cfg.res;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * R) return;
  const int y = i / R, x = i - y * R;
  const size_t slot = (size_t)blockIdx.y * pitch_h * pitch_w;
  const long long xx = (long long)P.A[2] + (long long)x * P.A[0] + (long long)y * P.A[1];
  const long long yy = (long long)P.A[5] + (long long)x * P.A[3] + (long long)y * P.A[4];
  const long long xin = xx >> 16, yin = yy >> 16;
  int r = 0, g = 0, b = 0;
  if (xin >= 0 && xin < P.src_w && yin >= 0 && yin < P.src_h) {
    const int sx = P.flip ? P.src_w - 1 - (int)xin : (int)xin;
    const size_t at = slot + (size_t)yin * pitch_w + sx;
    const u32 px = P.blur_r >= 0 ? blurred[at] : load_rgb(src, at);
    r = px & 0xff; g = (px >> 8) & 0xff; b = (px >> 16) & 0xff;
    int mean = 0;
    if (contrast_index(P) >= 0)  // int(sum / count + 0.5) in double, as ImageStat + ImageEnhance.Contrast
      mean = (int)__dadd_rn(__ddiv_rn((double)sums[blockIdx.y], (double)((long long)P.src_h * P.src_w)), 0.5);
    apply_ops(P, 0, P.n_ops, mean, r, g, b);
  }
  float v[3] = {__fdiv_rn((float)r, 255.f), __fdiv_rn((float)g, 255.f), __fdiv_rn((float)b, 255.f)};
  if (cfg.pad > 0) {  // handataset.py:391-397: [0, pad) and [res - pad, res - 1) on both axes (the last row / column stays)
    const bool fy = y < cfg.pad || (y >= R - cfg.pad && y < R - 1);
    const bool fx = x < cfg.pad || (x >= R - cfg.pad && x < R - 1);
    if (fy || fx) v[0] = v[1] = v[2] = 0.f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) v[c] = __fdiv_rn(__fsub_rn(v[c], cfg.mean[c]), cfg.std[c]);
  if (cfg.channels_last) {
    float* o = out + ((size_t)blockIdx.y * R * R + i) * 3;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  } else {
    float* o = out + (size_t)blockIdx.y * 3 * R * R + i;
    o[0] = v[0]; o[(size_t)R * R] = v[1]; o[(size_t)2 * R * R] = v[2];
  }
}

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" long obman_imgstream_ws_bytes(int B, int pitch_h, int pitch_w) {
  return (long)(align256((size_t)B * sizeof(u64)) + align256((size_t)B * pitch_h * pitch_w * sizeof(u32)));
}

extern "C" int obman_imgstream_fwd(const uint8_t* src, int B, int pitch_h, int pitch_w, const obman_img_params* params,
                                   int max_blur_r, int any_contrast, int out_res, int channels_last, int black_pad,
                                   const float* mean3, const float* std3, void* ws, float* out, obman_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || out_res <= 0) return 0;
  if (max_blur_r > 8) return (int)hipErrorInvalidValue;  // LDS tile (32 + 6 (r + 1))^2 * 8 B must fit 64 KB
  u64* sums = (u64*)ws;
  u32* blurred = (u32*)((char*)ws + align256((size_t)B * sizeof(u64)));
  if (max_blur_r >= 0) {
    const int halo = 3 * (max_blur_r + 1), T = BLUR_TILE + 2 * halo;
    dim3 grid(obman_cdiv(pitch_w, BLUR_TILE), obman_cdiv(pitch_h, BLUR_TILE), B);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), (size_t)2 * T * T * sizeof(u32), stream, src, params, pitch_h, pitch_w, halo, blurred);
    OBMAN_LAUNCH_CHECK();
  }
  if (any_contrast) {
    hipError_t e = obman_fill_u32(sums, 0u, (size_t)2 * B, stream);
    if (e != hipSuccess) return (int)e;
    const int blocks = min(pitch_h, 64);  // rows are dealt round-robin to the blocks of a sample
    hipLaunchKernelGGL(mean_kernel, dim3(blocks, B), dim3(256), 0, stream, src, blurred, params, pitch_h, pitch_w, sums);
    OBMAN_LAUNCH_CHECK();
  }
  WarpCfg cfg;
  cfg.res = out_res; cfg.channels_last = channels_last; cfg.pad = black_pad;
  for (int c = 0; c < 3; ++c) { cfg.mean[c] = mean3 ? mean3[c] : 0.5f; cfg.std[c] = std3 ? std3[c] : 1.f; }
  hipLaunchKernelGGL(warp_kernel, dim3(obman_cdiv((long)out_res * out_res, 256), B), dim3(256), 0, stream, src, blurred, params,
                     pitch_h, pitch_w, sums, cfg, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
