// K15 - the small fully connected layers of the heads at batch size ~64 (round 6).
//
// ManoBranch's regressor (512 -> 1024 -> 256 -> {33 | 144, 10}: manobranch.py:56-81,118-121) and AtlasBranch's translation / scale heads
// (512 -> 256 -> {3, 1}: atlasbranch.py:44-69,112-116) are nn.Linear layers over B = 64 rows.  As library GEMMs each is a
// launch-latency problem: 8 addmm + 16 mm launches per configs[2] step, 6 - 14 us each - and 34.5 us for the two [64,512] x [512,256]
// products whose bias epilogue lands on a ONE-workgroup hipBLASLt solution - plus the bias-gradient sums and ReLU / threshold
// kernels around them: ~0.33 ms of a 8.7 ms step for 0.2 GFLOP.  Here a layer is one launch forward (bias and ReLU fused) and two
// backward (data gradient; weight + bias gradient), fp32 FMAs on the VALU in a fixed order (deterministic), every operand read with
// 16-byte loads where its pitch allows:
//   forward     Y[r, n]  = act(sum_k X[r, k] W[n, k] + b[n])            grid (N / 16, B / 64): lane = row, wave = 4 columns
//   data grad   dX[r, k] = sum_n gZ[r, n] W[n, k]                        grid (K / 16, B / 64): lane = row, wave = 4 input channels
//   weight grad dW[n, k] = sum_r gZ[r, n] X[r, k],  db[n] = sum_r gZ[r, n]   grid (N / 16, K / 256): lane = 4 k, wave = 4 n
// with gZ = gY * (Y > 0) where the layer has a ReLU (the mask is recomputed from the stored output, as aten's threshold_backward).
// Rows beyond B and columns beyond N / K are clamped for loads and skipped for stores.  No MFMA: 64 x 512 x 1024 is 67 MFLOP.
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

constexpr int LIN_T = 256;

__device__ __forceinline__ float lin_gz(const float* __restrict__ gy, const float* __restrict__ y, size_t i, int relu) {
  const float g = gy[i];
  return (relu && !(y[i] > 0.f)) ? 0.f : g;
}

// lane = row (64 rows per block), wave w = output columns c0 + 4 w .. + 3
__global__ __launch_bounds__(LIN_T) void linear_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, const float* __restrict__ b,
                                                          int B, int K, int N, int relu, float* __restrict__ Y) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.y * 64 + lane, rc = r < B ? r : B - 1;
  const int n0 = blockIdx.x * 16 + wave * 4;
  const float* __restrict__ x = X + (size_t)rc * K;
  const float* w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = W + (size_t)(n0 + j < N ? n0 + j : N - 1) * K;  // wave-uniform rows of W: scalar loads
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  if (((K & 3) == 0) && ((reinterpret_cast<size_t>(X) | reinterpret_cast<size_t>(W)) & 15) == 0) {
    for (; k + 16 <= K; k += 16) {  // four 16-byte chunks of the row in flight
      float4 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const float4*>(x + k + 4 * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 wv = *reinterpret_cast<const float4*>(w[j] + k + 4 * u);
          acc[j] = __fmaf_rn(xv[u].w, wv.w, __fmaf_rn(xv[u].z, wv.z, __fmaf_rn(xv[u].y, wv.y, __fmaf_rn(xv[u].x, wv.x, acc[j]))));
        }
      }
    }
    for (; k + 4 <= K; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + k);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wv = *reinterpret_cast<const float4*>(w[j] + k);
        acc[j] = __fmaf_rn(xv.w, wv.w, __fmaf_rn(xv.z, wv.z, __fmaf_rn(xv.y, wv.y, __fmaf_rn(xv.x, wv.x, acc[j]))));
      }
    }
  }
  for (; k < K; ++k) {
    const float xv = x[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __fmaf_rn(xv, w[j][k], acc[j]);
  }
  if (r >= B) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + j;
    if (n < N) {
      float v = acc[j] + (b ? b[n] : 0.f);
      if (relu) v = fmaxf(v, 0.f);
      Y[(size_t)r * N + n] = v;
    }
  }
}

// dX[r, k0 .. k0+3] = sum_n gZ[r, n] W[n, k0 .. k0+3]; lane = row, wave w = input channels k0 = 16 blockIdx.x + 4 w
__global__ __launch_bounds__(LIN_T) void linear_bwd_x_kernel(const float* __restrict__ gY, const float* __restrict__ Yout, const float* __restrict__ W,
                                                            int B, int K, int N, int relu, float* __restrict__ dX) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = blockIdx.y * 64 + lane, rc = r < B ? r : B - 1;
  const int k0 = blockIdx.x * 16 + wave * 4;
  if (k0 >= K) return;
  const bool k4 = k0 + 4 <= K && (K & 3) == 0 && (reinterpret_cast<size_t>(W) & 15) == 0;
  const size_t row = (size_t)rc * N;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int n = 0;
  if (k4 && (N & 3) == 0 && ((reinterpret_cast<size_t>(gY) | reinterpret_cast<size_t>(Yout)) & 15) == 0) {
#pragma unroll 4
    for (; n + 4 <= N; n += 4) {
      float4 g = *reinterpret_cast<const float4*>(gY + row + n);
      if (relu) {
        const float4 y = *reinterpret_cast<const float4*>(Yout + row + n);
        g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f; g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
      }
      const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)(n + u) * K + k0);  // wave-uniform
        acc[0] = __fmaf_rn(gs[u], wv.x, acc[0]); acc[1] = __fmaf_rn(gs[u], wv.y, acc[1]);
        acc[2] = __fmaf_rn(gs[u], wv.z, acc[2]); acc[3] = __fmaf_rn(gs[u], wv.w, acc[3]);
      }
    }
  }
  for (; n < N; ++n) {
    const float g = lin_gz(gY, Yout, row + n, relu);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < K) acc[j] = __fmaf_rn(g, W[(size_t)n * K + k0 + j], acc[j]);
  }
  if (r >= B) return;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k0 + j < K) dX[(size_t)r * K + k0 + j] = acc[j];
}

// dW[n, k] = sum_r gZ[r, n] X[r, k]; db[n] = sum_r gZ[r, n].  Block = 16 output rows n x 256 input channels k: lane = 4 consecutive
// k (coalesced 16-byte loads of X), wave w = rows n0 + 4 w .. + 3 (wave-uniform gZ values).  The rows r are walked in order.
__global__ __launch_bounds__(LIN_T) void linear_bwd_w_kernel(const float* __restrict__ gY, const float* __restrict__ Yout, const float* __restrict__ X,
                                                            int B, int K, int N, int relu, float* __restrict__ dW, float* __restrict__ db) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = blockIdx.x * 16 + wave * 4, k0 = blockIdx.y * 256 + lane * 4;
  if (n0 >= N) return;
  const bool vec = k0 + 4 <= K && (K & 3) == 0 && (reinterpret_cast<size_t>(X) & 15) == 0;
  float acc[4][4], bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
  int nn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) nn[j] = n0 + j < N ? n0 + j : N - 1;
#pragma unroll 4
  for (int r = 0; r < B; ++r) {
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = lin_gz(gY, Yout, (size_t)r * N + nn[j], relu);  // wave-uniform
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      const float4 t = *reinterpret_cast<const float4*>(X + (size_t)r * K + k0);
      xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < K) xv[e] = X[(size_t)r * K + k0 + e];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bs[j] += g[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] = __fmaf_rn(g[j], xv[e], acc[j][e]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n0 + j >= N) continue;
    if (vec) {
      *reinterpret_cast<float4*>(dW + (size_t)(n0 + j) * K + k0) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < K) dW[(size_t)(n0 + j) * K + k0 + e] = acc[j][e];
    }
    if (db && blockIdx.y == 0 && lane == 0) db[n0 + j] = bs[j];
  }
}

}  // namespace

extern "C" {

int obman_linear_fwd(const float* x, const float* w, const float* b, int B, int K, int N, int relu, float* y, obman_stream_t stream) {
  if (B <= 0 || K <= 0 || N <= 0 || !x || !w || !y) return -1;
  linear_fwd_kernel<<<dim3(obman_cdiv(N, 16), obman_cdiv(B, 64)), LIN_T, 0, (hipStream_t)stream>>>(x, w, b, B, K, N, relu, y);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_linear_bwd(const float* gy, const float* y, const float* x, const float* w, int B, int K, int N, int relu, float* dx, float* dw,
                     float* db, obman_stream_t stream) {
  if (B <= 0 || K <= 0 || N <= 0 || !gy || !x || !w || (relu && !y)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    linear_bwd_x_kernel<<<dim3(obman_cdiv(K, 16), obman_cdiv(B, 64)), LIN_T, 0, st>>>(gy, y, w, B, K, N, relu, dx);
    OBMAN_LAUNCH_CHECK();
  }
  if (dw) {
    linear_bwd_w_kernel<<<dim3(obman_cdiv(N, 16), obman_cdiv(K, 256)), LIN_T, 0, st>>>(gy, y, x, B, K, N, relu, dw, db);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
