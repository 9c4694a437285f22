// K15 - the small fully connected layers of the heads at batch size ~64 (round 6).
//
// ManoBranch's regressor (512 -> 1024 -> 256 -> {33 | 144, 10}: manobranch.py:56-81,118-121) and AtlasBranch's translation / scale heads
// (512 -> 256 -> {3, 1}: atlasbranch.py:44-69,112-116) are nn.Linear layers over B = 64 rows.  As library GEMMs each is a
// launch-latency problem: 8 addmm + 16 mm launches per configs[2] step, 6 - 14 us each - and 34.5 us for the two [64,512] x [512,256]
// products whose bias epilogue lands on a ONE-workgroup hipBLASLt solution - plus the bias-gradient sums and ReLU / threshold
// kernels around them: ~0.33 ms of a 8.7 ms step for 0.2 GFLOP.  Here a layer is one launch forward (bias and ReLU fused) and two
// backward (data gradient; weight + bias gradient), fp32 FMAs on the VALU in a fixed order (deterministic), all three through ONE
// LDS-tiled kernel (small_gemm_kernel below).  With gZ = gY * (Y > 0) where the layer has a ReLU (the mask is recomputed from the
// stored output, as aten's threshold_backward).  No MFMA: 64 x 512 x 1024 is 67 MFLOP.
#include "common.h"
#include "../../include/obman_hip.h"

namespace {

// One LDS-tiled kernel for the three products (first version: one lane per row reading its own row of X with 16-byte loads - 64
// cache lines per wave-load - ran 27 - 37 us per launch, three times the library GEMMs it was meant to replace).  C tile = 64 rows x
// 16 columns per 256-thread block, contraction in chunks of 32 staged through LDS with every global load contiguous across lanes
// along the operand's own contiguous dimension; the next chunk is in registers while the current one multiplies (two LDS buffers,
// one barrier per chunk).  In the compute phase lane = row, wave = 4 columns: per 4 contraction steps one 16-byte LDS read of the
// lane's A row and four wave-uniform (broadcast) 16-byte reads of the B columns feed 16 FMAs.
//   MODE 0  forward      C[m, n] = act(sum_k X[m, k] W[n, k] + b[n])     rows m = batch,   cols n = outputs,        contraction K
//   MODE 1  data grad    C[m, k] = sum_n gZ[m, n] W[n, k]                rows m = batch,   cols = input channels,   contraction N
//   MODE 2  weight grad  C[n, k] = sum_r gZ[r, n] X[r, k]; db[n] = sum_r gZ[r, n]   rows n = outputs, cols = input channels, contraction B
// gZ = gY * (Y > 0) with a ReLU (mask recomputed from the stored output while staging), else gY.
constexpr int LIN_T = 256, LIN_TM = 64, LIN_TN = 16, LIN_TK = 32, LIN_P = 36;  // pitch 36 floats: 16-byte aligned rows, conflict-free b128 reads

struct LinArgs {
  const float *X, *W, *b, *gY, *Y;  // Y: the forward's output (mask) for MODE 1 / 2 with relu
  float *C, *db;
  int B, K, N, relu;
};

template <int MODE>
__device__ __forceinline__ float lin_a(const LinArgs& a, int m, int k, int M, int KC) {
  if (m >= M || k >= KC) return 0.f;
  if (MODE == 0) return a.X[(size_t)m * a.K + k];
  const size_t i = MODE == 1 ? (size_t)m * a.N + k : (size_t)k * a.N + m;
  const float g = a.gY[i];
  return (a.relu && !(a.Y[i] > 0.f)) ? 0.f : g;
}
template <int MODE>
__device__ __forceinline__ float lin_b(const LinArgs& a, int k, int n, int KC, int NC) {
  if (k >= KC || n >= NC) return 0.f;
  if (MODE == 0) return a.W[(size_t)n * a.K + k];
  if (MODE == 1) return a.W[(size_t)k * a.K + n];
  return a.X[(size_t)k * a.K + n];
}

template <int MODE>
__global__ __launch_bounds__(LIN_T) void small_gemm_kernel(LinArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][LIN_TM][LIN_P];
  __shared__ __attribute__((aligned(16))) float Bs[2][LIN_TN][LIN_P];
  const int M = MODE == 2 ? a.N : a.B, NC = MODE == 0 ? a.N : a.K, KC = MODE == 0 ? a.K : (MODE == 1 ? a.N : a.B);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * LIN_TM, n0 = blockIdx.x * LIN_TN;
  // staging tasks: A tile 64 x 32 = 8 elements per thread, B tile 32 x 16 = 2 per thread, consecutive along the CONTIGUOUS dimension
  int am[8], ak[8], bk[2], bn[2];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid * 8 + u;
    if (MODE == 2) { ak[u] = e / LIN_TM; am[u] = e % LIN_TM; }  // gZ[r, n]: contiguous along the tile's rows (n)
    else { am[u] = e / LIN_TK; ak[u] = e % LIN_TK; }            // X[m, k] / gZ[m, n]: contiguous along the contraction
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid * 2 + u;
    if (MODE == 0) { bn[u] = e / LIN_TK; bk[u] = e % LIN_TK; }  // W[n, k]: contiguous along the contraction
    else { bk[u] = e / LIN_TN; bn[u] = e % LIN_TN; }            // W[n, k] / X[r, k]: contiguous along the tile's columns
  }
  float ra[8], rb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) ra[u] = lin_a<MODE>(a, m0 + am[u], k0 + ak[u], M, KC);
#pragma unroll
    for (int u = 0; u < 2; ++u) rb[u] = lin_b<MODE>(a, k0 + bk[u], n0 + bn[u], KC, NC);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) As[buf][am[u]][ak[u]] = ra[u];
#pragma unroll
    for (int u = 0; u < 2; ++u) Bs[buf][bn[u]][bk[u]] = rb[u];
  };
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, rowsum = 0.f;
  const int nchunks = (KC + LIN_TK - 1) / LIN_TK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int cur = c & 1;
    if (c + 1 < nchunks) fetch((c + 1) * LIN_TK);
#pragma unroll
    for (int q = 0; q < LIN_TK / 4; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(&As[cur][lane][4 * q]);
      if (MODE == 2) rowsum += (av.x + av.y) + (av.z + av.w);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bv = *reinterpret_cast<const float4*>(&Bs[cur][wave * 4 + j][4 * q]);
        acc[j] = __fmaf_rn(av.w, bv.w, __fmaf_rn(av.z, bv.z, __fmaf_rn(av.y, bv.y, __fmaf_rn(av.x, bv.x, acc[j]))));
      }
    }
    if (c + 1 < nchunks) stash(cur ^ 1);
    __syncthreads();
  }
  const int m = m0 + lane;
  if (m >= M) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + wave * 4 + j;
    if (n >= NC) continue;
    float v = acc[j];
    if (MODE == 0) {
      if (a.b) v += a.b[n];
      if (a.relu) v = fmaxf(v, 0.f);
    }
    a.C[(size_t)m * NC + n] = v;
  }
  if (MODE == 2 && a.db && blockIdx.x == 0 && wave == 0) a.db[m] = rowsum;
}

}  // namespace

extern "C" {

int obman_linear_fwd(const float* x, const float* w, const float* b, int B, int K, int N, int relu, float* y, obman_stream_t stream) {
  if (B <= 0 || K <= 0 || N <= 0 || !x || !w || !y) return -1;
  LinArgs a{x, w, b, nullptr, nullptr, y, nullptr, B, K, N, relu};
  small_gemm_kernel<0><<<dim3(obman_cdiv(N, LIN_TN), obman_cdiv(B, LIN_TM)), LIN_T, 0, (hipStream_t)stream>>>(a);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_linear_bwd(const float* gy, const float* y, const float* x, const float* w, int B, int K, int N, int relu, float* dx, float* dw,
                     float* db, obman_stream_t stream) {
  if (B <= 0 || K <= 0 || N <= 0 || !gy || !x || !w || (relu && !y) || (db && !dw)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    LinArgs a{x, w, nullptr, gy, y, dx, nullptr, B, K, N, relu};
    small_gemm_kernel<1><<<dim3(obman_cdiv(K, LIN_TN), obman_cdiv(B, LIN_TM)), LIN_T, 0, st>>>(a);
    OBMAN_LAUNCH_CHECK();
  }
  if (dw) {
    LinArgs a{x, w, nullptr, gy, y, dw, db, B, K, N, relu};
    small_gemm_kernel<2><<<dim3(obman_cdiv(K, LIN_TN), obman_cdiv(N, LIN_TM)), LIN_T, 0, st>>>(a);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
