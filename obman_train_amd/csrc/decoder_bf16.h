// K6, bf16-MFMA flavour (BASELINE.json configs[2] precision) - included by decoder.hip inside namespace dec.
//
// What bounds it: with the contraction on v_mfma_f32_32x32x16_bf16 (16x the fp32-input rate) the matrix pipe is idle most of
// the time; the decoder becomes a STREAMING problem: 330 kflop per point but, at 16 050 points x 64 samples, gigabytes of
// per-point activations.  So this flavour is built around bytes and memory latency:
//   * activations that must be materialised (h2, h3 forward; the masked gradient gy2 backward) are stored as bf16 and moved in
//     16-byte pieces (8 channels per lane);
//   * the largest tensor of the backward, gy1 [R x 515], is NEVER written: the dA(gy1) GEMM tiles its rows as
//     (8 samples x 16 template vertices) and its epilogue reduces the two things layer 1 needs - P[b,c] = sum_n gy1 and
//     Q[n,c] = sum_b gy1 - from the accumulators (BatchNorm-1's S1/S2 follow from P, Q and the two small factors);
//   * every k-tile of an A operand is requested DEPTH tiles ahead into a register queue (the old kernel had one tile in
//     flight per block and one block per CU: each 32-deep k-step exposed a full Infinity-Cache / HBM round trip);
//   * per-channel constants of the fused BatchNorm / ReLU / BatchNorm-backward operand generators are staged in LDS once per
//     block; weights are bf16 [n][k] images cast once per call (wcast_kernel).
// A block is 128 rows x 320 columns (512 threads = 4 x 2 waves, wave tile 32 x 160): the generated A tile feeds every column
// of a 257-wide layer (515-wide: two column blocks).  LDS tiles are [row][k] bf16 with an 80-byte pitch (conflict-free
// ds_read_b128 fragments).  fp32 accumulation, fp32/fp64 BatchNorm statistics (of the ROUNDED stored values, so that the
// consumer's normalisation is exact for what it reads), fp32 master weights.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned short bfraw;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vectors stay in registers (HIP's uint4 class went to scratch)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int LP = 40;    // LDS row pitch of a [row][32 k] tile in bf16 elements (32 + 8 pad)
constexpr int LPT = 72;   // ... of a [channel][64 rows] tile of the weight-gradient kernel (64 + 8 pad)
constexpr int NTB = 512;  // 8 waves: 4 along M x 2 along N
constexpr int BKT = 64;   // rows per k-tile of the weight-gradient kernel

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {  // v_cvt_pk_bf16_f32, round to nearest even
  const f32x2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned lane_swap1(unsigned v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);
}
__device__ __forceinline__ void unpack8(const u32x4& w, float* o) {
  o[0] = bf_lo(w.x); o[1] = bf_hi(w.x); o[2] = bf_lo(w.y); o[3] = bf_hi(w.y);
  o[4] = bf_lo(w.z); o[5] = bf_hi(w.z); o[6] = bf_lo(w.w); o[7] = bf_hi(w.w);
}
__device__ __forceinline__ u32x4 pack8(const float* v) {
  u32x4 w;
  w.x = pack_bf16(v[0], v[1]); w.y = pack_bf16(v[2], v[3]); w.z = pack_bf16(v[4], v[5]); w.w = pack_bf16(v[6], v[7]);
  return w;
}

// out[n][k] (pitch Kp, zero beyond K) = transposed ? W[k][n] : W[n][k]
__global__ __launch_bounds__(256) void wcast_kernel(const float* __restrict__ W, int ld, int Nn, int K, int Kp, int transposed,
                                                    bfraw* __restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= (long)Nn * Kp) return;
  const int n = (int)(i / Kp), k = (int)(i - (long)n * Kp);
  auto at = [&](int kk) { return kk < K ? (transposed ? W[(size_t)kk * ld + n] : W[(size_t)n * ld + kk]) : 0.f; };
  *reinterpret_cast<unsigned*>(out + i) = pack_bf16(at(k), at(k + 1));
}

// ------------------------------------------------------------------------------------------------ row geometry
// linear: block x covers rows [128 x, 128 x + 128).  tiled (dA(gy1)): block x = (sample group bg of 8, vertex group ng of 16),
// row i of the block = (sample bg*8 + i/16, vertex ng*16 + i%16): an M-wave (32 rows) holds two samples x 16 vertices.
struct RowGeo {
  int R, N, B, NG, tiled;
  __device__ __forceinline__ void map(int blk, int i, int& b, int& n, long& r, bool& ok) const {
    if (tiled) {
      const int ng = blk % NG, bg = blk / NG;
      b = bg * 8 + (i >> 4);
      n = ng * 16 + (i & 15);
      ok = b < B && n < N;
      if (!ok) { b = 0; n = 0; }
      r = (long)b * N + n;
    } else {
      const int rr = blk * BM + i;  // R < 2^31 (obman_pointgen_* reject larger row counts)
      ok = rr < R;
      const int rc = ok ? rr : 0;
      r = rc;
      b = rc / N;
      n = rc - b * N;
    }
  }
  __host__ int blocks() const { return tiled ? NG * ((B + 7) / 8) : (int)(((long)R + BM - 1) / BM); }
};

// Workgroups are dealt to the 8 XCDs (each with its own 4 MB L2) round-robin by linear id.  -> a virtual id such that
// CONSECUTIVE virtual ids run on the same XCD at about the same time: blocks that read the same operand rows (the column
// tiles of one row block; the output tiles of one split-K chunk) are given consecutive virtual ids and meet in one L2
// instead of fetching the rows once per tile from HBM.
__device__ __forceinline__ int xcd_virtual_id(int lin, int total) {
  const int q = total >> 3, r = total & 7, x = lin & 7, s = lin >> 3;
  return x < r ? x * (q + 1) + s : r * (q + 1) + (x - r) * q + s;
}

// ------------------------------------------------------------------------------------------------ A operands (rows GEMM)
// A thread of the rows kernel owns ONE row and one 8-wide k chunk per tile (16-byte accesses).  load() only issues loads;
// fin() turns a landed chunk into 8 operand values using the per-channel constants staged in LDS (kcs = [NC][Kp]).  Columns
// >= K come out as exact zeros because their constants are staged as zeros and every padding column of a source array holds a
// finite value (the producers write zeros there: EpiStoreB / EpiMaskB up to the pitch, prep_kernel for Gx / Fx); rows beyond
// the end are zeroed once per chunk by the caller (Row::ok), not per element.
struct BGridFeat {  // a1[r,k] = relu(gamma[k] * (Gx[n,k] + Fx[b,k]) + beta[k])   (Gx, Fx: fp32 x-hat factors of layer 1)
  const float *Gx, *Fx, *gamma, *beta;
  int ld, K;
  static constexpr int NC = 2, DEPTH = 4;
  struct Row { const float *g, *f; bool ok; };
  struct Raw { float4 g0, g1, f0, f1; };
  __device__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < Kp; i += NTB) { const bool ok = i < K; kcs[i] = ok ? gamma[i] : 0.f; kcs[Kp + i] = ok ? beta[i] : 0.f; }
  }
  __device__ Row row(long, int b, int n, bool ok) const { return Row{Gx + (size_t)n * ld, Fx + (size_t)b * ld, ok}; }
  __device__ void load(Raw& q, const Row& w, int k) const {
    const int c = k <= ld - 8 ? k : ld - 8;
    q.g0 = *reinterpret_cast<const float4*>(w.g + c); q.g1 = *reinterpret_cast<const float4*>(w.g + c + 4);
    q.f0 = *reinterpret_cast<const float4*>(w.f + c); q.f1 = *reinterpret_cast<const float4*>(w.f + c + 4);
  }
  __device__ void fin(const Row& w, const float* kcs, int Kp, int k, const Raw& q, float* o) const {
    const float4 ga0 = *reinterpret_cast<const float4*>(kcs + k), ga1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 be0 = *reinterpret_cast<const float4*>(kcs + Kp + k), be1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float x[8] = {q.g0.x + q.f0.x, q.g0.y + q.f0.y, q.g0.z + q.f0.z, q.g0.w + q.f0.w,
                        q.g1.x + q.f1.x, q.g1.y + q.f1.y, q.g1.z + q.f1.z, q.g1.w + q.f1.w};
    const float ga[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
    const float be[8] = {be0.x, be0.y, be0.z, be0.w, be1.x, be1.y, be1.z, be1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaxf(__fmaf_rn(ga[j], x[j], be[j]), 0.f);
  }
};
struct BBnRelu {  // a[r,k] = relu(s[k] * H[r,k] + t[k]),  H stored bf16
  const bfraw* H;
  const float *s, *t;
  int ld, K;
  static constexpr int NC = 2, DEPTH = 4;
  struct Row { const bfraw* p; bool ok; };
  struct Raw { u32x4 h; };
  __device__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < Kp; i += NTB) { const bool ok = i < K; kcs[i] = ok ? s[i] : 0.f; kcs[Kp + i] = ok ? t[i] : 0.f; }
  }
  __device__ Row row(long r, int, int, bool ok) const { return Row{H + (size_t)r * ld, ok}; }
  __device__ void load(Raw& q, const Row& w, int k) const { q.h = *reinterpret_cast<const u32x4*>(w.p + (k <= ld - 8 ? k : ld - 8)); }
  __device__ void fin(const Row& w, const float* kcs, int Kp, int k, const Raw& q, float* o) const {
    float h[8];
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaxf(__fmaf_rn(kcs[k + j], h[j], kcs[Kp + k + j]), 0.f);
  }
};
// d(loss)/d(h) of a BatchNorm'd layer in the affine form gh = ka*gy + kb*h + kc (see bn_bwd_finalize_kernel), gy and h bf16
struct BGradH {
  const bfraw *GY, *H;
  const float *ka, *kb, *kc;
  int ld, K;
  static constexpr int NC = 3, DEPTH = 4;
  struct Row { long o; bool ok; };
  struct Raw { u32x4 gy, h; };
  __device__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < Kp; i += NTB) {
      const bool ok = i < K;
      kcs[i] = ok ? ka[i] : 0.f; kcs[Kp + i] = ok ? kb[i] : 0.f; kcs[2 * Kp + i] = ok ? kc[i] : 0.f;
    }
  }
  __device__ Row row(long r, int, int, bool ok) const { return Row{r * ld, ok}; }
  __device__ void load(Raw& q, const Row& w, int k) const {
    const long o = w.o + (k <= ld - 8 ? k : ld - 8);
    q.gy = *reinterpret_cast<const u32x4*>(GY + o);
    q.h = *reinterpret_cast<const u32x4*>(H + o);
  }
  __device__ void fin(const Row& w, const float* kcs, int Kp, int k, const Raw& q, float* o) const {
    float gy[8], h[8];
    unpack8(q.gy, gy);
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = __fmaf_rn(kcs[k + j], gy[j], __fmaf_rn(kcs[Kp + k + j], h[j], kcs[2 * Kp + k + j]));
  }
};
struct BGradH3 {  // gy3[r,o] = f * (g[r,:] . W4[:,o]) * (y3 > 0) regenerated from the 3-channel output gradient, then gh3
  const float *G, *W4;
  const bfraw* H;
  const float *s, *t, *ka, *kb, *kc;
  float f;
  int ld, K;
  static constexpr int NC = 8, DEPTH = 4;
  struct Row { const bfraw* p; float g0, g1, g2; bool ok; };
  struct Raw { u32x4 h; };
  // constants interleaved per channel, two 16-byte reads per element: {s, t, kb, kc} {ka*w0, ka*w1, ka*w2, -}
  //   gh3 = (s*h + t > 0 ? g . (ka*w) : 0) + kb*h + kc
  __device__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < Kp; i += NTB) {
      const bool ok = i < K;
      const float a = ok ? ka[i] : 0.f;
      float* q = kcs + (size_t)i * 8;
      q[0] = ok ? s[i] : 0.f; q[1] = ok ? t[i] : 0.f; q[2] = ok ? kb[i] : 0.f; q[3] = ok ? kc[i] : 0.f;
      q[4] = ok ? a * W4[i] : 0.f; q[5] = ok ? a * W4[K + i] : 0.f; q[6] = ok ? a * W4[2 * K + i] : 0.f; q[7] = 0.f;
    }
  }
  __device__ Row row(long r, int, int, bool ok) const {
    return Row{H + (size_t)r * ld, f * G[r * 3], f * G[r * 3 + 1], f * G[r * 3 + 2], ok};
  }
  __device__ void load(Raw& q, const Row& w, int k) const { q.h = *reinterpret_cast<const u32x4*>(w.p + (k <= ld - 8 ? k : ld - 8)); }
  __device__ void fin(const Row& w, const float* kcs, int, int k, const Raw& q, float* o) const {
    float h[8];
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 c0 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8 + 4);
      const float gy = __fmaf_rn(c0.x, h[j], c0.y) > 0.f ? (w.g0 * c1.x + w.g1 * c1.y + w.g2 * c1.z) : 0.f;
      o[j] = gy + __fmaf_rn(c0.z, h[j], c0.w);
    }
  }
};

// ------------------------------------------------------------------------------------------------ epilogues (rows GEMM)
// Accumulator layout of v_mfma_f32_32x32x16_bf16: lane l holds column (l & 31) of the 32x32 tile, rows
// (reg & 3) + 8 (reg >> 2) + 4 (l >> 5): registers 2p and 2p+1 are ADJACENT rows, lanes l and l^1 adjacent columns.  A bf16
// pair store therefore exchanges one value with lane l^1 (DPP): the even lane writes (row 2p: columns c, c+1), the odd lane
// (row 2p+1: columns c-1, c) - 4-byte stores, 64 contiguous bytes per 16 lanes.  Pair loads mirror it.
struct EpiCtx { int blk, lane, wm, wn, bn0; };

__device__ __forceinline__ unsigned pair_exchange(unsigned pk, bool odd) {
  // pk = (own value of row 2p, own value of row 2p+1) for this lane's column -> the word this lane stores:
  // even: (row 2p: own, neighbour's)   odd: (row 2p+1: neighbour's, own)
  const unsigned recv = lane_swap1(odd ? (pk & 0xffffu) : (pk >> 16));
  return odd ? ((pk & 0xffff0000u) | recv) : ((pk & 0xffffu) | (recv << 16));
}
__device__ __forceinline__ void pair_unexchange(unsigned w, bool odd, float& v0, float& v1) {
  // w = word loaded by this lane (even: row 2p, columns c,c+1; odd: row 2p+1, columns c-1,c) -> own column's rows 2p, 2p+1
  const unsigned recv = lane_swap1(odd ? (w & 0xffffu) : (w >> 16));
  v0 = odd ? __uint_as_float(recv << 16) : bf_lo(w);
  v1 = odd ? bf_hi(w) : __uint_as_float(recv << 16);
}

// moments / sums of one column held by (two lane halves) x (four M-waves) -> dst[(blk*Nc + col)*2 + {0,1}] (fp64), fixed order
template <int WN>
__device__ __forceinline__ void reduce_cols(double (&d1)[WN], double (&d2)[WN], const EpiCtx& c, int Nc, double* __restrict__ dst, char* smem) {
  const int cl = c.bn0 + c.wn * 32 * WN + (c.lane & 31);
#pragma unroll
  for (int j = 0; j < WN; ++j) { d1[j] += __shfl_xor(d1[j], 32, 64); d2[j] += __shfl_xor(d2[j], 32, 64); }
  double* red = reinterpret_cast<double*>(smem);  // [3 waves][2 wn][WN][32 lanes][2]: the tiles are dead after the last barrier
  __syncthreads();
  if (c.wm > 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < WN; ++j) { double* q = red + (((((c.wm - 1) * 2 + c.wn) * WN + j) * 32) + c.lane) * 2; q[0] = d1[j]; q[1] = d2[j]; }
  }
  __syncthreads();
  if (c.wm == 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      if (cl + 32 * j < Nc) {
        double a = d1[j], b = d2[j];
#pragma unroll
        for (int w = 0; w < 3; ++w) { const double* q = red + ((((w * 2 + c.wn) * WN + j) * 32) + c.lane) * 2; a += q[0]; b += q[1]; }
        double* o = dst + ((size_t)c.blk * Nc + cl + 32 * j) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
  }
}

struct EpiStoreB {  // C[r,n] = bf16(acc + bias[n]); fp64 column moments (sum, sum of squares of the STORED values) per row block
  bfraw* C;
  const float* bias;
  double* moments;  // [row_blocks][Nc][2] or null
  int ldc, Nc;
  template <int WN>
  __device__ __forceinline__ void finish(const f32x16 (&acc)[WN], const EpiCtx& c, const RowGeo& geo, char* smem) const {
    const int cl = c.bn0 + c.wn * 32 * WN + (c.lane & 31);
    const bool odd = c.lane & 1;
    float bv[WN];
    float s1[WN], s2[WN];  // 16 values per lane in fp32, fp64 across lanes / waves / blocks
#pragma unroll
    for (int j = 0; j < WN; ++j) { bv[j] = (bias && cl + 32 * j < Nc) ? bias[cl + 32 * j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int i0 = c.wm * 32 + acc_row(2 * p, c.lane);
      int b, n; long r0; bool ok0;
      geo.map(c.blk, i0, b, n, r0, ok0);
      const bool ok1 = ok0 && (geo.tiled ? n + 1 < geo.N : r0 + 1 < geo.R);  // row i0 + 1: the next vertex of the same sample
      bfraw* dst = C + (size_t)(odd ? r0 + 1 : r0) * ldc + (cl & ~1);
      const bool okw = odd ? ok1 : ok0;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bool cok = cl + 32 * j < Nc;
        const unsigned pk = pack_bf16(cok ? acc[j][2 * p] + bv[j] : 0.f, cok ? acc[j][2 * p + 1] + bv[j] : 0.f);
        const float v0 = ok0 ? bf_lo(pk) : 0.f, v1 = ok1 ? bf_hi(pk) : 0.f;
        s1[j] += v0 + v1;
        s2[j] = __fmaf_rn(v0, v0, __fmaf_rn(v1, v1, s2[j]));
        const unsigned w = pair_exchange(pk, odd);
        if (okw && (cl & ~1) + 32 * j < ldc) *reinterpret_cast<unsigned*>(dst + 32 * j) = w;
      }
    }
    if (moments) {
      double d1[WN], d2[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) { d1[j] = (double)s1[j]; d2[j] = (double)s2[j]; }
      reduce_cols<WN>(d1, d2, c, Nc, moments, smem);
    }
  }
};

struct EpiMaskB {  // C = bf16(acc * (y > 0)), y = s*H+t; column sums S1 = sum C, S2 = sum C * xhat, xhat = (H - mean) * rstd
  bfraw* C;
  const bfraw* H;  // same pitch as C
  double* sums;    // [row_blocks][Nc][2]
  const float *s, *t, *mean, *rstd;
  int ldc, Nc;
  template <int WN>
  __device__ __forceinline__ void finish(const f32x16 (&acc)[WN], const EpiCtx& c, const RowGeo& geo, char* smem) const {
    const int cl = c.bn0 + c.wn * 32 * WN + (c.lane & 31);
    const bool odd = c.lane & 1;
    float cs[WN], ct[WN], cm[WN], cr[WN];
    float p1[WN], p2[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const bool cok = cl + 32 * j < Nc;
      const int cc = cok ? cl + 32 * j : 0;
      cs[j] = cok ? s[cc] : 0.f; ct[j] = cok ? t[cc] : 0.f; cm[j] = cok ? mean[cc] : 0.f; cr[j] = cok ? rstd[cc] : 0.f;
      p1[j] = 0.f; p2[j] = 0.f;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int i0 = c.wm * 32 + acc_row(2 * p, c.lane);
      int b, n; long r0; bool ok0;
      geo.map(c.blk, i0, b, n, r0, ok0);
      const bool ok1 = ok0 && r0 + 1 < geo.R;
      const size_t ro = (size_t)(odd ? r0 + 1 : r0) * ldc + (cl & ~1);
      const bool okw = odd ? ok1 : ok0;
      unsigned hw[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) hw[j] = (okw && (cl & ~1) + 32 * j < ldc) ? *reinterpret_cast<const unsigned*>(H + ro + 32 * j) : 0u;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bool cok = cl + 32 * j < Nc;
        float h0, h1;
        pair_unexchange(hw[j], odd, h0, h1);
        const float g0 = (cok && ok0 && __fmaf_rn(cs[j], h0, ct[j]) > 0.f) ? acc[j][2 * p] : 0.f;
        const float g1 = (cok && ok1 && __fmaf_rn(cs[j], h1, ct[j]) > 0.f) ? acc[j][2 * p + 1] : 0.f;
        const unsigned pk = pack_bf16(g0, g1);
        const float v0 = bf_lo(pk), v1 = bf_hi(pk);
        p1[j] += v0 + v1;
        p2[j] = __fmaf_rn(v0, (h0 - cm[j]) * cr[j], __fmaf_rn(v1, (h1 - cm[j]) * cr[j], p2[j]));
        const unsigned w = pair_exchange(pk, odd);
        if (okw && (cl & ~1) + 32 * j < ldc) *reinterpret_cast<unsigned*>(C + ro + 32 * j) = w;
      }
    }
    double d1[WN], d2[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) { d1[j] = (double)p1[j]; d2[j] = (double)p2[j]; }
    reduce_cols<WN>(d1, d2, c, Nc, sums, smem);
  }
};

// dA(gy1) without gy1: rows tiled (8 samples x 16 vertices).  gy1 = acc * (y1 > 0), y1 = gamma*xhat1 + beta, xhat1 = Gx[n]+Fx[b].
//   Pp[ng][b][c] = sum over the block's 16 vertices   (each M-wave owns two samples: in-wave)
//   Qp[bg][n][c] = sum over the block's 8 samples      (two per wave in-lane, the four M-waves through LDS in wave order)
struct EpiL1B {
  float *Pp, *Qp;  // [NG][B][ld], [B/8][N][ld]
  const float *Gx, *Fx, *gamma, *beta;
  int ld, Nc;
  template <int WN>
  __device__ __forceinline__ void finish(const f32x16 (&acc)[WN], const EpiCtx& c, const RowGeo& geo, char* smem) const {
    const int cl = c.bn0 + c.wn * 32 * WN + (c.lane & 31), h = c.lane >> 5;
    const int ng = c.blk % geo.NG, bg = c.blk / geo.NG;
    const int b0 = bg * 8 + c.wm * 2;
    const bool bok0 = b0 < geo.B, bok1 = b0 + 1 < geo.B;
    float q[WN][8];
    float pa[WN], pb[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const bool cok = cl + 32 * j < Nc;
      const int cc = cok ? cl + 32 * j : 0;
      const float ga = cok ? gamma[cc] : 0.f, be = cok ? beta[cc] : 0.f;
      const float fa = (cok && bok0) ? Fx[(size_t)b0 * ld + cc] : 0.f, fb = (cok && bok1) ? Fx[(size_t)(b0 + 1) * ld + cc] : 0.f;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // register r: vertex v of sample b0, register r+8: the same vertex of sample b0+1
        const int v = (r & 3) + 8 * (r >> 2) + 4 * h, n = ng * 16 + v;
        const bool nok = n < geo.N && cok;
        const float gx = nok ? Gx[(size_t)n * ld + cc] : 0.f;
        const float ya = __fmaf_rn(ga, gx + fa, be), yb = __fmaf_rn(ga, gx + fb, be);
        const float ga_ = (nok && bok0 && ya > 0.f) ? acc[j][r] : 0.f;
        const float gb_ = (nok && bok1 && yb > 0.f) ? acc[j][r + 8] : 0.f;
        q[j][r] = ga_ + gb_;
        sa += ga_;
        sb += gb_;
      }
      pa[j] = sa + __shfl_xor(sa, 32, 64);
      pb[j] = sb + __shfl_xor(sb, 32, 64);
    }
    // P: lane half 0 writes sample b0, half 1 sample b0 + 1 (both halves hold both totals)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int col = cl + 32 * j;
      if (col < Nc && (h ? bok1 : bok0)) Pp[((size_t)ng * geo.B + b0 + h) * ld + col] = h ? pb[j] : pa[j];
    }
    // Q: M-waves 1..3 park their partials in LDS, wave 0 adds them in wave order and stores
    float* red = reinterpret_cast<float*>(smem);  // [3][2 wn][WN][8][64]
    __syncthreads();
    if (c.wm > 0) {
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int r = 0; r < 8; ++r) red[((((c.wm - 1) * 2 + c.wn) * WN + j) * 8 + r) * 64 + c.lane] = q[j][r];
    }
    __syncthreads();
    if (c.wm == 0) {
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = cl + 32 * j;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float a = q[j][r];
#pragma unroll
          for (int w = 0; w < 3; ++w) a += red[(((w * 2 + c.wn) * WN + j) * 8 + r) * 64 + c.lane];
          const int n = ng * 16 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (col < Nc && n < geo.N) Qp[((size_t)bg * geo.N + n) * ld + col] = a;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------ rows GEMM
// C[rows x Nc] = Aop[rows x K] * Wb^T, Wb = bf16 [Nc][Kp] image.  1-D grid of row blocks x column blocks (of 64*WN); the
// column blocks of a row block are neighbours on one XCD (xcd_virtual_id).
template <class AOp, class Epi, int WN>
__global__ __launch_bounds__(NTB) void rows_bf16_kernel(AOp aop, const bfraw* __restrict__ Wb, int Kp, int Nc, Epi epi, RowGeo geo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BNW = 64 * WN;
  constexpr int BCH = (256 * WN + NTB - 1) / NTB;  // 16-byte B chunks per thread and tile
  constexpr int DA = AOp::DEPTH, DB = 2, UN = DA % 2 == 0 ? DA : 2 * DA;  // unroll = lcm(DA, DB)
  bfraw* As = reinterpret_cast<bfraw*>(smem);                           // [2][BM][LP]
  bfraw* Bs = As + 2 * BM * LP;                                        // [2][BNW][LP]
  float* kcs = reinterpret_cast<float*>(Bs + 2 * BNW * LP);            // [AOp::NC][Kp]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int ncol = (Nc + BNW - 1) / BNW, vid = xcd_virtual_id(blockIdx.x, gridDim.x), rblk = vid / ncol;
  const int bn0 = (vid - rblk * ncol) * BNW;
  const int arow = tid >> 2, kq = (tid & 3) * 8;  // A staging: one row, 8 consecutive k
  typename AOp::Row row;
  {
    int b, n; long r; bool ok;
    geo.map(rblk, arow, b, n, r, ok);
    row = aop.row(r, b, n, ok);
  }
  aop.stage(kcs, Kp, tid);
  typename AOp::Raw qa[DA];
  u32x4 qb[DB][BCH];
  const int nk = Kp / BK;
  // B chunk c of a tile: row n = c / 4, 8 k at (c % 4) * 8.  Every thread loads BCH chunks (the surplus ones of the last
  // round re-read chunk 0 and are not written to LDS): no divergent load, so the compiler can count outstanding loads.
  const bfraw* bsrc[BCH];
#pragma unroll
  for (int j = 0; j < BCH; ++j) {
    const int c = tid + NTB * j, cc = c < 256 * WN ? c : 0, n = bn0 + (cc >> 2);
    bsrc[j] = Wb + (size_t)(n < Nc ? n : Nc - 1) * Kp + (cc & 3) * 8;
  }
  auto loadB = [&](u32x4* q, int k0) {
#pragma unroll
    for (int j = 0; j < BCH; ++j) q[j] = *reinterpret_cast<const u32x4*>(bsrc[j] + k0);
  };
  auto stash = [&](int buf, int kt, const typename AOp::Raw& ra, const u32x4* rb) {
    float v[8];
    aop.fin(row, kcs, Kp, kt * BK + kq, ra, v);
    u32x4 pk = pack8(v);
    if (!row.ok) pk = u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(As + ((size_t)buf * BM + arow) * LP + kq) = pk;
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const int c = tid + NTB * j;
      if (c < 256 * WN) *reinterpret_cast<u32x4*>(Bs + ((size_t)buf * BNW + (c >> 2)) * LP + (c & 3) * 8) = rb[j];
    }
  };
  const int fk = (lane >> 5) * 8, fr = lane & 31;
  f32x16 acc[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  auto mma = [&](int cur) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + ((size_t)cur * BM + wm * 32 + fr) * LP + ks * 16 + fk);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + ((size_t)cur * BNW + wn * 32 * WN + j * 32 + fr) * LP + ks * 16 + fk);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
  };
  // prologue: tiles 0 .. DA-1 (A) / 0 .. DB-1 (B) requested; tile 0 transformed into LDS buffer 0
#pragma unroll
  for (int d = 0; d < DB; ++d)
    if (d < nk) loadB(qb[d], d * BK);
#pragma unroll
  for (int d = 0; d < DA; ++d)
    if (d < nk) aop.load(qa[d], row, d * BK + kq);
  __syncthreads();  // constants staged
  stash(0, 0, qa[0], qb[0]);
  if (DB < nk) loadB(qb[0], DB * BK);
  if (DA < nk) aop.load(qa[0], row, DA * BK + kq);
  __syncthreads();
  // Queue slots are compile-time (tile t lives in slot t % DEPTH), so the loop is unrolled by UN = lcm(DA, DB).
  // MAIN phase: whole groups of UN iterations in which every tile index touched is valid - no branch between a load and its
  // use, hence the compiler waits with vmcnt(#younger loads) and DEPTH-1 tiles really stay in flight (with a conditional load
  // anywhere in the loop it falls back to vmcnt(0): one exposed Infinity-Cache / HBM round trip per k-tile).  Within an
  // iteration the B tile (weights, L2) is requested BEFORE the A tile: the counter is in-order, and the next iteration's wait
  // for that B tile must not also drain the deep A request behind it.
  // (Tried and dropped, profiles/r02_kernels.md: running the MFMA and transform phases in opposite order in the two waves of a
  // SIMD - spills; interleaving them inside a wave with sched_group_barrier - no change in time.)
  const int main_end = nk - 1 - DA > 0 ? ((nk - 1 - DA) / UN) * UN : 0;
  int kt0 = 0;
  for (; kt0 < main_end; kt0 += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kt = kt0 + u, cur = u & 1;  // kt0 is even
      const int sa = (u + 1) % DA, sb = (u + 1) % DB;
      mma(cur);
      stash(cur ^ 1, kt + 1, qa[sa], qb[sb]);
      loadB(qb[sb], (kt + 1 + DB) * BK);
      aop.load(qa[sa], row, (kt + 1 + DA) * BK + kq);
      __syncthreads();
    }
  }
  // TAIL phase (at most UN + DA + 1 iterations; for the production shapes it requests nothing new): guarded
  for (; kt0 < nk; kt0 += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        const int cur = kt & 1;
        const int sa = (u + 1) % DA, sb = (u + 1) % DB;
        mma(cur);
        if (kt + 1 < nk) {
          stash(cur ^ 1, kt + 1, qa[sa], qb[sb]);
          if (kt + 1 + DB < nk) loadB(qb[sb], (kt + 1 + DB) * BK);
          if (kt + 1 + DA < nk) aop.load(qa[sa], row, (kt + 1 + DA) * BK + kq);
        }
        __syncthreads();
      }
    }
  }
  const EpiCtx ctx{rblk, lane, wm, wn, bn0};
  epi.template finish<WN>(acc, ctx, geo, smem);
}

// (The first-generation weight-gradient GEMM of this file - tn_bf16_kernel and its T* operand generators: (8 rows) x (2 channels)
// strips, k-major LDS images written through a VALU transpose - was removed in round 6: decoder_tn2.h covers every width with
// WN = 2 or 5 tiles per wave.  git history: "bf16 decoder" of round 2.)
