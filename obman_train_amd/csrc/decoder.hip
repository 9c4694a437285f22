// K6 - AtlasNet PointGenCon decoder (atlasutils.py:42-75 + atlasbranch.py:117-132), gfx950, fp32 MFMA.
//
//   x[b,:,n] = (grid[n] (3) | feature[b] (C1-3))          -> never materialised (84 MB at bs 64)
//   h1 = W1 x + b1 = G[n,:] + F[b,:]                       -> two small factors, BN-1 batch statistics in
//                                                             closed form (mean and variance over the product
//                                                             set B x N are sums of the two factors' moments)
//   a1 = relu(bn1(h1))                                      -> generated ON THE FLY inside GEMM-2's A-tile loader
//   h2 = W2 a1 + b2   (R x 515 x 257, R = B*N rows)         -> v_mfma_f32_32x32x2_f32, fp64 column moments in the
//   a2 = relu(bn2(h2))                                         epilogue (BN-2 needs batch statistics)
//   h3 = W3 a2 + b3   (R x 257 x 128)                       -> A-tile loader applies BN-2 + ReLU on load
//   out = 200 * (W4 relu(bn3(h3)) + b4)   (K=128, N=3)      -> VALU row kernel
//
// Backward mirrors this: every BN/ReLU backward is folded into the operand loaders of the data-gradient
// GEMMs (rows x Cout x Cin) and weight-gradient GEMMs (contraction over the R rows, split-K, fixed-order
// reduction => deterministic); only h2, h3 (forward) and the masked gradients gy2, gy1 are materialised.
// fp32 in / fp32 accumulate MFMA (exact fp32 fma chains): 157 TF peak = the fp32 vector rate, but it
// leaves the VALU free for the fused loaders/epilogues.  Tile: 128 x 64 x 32, 4 waves (2x2), each wave
// 64 x 32 = two 32x32 accumulators; LDS tiles k-major with +1 padding (conflict-free b32 reads/writes).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace dec {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 64, BK = 32, NT = 256;

__host__ __device__ inline int pad16(int c) { return (c + 15) / 16 * 16; }

// ------------------------------------------------------------------------------------------------ A operands
// Every operand is generated in two phases so that the global loads of tile t+1 are in flight while the MFMAs
// of tile t run: raw() issues branch-free loads (indices clamped, never predicated), fin() applies the fused
// BN / ReLU / BN-backward arithmetic AFTER the MFMA loop.  kc() fetches the per-channel constants.
__device__ __forceinline__ void split_row(int r, int N, int bhint, int& b, int& n) {
  b = bhint;
  n = r - b * N;
  while (n >= N) { n -= N; ++b; }
}

struct AGridFeat {  // a1[r,k] = relu(gamma[k] * (Gx[n,k] + Fx[b,k]) + beta[k]),  r = b*N + n  (Gx, Fx are x-hat factors)
  const float *Gx, *Fx, *gamma, *beta;
  int N, ld, R, K;
  int ps;  // per-sample grid: Gx is [R,ld], indexed by the row instead of the template vertex
  struct Row { int g, f; bool ok; };
  struct KC { float ga, be; bool ok; };
  struct Raw { float a, b; };
  __device__ Row row(int r, int bhint) const {
    const bool ok = r < R;
    int b, n;
    split_row(ok ? r : R - 1, N, ok ? bhint : (R - 1) / N, b, n);
    return Row{(ps ? (ok ? r : R - 1) : n) * ld, b * ld, ok};
  }
  static constexpr int NC = 2;  // per-channel constants: kcv(j, k), j < NC (the rows GEMM stages them per k-tile in LDS)
  __device__ float kcv(int j, int k) const { return k < K ? (j == 0 ? gamma : beta)[k] : 0.f; }
  __device__ static KC make(const float* v, bool ok) { return KC{v[0], v[1], ok}; }
  __device__ KC kc(int k) const { const bool ok = k < K; const int c = ok ? k : 0; return KC{gamma[c], beta[c], ok}; }
  __device__ Raw raw(const Row& w, int k) const { const int c = k < K ? k : 0; return Raw{Gx[w.g + c], Fx[w.f + c]}; }
  __device__ void raw4(const Row& w, int k, Raw* o) const {  // k % 4 == 0; columns >= K are masked by kc().ok
    const int c = k <= ld - 4 ? k : ld - 4;
    const float4 a = *reinterpret_cast<const float4*>(Gx + w.g + c), b = *reinterpret_cast<const float4*>(Fx + w.f + c);
    o[0] = Raw{a.x, b.x}; o[1] = Raw{a.y, b.y}; o[2] = Raw{a.z, b.z}; o[3] = Raw{a.w, b.w};
  }
  __device__ float fin(const Row& w, const KC& c, const Raw& x) const {
    const float v = fmaxf(__fmaf_rn(c.ga, x.a + x.b, c.be), 0.f);
    return (w.ok && c.ok) ? v : 0.f;
  }
};
struct ABnRelu {  // a[r,k] = relu(s[k] * H[r,k] + t[k])
  const float *H, *s, *t;
  int ld, R, K;
  struct Row { long o; bool ok; };
  struct KC { float s, t; bool ok; };
  struct Raw { float h; };
  __device__ Row row(int r, int) const { const bool ok = r < R; return Row{(long)(ok ? r : 0) * ld, ok}; }
  static constexpr int NC = 2;
  __device__ float kcv(int j, int k) const { return k < K ? (j == 0 ? s : t)[k] : 0.f; }
  __device__ static KC make(const float* v, bool ok) { return KC{v[0], v[1], ok}; }
  __device__ KC kc(int k) const { const bool ok = k < K; const int c = ok ? k : 0; return KC{s[c], t[c], ok}; }
  __device__ Raw raw(const Row& w, int k) const { return Raw{H[w.o + (k < K ? k : 0)]}; }
  __device__ void raw4(const Row& w, int k, Raw* o) const {
    const float4 a = *reinterpret_cast<const float4*>(H + w.o + (k <= ld - 4 ? k : ld - 4));
    o[0] = Raw{a.x}; o[1] = Raw{a.y}; o[2] = Raw{a.z}; o[3] = Raw{a.w};
  }
  __device__ float fin(const Row& w, const KC& c, const Raw& x) const {
    const float v = fmaxf(__fmaf_rn(c.s, x.h, c.t), 0.f);
    return (w.ok && c.ok) ? v : 0.f;
  }
};
struct APlain {  // a[r,k] = X[r,k]
  const float* X;
  int ld, R, K;
  struct Row { long o; bool ok; };
  struct KC { bool ok; };
  struct Raw { float x; };
  __device__ Row row(int r, int) const { const bool ok = r < R; return Row{(long)(ok ? r : 0) * ld, ok}; }
  static constexpr int NC = 0;
  __device__ float kcv(int, int) const { return 0.f; }
  __device__ static KC make(const float*, bool ok) { return KC{ok}; }
  __device__ KC kc(int k) const { return KC{k < K}; }
  __device__ Raw raw(const Row& w, int k) const { return Raw{X[w.o + (k < K ? k : 0)]}; }
  __device__ void raw4(const Row& w, int k, Raw* o) const {  // rows of X need not be 16-byte aligned: scalar loads
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = raw(w, k + j);
  }
  __device__ float fin(const Row& w, const KC& c, const Raw& x) const { return (w.ok && c.ok) ? x.x : 0.f; }
};
// d(loss)/d(h) of a BatchNorm'd layer from the masked upstream gradient gy = d/d(y) * (y>0):
//   train: gh = k1 * (gy - k2 - xhat * k3),  k1 = gamma*rstd, k2 = mean_r(gy), k3 = mean_r(gy*xhat),  xhat = (h - mean) * rstd
//   eval : gh = k1 * gy
// bn_bwd_finalize_kernel folds this per channel into the affine form gh = ka * gy + kb * h + kc
//   ka = k1,  kb = -k1 * rstd * k3,  kc = k1 * (mean * rstd * k3 - k2)        (kb = kc = 0 in eval)
// so an operand element costs two FMAs and three per-channel constants.
struct AGradH {  // gy materialised
  const float *GY, *H, *ka, *kb, *kc_;
  int ld, R, K;
  struct Row { long o; bool ok; };
  struct KC { float a, b, c; bool ok; };
  struct Raw { float gy, h; };
  __device__ Row row(int r, int) const { const bool ok = r < R; return Row{(long)(ok ? r : 0) * ld, ok}; }
  static constexpr int NC = 3;
  __device__ float kcv(int j, int k) const { return k < K ? (j == 0 ? ka : (j == 1 ? kb : kc_))[k] : 0.f; }
  __device__ static KC make(const float* v, bool ok) { return KC{v[0], v[1], v[2], ok}; }
  __device__ KC kc(int k) const {
    const bool ok = k < K; const int c = ok ? k : 0;
    return KC{ka[c], kb[c], kc_[c], ok};
  }
  __device__ Raw raw(const Row& w, int k) const { const int c = k < K ? k : 0; return Raw{GY[w.o + c], H[w.o + c]}; }
  __device__ void raw4(const Row& w, int k, Raw* o) const {
    const int c = k <= ld - 4 ? k : ld - 4;
    const float4 a = *reinterpret_cast<const float4*>(GY + w.o + c), b = *reinterpret_cast<const float4*>(H + w.o + c);
    o[0] = Raw{a.x, b.x}; o[1] = Raw{a.y, b.y}; o[2] = Raw{a.z, b.z}; o[3] = Raw{a.w, b.w};
  }
  __device__ float fin(const Row& w, const KC& c, const Raw& x) const {
    const float v = __fmaf_rn(c.a, x.gy, __fmaf_rn(c.b, x.h, c.c));
    return (w.ok && c.ok) ? v : 0.f;
  }
};
struct AGradH3 {  // gy3 regenerated from the 3-channel output gradient: gy3[r,o] = f*(g[r,:].W4[:,o]) * (y3 > 0)
  const float *G, *W4, *H, *s, *t, *ka, *kb, *kc_;
  float f;
  int ld, R, K;
  struct Row { long o; float g0, g1, g2; bool ok; };
  struct KC { float s, t, b, c, w0, w1, w2; bool ok; };  // w_j = ka * W4[j]: gh3 = (y3 > 0 ? g . w : 0) + (kb * h + kc)
  struct Raw { float h; };
  __device__ Row row(int r, int) const {
    const bool ok = r < R;
    const long rr = ok ? r : 0;
    return Row{rr * ld, f * G[rr * 3], f * G[rr * 3 + 1], f * G[rr * 3 + 2], ok};
  }
  static constexpr int NC = 7;
  __device__ float kcv(int j, int k) const {
    if (k >= K) return 0.f;
    switch (j) {
      case 0: return s[k];
      case 1: return t[k];
      case 2: return kb[k];
      case 3: return kc_[k];
      default: return ka[k] * W4[(j - 4) * K + k];
    }
  }
  __device__ static KC make(const float* v, bool ok) { return KC{v[0], v[1], v[2], v[3], v[4], v[5], v[6], ok}; }
  __device__ KC kc(int k) const {
    const bool ok = k < K; const int c = ok ? k : 0;
    const float a = ka[c];
    return KC{s[c], t[c], kb[c], kc_[c], a * W4[c], a * W4[K + c], a * W4[2 * K + c], ok};
  }
  __device__ Raw raw(const Row& w, int k) const { return Raw{H[w.o + (k < K ? k : 0)]}; }
  __device__ void raw4(const Row& w, int k, Raw* o) const {
    const float4 a = *reinterpret_cast<const float4*>(H + w.o + (k <= ld - 4 ? k : ld - 4));
    o[0] = Raw{a.x}; o[1] = Raw{a.y}; o[2] = Raw{a.z}; o[3] = Raw{a.w};
  }
  __device__ float fin(const Row& w, const KC& c, const Raw& x) const {
    const float gy = __fmaf_rn(c.s, x.h, c.t) > 0.f ? (w.g0 * c.w0 + w.g1 * c.w1 + w.g2 * c.w2) : 0.f;
    const float v = gy + __fmaf_rn(c.b, x.h, c.c);
    return (w.ok && c.ok) ? v : 0.f;
  }
};

// ------------------------------------------------------------------------------------------------ epilogues
struct EpiStore {  // C[r, n] = acc + bias[n]; optional fp64 column moments (sum, sum of squares) per row-block
  float* C;
  const float* bias;
  double* moments;  // [row_blocks][mstride][2] or null
  int ldc, R, Nc;
  int mstride;      // channels per row block in `moments` (>= Nc)
};
struct EpiMaskStats {  // C = acc * (y > 0); per row-block column sums S1 = sum(C), S2 = sum(C * xhat)
  float* C;
  double* sums;  // [row_blocks][Nc][2]
  int ldc, R, Nc;
  int mode;      // 0: y = s*H+t, xhat = (H-mean)*rstd ; 1: xhat = Gx[n]+Fx[b], y = gamma*xhat+beta
  const float *H, *s, *t, *mean, *rstd;   // mode 0 (ld = ldc)
  const float *Gx, *Fx, *gamma, *beta;    // mode 1 (ld = ldc)
  int N;
  int ps;        // mode 1 with a per-sample grid: Gx indexed by the row
  int sstride;   // channels per row block in `sums` (>= Nc)
};

template <class A> __device__ __forceinline__ int rows_N(const A&) { return 1 << 30; }
__device__ __forceinline__ int rows_N(const AGridFeat& a) { return a.N; }

__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  Blocks that
// share an operand (the column blocks of one row block; the output tiles of one row chunk) get consecutive slots on ONE XCD,
// so the shared operand is fetched from HBM once and re-read from that L2 instead of once per block from HBM / MALL.
// id -> (group, member): group = (slot / members) * 8 + xcd, member = slot % members; grids are padded to 8 * ceil(groups / 8).
struct XcdOrder { int group, member; };
__device__ __forceinline__ XcdOrder xcd_order(int id, int members, int aware = 1) {
  if (!aware) return XcdOrder{id / members, id % members};  // OBMAN_DEC_XCD=0: plain group-major order (A/B knob)
  const int xcd = id & 7, slot = id >> 3;
  return XcdOrder{(slot / members) * 8 + xcd, slot % members};
}
inline int xcd_aware() {
  static const int v = [] { const char* e = getenv("OBMAN_DEC_XCD"); return e ? atoi(e) : 1; }();
  return v;
}
inline unsigned xcd_grid(long groups, int members) { return (unsigned)(((groups + 7) / 8) * 8 * members); }

// LDS tiles (k-major, +1 pad): As[buf][k][m], Bs[buf][k][n]; Sa / Sb: one extra operand column each (the "side" products)
struct Tiles {
  float As[2][BK][BM + 1];
  float Bs[2][BK][BN + 1];
  float Sa[2][BK], Sb[2][BK];
  float Kc[2][8][BK];  // the A operand's per-channel constants of a k-tile (rows GEMM): [constant][k]
};

// The weight-gradient kernel stages its tiles in load order (the contraction index is the row): a thread writes four consecutive
// channels of a row with ONE 16-byte store; rows are 16-byte aligned (pad 4).  With the +1 pad and four 4-byte stores per thread
// half of the kernel's LDS cycles were bank conflicts (PMC SQ_LDS_BANK_CONFLICT 1.6e7 of 3.3e7 active, profiles/r02_kernels.md).
struct TilesT {
  float As[2][BK][BM + 4];
  float Bs[2][BK][BN + 4];
  float Sa[2][BK], Sb[2][BK];
};

// 257 = 4 x 64 + 1 and 515 = 8 x 64 + 3: padding the output to whole 64-column tiles would spend a fifth (a ninth) of the
// blocks - and at 64 x 642 points a whole third block round - on one (three) live columns.  Instead the tiled dimension is cut
// to its whole tiles and each of the first `side` members of a group computes ONE leftover column beside its MFMAs, on the
// VALU, from the operand tile that is in LDS anyway: 4096 FMAs per k-tile and block against 262 144 on the matrix pipe.
struct TileSplit { int tiles, side; };
__host__ __device__ inline TileSplit tile_split(int n, int tile) {
  const int full = n / tile, tail = n - full * tile;
  if (full >= 1 && tail >= 1 && tail <= 3 && tail <= full) return TileSplit{full, tail};
  return TileSplit{(n + tile - 1) / tile, 0};
}

// C[R x Nc] = Aop[R x K] * B, B given either as W[n][k] (B_NK, ldb = row stride of W) or W[k][n] (B_KN).
template <class AOp, bool B_NK, class Epi, int DBG = 0>  // DBG: ablation builds only (-DOBMAN_ABLATION), wrong results by design
__global__ __launch_bounds__(NT, 3) void gemm_rows_kernel(AOp aop, const float* __restrict__ Bw, int ldb, int K, int Nc, Epi epi, int variant) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Tiles& T = *reinterpret_cast<Tiles*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const TileSplit cs = tile_split(Nc, BN);
  const XcdOrder bo = xcd_order(blockIdx.x, cs.tiles, !(variant & 8));  // group = row block, member = column block
  if ((long)bo.group * BM >= aop.R) return;
  const int bm0 = bo.group * BM, bn0 = bo.member * BN;
  const bool side = bo.member < cs.side;       // this block also owns output column sc (block-uniform)
  const int sc = cs.tiles * BN + (side ? bo.member : 0);

  // A staging: thread = 4 consecutive k (one 16-byte load per source array) x 4 rows (rm, rm+32, rm+64, rm+96)
  const int kq = (tid & 7) * 4, rm = tid >> 3;
  typename AOp::Row rows[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = bm0 + rm + 32 * p;
    rows[p] = aop.row(r, (r < aop.R ? r : aop.R - 1) / rows_N(aop));
  }
  typename AOp::Raw ra[4][4];
  typename AOp::KC kcur[4];
  float rb[8], rws = 0.f;
  const int ak = tid & 31;  // B staging keeps the scalar mapping (weight rows are not 16-byte aligned)
  // Per-channel constants of the A operand (BN scale / shift, folded BN-backward coefficients, layer-4 weights: up to 7 per
  // channel) travel through LDS: thread (constant tid / 32, channel tid % 32) loads ONE value per k-tile, two tiles ahead, and
  // every thread reads the NC x 4 values of its four channels with NC 16-byte LDS reads when it transforms the tile - instead of
  // 4 x NC global loads per thread and k-tile (32 for the gy2 GEMM, against its 32 MFMAs).
  constexpr int NC = AOp::NC;
  const bool kc_stager = tid < NC * BK;
  const int kc_j = tid / BK, kc_k = tid % BK;
  float kc_reg = 0.f;
  auto fetch_kc = [&](int k0) { if (NC > 0 && kc_stager) kc_reg = aop.kcv(kc_j, k0 + kc_k); };
  auto put_kc = [&](int buf) { if (NC > 0 && kc_stager) T.Kc[buf][kc_j][kc_k] = kc_reg; };
  auto read_kc = [&](int buf, int k0) {  // -> kcur[0..3] for the channels k0 + kq .. + 3
    float v[4][NC > 0 ? NC : 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 q = *reinterpret_cast<const float4*>(&T.Kc[buf][c][kq]);
      v[0][c] = q.x; v[1][c] = q.y; v[2][c] = q.z; v[3][c] = q.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) kcur[j] = AOp::make(v[j], k0 + kq + j < K);
  };
  auto fetch = [&](int k0) {  // loads only: nothing here consumes a loaded value
    if (side) {  // the side column's weights of this k-tile (every lane loads, lanes 0..31 of wave 0 stage them)
      const int k = k0 + ak < K ? k0 + ak : 0;
      rws = B_NK ? Bw[(size_t)sc * ldb + k] : Bw[(size_t)k * ldb + sc];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) aop.raw4(rows[p], k0 + kq, ra[p]);
    if (B_NK) {  // W[n][k]: consecutive lanes along k
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int n = bn0 + (tid >> 5) + 8 * p, k = k0 + ak;
        rb[p] = Bw[(size_t)(n < Nc ? n : 0) * ldb + (k < K ? k : 0)];
      }
    } else {  // W[k][n]: consecutive lanes along n
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int n = bn0 + (tid & 63), k = k0 + (tid >> 6) + 4 * p;
        rb[p] = Bw[(size_t)(k < K ? k : 0) * ldb + (n < Nc ? n : 0)];
      }
    }
  };
  const bool side_stage = side && tid < BK;
  auto stash_side = [&](int buf, int k0) {
    if (side_stage) T.Sb[buf][tid] = k0 + tid < K ? rws : 0.f;
  };
  auto stash_part = [&](int buf, int k0, int q) {  // q = 0..7: two A elements + one B element
    const int p = q >> 1, j0 = (q & 1) * 2;
    if (q == 0) {
      stash_side(buf, k0);
      read_kc(buf, k0);
    }
    if (DBG & 1) {  // no operand transform: the first loaded word as it is
      T.As[buf][kq + j0][rm + 32 * p] = *reinterpret_cast<const float*>(&ra[p][j0]);
      T.As[buf][kq + j0 + 1][rm + 32 * p] = *reinterpret_cast<const float*>(&ra[p][j0 + 1]);
    } else {
      T.As[buf][kq + j0][rm + 32 * p] = aop.fin(rows[p], kcur[j0], ra[p][j0]);
      T.As[buf][kq + j0 + 1][rm + 32 * p] = aop.fin(rows[p], kcur[j0 + 1], ra[p][j0 + 1]);
    }
    if (B_NK) {
      const int n = bn0 + (tid >> 5) + 8 * q, k = k0 + ak;
      T.Bs[buf][ak][(tid >> 5) + 8 * q] = (n < Nc && k < K) ? rb[q] : 0.f;
    } else {
      const int n = bn0 + (tid & 63), k = k0 + (tid >> 6) + 4 * q;
      T.Bs[buf][(tid >> 6) + 4 * q][tid & 63] = (n < Nc && k < K) ? rb[q] : 0.f;
    }
  };
  auto stash = [&](int buf, int k0) {
    stash_side(buf, k0);
    read_kc(buf, k0);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) T.As[buf][kq + j][rm + 32 * p] = aop.fin(rows[p], kcur[j], ra[p][j]);
    if (B_NK) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int n = bn0 + (tid >> 5) + 8 * p, k = k0 + ak;
        T.Bs[buf][ak][(tid >> 5) + 8 * p] = (n < Nc && k < K) ? rb[p] : 0.f;
      }
    } else {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int n = bn0 + (tid & 63), k = k0 + (tid >> 6) + 4 * p;
        T.Bs[buf][(tid >> 6) + 4 * p][tid & 63] = (n < Nc && k < K) ? rb[p] : 0.f;
      }
    }
  };

  f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
  const int nk = (K + BK - 1) / BK;
  // A wave whose 32 output columns all lie beyond Nc (the 257th / 515th column leaves 1 / 3 live columns in the last 64-wide
  // block) issues no MFMA and reads no fragment: it only helps staging.  The matrix pipe it would have burnt goes to the
  // other blocks of the CU.
  const bool live = bn0 + wn * 32 < Nc;
  const int arow = wm * 64 + (lane & 31), bcol = wn * 32 + (lane & 31), kh = lane >> 5;
  // side product: thread = (row srow, k half skh) of the block's 128 x 32 operand tile, one FMA per MFMA step
  const int srow = tid & (BM - 1), skh = (tid >> 7) * (BK / 2);
  float sacc = 0.f;
  fetch_kc(0);
  fetch(0);
  put_kc(0);
  fetch_kc(BK);
  __syncthreads();  // constants of tile 0 visible
  stash(0, 0);
  put_kc(1);        // constants of tile 1: read while tile 0 is multiplied
  __syncthreads();
  auto sweep = [&](auto side_c) {
    constexpr bool SIDE = decltype(side_c)::value;
    for (int kt = 0; kt + 1 < nk; ++kt) {  // every tile but the last: the next tile's transform rides behind the MFMAs
      const int cur = kt & 1;
      fetch_kc((kt + 2) * BK);  // constants of tile kt + 2 -> Kc[cur] (last read while tile kt was transformed, one sweep ago)
      if (!(DBG & 4)) fetch((kt + 1) * BK);
      // The next tile's operand transform + LDS writes are spread over MFMA steps 4..11 (one eighth each): the loads were
      // issued at the top of the sweep (landed by step 4) and ~12 VALU/LDS instructions fit in the shadow of every MFMA
      // pair, so the matrix pipe never waits for a write/barrier phase that all co-resident blocks would hit together.
      // (No run-time condition inside the 16 steps: one scheduling region, fragment reads can move ahead of the MFMAs.)
      if (live) {
        // fragments are double-buffered in registers: step i+1's LDS reads are issued before step i's MFMAs.  (Running them
        // two steps ahead behind a sched_barrier changed the waits from lgkmcnt(0) to lgkmcnt(4..8) but not the time: r02 notes.)
        float fb = T.Bs[cur][kh][bcol], fa0 = T.As[cur][kh][arow], fa1 = T.As[cur][kh][arow + 32];
#pragma unroll
        for (int step = 0; step < BK / 2; ++step) {
          const int kn = (step + 1 < BK / 2 ? step + 1 : step) * 2;
          const float nb = (DBG & 8) ? fb : T.Bs[cur][kn + kh][bcol];
          const float na0 = (DBG & 8) ? fa0 : T.As[cur][kn + kh][arow];
          const float na1 = (DBG & 8) ? fa1 : T.As[cur][kn + kh][arow + 32];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc1, 0, 0, 0);
          fb = nb; fa0 = na0; fa1 = na1;
          if (SIDE) sacc = __fmaf_rn(T.As[cur][skh + step][srow], T.Sb[cur][skh + step], sacc);
          constexpr int S0 = (DBG & 16) ? 8 : ((DBG & 32) ? 6 : 4);
          if (!(DBG & 2) && step >= S0 && step < S0 + 8) stash_part(cur ^ 1, (kt + 1) * BK, step - S0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) stash_part(cur ^ 1, (kt + 1) * BK, q);
      }
      put_kc(cur);
      __syncthreads();
    }
  };
  if (side) sweep(std::true_type{});
  else sweep(std::false_type{});
  if (live) {  // last tile: only the k-steps that hold real columns (K = 515 -> 2 of 16, K = 257 -> 1 of 16)
    const int cur = (nk - 1) & 1, steps = (K - (nk - 1) * BK + 1) / 2;
    for (int step = 0; step < steps; ++step) {
      const float fb = T.Bs[cur][2 * step + kh][bcol], fa0 = T.As[cur][2 * step + kh][arow], fa1 = T.As[cur][2 * step + kh][arow + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc1, 0, 0, 0);
    }
  }
  if (side) {  // last tile of the side column (columns >= K of either operand are staged as zeros)
    const int cur = (nk - 1) & 1;
#pragma unroll
    for (int step = 0; step < BK / 2; ++step) sacc = __fmaf_rn(T.As[cur][skh + step][srow], T.Sb[cur][skh + step], sacc);
  }
  __syncthreads();
  epi.finish(acc0, acc1, bm0 + wm * 64, bn0 + wn * 32, lane, wm, wn, bo.group, smem);
  if (side) {  // the two k halves of a row meet in LDS (tiles and the epilogue's scratch are dead), then the column's epilogue
    float* sred = reinterpret_cast<float*>(smem);
    __syncthreads();
    if (tid >= BM) sred[tid - BM] = sacc;
    __syncthreads();
    const float v = tid < BM ? sacc + sred[tid] : 0.f;
    __syncthreads();
    epi.finish_side(v, bm0 + srow, sc, tid, bo.group, smem);
  }
}

// ---- epilogue bodies (members defined here to keep the kernel readable)
// block sum of (s1, s2) held by threads 0..127 (zeros elsewhere) -> dst[0..1]; every thread of the block calls it
__device__ __forceinline__ void side_reduce(double s1, double s2, int tid, char* smem, double* dst) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  double* red = reinterpret_cast<double*>(smem);
  if (tid == 64) { red[0] = s1; red[1] = s2; }
  __syncthreads();
  if (tid == 0) {
    dst[0] = s1 + red[0];
    dst[1] = s2 + red[1];
  }
}
struct EpiStoreImpl : EpiStore {
  // side column `col` of the block's rows: thread tid < 128 holds the finished dot product of row r
  __device__ __forceinline__ void finish_side(float acc, int r, int col, int tid, int rblk, char* smem) const {
    const bool ok = tid < BM && r < R;
    const float v = acc + (bias ? bias[col] : 0.f);
    if (ok) C[(size_t)r * ldc + col] = v;
    if (moments) side_reduce(ok ? (double)v : 0.0, ok ? (double)v * (double)v : 0.0, tid, smem, moments + ((size_t)rblk * mstride + col) * 2);
  }
  __device__ __forceinline__ void finish(const f32x16& a0, const f32x16& a1, int r0, int c0, int lane, int wm, int wn, int rblk, char* smem) const {
    const int col = c0 + (lane & 31);
    const float bv = (bias && col < Nc) ? bias[col] : 0.f;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int r = r0 + t * 32 + acc_row(reg, lane);
        const float v = (t == 0 ? a0[reg] : a1[reg]) + bv;
        if (r < R && col < Nc) {
          C[(size_t)r * ldc + col] = v;
          s1 += (double)v;
          s2 += (double)v * (double)v;
        }
      }
    }
    if (moments) {
      // combine lane l and l^32 (same column, other rows), then the two M-waves through LDS
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      double* red = reinterpret_cast<double*>(smem);  // tiles are dead after the last barrier
      __syncthreads();
      if (wm == 1 && lane < 32) { red[(wn * 32 + lane) * 2] = s1; red[(wn * 32 + lane) * 2 + 1] = s2; }
      __syncthreads();
      if (wm == 0 && lane < 32 && col < Nc) {
        double* dst = moments + ((size_t)rblk * mstride + col) * 2;
        dst[0] = s1 + red[(wn * 32 + lane) * 2];
        dst[1] = s2 + red[(wn * 32 + lane) * 2 + 1];
      }
    }
  }
};

struct EpiMaskStatsImpl : EpiMaskStats {
  __device__ __forceinline__ void finish_side(float acc, int r, int col, int tid, int rblk, char* smem) const {
    const bool ok = tid < BM && r < R;
    float v = 0.f, xh = 0.f;
    if (ok) {
      float y;
      if (mode == 0) {
        const float h = H[(size_t)r * ldc + col];
        y = __fmaf_rn(s[col], h, t[col]);
        xh = (h - mean[col]) * rstd[col];
      } else {
        const int b = r / N, n = r - b * N;
        xh = Gx[(size_t)(ps ? r : n) * ldc + col] + Fx[(size_t)b * ldc + col];
        y = __fmaf_rn(gamma[col], xh, beta[col]);
      }
      v = y > 0.f ? acc : 0.f;
      C[(size_t)r * ldc + col] = v;
    }
    side_reduce((double)v, (double)v * (double)xh, tid, smem, sums + ((size_t)rblk * sstride + col) * 2);
  }
  // One 32x32 accumulator tile: C = acc * (y > 0) and this lane's column partials S1 = sum C, S2 = sum C * xhat.
  // The 16 rows a lane holds are r0 + (reg&3) + 8 (reg>>2) + 4 (lane>>5); their (sample, vertex) split advances
  // incrementally from one division per tile.  Partials stay fp32 inside the lane (16 terms), fp64 across lanes/blocks.
  struct Part { float s1, s2; };
  struct Cst { float s, t, m, r; };
  __device__ __forceinline__ Part tile(const f32x16& a, int r0, int col, bool cok, int lane, const Cst c) const {
    float s1 = 0.f, s2 = 0.f;
    const float c_s = c.s, c_t = c.t, c_m = c.m, c_r = c.r;
    const int rb = r0 + 4 * (lane >> 5);
    int b = 0, n = 0;
    if (mode != 0) { b = rb / N; n = rb - b * N; }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int off = (reg & 3) + 8 * (reg >> 2);
      const int r = rb + off;
      int bb = b, nn = n + off;
      if (mode != 0) while (nn >= N) { nn -= N; ++bb; }
      if (r < R && cok) {
        float xh, y;
        if (mode == 0) {
          const float h = H[(size_t)r * ldc + col];
          y = __fmaf_rn(c_s, h, c_t);
          xh = (h - c_m) * c_r;
        } else {
          xh = Gx[(size_t)(ps ? r : nn) * ldc + col] + Fx[(size_t)bb * ldc + col];
          y = __fmaf_rn(c_s, xh, c_t);
        }
        const float v = y > 0.f ? a[reg] : 0.f;
        C[(size_t)r * ldc + col] = v;
        s1 += v;
        s2 = __fmaf_rn(v, xh, s2);
      }
    }
    return Part{s1, s2};
  }
  __device__ __forceinline__ Cst consts(int col, bool cok) const {
    Cst c{0.f, 0.f, 0.f, 0.f};
    if (cok) {
      if (mode == 0) c = Cst{s[col], t[col], mean[col], rstd[col]};
      else c = Cst{gamma[col], beta[col], 0.f, 0.f};
    }
    return c;
  }
  // fp32 kernel: wave tile = two stacked 32x32 tiles, two M-waves per block
  __device__ __forceinline__ void finish(const f32x16& a0, const f32x16& a1, int r0, int c0, int lane, int wm, int wn, int rblk, char* smem) const {
    const int col = c0 + (lane & 31);
    const bool cok = col < Nc;
    const Cst c = consts(col, cok);
    const Part p0 = tile(a0, r0, col, cok, lane, c), p1 = tile(a1, r0 + 32, col, cok, lane, c);
    double s1 = (double)p0.s1 + (double)p1.s1, s2 = (double)p0.s2 + (double)p1.s2;
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    double* red = reinterpret_cast<double*>(smem);
    __syncthreads();
    if (wm == 1 && lane < 32) { red[(wn * 32 + lane) * 2] = s1; red[(wn * 32 + lane) * 2 + 1] = s2; }
    __syncthreads();
    if (wm == 0 && lane < 32 && cok) {
      double* dst = sums + ((size_t)rblk * sstride + col) * 2;
      dst[0] = s1 + red[(wn * 32 + lane) * 2];
      dst[1] = s2 + red[(wn * 32 + lane) * 2 + 1];
    }
  }
};

// ------------------------------------------------------------------------------------------------ weight gradients
// C[M x Nc] (+)= sum_r Aop[r, m] * Bop[r, n] over a chunk of rows; partial results per row-chunk (split-K), summed
// afterwards in chunk order (deterministic).  Tiles: the contraction index is the row r.
template <class AOp, class BOp>
__global__ __launch_bounds__(NT) void gemm_tn_kernel(AOp aop, BOp bop, int M, int Nc, int R, int rows_per_chunk, float* __restrict__ part, int order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TilesT& T = *reinterpret_cast<TilesT*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // Output rows / columns beyond the whole tiles (M = 257: one row; Nc = 515: three columns) are "side" products on the VALU
  // (tile_split): tile (mi, nj) also accumulates side row mi < ms.side against its 64 columns and side column nj < ns.side against
  // its 128 rows - both from the operand tiles it has in LDS plus ONE extra operand column per k-tile - and, when it has both,
  // their corner element.  257 x 515 is then 2 x 8 = 16 tiles instead of 3 x 9 = 27.
  const TileSplit ms = tile_split(M, BM), ns = tile_split(Nc, BN);
  const int mt = ms.tiles, ntile = mt * ns.tiles;
  const XcdOrder bo = xcd_order(blockIdx.x, ntile, order);  // group = row chunk, member = output tile
  if ((long)bo.group * rows_per_chunk >= R) return;
  const int mi = bo.member % mt, nj = bo.member / mt;
  const int bm0 = mi * BM, bn0 = nj * BN;
  const bool hasrow = mi < ms.side, hascol = nj < ns.side;  // block-uniform
  const int m_s = ms.tiles * BM + (hasrow ? mi : 0), n_s = ns.tiles * BN + (hascol ? nj : 0);
  const bool side_a = hasrow && tid < BK, side_b = hascol && tid >= 64 && tid < 64 + BK;  // who stages Sa[k] / Sb[k], k = tid % 32
  const typename AOp::KC kcsa = aop.kc(m_s);
  const typename BOp::KC kcsb = bop.kc(n_s);
  typename AOp::Row rowsa;
  typename BOp::Row rowsb;
  typename AOp::Raw rsa;
  typename BOp::Raw rsb;
  const int rbeg = bo.group * rows_per_chunk, rend = min(R, rbeg + rows_per_chunk);
  // staging: A tile [32 rows][128 m]: thread = 4 consecutive channels (16-byte loads) x rows (tid>>5) + 8p, p < 4;
  // B tile [32 rows][64 n]: 4 channels x rows (tid>>4) + 16p, p < 2.  A thread's channels are fixed for the whole
  // sweep: their constants are loaded once.
  const int ma = (tid & 31) * 4, ra0 = tid >> 5, nb = (tid & 15) * 4, rb0 = tid >> 4;
  typename AOp::KC kca[4];
  typename BOp::KC kcb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { kca[j] = aop.kc(bm0 + ma + j); kcb[j] = bop.kc(bn0 + nb + j); }
  typename AOp::Raw ra[4][4];
  typename BOp::Raw rb[2][4];
  typename AOp::Row rowa[4];
  typename BOp::Row rowb[2];
  auto fetch = [&](int r0) {
    const int bha = r0 / rows_N(aop), bhb = r0 / rows_N(bop);
    if (side_a) {
      const int r = r0 + tid;
      rowsa = aop.row(r < rend ? r : 0x7ffffff0, bha);
      rsa = aop.raw(rowsa, m_s);
    }
    if (side_b) {
      const int r = r0 + tid - 64;
      rowsb = bop.row(r < rend ? r : 0x7ffffff0, bhb);
      rsb = bop.raw(rowsb, n_s);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = r0 + ra0 + 8 * p;
      rowa[p] = aop.row(r < rend ? r : 0x7ffffff0, bha);
      aop.raw4(rowa[p], bm0 + ma, ra[p]);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = r0 + rb0 + 16 * p;
      rowb[p] = bop.row(r < rend ? r : 0x7ffffff0, bhb);
      bop.raw4(rowb[p], bn0 + nb, rb[p]);
    }
  };
  auto stash_side = [&](int buf) {
    if (side_a) T.Sa[buf][tid] = aop.fin(rowsa, kcsa, rsa);
    if (side_b) T.Sb[buf][tid - 64] = bop.fin(rowsb, kcsb, rsb);
  };
  auto put_a = [&](int buf, int p) {
    *reinterpret_cast<float4*>(&T.As[buf][ra0 + 8 * p][ma]) = make_float4(aop.fin(rowa[p], kca[0], ra[p][0]), aop.fin(rowa[p], kca[1], ra[p][1]),
                                                                        aop.fin(rowa[p], kca[2], ra[p][2]), aop.fin(rowa[p], kca[3], ra[p][3]));
  };
  auto put_b = [&](int buf, int p) {
    *reinterpret_cast<float4*>(&T.Bs[buf][rb0 + 16 * p][nb]) = make_float4(bop.fin(rowb[p], kcb[0], rb[p][0]), bop.fin(rowb[p], kcb[1], rb[p][1]),
                                                                         bop.fin(rowb[p], kcb[2], rb[p][2]), bop.fin(rowb[p], kcb[3], rb[p][3]));
  };
  auto stash_part = [&](int buf, int q) {  // q = 0..7: A row groups at even q, B row groups at q = 1 and 5, side columns at q = 3
    if ((q & 1) == 0) put_a(buf, q >> 1);
    else if (q == 1 || q == 5) put_b(buf, q >> 2);
    else if (q == 3) stash_side(buf);
  };
  auto stash = [&](int buf) {
    stash_side(buf);
#pragma unroll
    for (int p = 0; p < 4; ++p) put_a(buf, p);
#pragma unroll
    for (int p = 0; p < 2; ++p) put_b(buf, p);
  };
  f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
  const int nk = (rend - rbeg + BK - 1) / BK;
  if (nk > 0) {
    fetch(rbeg);
    stash(0);
  }
  __syncthreads();
  // Without side products (a dimension that tile_split leaves padded): waves whose output rows or columns are all beyond the
  // matrix issue no MFMA for them.
  const bool ncol = bn0 + wn * 32 < Nc;
  const bool live0 = ncol && bm0 + wm * 64 < M, live1 = ncol && bm0 + wm * 64 + 32 < M;
  // side accumulators: row side = thread (column tid % 64, k quarter tid / 64), 8 k per tile; column side = thread (row tid % 128,
  // k half tid / 128), 16 k per tile; corner = lane k of wave 0
  const int scol = tid & 63, sk4 = (tid >> 6) * (BK / 4), srow = tid & (BM - 1), sk2 = (tid >> 7) * (BK / 2);
  float acc_r = 0.f, acc_c = 0.f, acc_x = 0.f;
  auto side_step = [&](int cur, int step) {
    if (hascol) acc_c = __fmaf_rn(T.As[cur][sk2 + step][srow], T.Sb[cur][sk2 + step], acc_c);
    if (hasrow && step < BK / 4) acc_r = __fmaf_rn(T.Sa[cur][sk4 + step], T.Bs[cur][sk4 + step][scol], acc_r);
  };
  auto sweep = [&](auto side_c) {
    constexpr bool SIDE = decltype(side_c)::value;
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) fetch(rbeg + (kt + 1) * BK);
      const int arow = wm * 64 + (lane & 31), bcol = wn * 32 + (lane & 31), kh = lane >> 5;
      const bool more = kt + 1 < nk;
      if (live0) {
        float fb = T.Bs[cur][kh][bcol], fa0 = T.As[cur][kh][arow], fa1 = T.As[cur][kh][arow + 32];
#pragma unroll
        for (int step = 0; step < BK / 2; ++step) {  // register double-buffered fragments; next tile's transform spread over steps 4..11
          const int kn = (step + 1 < BK / 2 ? step + 1 : step) * 2;
          const float nb = T.Bs[cur][kn + kh][bcol];
          const float na0 = T.As[cur][kn + kh][arow];
          const float na1 = T.As[cur][kn + kh][arow + 32];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb, acc0, 0, 0, 0);
          if (live1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb, acc1, 0, 0, 0);
          fb = nb; fa0 = na0; fa1 = na1;
          if (SIDE) side_step(cur, step);
          if (step >= 4 && step < 12 && more) stash_part(cur ^ 1, step - 4);
        }
      } else {  // a wave without live output still carries its share of the side products and of the staging
        if (SIDE) {
#pragma unroll
          for (int step = 0; step < BK / 2; ++step) side_step(cur, step);
        }
        if (more) {
#pragma unroll
          for (int q = 0; q < 8; ++q) stash_part(cur ^ 1, q);
        }
      }
      if (SIDE && hasrow && hascol && tid < BK) acc_x = __fmaf_rn(T.Sa[cur][tid], T.Sb[cur][tid], acc_x);
      __syncthreads();
    }
  };
  if (hasrow || hascol) sweep(std::true_type{});
  else sweep(std::false_type{});
  float* dst = part + (size_t)bo.group * M * Nc;
  const int col = bn0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = bm0 + wm * 64 + t * 32 + acc_row(reg, lane);
      if (m < M && col < Nc) dst[(size_t)m * Nc + col] = t == 0 ? acc0[reg] : acc1[reg];
    }
  if (hasrow || hascol) {  // partial side sums meet in LDS (the tiles are dead: the sweep ended on a barrier), fixed order
    float* red = reinterpret_cast<float*>(smem);  // [256] row-side partials, [256] column-side partials
    red[tid] = acc_r;
    red[NT + tid] = acc_c;
    if (tid < BK) {  // corner: the 32 lane partials of wave 0
      float x = acc_x;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
      if (tid == 0 && hasrow && hascol) dst[(size_t)m_s * Nc + n_s] = x;
    }
    __syncthreads();
    if (hasrow && tid < BN && bn0 + tid < Nc) dst[(size_t)m_s * Nc + bn0 + tid] = ((red[tid] + red[64 + tid]) + red[128 + tid]) + red[192 + tid];
    if (hascol && tid < BM && bm0 + tid < M) dst[(size_t)(bm0 + tid) * Nc + n_s] = red[NT + tid] + red[NT + BM + tid];
  }
}

// out[i] = scale * sum_c part[c][i]  (fixed chunk order)
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const float* __restrict__ part, int chunks, long n, float scale, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(size_t)c * n + i];
  out[i] = scale * s;
}


#include "decoder_bf16.h"
#include "decoder_rows2.h"
#include "decoder_rows3.h"
#include "decoder_tn2.h"
#include "decoder_rows2f.h"
#include "decoder_tn3.h"

}  // namespace dec

// ================================================================================================ small kernels
namespace dec {

constexpr int PREP_COLS = 4;   // channels per block of prep_ps_kernel (per-sample grids)
constexpr int PREP_SCOLS = 2;  // channels per block of prep_kernel (shared template grid): 264 blocks at ld1 = 528
constexpr int PREP_NT = 1024;  // threads of prep_kernel: 16 waves walk the 64 samples / the template vertices of a block's channels

// G[n,c] = W1[c,0:3].grid[n] with a pinned operation order: the statistics pass and l1_fill_kernel must see the same values
__device__ __forceinline__ float l1_gval(float w0, float w1, float w2, float g0, float g1, float g2) {
  return __fmaf_rn(w2, g2, __fmaf_rn(w1, g1, w0 * g0));
}

// Layer 1 in factored form + closed-form BN-1 statistics.  One block = COLS output channels.
//   G[n,c] = W1[c,0:3].grid[n],  F[b,c] = b1[c] + W1[c,3:].feat[b]
//   train: mean = mean_n G + mean_b F, var = var_n G + var_b F (biased; exact for the B x N product set)
//   Gx = rstd*(G - gm), Fx = rstd*(F - fm)  with gm + fm = mean  =>  xhat1[b,n,c] = Gx[n,c] + Fx[b,c]
// This kernel leaves the per-channel constants (gmean1 = gm, rstd1), Fx and the pre-scaled Fy = gamma Fx + beta; the [N, ld1] arrays
// Gx and Gy = gamma Gx are l1_fill_kernel's.  (Until r06 a block also wrote its 4 channels of Gx - 16-byte pieces of 2 112-byte rows
// from 132 blocks - and prescale_l1_kernel read Gx back to write Gy: 220 + 61 us at N = 64 050.)
template <int COLS>
__global__ __launch_bounds__(PREP_NT) void prep_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                   const float* __restrict__ grid, const float* __restrict__ feat, int B, int N,
                                                   int C1, int ld1, int training, float eps, float momentum,
                                                   float* __restrict__ rmean, float* __restrict__ rvar, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float* __restrict__ Fx, float* __restrict__ Fy,
                                                   float* __restrict__ mean1, float* __restrict__ rstd1, float* __restrict__ gmean1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Cf = C1 - 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = PREP_NT / 64, NPART = NW / COLS;  // 16 waves: NPART per channel in the statistics pass
  double* sPart = reinterpret_cast<double*>(smem);      // [COLS][NPART][2]: partial (sum, sum of squares) over n
  float* sW = reinterpret_cast<float*>(sPart + COLS * NPART * 2);  // [COLS][C1]
  float* sF = sW + COLS * C1;                      // [COLS][B]
  float* sStat = sF + COLS * B;                    // [COLS][4]: gm, fm, rstd
  const int c0 = blockIdx.x * COLS;
  for (int i = tid; i < COLS * C1; i += PREP_NT) {
    const int c = c0 + i / C1;
    sW[i] = c < C1 ? W1[(size_t)c * C1 + i % C1] : 0.f;
  }
  __syncthreads();
  for (int b = wave; b < B; b += NW) {  // one wave per sample: lanes stride over the feature
    float acc[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < Cf; k0 += 512) {  // 8 independent loads in flight per lane
      float f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int k = k0 + lane + 64 * u; f[u] = k < Cf ? feat[(size_t)b * Cf + k] : 0.f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + lane + 64 * u;
        if (k < Cf) {
#pragma unroll
          for (int j = 0; j < COLS; ++j) acc[j] = __fmaf_rn(f[u], sW[j * C1 + 3 + k], acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      const float r = obman_wave_sum(acc[j]);
      if (lane == 0) sF[j * B + b] = r + ((c0 + j < C1) ? b1[c0 + j] : 0.f);
    }
  }
  __syncthreads();
  // per-channel statistics in fp64: wave = (channel j, part); the parts split the template vertices and meet in LDS (fixed order).
  // G[n,c] (3 MACs) is recomputed instead of being staged: N reaches 64 050 (25 x 2562).  Four vertices' coordinates are requested
  // before the first sum (a lane's vertices keep their order).
  {
    const int j = wave % COLS, part = wave / COLS;
    const float w0 = sW[j * C1], w1 = sW[j * C1 + 1], w2 = sW[j * C1 + 2];
    double sg = 0, sgg = 0;
    constexpr int STEP = 64 * NPART;
    for (int n = part * 64 + lane; n < N; n += 4 * STEP) {
      float g[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = n + u * STEP < N ? n + u * STEP : n;
        g[u][0] = grid[nn * 3]; g[u][1] = grid[nn * 3 + 1]; g[u][2] = grid[nn * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (n + u * STEP < N) { const double v = l1_gval(w0, w1, w2, g[u][0], g[u][1], g[u][2]); sg += v; sgg += v * v; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sg += __shfl_xor(sg, off, 64); sgg += __shfl_xor(sgg, off, 64); }
    if (lane == 0) { sPart[(j * NPART + part) * 2] = sg; sPart[(j * NPART + part) * 2 + 1] = sgg; }
  }
  __syncthreads();
  for (int j = wave; j < COLS; j += NW) {
    const int c = c0 + j;
    double sg = 0, sgg = 0, sf = 0, sff = 0;
    for (int q = 0; q < NPART; ++q) { sg += sPart[(j * NPART + q) * 2]; sgg += sPart[(j * NPART + q) * 2 + 1]; }
    for (int b = lane; b < B; b += 64) { const double v = sF[j * B + b]; sf += v; sff += v * v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { sf += __shfl_xor(sf, off, 64); sff += __shfl_xor(sff, off, 64); }
    if (lane == 0 && c < C1) {
      float gm, fm, var;
      if (training) {
        const double mg = sg / N, mf = sf / B;
        const double v = (sgg / N - mg * mg) + (sff / B - mf * mf);
        gm = (float)mg; fm = (float)mf; var = (float)(v > 0 ? v : 0);
        if (rmean) {
          const double R = (double)B * N;
          rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)(mg + mf);
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(v * (R / (R > 1 ? R - 1 : 1)));
        }
      } else {
        gm = rmean[c]; fm = 0.f; var = rvar[c];
      }
      const float rs = 1.f / sqrtf(var + eps);
      sStat[j * 4] = gm; sStat[j * 4 + 1] = fm; sStat[j * 4 + 2] = rs;
      mean1[c] = gm + fm;
      rstd1[c] = rs;
      gmean1[c] = gm;
    }
  }
  __syncthreads();
  // columns C1 .. ld1-1 (the pitch padding, covered by the extra blocks of the grid) are written as zeros: the bf16 flavour's
  // operand generators read whole 8-wide chunks and rely on finite padding
  for (int i = tid; i < COLS * B; i += PREP_NT) {
    const int b = i / COLS, j = i % COLS, c = c0 + j;
    if (c < ld1) {
      const float fx = c < C1 ? (sF[j * B + b] - sStat[j * 4 + 1]) * sStat[j * 4 + 2] : 0.f;
      Fx[(size_t)b * ld1 + c] = fx;
      if (Fy) Fy[(size_t)b * ld1 + c] = c < C1 ? __fmaf_rn(gamma[c], fx, beta[c]) : 0.f;
    }
  }
}

// The [N, ld1] halves of layer 1's factors, whole rows at a time: Gx[n,c] = rstd (G[n,c] - gm) and (shared-template flavours) the
// pre-scaled Gy = gamma Gx that the h2 GEMM and the layer-2 weight gradient regenerate a1 = relu(Gy[n] + Fy[b]) from, plus Gy's
// sentinel row N (-3e38: rows outside the problem read it, relu(-3e38 + Fy) = 0 exactly).  The per-channel constants sit in LDS
// (6 x ld1 floats); a thread forms four channels of a row and stores them as 16 bytes.
constexpr int L1F_NT = 256, L1F_ITEMS = 16;  // items (four channels of a row) per thread: up to 16, fewer on small templates (>= ~1024 blocks)
__global__ __launch_bounds__(L1F_NT) void l1_fill_kernel(const float* __restrict__ W1, const float* __restrict__ grid,
                                                         const float* __restrict__ gmean1, const float* __restrict__ rstd1,
                                                         const float* __restrict__ gamma, int N, int C1, int ld1, int items,
                                                         float* __restrict__ Gx, float* __restrict__ Gy) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [6][ld1]: w0, w1, w2, gm, rstd, gamma
  const int tid = threadIdx.x;
  for (int c = tid; c < ld1; c += L1F_NT) {
    const bool ok = c < C1;
    sc[c] = ok ? W1[(size_t)c * C1] : 0.f; sc[ld1 + c] = ok ? W1[(size_t)c * C1 + 1] : 0.f; sc[2 * ld1 + c] = ok ? W1[(size_t)c * C1 + 2] : 0.f;
    sc[3 * ld1 + c] = ok ? gmean1[c] : 0.f; sc[4 * ld1 + c] = ok ? rstd1[c] : 0.f; sc[5 * ld1 + c] = ok && Gy ? gamma[c] : 0.f;
  }
  __syncthreads();
  const unsigned q4 = (unsigned)ld1 >> 2, total = (unsigned)N * q4;
  const unsigned i0 = blockIdx.x * (unsigned)(L1F_NT * items) + tid;
#pragma unroll 4
  for (int k = 0; k < items; ++k) {
    const unsigned i = i0 + k * L1F_NT;
    if (i >= total) break;
    const unsigned n = i / q4, c = (i - n * q4) * 4;
    const float g0 = grid[n * 3], g1 = grid[n * 3 + 1], g2 = grid[n * 3 + 2];
    const float4 w0 = *reinterpret_cast<const float4*>(sc + c), w1 = *reinterpret_cast<const float4*>(sc + ld1 + c),
                 w2 = *reinterpret_cast<const float4*>(sc + 2 * ld1 + c), gm = *reinterpret_cast<const float4*>(sc + 3 * ld1 + c),
                 rs = *reinterpret_cast<const float4*>(sc + 4 * ld1 + c);
    float4 x;
    x.x = (int)c < C1 ? (l1_gval(w0.x, w1.x, w2.x, g0, g1, g2) - gm.x) * rs.x : 0.f;
    x.y = (int)c + 1 < C1 ? (l1_gval(w0.y, w1.y, w2.y, g0, g1, g2) - gm.y) * rs.y : 0.f;
    x.z = (int)c + 2 < C1 ? (l1_gval(w0.z, w1.z, w2.z, g0, g1, g2) - gm.z) * rs.z : 0.f;
    x.w = (int)c + 3 < C1 ? (l1_gval(w0.w, w1.w, w2.w, g0, g1, g2) - gm.w) * rs.w : 0.f;
    *reinterpret_cast<float4*>(Gx + (size_t)n * ld1 + c) = x;
    if (Gy) {
      const float4 ga = *reinterpret_cast<const float4*>(sc + 5 * ld1 + c);
      *reinterpret_cast<float4*>(Gy + (size_t)n * ld1 + c) = make_float4(ga.x * x.x, ga.y * x.y, ga.z * x.z, ga.w * x.w);
    }
  }
  if (Gy && blockIdx.x == 0)
    for (int c = tid; c < ld1; c += L1F_NT) Gy[(size_t)N * ld1 + c] = -3.0e38f;
}

constexpr int L1PS_ROWS = 128;  // rows per block of the per-sample-grid layer-1 backward

// Layer 1 for a PER-SAMPLE grid (AtlasBranch.forward, atlasbranch.py:78-108: every sample gets its own points on the sphere).
// h1[b,n,c] = G[b,n,c] + F[b,c] with G = W1[c,0:3].grid[b,n]: the two-factor form survives (Gx is just [R,ld] now, indexed
// by the row), but the batch statistics are no longer additive - the per-sample means of G correlate with F:
//   E[h] = mean_r G + mean_b F,   E[h^2] = (sum_r G^2 + 2 sum_b F_b sum_n G_bn + N sum_b F_b^2) / R.
// One block = PREP_COLS channels; one wave per channel walks the rows sample by sample (fp64 sums, fixed order).
__global__ __launch_bounds__(256) void prep_ps_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                      const float* __restrict__ grid, const float* __restrict__ feat, int B, int N,
                                                      int C1, int ld1, int training, float eps, float momentum,
                                                      float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ Gx,
                                                      float* __restrict__ Fx, float* __restrict__ mean1, float* __restrict__ rstd1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Cf = C1 - 3, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* sW = reinterpret_cast<float*>(smem);          // [PREP_COLS][C1]
  float* sF = sW + PREP_COLS * C1;                      // [PREP_COLS][B]
  float* sStat = sF + PREP_COLS * B;                    // [PREP_COLS][4]: gm, fm, rstd
  const int c0 = blockIdx.x * PREP_COLS;
  const long R = (long)B * N;
  for (int i = tid; i < PREP_COLS * C1; i += 256) {
    const int c = c0 + i / C1;
    sW[i] = c < C1 ? W1[(size_t)c * C1 + i % C1] : 0.f;
  }
  __syncthreads();
  auto Gval = [&](int j, long r) {
    return sW[j * C1] * grid[r * 3] + sW[j * C1 + 1] * grid[r * 3 + 1] + sW[j * C1 + 2] * grid[r * 3 + 2];
  };
  for (int b = wave; b < B; b += 4) {  // F[b,c] = b1[c] + W1[c,3:].feat[b]: one wave per sample
    float acc[PREP_COLS];
#pragma unroll
    for (int j = 0; j < PREP_COLS; ++j) acc[j] = 0.f;
    for (int k = lane; k < Cf; k += 64) {
      const float f = feat[(size_t)b * Cf + k];
#pragma unroll
      for (int j = 0; j < PREP_COLS; ++j) acc[j] = __fmaf_rn(f, sW[j * C1 + 3 + k], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < PREP_COLS; ++j) {
      const float r = obman_wave_sum(acc[j]);
      if (lane == 0) sF[j * B + b] = r + ((c0 + j < C1) ? b1[c0 + j] : 0.f);
    }
  }
  __syncthreads();
  auto wsum = [](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
  };
  for (int j = wave; j < PREP_COLS; j += 4) {
    const int c = c0 + j;
    double sg = 0, sgg = 0, cross = 0, sf = 0, sff = 0;
    for (int b = 0; b < B; ++b) {
      double g1 = 0, g2 = 0;
      for (int n = lane; n < N; n += 64) { const double v = Gval(j, (long)b * N + n); g1 += v; g2 += v * v; }
      g1 = wsum(g1); g2 = wsum(g2);
      const double f = sF[j * B + b];
      sg += g1; sgg += g2; cross += f * g1; sf += f; sff += f * f;
    }
    if (lane == 0 && c < C1) {
      float gm, fm, var;
      if (training) {
        const double mg = sg / (double)R, mf = sf / B;
        const double mean = mg + mf;
        double v = (sgg + 2.0 * cross + (double)N * sff) / (double)R - mean * mean;
        if (v < 0) v = 0;
        gm = (float)mg; fm = (float)mf; var = (float)v;
        if (rmean) {
          rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(v * ((double)R / (R > 1 ? R - 1 : 1)));
        }
      } else {
        gm = rmean[c]; fm = 0.f; var = rvar[c];
      }
      const float rs = 1.f / sqrtf(var + eps);
      sStat[j * 4] = gm; sStat[j * 4 + 1] = fm; sStat[j * 4 + 2] = rs;
      mean1[c] = gm + fm;
      rstd1[c] = rs;
    }
  }
  __syncthreads();
  for (long i = tid; i < (long)PREP_COLS * R; i += 256) {
    const long r = i / PREP_COLS;
    const int j = (int)(i % PREP_COLS);
    if (c0 + j < C1) Gx[(size_t)r * ld1 + c0 + j] = (Gval(j, r) - sStat[j * 4]) * sStat[j * 4 + 2];
  }
  for (int i = tid; i < PREP_COLS * B; i += 256) {
    const int b = i / PREP_COLS, j = i % PREP_COLS;
    if (c0 + j < C1) Fx[(size_t)b * ld1 + c0 + j] = (sF[j * B + b] - sStat[j * 4 + 1]) * sStat[j * 4 + 2];
  }
}

// Per-sample-grid layer-1 backward, pass over gy1 [R,ld1] and Gx [R,ld1]: thread = channel, block = L1PS_ROWS rows of one
// sample.  gh1 = ca*gy1 + cb*xhat1 + cc (coefficients from l1ps_coef_kernel);  PdF[b][chunk][c] = sum_n gh1,
// Pw[b][chunk][j][c] = sum_n gh1 * grid[b,n,j].
__global__ __launch_bounds__(256) void l1ps_reduce_kernel(const float* __restrict__ GY1, const float* __restrict__ Gx,
                                                          const float* __restrict__ Fx, const float* __restrict__ grid,
                                                          const float* __restrict__ ca, const float* __restrict__ cb,
                                                          const float* __restrict__ cc, int ld1, int N, int C1, int chunks,
                                                          float* __restrict__ PdF, float* __restrict__ Pw) {
  const int c = blockIdx.z * 256 + threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
  if (c >= C1) return;
  const int n0 = chunk * L1PS_ROWS, n1 = min(N, n0 + L1PS_ROWS);
  const float a_ = ca[c], b_ = cb[c], c_ = cc[c], fx = Fx[(size_t)b * ld1 + c];
  float d = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
  for (int n = n0; n < n1; ++n) {
    const size_t r = (size_t)b * N + n;
    const float gh = __fmaf_rn(a_, GY1[r * ld1 + c], __fmaf_rn(b_, Gx[r * ld1 + c] + fx, c_));
    d += gh;
    w0 = __fmaf_rn(gh, grid[r * 3], w0); w1 = __fmaf_rn(gh, grid[r * 3 + 1], w1); w2 = __fmaf_rn(gh, grid[r * 3 + 2], w2);
  }
  const size_t slot = (size_t)b * chunks + chunk;
  PdF[slot * ld1 + c] = d;
  Pw[(slot * 3 + 0) * ld1 + c] = w0; Pw[(slot * 3 + 1) * ld1 + c] = w1; Pw[(slot * 3 + 2) * ld1 + c] = w2;
}
// sums [blocks][C1][2] (S1 = sum gy1, S2 = sum gy1*xhat1) -> g_gamma1 = S2, g_beta1 = S1 and the coefficients of
// gh1 = k1*(gy1 - S1/R - xhat1*S2/R) = ca*gy1 + cb*xhat1 + cc  (eval: ca = k1, cb = cc = 0).  One wave per channel.
__global__ __launch_bounds__(256) void l1ps_coef_kernel(const double* __restrict__ sums, int blocks, long R, int C, int training,
                                                        const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                        float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                        float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int b = lane; b < blocks; b += 64) { s1 += sums[((size_t)b * C + c) * 2]; s2 += sums[((size_t)b * C + c) * 2 + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
  if (lane != 0) return;
  g_gamma[c] = (float)s2;
  g_beta[c] = (float)s1;
  const float k1 = gamma[c] * rstd[c];
  ca[c] = k1;
  cb[c] = training ? (float)(-(double)k1 * s2 / (double)R) : 0.f;
  cc[c] = training ? (float)(-(double)k1 * s1 / (double)R) : 0.f;
}
// dF[b,c] = sum_chunk PdF, g_b1[c] = sum_b dF[b,c], gW1[c, 0:3] = sum_{b,chunk} Pw   (fixed order).  Thread = channel.
__global__ __launch_bounds__(256) void l1ps_finalize_kernel(const float* __restrict__ PdF, const float* __restrict__ Pw, int ld1, int B,
                                                            int C1, int chunks, float* __restrict__ dF, float* __restrict__ g_b1,
                                                            float* __restrict__ gW1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C1) return;
  float gb = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
  for (int b = 0; b < B; ++b) {
    float d = 0.f;
    for (int k = 0; k < chunks; ++k) {
      const size_t slot = (size_t)b * chunks + k;
      d += PdF[slot * ld1 + c];
      w0 += Pw[(slot * 3 + 0) * ld1 + c]; w1 += Pw[(slot * 3 + 1) * ld1 + c]; w2 += Pw[(slot * 3 + 2) * ld1 + c];
    }
    dF[(size_t)b * ld1 + c] = d;
    gb += d;
  }
  g_b1[c] = gb;
  gW1[(size_t)c * C1] = w0; gW1[(size_t)c * C1 + 1] = w1; gW1[(size_t)c * C1 + 2] = w2;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Pre-reduction of per-row-block partials when there are thousands of them (25-patch templates: 8 025 row blocks, 32 100
// layer-4 blocks): out[seg][col] = sum of in[r][col] over the rows of segment seg, lanes along the contiguous columns,
// fixed order.  The one-wave-per-channel finalize kernels below then see <= 64 rows instead of striding through megabytes.
constexpr int PRE_SEGMENTS = 64, PRE_MIN_ROWS = 512;
constexpr int PRE_SEGMENTS_NARROW = 256;  // arrays of <= 1024 columns (per-channel partials): 3 column blocks x 64 segments = 192
// segment count of pre_reduce for an array of `cols` columns: allocation (bwd_ws / fwd_ws) and launch share this rule
constexpr int pre_segments(long cols) { return cols <= 1024 ? PRE_SEGMENTS_NARROW : PRE_SEGMENTS; }
                                         // blocks walked 125 rows each in 32 us at 8 032 row blocks; 768 blocks of 32 rows do it in a third
template <class T>
__global__ __launch_bounds__(256) void colsum_segments_kernel(const T* __restrict__ in, long rows, int cols, int rows_per_seg,
                                                              T* __restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const long r0 = (long)blockIdx.y * rows_per_seg, r1 = r0 + rows_per_seg < rows ? r0 + rows_per_seg : rows;
  T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += in[r * cols + col]; a1 += in[(r + 1) * cols + col]; a2 += in[(r + 2) * cols + col]; a3 += in[(r + 3) * cols + col];
  }
  for (; r < r1; ++r) a0 += in[r * cols + col];
  out[(size_t)blockIdx.y * cols + col] = (a0 + a1) + (a2 + a3);
}

// moments [blocks][C][2] (sum, sum sq over each row block) -> mean/rstd, affine (s,t) of y = s*h + t, running stats.
// One wave per channel: lanes stride over the row blocks (fixed lane->block mapping + xor tree => deterministic).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ moments, int blocks, long R, int C, int training,
                                                          float eps, float momentum, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ s, float* __restrict__ t) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  float m, v;
  if (training) {
    double s1 = 0, s2 = 0;
    for (int b = lane; b < blocks; b += 64) { s1 += moments[((size_t)b * C + c) * 2]; s2 += moments[((size_t)b * C + c) * 2 + 1]; }
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
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
  s[c] = gamma[c] * rs;
  t[c] = beta[c] - m * gamma[c] * rs;
}

// out[r, 0:3] = f * (b4 + W4 . relu(s3*h3[r]+t3)); 32 lanes per row, 2 rows per wave pass.  Lane sub owns the four CONSECUTIVE
// channels 4 sub .. 4 sub + 3 (C3 <= 128): one 16-byte (fp32) or 8-byte (bf16) load per row.
__device__ __forceinline__ void ld4act(const float* p, float* o) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld4act(const bfraw* p, float* o) {
  const u32x2 v = *reinterpret_cast<const u32x2*>(p);
  o[0] = bf_lo(v.x); o[1] = bf_hi(v.x); o[2] = bf_lo(v.y); o[3] = bf_hi(v.y);
}
template <class HT>
__global__ __launch_bounds__(256) void l4_fwd_kernel(const HT* __restrict__ H3, int ld3, const float* __restrict__ s3,
                                                     const float* __restrict__ t3, const float* __restrict__ W4,
                                                     const float* __restrict__ b4, float f, long R, int C3, float* __restrict__ out) {
  const int tid = threadIdx.x, sub = tid & 31, c0 = sub * 4;
  const long r0 = ((long)blockIdx.x * 256 + tid) >> 5;
  const long stride = ((long)gridDim.x * 256) >> 5;
  float cs[4], ct[4], w0[4], w1[4], w2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool ok = c0 + j < C3;
    const int c = ok ? c0 + j : 0;
    cs[j] = ok ? s3[c] : 0.f; ct[j] = ok ? t3[c] : 0.f;
    w0[j] = ok ? W4[c] : 0.f; w1[j] = ok ? W4[C3 + c] : 0.f; w2[j] = ok ? W4[2 * C3 + c] : 0.f;
  }
  const int cl = c0 + 4 <= ld3 ? c0 : 0;  // ld3 is a multiple of 16: a lane's four channels are inside the pitch or wholly masked
  // four rows per step, every load issued before the first use (one 8- or 16-byte load in flight per lane left this pass at
  // 2.5 TB/s); rows are still finished in their original order
  for (long rq = r0; rq < R; rq += 4 * stride) {
    float hq[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = rq + u * stride;
      ld4act(H3 + (size_t)(rr < R ? rr : rq) * ld3 + cl, hq[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
    const long r = rq + u * stride;
    if (r >= R) break;
    const float (&h)[4] = hq[u];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // channels beyond C3 (pitch padding of the fp32 flavour is not initialised): select, do not multiply
      const float a = c0 + j < C3 ? fmaxf(__fmaf_rn(cs[j], h[j], ct[j]), 0.f) : 0.f;
      a0 = __fmaf_rn(a, w0[j], a0); a1 = __fmaf_rn(a, w1[j], a1); a2 = __fmaf_rn(a, w2[j], a2);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64);
    }
    if (sub == 0) {
      out[r * 3] = f * (a0 + b4[0]);
      out[r * 3 + 1] = f * (a1 + b4[1]);
      out[r * 3 + 2] = f * (a2 + b4[2]);
    }
    }
  }
}

// Layer-4 backward over a chunk of rows.  Block = 128 threads = 64 channel PAIRS x 2 row halves (4-byte loads of bf16 pairs,
// 8-byte of fp32); the two halves are combined through LDS in fixed order.  Partials: sums[blk][C3][2] (S1,S2 fp64),
// gw[blk][3*C3 + 4] (gW4 rows then gb4).
__device__ __forceinline__ void ld2act(const float* p, float& a, float& b) { const float2 v = *reinterpret_cast<const float2*>(p); a = v.x; b = v.y; }
__device__ __forceinline__ void ld2act(const bfraw* p, float& a, float& b) { const unsigned v = *reinterpret_cast<const unsigned*>(p); a = bf_lo(v); b = bf_hi(v); }
template <class HT>
__global__ __launch_bounds__(128) void l4_bwd_kernel(const float* __restrict__ G, const HT* __restrict__ H3, int ld3,
                                                     const float* __restrict__ s3, const float* __restrict__ t3,
                                                     const float* __restrict__ mean3, const float* __restrict__ rstd3,
                                                     const float* __restrict__ W4, float f, long R, int C3, int rows_per_blk,
                                                     double* __restrict__ sums, float* __restrict__ gw) {
  const int pair = threadIdx.x & 63, half = threadIdx.x >> 6, c = pair * 2;
  const long rbeg = (long)blockIdx.x * rows_per_blk, rend = min(R, rbeg + rows_per_blk);
  float cs[2], ct[2], cm[2], cr[2], w0[2], w1[2], w2[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const bool ok = c + e < C3;
    const int cc = ok ? c + e : 0;
    cs[e] = ok ? s3[cc] : 0.f; ct[e] = ok ? t3[cc] : 0.f; cm[e] = ok ? mean3[cc] : 0.f; cr[e] = ok ? rstd3[cc] : 0.f;
    w0[e] = ok ? W4[cc] : 0.f; w1[e] = ok ? W4[C3 + cc] : 0.f; w2[e] = ok ? W4[2 * C3 + cc] : 0.f;
  }
  const int cl = c + 2 <= ld3 ? c : 0;
  double S1[2] = {0, 0}, S2[2] = {0, 0};
  float ga[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, gb[3] = {0.f, 0.f, 0.f};
  for (long rq = rbeg + half; rq < rend; rq += 8) {  // four rows per step, loads first (rows finish in their original order)
    float gq[4][3], hq[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = rq + 2 * u < rend ? rq + 2 * u : rq;
      gq[u][0] = G[rr * 3]; gq[u][1] = G[rr * 3 + 1]; gq[u][2] = G[rr * 3 + 2];
      ld2act(H3 + (size_t)rr * ld3 + cl, hq[u][0], hq[u][1]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
    if (rq + 2 * u >= rend) break;
    const float g0 = f * gq[u][0], g1 = f * gq[u][1], g2 = f * gq[u][2];
    const float (&h)[2] = hq[u];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float y = __fmaf_rn(cs[e], h[e], ct[e]);
      const float a = fmaxf(y, 0.f);
      const float gy = y > 0.f ? (g0 * w0[e] + g1 * w1[e] + g2 * w2[e]) : 0.f;
      S1[e] += (double)gy;
      S2[e] += (double)gy * (double)((h[e] - cm[e]) * cr[e]);
      ga[e][0] = __fmaf_rn(g0, a, ga[e][0]); ga[e][1] = __fmaf_rn(g1, a, ga[e][1]); ga[e][2] = __fmaf_rn(g2, a, ga[e][2]);
    }
    gb[0] += g0; gb[1] += g1; gb[2] += g2;
    }
  }
  __shared__ double sd[64][4];
  __shared__ float sf[64][9];
  if (half == 1) {
    sd[pair][0] = S1[0]; sd[pair][1] = S2[0]; sd[pair][2] = S1[1]; sd[pair][3] = S2[1];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int k = 0; k < 3; ++k) sf[pair][e * 3 + k] = ga[e][k];
    sf[pair][6] = gb[0]; sf[pair][7] = gb[1]; sf[pair][8] = gb[2];
  }
  __syncthreads();
  if (half == 0) {
    float* dst = gw + (size_t)blockIdx.x * (3 * C3 + 4);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (c + e < C3) {
        sums[((size_t)blockIdx.x * C3 + c + e) * 2] = S1[e] + sd[pair][2 * e];
        sums[((size_t)blockIdx.x * C3 + c + e) * 2 + 1] = S2[e] + sd[pair][2 * e + 1];
        dst[c + e] = ga[e][0] + sf[pair][e * 3];
        dst[C3 + c + e] = ga[e][1] + sf[pair][e * 3 + 1];
        dst[2 * C3 + c + e] = ga[e][2] + sf[pair][e * 3 + 2];
      }
    }
    if (pair == 0) { dst[3 * C3] = gb[0] + sf[0][6]; dst[3 * C3 + 1] = gb[1] + sf[0][7]; dst[3 * C3 + 2] = gb[2] + sf[0][8]; }
  }
}

// sums [blocks][C][2] -> g_gamma = S2, g_beta = S1, and the affine gh coefficients (see AGradH): with k1 = gamma*rstd,
// k2 = S1/R, k3 = S2/R (0 in eval):  ka = k1, kb = -k1*rstd*k3, kc = k1*(mean*rstd*k3 - k2), formed in fp64.
// conv-bias gradient gb = sum_r gh = (train ? 0 : k1*S1).  One wave per channel.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ sums, int blocks, long R, int C, int training,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                              float* __restrict__ g_bias, float* __restrict__ k1,
                                                              float* __restrict__ k2, float* __restrict__ k3) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int b = lane; b < blocks; b += 64) { s1 += sums[((size_t)b * C + c) * 2]; s2 += sums[((size_t)b * C + c) * 2 + 1]; }
  s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
  if (lane != 0) return;
  g_gamma[c] = (float)s2;
  g_beta[c] = (float)s1;
  const float kk = gamma[c] * rstd[c];
  const double m2 = training ? s1 / (double)R : 0.0, m3 = training ? s2 / (double)R : 0.0, rs = (double)rstd[c];
  k1[c] = kk;                                                  // ka
  k2[c] = (float)(-(double)kk * rs * m3);                      // kb
  k3[c] = (float)((double)kk * ((double)mean[c] * rs * m3 - m2));  // kc
  if (g_bias) g_bias[c] = training ? 0.f : kk * (float)s1;
}

// ---- layer 4, C3 == 128 (the production width), round 5: SIXTEEN lanes per row, each moving its 8 consecutive channels as one
// 16-byte (bf16) / two 16-byte (fp32) loads, so a wave-level load carries four whole rows; the cross-lane sums stay inside a
// 16-lane DPP row (quad_perm / row_half_mirror / row_mirror adds - no ds_bpermute).  The kernels above move 8 / 4 bytes per
// lane and reduce a row with 15 LDS-crossbar shuffles (l4_fwd: a third of its cycles stalled on LDS issue) or issue one
// 256-byte load instruction per row (l4_bwd: 3 M load instructions at 16 050 x 64 rows): 110 + 121 us against ~50 + 50 us of
// HBM time (profiles/r05_kernels.md).  Same arithmetic per element; the backward's per-channel sums are fp32 per lane (at most 64 rows) and fp64 across lanes and blocks instead of fp64 per row.
__device__ __forceinline__ void ld8act(const float* p, float* o) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8act(const bfraw* p, float* o) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  o[0] = bf_lo(v.x); o[1] = bf_hi(v.x); o[2] = bf_lo(v.y); o[3] = bf_hi(v.y);
  o[4] = bf_lo(v.z); o[5] = bf_hi(v.z); o[6] = bf_lo(v.w); o[7] = bf_hi(v.w);
}
__device__ __forceinline__ float row16_sum(float v) {  // sum over the 16 lanes of a DPP row, every lane gets it
  v += obman_dpp<0xB1>(v);
  v += obman_dpp<0x4E>(v);
  v += obman_dpp<0x141>(v);
  v += obman_dpp<0x140>(v);
  return v;
}
constexpr int L4W_C = 128;
template <class HT>
__global__ __launch_bounds__(256) void l4w_fwd_kernel(const HT* __restrict__ H3, const float* __restrict__ s3, const float* __restrict__ t3,
                                                      const float* __restrict__ W4, const float* __restrict__ b4, float f, long R,
                                                      float* __restrict__ out) {
  const int tid = threadIdx.x, sub = tid & 15, c0 = sub * 8;
  float cs[8], ct[8], w0[8], w1[8], w2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cs[j] = s3[c0 + j]; ct[j] = t3[c0 + j];
    w0[j] = W4[c0 + j]; w1[j] = W4[L4W_C + c0 + j]; w2[j] = W4[2 * L4W_C + c0 + j];
  }
  const float bb0 = b4[0], bb1 = b4[1], bb2 = b4[2];
  const long r0 = ((long)blockIdx.x * 256 + tid) >> 4, stride = ((long)gridDim.x * 256) >> 4;
  for (long rq = r0; rq < R; rq += 4 * stride) {  // four rows per lane and step, every load issued before the first use
    float hq[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long rr = rq + u * stride;
      ld8act(H3 + (size_t)(rr < R ? rr : rq) * L4W_C + c0, hq[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long r = rq + u * stride;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = fmaxf(__fmaf_rn(cs[j], hq[u][j], ct[j]), 0.f);
        a0 = __fmaf_rn(a, w0[j], a0); a1 = __fmaf_rn(a, w1[j], a1); a2 = __fmaf_rn(a, w2[j], a2);
      }
      a0 = row16_sum(a0); a1 = row16_sum(a1); a2 = row16_sum(a2);
      if (sub < 3 && r < R) out[r * 3 + sub] = f * ((sub == 0 ? a0 : (sub == 1 ? a1 : a2)) + (sub == 0 ? bb0 : (sub == 1 ? bb1 : bb2)));
    }
  }
}
// sums [blocks][128][2] fp64 (S1 = sum gy3, S2 = sum gy3 * xhat3), gw [blocks][3 * 128 + 4] as l4_bwd_kernel.  Block = 256 threads =
// 16 row lanes x 16 channel octets; rows [blockIdx.x * rows_per_blk, + rows_per_blk), rows_per_blk <= L4W_MAX_ROWS so that a lane adds
// at most 64 rows in fp32 before the fixed-order fp64 sum over the block's 16 row lanes (and, outside, over the blocks).
constexpr int L4W_MAX_ROWS = 1024;
__device__ __forceinline__ void unpack8act(const u32x4& v, float* o) {
  o[0] = bf_lo(v.x); o[1] = bf_hi(v.x); o[2] = bf_lo(v.y); o[3] = bf_hi(v.y);
  o[4] = bf_lo(v.z); o[5] = bf_hi(v.z); o[6] = bf_lo(v.w); o[7] = bf_hi(v.w);
}
template <class HT> struct L4Raw;
template <> struct L4Raw<bfraw> {
  u32x4 v;
  __device__ __forceinline__ void load(const bfraw* p) { v = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void get(float* o) const { unpack8act(v, o); }
  __device__ __forceinline__ void get2(f32x2v* o) const {
    o[0] = f32x2v{bf_lo(v.x), bf_hi(v.x)}; o[1] = f32x2v{bf_lo(v.y), bf_hi(v.y)};
    o[2] = f32x2v{bf_lo(v.z), bf_hi(v.z)}; o[3] = f32x2v{bf_lo(v.w), bf_hi(v.w)};
  }
};
template <> struct L4Raw<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
  __device__ __forceinline__ void get(float* o) const { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; }
  __device__ __forceinline__ void get2(f32x2v* o) const {
    o[0] = f32x2v{a.x, a.y}; o[1] = f32x2v{a.z, a.w}; o[2] = f32x2v{b.x, b.y}; o[3] = f32x2v{b.z, b.w};
  }
};
// The per-element arithmetic runs on channel PAIRS (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): ~17 vector instructions per
// element made this pass VALU-bound (r06: 118 us at R = 1.03 M rows for 263 MB of bf16 activations, 364 us at 4.1 M), the packed
// form issues ~9.  Element-wise the operations and their order are the scalar ones.
template <class HT>
__global__ __launch_bounds__(256, 2) void l4w_bwd_kernel(const float* __restrict__ G, const HT* __restrict__ H3, const float* __restrict__ s3,
                                                         const float* __restrict__ t3, const float* __restrict__ mean3,
                                                         const float* __restrict__ rstd3, const float* __restrict__ W4, float f, long R,
                                                         int rows_per_blk, double* __restrict__ sums, float* __restrict__ gw) {
  const int tid = threadIdx.x, sub = tid & 15, rl = tid >> 4, c0 = sub * 8;
  const long rbeg = (long)blockIdx.x * rows_per_blk, rend = min(R, rbeg + rows_per_blk);
  f32x2v cs[4], ct[4], cm[4], cr[4], w0[4], w1[4], w2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + 2 * j;
    cs[j] = f32x2v{s3[c], s3[c + 1]}; ct[j] = f32x2v{t3[c], t3[c + 1]}; cm[j] = f32x2v{mean3[c], mean3[c + 1]};
    cr[j] = f32x2v{rstd3[c], rstd3[c + 1]};
    w0[j] = f32x2v{W4[c], W4[c + 1]}; w1[j] = f32x2v{W4[L4W_C + c], W4[L4W_C + c + 1]}; w2[j] = f32x2v{W4[2 * L4W_C + c], W4[2 * L4W_C + c + 1]};
  }
  f32x2v p1[4], p2[4], ga[4][3];
  float gb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) { p1[j] = f32x2v{0.f, 0.f}; p2[j] = p1[j]; ga[j][0] = ga[j][1] = ga[j][2] = p1[j]; }
  // two rows per lane and step (16 apart); the NEXT step's rows are requested before this step's arithmetic (a lane's rows keep
  // their order: rq, rq + 16, rq + 32, ..)
  float gq[2][3], gn[2][3];
  L4Raw<HT> hq[2], hn[2];
  auto request = [&](long rq, L4Raw<HT>(&h)[2], float(&g)[2][3]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long r = rq + 16 * u;
      const bool ok = r < rend;
      const long rr = ok ? r : rbeg;
      g[u][0] = ok ? G[rr * 3] : 0.f; g[u][1] = ok ? G[rr * 3 + 1] : 0.f; g[u][2] = ok ? G[rr * 3 + 2] : 0.f;  // g = 0: the row adds nothing
      h[u].load(H3 + (size_t)rr * L4W_C + c0);
    }
  };
  request(rbeg + rl, hq, gq);
  for (long rq = rbeg + rl; rq < rend; rq += 32) {
    request(rq + 32, hn, gn);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float g0 = f * gq[u][0], g1 = f * gq[u][1], g2 = f * gq[u][2];
      f32x2v hv[4];
      hq[u].get2(hv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2v y = __builtin_elementwise_fma(cs[j], hv[j], ct[j]);
        const f32x2v d = __builtin_elementwise_fma(w2[j], f32x2v{g2, g2}, __builtin_elementwise_fma(w1[j], f32x2v{g1, g1}, w0[j] * g0));
        const f32x2v a = f32x2v{fmaxf(y[0], 0.f), fmaxf(y[1], 0.f)};
        const f32x2v gy = f32x2v{y[0] > 0.f ? d[0] : 0.f, y[1] > 0.f ? d[1] : 0.f};
        p1[j] += gy;
        p2[j] = __builtin_elementwise_fma(gy, (hv[j] - cm[j]) * cr[j], p2[j]);
        ga[j][0] = __builtin_elementwise_fma(f32x2v{g0, g0}, a, ga[j][0]);
        ga[j][1] = __builtin_elementwise_fma(f32x2v{g1, g1}, a, ga[j][1]);
        ga[j][2] = __builtin_elementwise_fma(f32x2v{g2, g2}, a, ga[j][2]);
      }
      gb[0] += g0; gb[1] += g1; gb[2] += g2;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { hq[u] = hn[u]; gq[u][0] = gn[u][0]; gq[u][1] = gn[u][1]; gq[u][2] = gn[u][2]; }
  }
  // 16 row lanes hold partials of the same channels: fixed-order sum through LDS (row lane 0 adds 1 .. 15 in order), fp64 for S1 / S2
  __shared__ float sp[15][16][16];  // [row lane - 1][octet][p1[8] | p2[8]]
  __shared__ float sf[15][16][24];
  __shared__ float sg[16][3];
  if (rl > 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sp[rl - 1][sub][j] = p1[j >> 1][j & 1]; sp[rl - 1][sub][8 + j] = p2[j >> 1][j & 1]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sf[rl - 1][sub][3 * j] = ga[j >> 1][0][j & 1]; sf[rl - 1][sub][3 * j + 1] = ga[j >> 1][1][j & 1]; sf[rl - 1][sub][3 * j + 2] = ga[j >> 1][2][j & 1];
    }
  }
  if (sub == 0) { sg[rl][0] = gb[0]; sg[rl][1] = gb[1]; sg[rl][2] = gb[2]; }
  __syncthreads();
  if (rl == 0) {
    double S1[8], S2[8];
    float gs[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      S1[j] = (double)p1[j >> 1][j & 1]; S2[j] = (double)p2[j >> 1][j & 1];
      gs[j][0] = ga[j >> 1][0][j & 1]; gs[j][1] = ga[j >> 1][1][j & 1]; gs[j][2] = ga[j >> 1][2][j & 1];
    }
    for (int w = 0; w < 15; ++w) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { S1[j] += (double)sp[w][sub][j]; S2[j] += (double)sp[w][sub][8 + j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { gs[j][0] += sf[w][sub][3 * j]; gs[j][1] += sf[w][sub][3 * j + 1]; gs[j][2] += sf[w][sub][3 * j + 2]; }
    }
    float* dst = gw + (size_t)blockIdx.x * (3 * L4W_C + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sums[((size_t)blockIdx.x * L4W_C + c0 + j) * 2] = S1[j];
      sums[((size_t)blockIdx.x * L4W_C + c0 + j) * 2 + 1] = S2[j];
      dst[c0 + j] = gs[j][0]; dst[L4W_C + c0 + j] = gs[j][1]; dst[2 * L4W_C + c0 + j] = gs[j][2];
    }
    if (sub < 3) {
      float e = 0.f;
      for (int w = 0; w < 16; ++w) e += sg[w][sub];
      dst[3 * L4W_C + sub] = e;
    }
  }
}

// partial layer-4 weight gradients [blocks][3*C3+4] -> gW4 [3][C3], gb4 [3].  One wave per output element.
__global__ __launch_bounds__(256) void l4_bwd_finalize_kernel(const float* __restrict__ gw, int blocks, int C3, float* __restrict__ gW4,
                                                              float* __restrict__ gb4) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= 3 * C3 + 3) return;
  double s = 0;
  for (int b = lane; b < blocks; b += 64) s += gw[(size_t)b * (3 * C3 + 4) + i];
  s = wave_sum_f64(s);
  if (lane != 0) return;
  if (i < 3 * C3) gW4[i] = (float)s; else gb4[i - 3 * C3] = (float)s;
}

// P[b,c] = sum_n GY1[b,n,c] and Q[n,c] = sum_b GY1[b,n,c] from ONE read of GY1 (the largest tensor of the backward).
// Block = (tile of 16*S vertices, group of 16 samples, 256 channels); thread = channel (coalesced rows).  Per 16-vertex
// sub-tile a thread issues 16 independent loads per sample and keeps 16 vertex sums + 16 sample sums in registers;
// partial results go to Qp[group][n][c] and Pp[tile][b][c] and are summed in index order by l1_reduce2 (deterministic).
constexpr int L1_V = 16, L1_B = 16;
__global__ __launch_bounds__(256) void l1_reduce_kernel(const float* __restrict__ GY1, int ld1, int B, int N, int C1, int S,
                                                        float* __restrict__ Pp, float* __restrict__ Qp) {
  const int c = blockIdx.z * 256 + threadIdx.x;
  if (c >= C1) return;
  const int tile = blockIdx.x, g = blockIdx.y, b0 = g * L1_B;
  float pb[L1_B];
#pragma unroll
  for (int k = 0; k < L1_B; ++k) pb[k] = 0.f;
  for (int sub = 0; sub < S; ++sub) {
    const int n0 = (tile * S + sub) * L1_V;
    if (n0 >= N) break;
    float q[L1_V];
#pragma unroll
    for (int i = 0; i < L1_V; ++i) q[i] = 0.f;
#pragma unroll
    for (int k = 0; k < L1_B; ++k) {
      if (b0 + k < B) {
        const float* src = GY1 + ((size_t)(b0 + k) * N + n0) * ld1 + c;
        float v[L1_V];
#pragma unroll
        for (int i = 0; i < L1_V; ++i) v[i] = n0 + i < N ? src[(size_t)i * ld1] : 0.f;
#pragma unroll
        for (int i = 0; i < L1_V; ++i) { q[i] += v[i]; pb[k] += v[i]; }
      }
    }
#pragma unroll
    for (int i = 0; i < L1_V; ++i)
      if (n0 + i < N) Qp[((size_t)g * N + n0 + i) * ld1 + c] = q[i];
  }
#pragma unroll
  for (int k = 0; k < L1_B; ++k)
    if (b0 + k < B) Pp[((size_t)tile * B + b0 + k) * ld1 + c] = pb[k];
}
// rows 0..B-1: P[b,c] = sum_tile Pp[tile][b][c]; rows B..B+N-1: Q[n,c] = sum_group Qp[group][n][c]
__global__ __launch_bounds__(256) void l1_reduce2_kernel(const float* __restrict__ Pp, const float* __restrict__ Qp, int ld1, int B, int N,
                                                         int C1, int tiles, int groups, float* __restrict__ P, float* __restrict__ Q) {
  const int c = blockIdx.y * 256 + threadIdx.x, row = blockIdx.x;
  if (c >= C1) return;
  float s = 0.f;
  if (row < B) {
    for (int t = 0; t < tiles; t += 8) {  // eight partials requested at once, added in tile order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = t + u < tiles ? Pp[((size_t)(t + u) * B + row) * ld1 + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (t + u < tiles) s += v[u];
    }
    P[(size_t)row * ld1 + c] = s;
  } else {
    const int n = row - B;
    for (int g = 0; g < groups; ++g) s += Qp[((size_t)g * N + n) * ld1 + c];
    Q[(size_t)n * ld1 + c] = s;
  }
}

// With ONE sample group (B <= 64 on the second-generation data-gradient kernels) Q IS the single partial: only the B rows of P are
// summed and the consumers read Qp in place (r06: the copy was 37 us at N = 16 050, 93 us at 64 050).  Returns where Q lives.
const float* launch_l1_reduce2(const float* Pp, const float* Qp, int ld1, int B, int N, int C1, int tiles, int groups, float* P, float* Q,
                               hipStream_t st) {
  const int rows = groups == 1 ? B : B + N;
  l1_reduce2_kernel<<<dim3(rows, obman_cdiv(C1, 256)), 256, 0, st>>>(Pp, Qp, ld1, B, N, C1, tiles, groups, P, Q);
  return groups == 1 ? Qp : Q;
}

// BN-1 backward in factored form.  Block = 64 channels x 16 row groups (coalesced along channels; the rows of P/Fx
// (B) and Q/Gx (N) are split over the 16 groups and merged through LDS in group order => deterministic).
// Produces g_gamma1, g_beta1, dF [B,ld1], dG [N,ld1], g_b1 = sum_b dF, and gW1[:, 0:3] = dG^T grid.
constexpr int L1F_CH = 64;  // 16 channels x 64 row groups per block (33 blocks instead of 9) measured 151 us instead of 35 at 642 points: kept at 64
__global__ __launch_bounds__(1024) void l1_finalize_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                           const float* __restrict__ Gx, const float* __restrict__ Fx, int ld1,
                                                           int B, int N, int C1, int training, const float* __restrict__ gamma,
                                                           const float* __restrict__ rstd1, const float* __restrict__ grid,
                                                           float* __restrict__ g_gamma, float* __restrict__ g_beta,
                                                           float* __restrict__ g_b1, float* __restrict__ gW1, float* __restrict__ dF,
                                                           float* __restrict__ dG) {
  constexpr int CH = L1F_CH, RG = 1024 / L1F_CH;
  const int cl = threadIdx.x % CH, rg = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  const bool ok = c < C1;
  __shared__ double red[RG][4][CH];
  __shared__ float redf[RG][4][CH];
  double s1 = 0, s2 = 0, sgx = 0, sfx = 0;
  if (ok) {
    for (int b = rg; b < B; b += RG) {
      const double p = P[(size_t)b * ld1 + c], fx = Fx[(size_t)b * ld1 + c];
      s1 += p; s2 += p * fx; sfx += fx;
    }
    // eight rows per round, all sixteen loads issued before the first use: the plain loop exposed one L2 round trip per row
    // (40 rows per thread at 642 vertices: 35 us for a kernel that moves 2.7 MB)
    int n = rg;
    for (; n + 7 * RG < N; n += 8 * RG) {
      float gx[8], q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { gx[u] = Gx[(size_t)(n + u * RG) * ld1 + c]; q[u] = Q[(size_t)(n + u * RG) * ld1 + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s2 += (double)gx[u] * (double)q[u]; sgx += (double)gx[u]; }
    }
    for (; n < N; n += RG) {
      const double gx = Gx[(size_t)n * ld1 + c];
      s2 += gx * (double)Q[(size_t)n * ld1 + c];
      sgx += gx;
    }
  }
  red[rg][0][cl] = s1; red[rg][1][cl] = s2; red[rg][2][cl] = sgx; red[rg][3][cl] = sfx;
  __syncthreads();
  s1 = s2 = sgx = sfx = 0;
  for (int g = 0; g < RG; ++g) { s1 += red[g][0][cl]; s2 += red[g][1][cl]; sgx += red[g][2][cl]; sfx += red[g][3][cl]; }
  const double R = (double)B * N;
  const float k1 = ok ? gamma[c] * rstd1[c] : 0.f;
  const float k2 = training ? (float)(s1 / R) : 0.f, k3 = training ? (float)(s2 / R) : 0.f;
  float gb = 0.f, w0 = 0.f, w1 = 0.f, w2 = 0.f;
  if (ok) {
    for (int b = rg; b < B; b += RG) {
      const float v = k1 * (P[(size_t)b * ld1 + c] - N * k2 - k3 * ((float)sgx + N * Fx[(size_t)b * ld1 + c]));
      dF[(size_t)b * ld1 + c] = v;
      gb += v;
    }
    int n = rg;
    for (; n + 7 * RG < N; n += 8 * RG) {  // same rows in the same order as the tail loop below, loads first
      float q[8], gx[8], g0[8], g1[8], g2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int nn = n + u * RG;
        q[u] = Q[(size_t)nn * ld1 + c]; gx[u] = Gx[(size_t)nn * ld1 + c];
        g0[u] = grid[nn * 3]; g1[u] = grid[nn * 3 + 1]; g2[u] = grid[nn * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float v = k1 * (q[u] - B * k2 - k3 * (B * gx[u] + (float)sfx));
        dG[(size_t)(n + u * RG) * ld1 + c] = v;
        w0 = __fmaf_rn(v, g0[u], w0); w1 = __fmaf_rn(v, g1[u], w1); w2 = __fmaf_rn(v, g2[u], w2);
      }
    }
    for (; n < N; n += RG) {
      const float v = k1 * (Q[(size_t)n * ld1 + c] - B * k2 - k3 * (B * Gx[(size_t)n * ld1 + c] + (float)sfx));
      dG[(size_t)n * ld1 + c] = v;
      w0 = __fmaf_rn(v, grid[n * 3], w0); w1 = __fmaf_rn(v, grid[n * 3 + 1], w1); w2 = __fmaf_rn(v, grid[n * 3 + 2], w2);
    }
  }
  redf[rg][0][cl] = gb; redf[rg][1][cl] = w0; redf[rg][2][cl] = w1; redf[rg][3][cl] = w2;
  __syncthreads();
  if (rg == 0 && ok) {
    gb = w0 = w1 = w2 = 0.f;
    for (int g = 0; g < RG; ++g) { gb += redf[g][0][cl]; w0 += redf[g][1][cl]; w1 += redf[g][2][cl]; w2 += redf[g][3][cl]; }
    g_gamma[c] = (float)s2;
    g_beta[c] = (float)s1;
    g_b1[c] = gb;
    gW1[(size_t)c * C1] = w0; gW1[(size_t)c * C1 + 1] = w1; gW1[(size_t)c * C1 + 2] = w2;
  }
}

// The same computation for large templates (N > L1_SPLIT_N: 25 patches = 16 050 vertices), where nine 64-channel blocks
// would crawl over N rows on nine CUs (0.86 ms at configs[2]).  dG[n,c] = k1 (Q - B k2 - k3 (B Gx + sum_b Fx)) is affine in the
// per-channel constants, so gW1[c, 0:3] = sum_n dG[n,c] grid[n] follows from row sums that need NO constant:
//   l1_seg_stats : ONE pass over Q and Gx, per (row segment, 64 channels) block, fp64:
//                  sum_n Gx Q, sum_n Gx, sum_n Q grid_k, sum_n Gx grid_k (k < 3)   -> seg[segment][8][C]
//   l1_seg_final : per channel, the segment sums in fixed order, the B sample rows, sum_n grid; constants, dF, g_b1, g_gamma1,
//                  g_beta1 and gW1[:, 0:3] (combined in fp64)
// (Until r06 a second pass re-read Q and Gx to form dG . grid with the constants in hand, and a third launch summed its
// segment partials: 63 + 111 + 11 us at N = 64 050.)
constexpr int L1_SPLIT_N = 2048, L1_SEG_ROWS = 512, L1_SEG_Q = 8;
__global__ __launch_bounds__(1024) void l1_seg_stats_kernel(const float* __restrict__ Q, const float* __restrict__ Gx,
                                                            const float* __restrict__ grid, int ld1, int N, int C1, double* __restrict__ seg) {
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const int n0 = blockIdx.y * L1_SEG_ROWS, n1 = min(N, n0 + L1_SEG_ROWS);
  __shared__ double red[16][4][64];
  double a[L1_SEG_Q];
#pragma unroll
  for (int q = 0; q < L1_SEG_Q; ++q) a[q] = 0;
  if (c < C1)
    for (int n = n0 + rg; n < n1; n += 64) {  // four rows' operands requested before the first sum; a lane's rows keep their order
      float qv[4], gv[4], g0[4], g1[4], g2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int nn = n + 16 * u < n1 ? n + 16 * u : n;
        qv[u] = Q[(size_t)nn * ld1 + c]; gv[u] = Gx[(size_t)nn * ld1 + c];
        g0[u] = grid[nn * 3]; g1[u] = grid[nn * 3 + 1]; g2[u] = grid[nn * 3 + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (n + 16 * u < n1) {
          const double q = qv[u], gx = gv[u];
          a[0] += gx * q; a[1] += gx;
          a[2] += q * (double)g0[u]; a[3] += q * (double)g1[u]; a[4] += q * (double)g2[u];
          a[5] += gx * (double)g0[u]; a[6] += gx * (double)g1[u]; a[7] += gx * (double)g2[u];
        }
      }
    }
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // the 16 row groups meet in LDS in group order, four quantities at a time
    if (half) __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) red[rg][q][cl] = a[4 * half + q];
    __syncthreads();
    if (rg < 4 && c < C1) {
      double t = 0;
      for (int g = 0; g < 16; ++g) t += red[g][rg][cl];
      seg[((size_t)blockIdx.y * L1_SEG_Q + 4 * half + rg) * C1 + c] = t;
    }
  }
}
__global__ __launch_bounds__(1024) void l1_seg_final_kernel(const float* __restrict__ P, const float* __restrict__ Fx, int ld1, int B, int N,
                                                            int C1, int training, const float* __restrict__ gamma,
                                                            const float* __restrict__ rstd1, const float* __restrict__ grid,
                                                            const double* __restrict__ seg, int nseg, float* __restrict__ g_gamma,
                                                            float* __restrict__ g_beta, float* __restrict__ g_b1, float* __restrict__ dF,
                                                            float* __restrict__ gW1) {
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  const bool ok = c < C1;
  __shared__ double sgrid[16][3];
  __shared__ float redf[16][64];
  // sum_n grid[n] (channel independent): every thread a strided share, waves meet in LDS in wave order
  {
    double t0 = 0, t1 = 0, t2 = 0;
    for (int n = threadIdx.x; n < N; n += 1024) { t0 += grid[n * 3]; t1 += grid[n * 3 + 1]; t2 += grid[n * 3 + 2]; }
    t0 = wave_sum_f64(t0); t1 = wave_sum_f64(t1); t2 = wave_sum_f64(t2);
    if (cl == 0) { sgrid[rg][0] = t0; sgrid[rg][1] = t1; sgrid[rg][2] = t2; }
  }
  // channel sums, one quantity per row group (fixed order): groups 0 .. 7 the eight segment quantities, 8 .. 10 the sample sums
  __shared__ double tot[11][64];
  if (ok && rg < 11) {
    double t = 0;
    if (rg < L1_SEG_Q) {
      for (int sg = 0; sg < nseg; sg += 8) {  // eight partials requested at once, added in segment order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sg + u < nseg ? seg[((size_t)(sg + u) * L1_SEG_Q + rg) * C1 + c] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
    } else {
      for (int b = 0; b < B; b += 8) {
        float pv[8], fv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bb = b + u < B ? b + u : b;
          pv[u] = P[(size_t)bb * ld1 + c]; fv[u] = Fx[(size_t)bb * ld1 + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (b + u < B) t += rg == 8 ? (double)pv[u] : (rg == 9 ? (double)pv[u] * (double)fv[u] : (double)fv[u]);
      }
    }
    tot[rg][cl] = t;
  }
  __syncthreads();
  double sg0 = 0, sg1 = 0, sg2 = 0;
  for (int g = 0; g < 16; ++g) { sg0 += sgrid[g][0]; sg1 += sgrid[g][1]; sg2 += sgrid[g][2]; }
  const double s1 = ok ? tot[8][cl] : 0.0, sfx = ok ? tot[10][cl] : 0.0;
  const double s2 = ok ? tot[9][cl] + tot[0][cl] : 0.0;  // sum_b P Fx + sum_n Gx Q
  const double sgx = ok ? tot[1][cl] : 0.0;
  const double R = (double)B * N;
  const float k1 = ok ? gamma[c] * rstd1[c] : 0.f;
  const float k2 = training ? (float)(s1 / R) : 0.f, k3 = training ? (float)(s2 / R) : 0.f;
  float gb = 0.f;
  if (ok)
    for (int b = rg; b < B; b += 16) {
      const float v = k1 * (P[(size_t)b * ld1 + c] - N * k2 - k3 * ((float)sgx + N * Fx[(size_t)b * ld1 + c]));
      dF[(size_t)b * ld1 + c] = v;
      gb += v;
    }
  redf[rg][cl] = gb;
  __syncthreads();
  if (rg == 0 && ok) {
    gb = 0.f;
    for (int g = 0; g < 16; ++g) gb += redf[g][cl];
    g_gamma[c] = (float)s2;
    g_beta[c] = (float)s1;
    g_b1[c] = gb;
    // gW1[c,k] = k1 (sum_n Q grid_k - B k2 sum_n grid_k - k3 (B sum_n Gx grid_k + sum_b Fx sum_n grid_k))
    const double sgk[3] = {sg0, sg1, sg2};
#pragma unroll
    for (int k = 0; k < 3; ++k)
      gW1[(size_t)c * C1 + k] =
          (float)((double)k1 * (tot[2 + k][cl] - (double)B * k2 * sgk[k] - (double)k3 * ((double)B * tot[5 + k][cl] + sfx * sgk[k])));
  }
}

// g_feat[b,k] = sum_c dF[b,c] * W1[c, 3+k]: 64 samples x 512 features, contraction 515 - far too skinny for the tiled
// MFMA kernel (one row block, 17 serial k-tiles); thread = (b, k), dF broadcast within the wave, W1 coalesced over k.
__global__ __launch_bounds__(256) void gfeat_kernel(const float* __restrict__ dF, int ld1, const float* __restrict__ W1, int C1, int B,
                                                    float* __restrict__ g_feat) {
  // g_feat[b,k] = sum_c dF[b,c] W1[c,3+k].  Block = 64 features x 4 channel quarters (one per wave), sample = blockIdx.y: a
  // quarter is ~129 channels = 8 batches of 16 independent strided loads (one thread per feature walking all 515 channels was
  // 33 dependent batches: 16 us at 64 samples); the quarters meet in LDS in fixed order.
  __shared__ float red[3][64];
  const int Cf = C1 - 3, kk = threadIdx.x & 63, cq = threadIdx.x >> 6, k = blockIdx.x * 64 + kk, b = blockIdx.y;
  const int per = (C1 + 3) / 4, cbeg = cq * per, cend = cbeg + per < C1 ? cbeg + per : C1;
  const float* d = dF + (size_t)b * ld1;
  const float* w = W1 + 3 + (k < Cf ? k : 0);
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  int c = cbeg;
  for (; c + 15 < cend; c += 16) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u] = __fmaf_rn(d[c + u], w[(size_t)(c + u) * C1], acc[u]);
  }
  for (; c < cend; ++c) acc[0] = __fmaf_rn(d[c], w[(size_t)c * C1], acc[0]);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int u = 0; u < 16; u += 4) { a0 += acc[u]; a1 += acc[u + 1]; a2 += acc[u + 2]; a3 += acc[u + 3]; }
  const float part = (a0 + a1) + (a2 + a3);
  if (cq) red[cq - 1][kk] = part;
  __syncthreads();
  if (cq == 0 && k < Cf) g_feat[(size_t)b * Cf + k] = ((part + red[0][kk]) + red[1][kk]) + red[2][kk];
}

// out[m*ldo + off + n] = sum_c part[c][m][n].  Fixed summation order (run-to-run reproducible): wave q of a block sums the chunks
// q, q+4, q+8, ... of 64 consecutive elements, eight independent loads in flight at a time, and the four partial sums are
// combined as ((s0 + s1) + s2) + s3.  (One thread per element walking all chunks with a load -> add dependency per chunk took
// 33 us for 56 chunks x 132 k elements = 30 MB: ~0.9 TB/s, latency-bound.)
constexpr int RTN_ELEMS = 64;
// `transposed`: the product was formed as [Nc-major] = out^T (the bf16 layer-2 weight gradient), out[n*ldo + off + m] receives it.
__global__ __launch_bounds__(256) void reduce_tn_kernel(const float* __restrict__ part, int chunks, int M, int Nc, int ldo, int off,
                                                        float* __restrict__ out, int transposed = 0) {
  __shared__ float red[3][RTN_ELEMS];
  const int q = threadIdx.x >> 6;
  const long total = (long)M * Nc, i = (long)blockIdx.x * RTN_ELEMS + (threadIdx.x & 63);
  const bool ok = i < total;
  const float* src = part + (ok ? i : 0);
  float s = 0.f;
  int c = q;
  for (; c + 28 < chunks; c += 32) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(c + 4 * j) * total];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
  }
  for (; c < chunks; c += 4) s += src[(size_t)c * total];
  if (q) red[q - 1][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && ok) {
    s = ((s + red[0][threadIdx.x]) + red[1][threadIdx.x]) + red[2][threadIdx.x];
    if (transposed) out[(size_t)(i % Nc) * ldo + off + (i / Nc)] = s;
    else out[(size_t)(i / Nc) * ldo + off + (i % Nc)] = s;
  }
}

}  // namespace dec

// ================================================================================================ orchestration
namespace {
using namespace dec;

struct Dims {
  int B, N, C1, C2, C3, ld1, ld2, ld3, rb;  // rb = row blocks of the rows-GEMMs
  int bf16;                                 // contraction on the bf16 matrix pipe (operands rounded to bf16, fp32 accumulate)
  int ps;                                   // per-sample grid [B,N,3] (AtlasBranch.forward): Gx is [R,ld1]
  long R;
};
inline int kpad(int K) { return (K + BK - 1) / BK * BK; }       // k extent of a bf16 weight image (zero padded)
inline int wide_wn(int Nc) { return Nc > 128 ? 5 : 2; }         // 32-column MFMA tiles per wave of the bf16 kernels (block = 64 * WN columns)
Dims dims_of(const obman_pointgen_params* p) {
  Dims d;
  d.B = p->B; d.N = p->N; d.C1 = p->C1; d.C2 = p->C1 / 2; d.C3 = p->C1 / 4;
  d.ld1 = pad16(d.C1); d.ld2 = pad16(d.C2); d.ld3 = pad16(d.C3);
  d.R = (long)d.B * d.N;
  d.rb = (int)((d.R + BM - 1) / BM);
  d.bf16 = p->mfma_bf16 ? 1 : 0;
  if (d.bf16) {  // the layer-2 GEMM of the bf16 flavour runs on (8 samples x 16 vertices) row tiles: a few more blocks
    const int tb = ((d.N + 15) / 16) * ((d.B + 7) / 8);
    if (tb > d.rb) d.rb = tb;
  }
  d.ps = p->grid_per_sample ? 1 : 0;
  return d;
}
constexpr int L4_ROWS = 32;   // rows per block of the layer-4 backward (1284 blocks at 64 x 642 points)
constexpr int TN_CHUNK_ROWS = 1024;  // rows per split-K chunk of the weight-gradient GEMMs

// The branch-free k loop of the rows2 kernels (decoder_rows2.h) requests up to R2Depth k-steps past the end of a row; for the
// LAST row of an operand that is up to 8 x 64 bytes past the end of its array, and the scalar k offset of a raw buffer load is not
// part of the hardware range check.  Every operand lives inside one of the two arenas below and is followed by other arrays, so
// those (never consumed) reads stay inside the arena; the tail padding keeps that true whatever the order of the arrays.
constexpr long WS_TAIL_FLOATS = 256;  // 1 KB
// forward workspace (kept for the backward), float offsets
struct FwdWs {
  long Gx, Fx, Gy, Fy, gm1, mean1, rstd1, H2, mean2, rstd2, s2, t2, H3, mean3, rstd3, s3, t3, moments, mred, wb2, wb3, total;
};
FwdWs fwd_ws(const Dims& d) {
  FwdWs w; long o = 0;
  auto take = [&](long n) { long at = o; o += (n + 15) / 16 * 16; return at; };
  w.Gx = take((d.ps ? d.R : (long)d.N) * d.ld1); w.Fx = take((long)d.B * d.ld1); w.mean1 = take(d.ld1); w.rstd1 = take(d.ld1);
  const int act = d.bf16 ? 2 : 1;  // bf16 flavour: activations stored as bf16 (two per float slot)
  w.H2 = take(d.R * d.ld2 / act); w.mean2 = take(d.ld2); w.rstd2 = take(d.ld2); w.s2 = take(d.ld2); w.t2 = take(d.ld2);
  w.H3 = take(d.R * d.ld3 / act); w.mean3 = take(d.ld3); w.rstd3 = take(d.ld3); w.s3 = take(d.ld3); w.t3 = take(d.ld3);
  w.moments = take((long)d.rb * d.C2 * 2 * 2);  // doubles
  w.mred = take(d.rb > PRE_MIN_ROWS ? (long)PRE_SEGMENTS_NARROW * d.C2 * 2 * 2 : 0);  // doubles: pre-reduced moments
  w.wb2 = take(d.bf16 ? ((long)d.C2 * kpad(d.C1) + 1) / 2 : 0);  // bf16 [C2][kpad(C1)] image of W2
  w.wb3 = take(d.bf16 ? ((long)d.C3 * kpad(d.C2) + 1) / 2 : 0);
  // pre-scaled layer-1 factors (Gy = gamma Gx: l1_fill_kernel, Fy = gamma Fx + beta: prep_kernel): the bf16 flavour and the second-generation fp32 kernels (shared template grid)
  w.Gy = take(d.bf16 || !d.ps ? (long)(d.N + 1) * d.ld1 : 0);
  w.Fy = take(d.bf16 || !d.ps ? (long)d.B * d.ld1 : 0);
  w.gm1 = take(d.ps ? 0 : d.ld1);  // the grid factor's channel means (prep_kernel -> l1_fill_kernel)
  w.total = o + WS_TAIL_FLOATS;
  return w;
}
// rows per split-K chunk of the fp32 weight-gradient kernel.  768 block slots (256 CUs x 3 blocks of 49.7 KB LDS): aim at just
// under TWO full rounds - 27 tiles x 40 chunks = 1080 blocks ran a full round plus a 41 % one; chunk count a multiple of 8
// (whole chunks are dealt to the 8 XCDs), never below 128 rows.
int tn_chunk_rows(int M, int Nc, long R, int bn = BN) {
  const long tiles = (long)tile_split(M, BM).tiles * tile_split(Nc, bn).tiles;
  long want = (2 * 768 / tiles) / 8 * 8;
  if (want < 8) want = 8;
  long rows = (R + want - 1) / want;
  if (rows < 128) rows = 128;
  return (int)rows;
}
// split-K chunks of the bf16 weight-gradient kernel: output tiles x chunks <= 512 = two full rounds of the 256 CUs (one
// 512-thread block per CU), at least two k-tiles per chunk
int tn_bf16_chunks(int M, int Nc, int N, int Bsz, int bn) {
  const long tiles = (long)((M + BM - 1) / BM) * ((Nc + bn - 1) / bn);
  const long ntiles = (long)((Bsz + 7) / 8) * ((N + 7) / 8);
  long want = 512 / tiles;
  if (want < 1) want = 1;
  if (want > (ntiles + 1) / 2) want = (ntiles + 1) / 2;
  if (want < 1) want = 1;
  const long per = (ntiles + want - 1) / want;
  return (int)((ntiles + per - 1) / per);
}
// round 6: the wide (256 x 288) tile of decoder_tn2.h for the transposed layer-2 weight gradient (M = C1 = 515 = 2 x 256 + 3 rows,
// Nc = C2 = 257 columns): full row tiles in tn2w_bf16_kernel, the M % 256 <= 3 remainder rows in gh2_inplace_side_kernel.
// One round of blocks (one 512-thread block per CU).  OBMAN_DEC_TN2W=0: the 128 x 320 tiles of tn2_bf16_kernel (A/B).
struct Tn2wPlan { bool use; int mt, ns, chunks, tiles_per_chunk; };
Tn2wPlan tn2w_plan(int M, int Nc, int N, int Bsz) {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_TN2W"); return e ? atoi(e) : 1; }();
  Tn2wPlan p{false, 0, 0, 0, 0};
  if (!on || M < TW_BM || M % TW_BM > 3 || Nc > TW_BN || Nc <= 160) return p;
  p.mt = M / TW_BM;
  p.ns = M % TW_BM;
  const long ntiles = (long)((Bsz + 7) / 8) * ((N + 3) / 4);
  long want = 256 / p.mt;
  if (want > ntiles) want = ntiles;
  if (want < 1) want = 1;
  p.tiles_per_chunk = (int)((ntiles + want - 1) / want);
  p.chunks = (int)((ntiles + p.tiles_per_chunk - 1) / p.tiles_per_chunk);
  p.use = true;
  return p;
}
long tn2w_part_floats(const Tn2wPlan& p, int Nc) { return (long)p.chunks * p.mt * TW_BM * Nc + (long)GH2S_BLOCKS * 3 * Nc; }
// l1_reduce geometry: vertex sub-tiles per block such that tiles x sample-groups x channel-tiles ~ 1024 blocks
struct L1Geo { int S, tiles, groups; };
L1Geo l1_geo(const Dims& d) {
  L1Geo g;
  if (d.bf16) {  // EpiL1B: one partial P per group of 16 vertices, one partial Q per group of 8 samples; the rows2 dA kernel (EpiL1B2)
    // writes one partial P per persistent block of a sample group (<= CU count, <= N / 4) and one partial Q per 64 samples
    g.S = 1; g.tiles = (d.N + 15) / 16; g.groups = (d.B + 7) / 8;
    const int r2 = (d.N + 3) / 4 < 1024 ? (d.N + 3) / 4 : 1024;
    if (g.tiles < r2) g.tiles = r2;
    return g;
  }
  g.groups = (d.B + L1_B - 1) / L1_B;
  const int sub = (d.N + L1_V - 1) / L1_V, ct = (d.C1 + 255) / 256;
  const int want = 1024 / (g.groups * ct) > 0 ? 1024 / (g.groups * ct) : 1;
  g.S = (sub + want - 1) / want;
  if (g.S < 1) g.S = 1;
  g.tiles = (sub + g.S - 1) / g.S;
  return g;
}
int device_cus();
struct BwdWs {
  long GY2, GY1, sums, sred, k, l4p, l4red, P, Q, Pp, Qp, Ppre, dF, dG, seg, segw, tn, wt2, wt3, total;
  int chunks;
};
BwdWs bwd_ws(const Dims& d) {
  BwdWs w; long o = 0;
  auto take = [&](long n) { long at = o; o += (n + 15) / 16 * 16; return at; };
  w.chunks = (int)((d.R + TN_CHUNK_ROWS - 1) / TN_CHUNK_ROWS);
  const int l4b = (int)((d.R + L4_ROWS - 1) / L4_ROWS);
  w.GY2 = take(d.bf16 ? d.R * d.ld2 / 2 : d.R * d.ld2);
  w.GY1 = take(d.bf16 ? 0 : d.R * d.ld1);  // bf16 flavour: gy1 is never materialised (EpiL1B)
  w.sums = take((long)(d.rb > l4b ? d.rb : l4b) * d.C1 * 2 * 2);
  w.sred = take(l4b > PRE_MIN_ROWS ? (long)PRE_SEGMENTS_NARROW * d.C1 * 2 * 2 : 0);  // doubles
  w.k = take(3 * d.ld1);
  w.l4p = take((long)l4b * (3 * d.C3 + 4));
  w.l4red = take(l4b > PRE_MIN_ROWS ? (long)PRE_SEGMENTS_NARROW * (3 * d.C3 + 4) : 0);
  w.P = take((long)d.B * d.ld1); w.Q = take((long)d.N * d.ld1); w.dF = take((long)d.B * d.ld1); w.dG = take((long)d.N * d.ld1);
  if (d.ps) {  // per-sample grid: partial dF / gW1[:, 0:3] per (sample, chunk of L1PS_ROWS rows)
    const long ch = (d.N + L1PS_ROWS - 1) / L1PS_ROWS;
    w.Pp = take((long)d.B * ch * d.ld1); w.Qp = take((long)d.B * ch * 3 * d.ld1);
  } else {
    const L1Geo g = l1_geo(d);
    long pp = (long)g.tiles * d.B * d.ld1, qp = (long)g.groups * d.N * d.ld1;
    if (!d.bf16) {  // the second-generation fp32 data-gradient kernel writes its own partials (F2EpiL1PQ): <= 8 CUs' waves per sample group
      const long nbg = (d.B + 7) / 8, nvt = (d.N + 3) / 4;
      long wpg = (long)device_cus() * 8 / nbg;
      if (wpg > nvt) wpg = nvt;
      if (wpg < 1) wpg = 1;
      if (wpg * d.B * d.ld1 > pp) pp = wpg * d.B * d.ld1;
      if (nbg * d.N * d.ld1 > qp) qp = nbg * d.N * d.ld1;
    }
    w.Pp = take(pp); w.Qp = take(qp);
  }
  w.Ppre = take(d.bf16 && l1_geo(d).tiles > PRE_MIN_ROWS ? (long)pre_segments((long)d.B * d.ld1) * d.B * d.ld1 : 0);  // same rule as pre_reduce
  {  // large-template layer-1 finalize: per-segment partials
    const long nseg = d.N > L1_SPLIT_N ? (d.N + L1_SEG_ROWS - 1) / L1_SEG_ROWS : 0;
    w.seg = take(nseg * L1_SEG_Q * d.C1 * 2);  // doubles
    w.segw = take(0);
  }
  {  // split-K partials: the largest of the three weight-gradient products
    auto need = [&](int M, int Nc, long R) {
      if (d.bf16 && R == d.R) {  // the bf16 flavour forms the layer-2 product transposed (backward_bf16)
        if (M == d.C2 && Nc == d.C1) { M = d.C1; Nc = d.C2; }
        long n1 = (long)tn_bf16_chunks(M, Nc, d.N, d.B, 64 * wide_wn(Nc)) * M * Nc;
        const Tn2wPlan wp = tn2w_plan(M, Nc, d.N, d.B);
        if (wp.use && tn2w_part_floats(wp, Nc) > n1) n1 = tn2w_part_floats(wp, Nc);
        return n1;
      }
      const int rows = tn_chunk_rows(M, Nc, R, BN);
      return ((R + rows - 1) / rows) * (long)M * Nc;
    };
    long a = need(d.C3, d.C2, d.R), b = need(d.C2, d.C1, d.R), c = need(d.C1, d.C1 - 3, d.B);
    if (b > a) a = b;
    if (c > a) a = c;
    w.tn = take(a);
  }
  w.wt2 = take(d.bf16 ? ((long)d.C1 * kpad(d.C2) + 1) / 2 : 0);  // bf16 [C1][kpad(C2)] image of W2^T (dA GEMM of layer 2)
  w.wt3 = take(d.bf16 ? ((long)d.C2 * kpad(d.C3) + 1) / 2 : 0);
  w.total = o + WS_TAIL_FLOATS;
  return w;
}

template <class AOp, bool B_NK, class Epi>
int launch_rows(const AOp& a, const float* W, int ldb, int K, int Nc, long R, const Epi& e, hipStream_t st) {
  dim3 grid(xcd_grid((R + BM - 1) / BM, tile_split(Nc, BN).tiles));
#ifdef OBMAN_ABLATION  // tools/ablate_gemm.sh: which resource bounds the main loop?  (results are wrong in every DBG != 0 build)
  static const int dbg = [] { const char* v = getenv("OBMAN_GEMM_DBG"); return v ? atoi(v) : 0; }();
#define OBMAN_DBG_CASE(D) \
  if (dbg == D) { gemm_rows_kernel<AOp, B_NK, Epi, D><<<grid, NT, sizeof(Tiles), st>>>(a, W, ldb, K, Nc, e, xcd_aware() ? 0 : 8); OBMAN_LAUNCH_CHECK(); return 0; }
  OBMAN_DBG_CASE(1) OBMAN_DBG_CASE(2) OBMAN_DBG_CASE(3) OBMAN_DBG_CASE(4) OBMAN_DBG_CASE(6) OBMAN_DBG_CASE(8) OBMAN_DBG_CASE(10) OBMAN_DBG_CASE(14) OBMAN_DBG_CASE(16) OBMAN_DBG_CASE(32)
#undef OBMAN_DBG_CASE
#endif
  gemm_rows_kernel<AOp, B_NK, Epi><<<grid, NT, sizeof(Tiles), st>>>(a, W, ldb, K, Nc, e, xcd_aware() ? 0 : 8);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
template <class AOp, class BOp>
int launch_tn(const AOp& a, const BOp& b, int M, int Nc, long R, int, float* part, float* out, int ldo, int off, hipStream_t st) {
  const int chunk_rows = tn_chunk_rows(M, Nc, R);
  const int chunks = (int)((R + chunk_rows - 1) / chunk_rows);
  dim3 grid(xcd_grid(chunks, tile_split(M, BM).tiles * tile_split(Nc, BN).tiles));
  gemm_tn_kernel<AOp, BOp><<<grid, NT, sizeof(TilesT), st>>>(a, b, M, Nc, (int)R, chunk_rows, part, xcd_aware());
  OBMAN_LAUNCH_CHECK();
  reduce_tn_kernel<<<obman_cdiv((long)M * Nc, RTN_ELEMS), 256, 0, st>>>(part, chunks, M, Nc, ldo, off, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
int launch_wcast(const float* W, int ld, int Nn, int K, int transposed, bfraw* out, hipStream_t st, int Kp_override = 0) {
  const int Kp = Kp_override ? Kp_override : kpad(K);
  wcast_kernel<<<obman_cdiv((long)Nn * Kp / 2, 256), 256, 0, st>>>(W, ld, Nn, K, Kp, transposed, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
template <class AOp, class Epi, int WN>
int launch_rows_bf16_wn(const AOp& a, const bfraw* Wb, int K, int Nc, const RowGeo& geo, const Epi& e, hipStream_t st) {
  const int Kp = kpad(K);
  const size_t lds = (size_t)2 * (BM + 64 * WN) * LP * sizeof(bfraw) + (size_t)AOp::NC * Kp * sizeof(float);
  // largest dynamic-LDS size already enabled for this instantiation, per device (a process may drive several GPUs)
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if ((int)lds > granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)rows_bf16_kernel<AOp, Epi, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store((int)lds, std::memory_order_relaxed);  // racing threads at worst repeat the (idempotent) call
  }
  const unsigned grid = (unsigned)geo.blocks() * (unsigned)((Nc + 64 * WN - 1) / (64 * WN));
  rows_bf16_kernel<AOp, Epi, WN><<<grid, NTB, lds, st>>>(a, Wb, Kp, Nc, e, geo);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
template <class AOp, class Epi>
int launch_rows_bf16(const AOp& a, const bfraw* Wb, int K, int Nc, const RowGeo& geo, const Epi& e, hipStream_t st) {
  return wide_wn(Nc) == 5 ? launch_rows_bf16_wn<AOp, Epi, 5>(a, Wb, K, Nc, geo, e, st)
                          : launch_rows_bf16_wn<AOp, Epi, 2>(a, Wb, K, Nc, geo, e, st);
}
// weight-gradient GEMMs (decoder_tn2.h): operands through 16-byte loads and transposing LDS reads; WN = 5 (320-column block tiles) for
// outputs wider than 128 columns, 2 below
template <class AOp, class BOp, int WN>
int launch_tn2_bf16_wn(const AOp& a, const BOp& b, int M, int Nc, int N, int Bsz, float* part, float* out, int ldo, int off, hipStream_t st,
                       int transposed) {
  constexpr int PA = BM + 32, PB = 64 * WN + (64 * WN % 128 == 0 ? 32 : 96);
  const size_t lds = (size_t)2 * T2_KT * (PA + PB) * sizeof(bfraw);
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if (!granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)tn2_bf16_kernel<AOp, BOp, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store(1, std::memory_order_relaxed);
  }
  const int chunks = tn_bf16_chunks(M, Nc, N, Bsz, 64 * WN);
  const long ntiles = (long)((Bsz + 7) / 8) * ((N + 7) / 8);
  const int tiles_per_chunk = (int)((ntiles + chunks - 1) / chunks);
  const unsigned grid = (unsigned)(((M + BM - 1) / BM) * ((Nc + 64 * WN - 1) / (64 * WN))) * (unsigned)chunks;
  tn2_bf16_kernel<AOp, BOp, WN><<<grid, NTB, lds, st>>>(a, b, M, Nc, N, Bsz, tiles_per_chunk, part);
  OBMAN_LAUNCH_CHECK();
  reduce_tn_kernel<<<obman_cdiv((long)M * Nc, RTN_ELEMS), 256, 0, st>>>(part, chunks, M, Nc, ldo, off, out, transposed);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
template <class AOp, class BOp>
int launch_tn2_bf16(const AOp& a, const BOp& b, int M, int Nc, int N, int Bsz, float* part, float* out, int ldo, int off, hipStream_t st,
                    int transposed) {
  return wide_wn(Nc) == 5 ? launch_tn2_bf16_wn<AOp, BOp, 5>(a, b, M, Nc, N, Bsz, part, out, ldo, off, st, transposed)
                          : launch_tn2_bf16_wn<AOp, BOp, 2>(a, b, M, Nc, N, Bsz, part, out, ldo, off, st, transposed);
}
// the wide tile: full 256-row tiles of the (transposed) product; the remainder rows' partials are in part_side [GH2S_BLOCKS][ns][Nc]
template <class AOp, class BOp>
int launch_tn2w_bf16(const AOp& a, const BOp& b, const Tn2wPlan& wp, int M, int Nc, int N, int Bsz, float* part, float* out, int ldo,
                     hipStream_t st) {
  const size_t lds = (size_t)2 * TW_KT * 2 * TW_P * sizeof(bfraw);
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if (!granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)tn2w_bf16_kernel<AOp, BOp>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store(1, std::memory_order_relaxed);
  }
  const int Mmain = wp.mt * TW_BM;
  tn2w_bf16_kernel<AOp, BOp><<<(unsigned)(wp.mt * wp.chunks), NTB, lds, st>>>(a, b, M, Nc, N, Bsz, wp.tiles_per_chunk, part);
  OBMAN_LAUNCH_CHECK();
  reduce_tn_kernel<<<obman_cdiv((long)Mmain * Nc, RTN_ELEMS), 256, 0, st>>>(part, wp.chunks, Mmain, Nc, ldo, 0, out, 1);
  OBMAN_LAUNCH_CHECK();
  if (wp.ns) {
    const float* side = part + (size_t)wp.chunks * Mmain * Nc;
    reduce_side_kernel<<<obman_cdiv((long)wp.ns * Nc, 64), 1024, 0, st>>>(side, GH2S_BLOCKS, wp.ns, Nc, ldo, Mmain, out);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}
// -> pointer / row count the finalize kernels should read: the partials themselves, or their 64-segment pre-reduction
template <class T>
int pre_reduce(const T*& part, int& rows, int cols, T* scratch, hipStream_t st) {
  if (rows <= PRE_MIN_ROWS) return 0;
  const int want = pre_segments(cols);
  const int per = (rows + want - 1) / want, segs = (rows + per - 1) / per;
  colsum_segments_kernel<T><<<dim3(obman_cdiv(cols, 256), segs), 256, 0, st>>>(part, rows, cols, per, scratch);
  OBMAN_LAUNCH_CHECK();
  part = scratch;
  rows = segs;
  return 0;
}

// layer 4: the 16-lanes-per-row kernels (l4w_*) serve the production width; OBMAN_DEC_L4W=0 = the first-generation kernels (A/B)
bool l4_wide(const Dims& d) {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_L4W"); return e ? atoi(e) : 1; }();
  return on != 0 && d.C3 == L4W_C && d.ld3 == L4W_C;
}
// rows per block of the layer-4 backward: L4_ROWS for the first-generation kernel (the workspace is sized for it); the wide kernel
// takes >= 64 rows and at most L4W_MAX_ROWS, in WHOLE rounds of the resident block slots (three 4-wave blocks per CU at its ~150
// registers): 1 004 blocks of 1 024 rows on 768 slots were a full round plus a 31 % one (r06: 85 -> see profiles/r06_kernels.md)
int device_cus();
int l4_rows(const Dims& d) {
  if (!l4_wide(d)) return L4_ROWS;
  const long slots = 3L * device_cus();
  const long rounds = (d.R + slots * L4W_MAX_ROWS - 1) / (slots * L4W_MAX_ROWS);
  long rows = (d.R + slots * rounds - 1) / (slots * rounds);
  rows = (rows + 31) / 32 * 32;
  return (int)(rows < 64 ? 64 : (rows > L4W_MAX_ROWS ? L4W_MAX_ROWS : rows));
}
// ---- second-generation rows GEMMs (decoder_rows2.h): persistent blocks, weights stationary in LDS
int device_cus() {
  static std::atomic<int> cus[MAX_DEVICES];
  const int dev = current_device();
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}
// the rows2 generators address their bf16 operands ([R, ld2] is the largest) with 32-bit byte offsets into a buffer descriptor
bool r2_addr32(const Dims& d) { return (size_t)d.R * d.ld2 * sizeof(bfraw) <= R2_PLAIN_MAX_BYTES && (size_t)d.R * d.ld3 * sizeof(bfraw) <= R2_PLAIN_MAX_BYTES; }
bool rows2_enabled() {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_ROWS2"); return e ? atoi(e) : 1; }();  // A/B knob
  return on != 0;
}
R2Geo r2_geo(const Dims& d, int Nc, int mode = 0) {
  R2Geo g{};
  g.R = (int)d.R; g.N = d.N; g.B = d.B; g.mode = mode;
  g.nvt = mode == 1 ? (d.N + 3) / 4 : (d.N + 31) / 32;
  g.nbg = mode == 1 ? (d.B + 63) / 64 : (d.B + 7) / 8;
  g.ngroups = Nc > R2_SIDE ? (Nc - R2_SIDE + R2_COLS - 1) / R2_COLS : 1;
  { const int last = Nc - (g.ngroups - 1) * R2_COLS; g.wside = last > R2_COLS ? last - R2_COLS : 0; }
  int target = device_cus() / g.ngroups;  // one block per CU (the weight slice takes most of a CU's LDS)
  if (target < 1) target = 1;
  g.spb = target / g.nbg;
  if (g.spb < 1) g.spb = 1;
  if (g.spb > g.nvt) g.spb = g.nvt;
  g.slots = g.spb * g.nbg;
  g.chunk = (g.nvt + g.spb - 1) / g.spb;
  return g;
}
template <class AOp, class Epi>
size_t r2_lds_bytes(int Kp, const R2Geo& geo) {
  size_t lds = (size_t)(R2_COLS + geo.wside) * (Kp + 8) * sizeof(bfraw) + (size_t)R2Lds<AOp>::floats(Kp) * sizeof(float) + (size_t)Epi::LDS_FLOATS * sizeof(float);
  const size_t flush = (size_t)(R2_WAVES - 1) * (R2_NT + 1) * 32 * 2 * sizeof(double);  // the end-of-block reduction re-uses the slice
  return lds < flush ? flush : lds;
}
constexpr size_t R2_LDS_LIMIT = 160 * 1024;  // per workgroup on gfx950; wider layers than that fits take the first-generation kernels
template <class AOp, class Epi>
int launch_rows2(const AOp& a, const bfraw* Wb, int Kp, int Nc, const R2Geo& geo, const Epi& e, hipStream_t st) {
  const int aop_floats = R2Lds<AOp>::floats(Kp);
  const size_t lds = r2_lds_bytes<AOp, Epi>(Kp, geo);
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if ((int)lds > granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)rows2_bf16_kernel<AOp, Epi>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store((int)lds, std::memory_order_relaxed);
  }
#ifdef OBMAN_ABLATION  // tools/ablate_gemm.sh build: the product library holds no measurement-only kernel
  if constexpr (std::is_same<AOp, BGridFeatPre>::value || std::is_same<AOp, BPlain>::value) {
    // measurement-only ablations of the k loop (wrong results; tools/archive/r03/r03_abl.sh): where does a k-step's time go?
    static const int abl = [] { const char* v = getenv("OBMAN_R2_ABL"); return v ? atoi(v) : 0; }();
    if (abl) {
      const unsigned g = (unsigned)(geo.ngroups * geo.slots);
      auto go = [&](auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        kern<<<g, R2_THREADS, lds, st>>>(a, Wb, Kp, Nc, e, geo, aop_floats);
      };
      switch (abl) {
        case 1: go(rows2_bf16_kernel<AOp, Epi, 1>); break;
        case 2: go(rows2_bf16_kernel<AOp, Epi, 2>); break;
        case 3: go(rows2_bf16_kernel<AOp, Epi, 3>); break;
        case 4: go(rows2_bf16_kernel<AOp, Epi, 4>); break;
        case 6: go(rows2_bf16_kernel<AOp, Epi, 6>); break;
        case 9: go(rows2_bf16_kernel<AOp, Epi, 9>); break;
        case 8: {
          go(rows2_bf16_kernel<AOp, Epi, 8>);
          (void)hipStreamSynchronize(st);
          static int calls = 0;
          if (++calls == 5) {
            static unsigned long long host[1024 * R2_WAVES * 8];
            (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(r2_dbg), sizeof(host));
            double sum[8] = {0};
            const int nw = (int)(g < 1024 ? g : 1024) * R2_WAVES;
            for (int i = 0; i < nw; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)host[(size_t)i * 8 + k];
            const double st_ = sum[5] > 0 ? sum[5] : 1, tl = sum[7] > 0 ? sum[7] : 1;
            fprintf(stderr, "R2DBG %s waves %d steps/wave %.0f tiles/wave %.1f | per k-step ticks: lds %.1f transform+vmwait %.1f mfma %.1f rest %.1f | per tile: epilogue %.0f | per wave total %.0f\n",
                    std::is_same<AOp, BPlain>::value ? "dA" : "h2", nw, st_ / nw, tl / nw, sum[0] / st_, sum[1] / st_, sum[2] / st_, sum[3] / st_, sum[4] / tl, sum[6] / nw);
          }
          break;
        }
        default: go(rows2_bf16_kernel<AOp, Epi, 5>); break;
      }
      OBMAN_LAUNCH_CHECK();
      return 0;
    }
  }
#endif
  rows2_bf16_kernel<AOp, Epi><<<(unsigned)(geo.ngroups * geo.slots), R2_THREADS, lds, st>>>(a, Wb, Kp, Nc, e, geo, aop_floats);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

// ---- third generation of the single-array rows GEMMs: A tile by LDS-DMA into a wave-private ring (decoder_rows3.h)
bool rows3_enabled() {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_ROWS3"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = rows2
  return on != 0;
}
template <class AOp, class Epi>
size_t r3_lds_bytes(int Kp, const R2Geo& geo, int nb) {
  return r2_lds_bytes<AOp, Epi>(Kp, geo) + (size_t)R2_WAVES * nb * R3_BLOCK_BYTES;
}
// ring depth (4 KB blocks per wave) that fits next to the weight slice and the epilogue's scratch: 3, 2, or 0 = take the rows2 kernel
template <class AOp, class Epi>
int r3_depth(int Kp, const R2Geo& geo) {
  if (!rows3_enabled() || (Kp & 15)) return 0;
  if (r3_lds_bytes<AOp, Epi>(Kp, geo, 3) <= R2_LDS_LIMIT) return 3;
  if (r3_lds_bytes<AOp, Epi>(Kp, geo, 2) <= R2_LDS_LIMIT) return 2;
  return 0;
}
template <class AOp, class Epi, int NB>
int launch_rows3_nb(const AOp& a, const bfraw* Wb, int Kp, int Nc, const R2Geo& geo, const Epi& e, hipStream_t st) {
  const size_t lds = r3_lds_bytes<AOp, Epi>(Kp, geo, NB);
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if ((int)lds > granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)rows3_bf16_kernel<AOp, Epi, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store((int)lds, std::memory_order_relaxed);
  }
  rows3_bf16_kernel<AOp, Epi, NB><<<(unsigned)(geo.ngroups * geo.slots), R2_THREADS, lds, st>>>(a, Wb, Kp, Nc, e, geo, R2Lds<AOp>::floats(Kp));
  OBMAN_LAUNCH_CHECK();
  return 0;
}
template <class AOp, class Epi>
int launch_rows3(const AOp& a, const bfraw* Wb, int Kp, int Nc, const R2Geo& geo, const Epi& e, int depth, hipStream_t st) {
  return depth == 3 ? launch_rows3_nb<AOp, Epi, 3>(a, Wb, Kp, Nc, geo, e, st) : launch_rows3_nb<AOp, Epi, 2>(a, Wb, Kp, Nc, geo, e, st);
}

// ---- second-generation fp32 rows GEMMs (decoder_rows2f.h)
bool rows2f_enabled() {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_ROWS2F"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = first-generation fp32 kernels
  return on != 0;
}
F2Geo f2_geo(const Dims& d, int Nc, int NT, int mode = 0) {
  F2Geo g{};
  const int cols = 32 * NT;
  g.R = (int)d.R; g.N = d.N; g.B = d.B; g.mode = mode;
  g.nvt = (d.N + 3) / 4;
  g.tiles = mode == 0 ? (int)((d.R + 31) / 32) : g.nvt * ((d.B + 7) / 8);
  g.ngroups = Nc > F2_SIDE ? (Nc - F2_SIDE + cols - 1) / cols : 1;
  { const int last = Nc - (g.ngroups - 1) * cols; g.wside = last > cols ? last - cols : 0; }
  // one block per CU.  A block of the last column group also carries the 1 .. 3 side columns on the VALU: measured per tile
  // (s_memtime in the k loop, profiles/r04_kernels.md) +19 % with one side column at NT = 2, +14.5 % with three at NT = 4 - the
  // last group gets proportionally more blocks (fewer tiles each) so that all groups finish together.
  const int cus = device_cus();
  const int need = (g.tiles + F2_WAVES - 1) / F2_WAVES;  // no block without a tile
  const int cap = need < d.rb ? need : d.rb;              // the per-block moment / sum partials are sized for d.rb row blocks (fwd_ws / bwd_ws)
  if (g.ngroups == 1) {
    g.slots = 0;
    g.slots_last = cus < cap ? cus : cap;
  } else {
    const double w = g.wside ? 1.0 + (0.14 + 0.05 * g.wside) * 2.0 / NT : 1.0;
    int last = (int)(cus * w / (g.ngroups - 1 + w) + 0.5);
    int main_ = (cus - last) / (g.ngroups - 1);
    if (main_ < 1) main_ = 1;
    if (last < 1) last = 1;
    g.slots = main_ < cap ? main_ : cap;
    g.slots_last = last < cap ? last : cap;
  }
  return g;
}
template <class AOp, class Epi, int NT>
size_t f2_lds_bytes(int Kp, const F2Geo& geo) {
  const size_t lds = ((size_t)(32 * NT + geo.wside) * (Kp + 4) + (size_t)AOp::lds_floats(Kp) + (size_t)Epi::LDS_FLOATS) * sizeof(float);
  return lds < f2_flush_bytes(NT) ? f2_flush_bytes(NT) : lds;
}
constexpr size_t F2_LDS_LIMIT = 160 * 1024;
template <class AOp, class Epi, int NT>
int launch_rows2f(const AOp& a, const float* W, int ldw, int w_kn, int K, int Nc, const F2Geo& geo, const Epi& e, hipStream_t st) {
  const int Kp = kpad8(K);
  const size_t lds = f2_lds_bytes<AOp, Epi, NT>(Kp, geo);
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if ((int)lds > granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)rows2f_kernel<AOp, Epi, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    granted[dev].store((int)lds, std::memory_order_relaxed);
  }
  rows2f_kernel<AOp, Epi, NT><<<(unsigned)((geo.ngroups - 1) * geo.slots + geo.slots_last), F2_THREADS, lds, st>>>(a, W, ldw, w_kn, K, Kp, Nc, e, geo, AOp::lds_floats(Kp));
  OBMAN_LAUNCH_CHECK();
#ifdef OBMAN_F2_TIMING
  {
    static int calls = 0;
    (void)hipStreamSynchronize(st);
    if (++calls % 8 == 5) {
      static unsigned long long host[2048 * 8];
      (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(f2_dbg), sizeof(host));
      const int nball = (geo.ngroups - 1) * geo.slots + geo.slots_last, nb = nball < 256 ? nball : 256;
      double st_[8] = {0}, mx[8] = {0};
      for (int i = 0; i < nb * 8; ++i) for (int k = 0; k < 6; ++k) { st_[k] += (double)host[i * 8 + k]; if ((double)host[i * 8 + k] > mx[k]) mx[k] = (double)host[i * 8 + k]; }
      double lastg[3] = {0}; int nl = 0;
      for (int i = 0; i < nb * 8; ++i) if ((int)host[i * 8 + 6] == geo.ngroups - 1) { lastg[0] += (double)host[i * 8 + 1]; lastg[1] += (double)host[i * 8 + 3]; lastg[2] += (double)host[i * 8 + 5]; ++nl; }
      const int nw = nb * 8;
      fprintf(stderr, "F2DBG K=%d Nc=%d NT=%d blocks=%d(+%d last) tiles=%d | per wave avg (max) ticks: stage %.0f (%.0f) kloop %.0f (%.0f) epi %.0f (%.0f) tiles %.1f (%.0f) total %.0f (%.0f) | per tile: kloop %.0f (per k-step %.1f) epi %.0f; last group: kloop/tile %.0f total %.0f\n",
              K, Nc, NT, nball, geo.slots_last, geo.tiles, st_[0] / nw, mx[0], st_[1] / nw, mx[1], st_[2] / nw, mx[2], st_[3] / nw, mx[3], st_[5] / nw, mx[5],
              st_[1] / st_[3], st_[1] / st_[3] / (Kp / 8), st_[2] / st_[3], nl ? lastg[0] / lastg[1] : 0.0, nl ? lastg[2] / nl : 0.0);
    }
  }
#endif
  return 0;
}
// 128 columns per block (4 accumulator tiles per wave) where that weight slice fits in LDS, else 64
template <class AOp, class Epi4, class Epi2>
bool f2_wide(const Dims& d, int K, int Nc) { return f2_lds_bytes<AOp, Epi4, 4>(kpad8(K), f2_geo(d, Nc, 4)) <= F2_LDS_LIMIT; }
template <class AOp, template <int> class Epi, bool WIDE_OK = true>
int launch_rows2f_auto(const Dims& d, const AOp& a, const float* W, int ldw, int w_kn, int K, int Nc, const Epi<4>& e4, const Epi<2>& e2, int mode2,
                       int* slots, hipStream_t st) {
  // 128 columns per block halve the operand loads per MFMA but double the tile: only with at least 4 tiles per wave (at 64 x 642
  // points a 128-column h3 GEMM has one tile per wave on 161 of the 256 CUs)
  if constexpr (WIDE_OK) if (f2_wide<AOp, Epi<4>, Epi<2>>(d, K, Nc) && f2_geo(d, Nc, 4, mode2).tiles >= 4 * F2_WAVES * device_cus() / f2_geo(d, Nc, 4, mode2).ngroups) {
    const F2Geo g = f2_geo(d, Nc, 4, mode2);
    if (slots) *slots = g.mrows();
    return launch_rows2f<AOp, Epi<4>, 4>(a, W, ldw, w_kn, K, Nc, g, e4, st);
  }
  const F2Geo g = f2_geo(d, Nc, 2, mode2);
  if (slots) *slots = g.mrows();
  return launch_rows2f<AOp, Epi<2>, 2>(a, W, ldw, w_kn, K, Nc, g, e2, st);
}
// layer-2 data gradient with P / Q partials from the accumulators (F2EpiL1PQ): row mode 2, every wave inside one sample group
F2Geo f2_geo_pq(const Dims& d, int Nc, int NT) {
  F2Geo g = f2_geo(d, Nc, NT, 2);  // block counts per column group as usual (the last group, which carries the side columns, has more)
  const int cus = device_cus(), nbg = (d.B + 7) / 8;
  if (g.ngroups == 1) g.slots_last = cus;       // f2_geo caps the block count by the tile count: here every wave counts
  else {
    const double w = g.wside ? 1.0 + (0.14 + 0.05 * g.wside) * 2.0 / NT : 1.0;
    int last = (int)(cus * w / (g.ngroups - 1 + w) + 0.5), main_ = (cus - last) / (g.ngroups - 1);
    g.slots = main_ < 1 ? 1 : main_;
    g.slots_last = last < 1 ? 1 : last;
  }
  auto per_group = [&](int slots) { int w = slots * F2_WAVES / nbg; return w > g.nvt ? g.nvt : w; };
  g.wpg = g.ngroups > 1 ? per_group(g.slots) : per_group(g.slots_last);
  g.wpg_last = per_group(g.slots_last);
  if (g.wpg < 1 || g.wpg_last < 1) g.wpg = g.wpg_last = 0;  // more sample groups than waves: the caller takes the materialised form
  return g;
}
// ---- second-generation fp32 weight-gradient GEMM (decoder_tn3.h)
bool tn3_enabled() {
  static const int on = [] { const char* e = getenv("OBMAN_DEC_TN3"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = first-generation gemm_tn_kernel
  return on != 0;
}
bool tn3_ok(int M, int Nc, int N) { return tn3_enabled() && N >= T3_KB && M >= T3_BM && M % T3_BM <= T3_SIDE && Nc >= T3_BN && Nc % T3_BN <= T3_SIDE; }
template <class AOp, class BOp>
int launch_tn3(const AOp& a, const BOp& b, int M, int Nc, long R, int N, float* part, float* out, int ldo, int off, hipStream_t st) {
  T3Geo g{};
  g.M = M; g.Nc = Nc; g.R = (int)R; g.N = N; g.B = (int)(R / N);
  g.mt = M / T3_BM; g.ms = M - g.mt * T3_BM; g.nt = Nc / T3_BN; g.ns = Nc - g.nt * T3_BN;
  const int tiles = g.mt * g.nt;
  // one block per CU; never more chunks than the first generation would use (the split-K partials share its workspace: bwd_ws)
  const int rows1 = tn_chunk_rows(M, Nc, R);
  const int max_chunks = (int)((R + rows1 - 1) / rows1);
  int chunks = device_cus() / tiles;
  if (chunks < 1) chunks = 1;
  if (chunks > max_chunks) chunks = max_chunks;
  long rows = (R + chunks - 1) / chunks;
  rows = (rows + T3_KB - 1) / T3_KB * T3_KB;
  g.chunk_rows = (int)rows;
  g.chunks = (int)((R + rows - 1) / rows);
#ifdef OBMAN_T3_DBG
  g.dbg = getenv("OBMAN_T3_DBG") ? atoi(getenv("OBMAN_T3_DBG")) : 0;
#endif
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if (!granted[dev].load(std::memory_order_relaxed)) {  // > 64 KB of dynamic LDS
    const hipError_t err = hipFuncSetAttribute((const void*)tn3_kernel<AOp, BOp>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(T3Tiles));
    if (err != hipSuccess) return (int)err;
    granted[dev].store(1, std::memory_order_relaxed);
  }
  tn3_kernel<AOp, BOp><<<(unsigned)(tiles * g.chunks), T3_THREADS, sizeof(T3Tiles), st>>>(a, b, g, part);
  OBMAN_LAUNCH_CHECK();
  reduce_tn_kernel<<<obman_cdiv((long)M * Nc, RTN_ELEMS), 256, 0, st>>>(part, g.chunks, M, Nc, ldo, off, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}
// the whole fp32 call takes the second-generation kernels or none of them (they share the pre-scaled factors and the zeroed pitch columns)
bool use_rows2f(const Dims& d) {
  if (d.bf16 || d.ps || !rows2f_enabled()) return false;
  return f2_lds_bytes<F2GridFeatPre, F2EpiStore<2>, 2>(kpad8(d.C1), f2_geo(d, d.C2, 2, 2)) <= F2_LDS_LIMIT &&
         f2_lds_bytes<F2BnRelu, F2EpiStore<2>, 2>(kpad8(d.C2), f2_geo(d, d.C3, 2)) <= F2_LDS_LIMIT &&
         f2_lds_bytes<F2GradH3, F2EpiMask<2>, 2>(kpad8(d.C3), f2_geo(d, d.C2, 2)) <= F2_LDS_LIMIT &&
         f2_lds_bytes<F2GradH, F2EpiL1<2>, 2>(kpad8(d.C2), f2_geo(d, d.C1, 2)) <= F2_LDS_LIMIT;
}

// ---- bf16 flavour: layers 2-4 of the forward (after prep_kernel) and the whole backward (decoder_bf16.h)
int forward_bf16(const obman_pointgen_params* p, const Dims& d, const FwdWs& w, float* out, float* ws, hipStream_t st) {
  const int tr = p->training;
  double* moments = reinterpret_cast<double*>(ws + w.moments);
  bfraw* H2 = reinterpret_cast<bfraw*>(ws + w.H2);
  bfraw* H3 = reinterpret_cast<bfraw*>(ws + w.H3);
  const RowGeo geo{(int)d.R, d.N, d.B, 0, 0};
  // layer 2 regenerates a1 from Gx[n] + Fx[b]: with (8 samples x 16 vertices) row tiles a block touches 16 rows of Gx and 8 of
  // Fx instead of 128 + 1..2 (2.2 GB of Gx fetches per call at 16 050 points with linear rows, profiles/r02_kernels.md)
  const RowGeo tiled{(int)d.R, d.N, d.B, (d.N + 15) / 16, 1};
  int rc;
  {  // h2 = W2 relu(bn1(h1)) + b2
    BGridFeat a{ws + w.Gx, ws + w.Fx, p->bn_w[0], p->bn_b[0], d.ld1, d.C1};
    EpiStoreB e{H2, p->b2, tr ? moments : nullptr, d.ld2, d.C2};
    bfraw* wb = reinterpret_cast<bfraw*>(ws + w.wb2);
    const double* mom = moments;
    int mrows;
    // the pre-scaled layer-1 factors Gy / Fy (+ the sentinel row) come from prep_kernel / l1_fill_kernel: read by the h2 GEMM below and
    // by the weight-gradient GEMM of the backward pass
    if (rows2_enabled() && r2_addr32(d) && r2_lds_bytes<BGridFeatPre, EpiStoreB2>(kpad16(d.C1), r2_geo(d, d.C2, 2)) <= R2_LDS_LIMIT) {
      const int Kp = kpad16(d.C1);
      // rows as (8 samples x 4 vertices) per wave: a load instruction touches 4 rows of the layer-1 grid factor and 8 of the
      // feature factor instead of 32 + 1 (the fp32 factors go through the texture path 64 B per clock and CU),
      // and the feature factor's 8 rows of the block live in LDS.  Factors pre-scaled by BatchNorm-1's gamma / beta: add + max per element
      // (Round 6, measured and dropped - VERDICT r05 task 1a: ONE wave per SIMD with 2 - 4 fragments per wave (256-thread blocks, up
      // to 512 registers, weight fragments / feature-factor vectors shared by the fragments: 4.25 - 3.9 instead of 5 instructions per
      // MFMA in the k loop).  Parity-green; 643 - 767 us against this kernel's 523: with nobody on the SIMD to hide them the k loop's
      // LDS round trips and operand waits run it at 96 - 106 cycles per MFMA (s_memtime; 32 is the pipe's rate) and the epilogue
      // (27 % of a tile) runs beside an idle matrix pipe; 3 and 4 fragments also spill (82 - 129 registers).  The kernel is
      // decoder_rows4.h of commit 12fc77d; numbers in profiles/r06_kernels.md section 2.)
      const R2Geo g2 = r2_geo(d, d.C2, 2);
      EpiStoreB2 e2{H2, p->b2, tr ? moments : nullptr, d.ld2, d.C2};
      BGridFeatPre ap{ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
      if ((rc = launch_wcast(p->w2, d.C1, d.C2, d.C1, 0, wb, st, Kp))) return rc;
      if ((rc = launch_rows2<BGridFeatPre, EpiStoreB2>(ap, wb, Kp, d.C2, g2, e2, st))) return rc;
      mrows = g2.slots;
    } else {
      if ((rc = launch_wcast(p->w2, d.C1, d.C2, d.C1, 0, wb, st))) return rc;
      if ((rc = launch_rows_bf16<BGridFeat, EpiStoreB>(a, wb, d.C1, d.C2, tiled, e, st))) return rc;
      mrows = tiled.blocks();
    }
    if (tr && (rc = pre_reduce<double>(mom, mrows, d.C2 * 2, reinterpret_cast<double*>(ws + w.mred), st))) return rc;
    bn_finalize_kernel<<<obman_cdiv(d.C2, 4), 256, 0, st>>>(mom, mrows, d.R, d.C2, tr, p->eps, p->momentum, p->bn_w[1], p->bn_b[1],
                                                               p->bn_rm[1], p->bn_rv[1], ws + w.mean2, ws + w.rstd2, ws + w.s2, ws + w.t2);
    OBMAN_LAUNCH_CHECK();
  }
  {  // h3 = W3 relu(bn2(h2)) + b3
    BBnRelu a{H2, ws + w.s2, ws + w.t2, d.ld2, d.C2};
    EpiStoreB e{H3, p->b3, tr ? moments : nullptr, d.ld3, d.C3};
    bfraw* wb = reinterpret_cast<bfraw*>(ws + w.wb3);
    const double* mom = moments;
    int mrows;
    if (rows2_enabled() && r2_addr32(d) && r2_lds_bytes<BBnRelu, EpiStoreB2>(kpad16(d.C2), r2_geo(d, d.C3)) <= R2_LDS_LIMIT) {
      const int Kp = kpad16(d.C2);
      const R2Geo g2 = r2_geo(d, d.C3);
      EpiStoreB2 e2{H3, p->b3, tr ? moments : nullptr, d.ld3, d.C3};
      if ((rc = launch_wcast(p->w3, d.C2, d.C3, d.C2, 0, wb, st, Kp))) return rc;
      const int depth = r3_depth<BBnRelu, EpiStoreB2>(Kp, g2);
      if (depth) rc = launch_rows3<BBnRelu, EpiStoreB2>(a, wb, Kp, d.C3, g2, e2, depth, st);
      else rc = launch_rows2<BBnRelu, EpiStoreB2>(a, wb, Kp, d.C3, g2, e2, st);
      if (rc) return rc;
      mrows = g2.slots;
    } else {
      if ((rc = launch_wcast(p->w3, d.C2, d.C3, d.C2, 0, wb, st))) return rc;
      if ((rc = launch_rows_bf16<BBnRelu, EpiStoreB>(a, wb, d.C2, d.C3, geo, e, st))) return rc;
      mrows = geo.blocks();
    }
    if (tr && (rc = pre_reduce<double>(mom, mrows, d.C3 * 2, reinterpret_cast<double*>(ws + w.mred), st))) return rc;
    bn_finalize_kernel<<<obman_cdiv(d.C3, 4), 256, 0, st>>>(mom, mrows, d.R, d.C3, tr, p->eps, p->momentum, p->bn_w[2], p->bn_b[2],
                                                               p->bn_rm[2], p->bn_rv[2], ws + w.mean3, ws + w.rstd3, ws + w.s3, ws + w.t3);
    OBMAN_LAUNCH_CHECK();
  }
  if (l4_wide(d)) l4w_fwd_kernel<bfraw><<<2048, 256, 0, st>>>(H3, ws + w.s3, ws + w.t3, p->w4, p->b4, p->out_factor, d.R, out);
  else l4_fwd_kernel<bfraw><<<2048, 256, 0, st>>>(H3, d.ld3, ws + w.s3, ws + w.t3, p->w4, p->b4, p->out_factor, d.R, d.C3, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

// everything of the backward up to (and including) P[b,c] = sum_n gy1, Q[n,c] = sum_b gy1; the caller continues with layer 1
int backward_bf16(const obman_pointgen_params* p, const Dims& d, const FwdWs& w, const BwdWs& v, const float* g_out, const float* ws,
                  float* ws2, const obman_pointgen_grads* g, const float** q_sum, hipStream_t st) {
  const int tr = p->training;
  double* sums = reinterpret_cast<double*>(ws2 + v.sums);
  float *k1 = ws2 + v.k, *k2 = k1 + d.ld1, *k3 = k2 + d.ld1;
  const float f = p->out_factor;
  const bfraw* H2 = reinterpret_cast<const bfraw*>(ws + w.H2);
  const bfraw* H3 = reinterpret_cast<const bfraw*>(ws + w.H3);
  bfraw* GY2 = reinterpret_cast<bfraw*>(ws2 + v.GY2);
  const RowGeo lin{(int)d.R, d.N, d.B, 0, 0};
  int rc;
  // ---- layer 4 + BN-3 statistics
  const int l4rows = l4_rows(d);
  const int l4b = obman_cdiv(d.R, l4rows);
  if (l4_wide(d))
    l4w_bwd_kernel<bfraw><<<l4b, 256, 0, st>>>(g_out, H3, ws + w.s3, ws + w.t3, ws + w.mean3, ws + w.rstd3, p->w4, f, d.R, l4rows, sums, ws2 + v.l4p);
  else
    l4_bwd_kernel<bfraw><<<l4b, 128, 0, st>>>(g_out, H3, d.ld3, ws + w.s3, ws + w.t3, ws + w.mean3, ws + w.rstd3, p->w4, f, d.R, d.C3, l4rows,
                                               sums, ws2 + v.l4p);
  OBMAN_LAUNCH_CHECK();
  {
    const float* lp = ws2 + v.l4p;
    int lrows = l4b;
    if ((rc = pre_reduce<float>(lp, lrows, 3 * d.C3 + 4, ws2 + v.l4red, st))) return rc;
    l4_bwd_finalize_kernel<<<obman_cdiv(3 * d.C3 + 3, 4), 256, 0, st>>>(lp, lrows, d.C3, g->w4, g->b4);
    OBMAN_LAUNCH_CHECK();
  }
  const double* sp = sums;
  int srows = l4b;
  if ((rc = pre_reduce<double>(sp, srows, d.C3 * 2, reinterpret_cast<double*>(ws2 + v.sred), st))) return rc;
  bn_bwd_finalize_kernel<<<obman_cdiv(d.C3, 4), 256, 0, st>>>(sp, srows, d.R, d.C3, tr, p->bn_w[2], ws + w.mean3, ws + w.rstd3, g->bn_w[2], g->bn_b[2],
                                                                 g->b3, k1, k2, k3);
  OBMAN_LAUNCH_CHECK();
  {  // gW3[o,c] = sum_r gh3[r,o] a2[r,c]
    T2GradH3 ta{g_out, p->w4, H3, ws + w.s3, ws + w.t3, k1, k2, k3, f, d.ld3, d.C3, d.N, d.B};
    T2BnRelu tb{H2, ws + w.s2, ws + w.t2, d.ld2, d.C2, d.N, d.B};
    if ((rc = launch_tn2_bf16<T2GradH3, T2BnRelu>(ta, tb, d.C3, d.C2, d.N, d.B, ws2 + v.tn, g->w3, d.C2, 0, st, 0))) return rc;
  }
  {  // gy2 = (gh3 W3) * (y2 > 0) stored bf16, BN-2 sums.  B[k = out channel][n = in channel] = W3[k][n]: the transposed image
    BGradH3 a{g_out, p->w4, H3, ws + w.s3, ws + w.t3, k1, k2, k3, f, d.ld3, d.C3};
    EpiMaskB e{GY2, H2, sums, ws + w.s2, ws + w.t2, ws + w.mean2, ws + w.rstd2, d.ld2, d.C2};
    bfraw* wt = reinterpret_cast<bfraw*>(ws2 + v.wt3);
    if (rows2_enabled() && r2_addr32(d) && r2_lds_bytes<BGradH3, EpiMaskB2>(kpad16(d.C3), r2_geo(d, d.C2)) <= R2_LDS_LIMIT) {
      const int Kp = kpad16(d.C3);
      const R2Geo g2 = r2_geo(d, d.C2);
      EpiMaskB2 e2{GY2, H2, sums, ws + w.s2, ws + w.t2, ws + w.mean2, ws + w.rstd2, d.ld2, d.C2};
      if ((rc = launch_wcast(p->w3, d.C2, d.C2, d.C3, 1, wt, st, Kp))) return rc;
      if ((rc = launch_rows2<BGradH3, EpiMaskB2>(a, wt, Kp, d.C2, g2, e2, st))) return rc;
      srows = g2.slots;
    } else {
      if ((rc = launch_wcast(p->w3, d.C2, d.C2, d.C3, 1, wt, st))) return rc;
      if ((rc = launch_rows_bf16<BGradH3, EpiMaskB>(a, wt, d.C3, d.C2, lin, e, st))) return rc;
      srows = lin.blocks();
    }
  }
  sp = sums;
  if ((rc = pre_reduce<double>(sp, srows, d.C2 * 2, reinterpret_cast<double*>(ws2 + v.sred), st))) return rc;
  bn_bwd_finalize_kernel<<<obman_cdiv(d.C2, 4), 256, 0, st>>>(sp, srows, d.R, d.C2, tr, p->bn_w[1], ws + w.mean2, ws + w.rstd2, g->bn_w[1], g->bn_b[1],
                                                                 g->b2, k1, k2, k3);
  OBMAN_LAUNCH_CHECK();
  const bool gh_plain = rows2_enabled() && (d.ld2 & 7) == 0 && d.ld2 <= 512 &&
                        (size_t)d.R * d.ld2 * sizeof(bfraw) <= R2_PLAIN_MAX_BYTES &&
                        r2_lds_bytes<BPlain, EpiL1B2>(kpad16(d.C2), r2_geo(d, d.C1, 1)) <= R2_LDS_LIMIT;
  {  // gW2[o,c] = sum_r gh2[r,o] a1[r,c], formed TRANSPOSED (M = the 515 channels of a1, one 320-wide tile for the 257 of gh2):
     // every operand is regenerated once per tile of the OTHER operand, and a1 (fp32 factors, add + fma + max per element) is the
     // expensive one - as the A operand of five 128 x 320 tiles it is generated once (640 channel columns per k-tile) and gh2 five
     // times (1 600), against 1 920 + 768 with gh2 as A on 3 x 2 tiles
    const Tn2wPlan wp = gh_plain ? tn2w_plan(d.C1, d.C2, d.N, d.B) : Tn2wPlan{false, 0, 0, 0, 0};
    if (gh_plain && wp.use) {
      // round 6: gh2 materialised in place AND, in the same pass, the products of the 3 odd channels of a1 (512 .. 514) with it
      // as VALU side sums; the 512 x 257 body of the product on two 256 x 288 tiles (decoder_tn2.h, "the WIDE tile")
      float* part = ws2 + v.tn;
      float* side = part + (size_t)wp.chunks * wp.mt * TW_BM * d.C2;
      const int m0 = wp.mt * TW_BM;
      if (wp.ns == 3)
        gh2_inplace_side_kernel<3><<<GH2S_BLOCKS, dim3(64, 4), 0, st>>>(GY2, H2, k1, k2, k3, d.R, d.ld2, d.C2, ws + w.Gy, ws + w.Fy, d.ld1, d.N, m0, d.C2, side);
      else if (wp.ns == 2)
        gh2_inplace_side_kernel<2><<<GH2S_BLOCKS, dim3(64, 4), 0, st>>>(GY2, H2, k1, k2, k3, d.R, d.ld2, d.C2, ws + w.Gy, ws + w.Fy, d.ld1, d.N, m0, d.C2, side);
      else if (wp.ns == 1)
        gh2_inplace_side_kernel<1><<<GH2S_BLOCKS, dim3(64, 4), 0, st>>>(GY2, H2, k1, k2, k3, d.R, d.ld2, d.C2, ws + w.Gy, ws + w.Fy, d.ld1, d.N, m0, d.C2, side);
      else
        gh2_inplace_kernel<<<obman_cdiv(d.R, GH2_ROWS), dim3(64, 4), 0, st>>>(GY2, H2, k1, k2, k3, d.R, d.ld2, d.C2);
      OBMAN_LAUNCH_CHECK();
      T2Pre ta{ws + w.Gy, ws + w.Fy, d.ld1, d.N, d.B};
      T2Plain tb{GY2, d.ld2, d.N, d.B};
      if ((rc = launch_tn2w_bf16<T2Pre, T2Plain>(ta, tb, wp, d.C1, d.C2, d.N, d.B, part, g->w2, d.C1, st))) return rc;
    } else if (gh_plain) {
      // gh2 materialised once (in place over gy2): both GEMMs below read it as a plain bf16 operand
      gh2_inplace_kernel<<<obman_cdiv(d.R, GH2_ROWS), dim3(64, 4), 0, st>>>(GY2, H2, k1, k2, k3, d.R, d.ld2, d.C2);
      OBMAN_LAUNCH_CHECK();
      T2Pre ta{ws + w.Gy, ws + w.Fy, d.ld1, d.N, d.B};
      T2Plain tb{GY2, d.ld2, d.N, d.B};
      if ((rc = launch_tn2_bf16<T2Pre, T2Plain>(ta, tb, d.C1, d.C2, d.N, d.B, ws2 + v.tn, g->w2, d.C1, 0, st, 1))) return rc;
    } else {
      T2Pre ta{ws + w.Gy, ws + w.Fy, d.ld1, d.N, d.B};
      T2GradH tb{GY2, H2, k1, k2, k3, d.ld2, d.C2, d.N, d.B};
      if ((rc = launch_tn2_bf16<T2Pre, T2GradH>(ta, tb, d.C1, d.C2, d.N, d.B, ws2 + v.tn, g->w2, d.C1, 0, st, 1))) return rc;
    }
  }
  const L1Geo lg = l1_geo(d);
  int l1_prow = lg.tiles, l1_groups = lg.groups;
  {  // dA(gy1) with the (8 samples x 16 vertices) row tiling: P / Q partials straight from the accumulators
    BGradH a{GY2, H2, k1, k2, k3, d.ld2, d.C2};
    EpiL1B e{ws2 + v.Pp, ws2 + v.Qp, ws + w.Gx, ws + w.Fx, p->bn_w[0], p->bn_b[0], d.ld1, d.C1};
    const RowGeo tiled{(int)d.R, d.N, d.B, lg.tiles, 1};
    bfraw* wt = reinterpret_cast<bfraw*>(ws2 + v.wt2);
    if (gh_plain) {
      const int Kp = kpad16(d.C2);
      const R2Geo g2 = r2_geo(d, d.C1, 1);
      EpiL1B2 e2{ws2 + v.Pp, ws2 + v.Qp, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
      BPlain ap{GY2, d.ld2, d.C2};
      if ((rc = launch_wcast(p->w2, d.C1, d.C1, d.C2, 1, wt, st, Kp))) return rc;
      if ((rc = launch_rows2<BPlain, EpiL1B2>(ap, wt, Kp, d.C1, g2, e2, st))) return rc;
      l1_prow = g2.spb;
      l1_groups = g2.nbg;
    } else if (rows2_enabled() && r2_addr32(d) && r2_lds_bytes<BGradH, EpiL1B2>(kpad16(d.C2), r2_geo(d, d.C1, 1)) <= R2_LDS_LIMIT) {
      const int Kp = kpad16(d.C2);
      const R2Geo g2 = r2_geo(d, d.C1, 1);
      EpiL1B2 e2{ws2 + v.Pp, ws2 + v.Qp, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
      if ((rc = launch_wcast(p->w2, d.C1, d.C1, d.C2, 1, wt, st, Kp))) return rc;
      if ((rc = launch_rows2<BGradH, EpiL1B2>(a, wt, Kp, d.C1, g2, e2, st))) return rc;
      l1_prow = g2.spb;
      l1_groups = g2.nbg;
    } else {
      if ((rc = launch_wcast(p->w2, d.C1, d.C1, d.C2, 1, wt, st))) return rc;
      if ((rc = launch_rows_bf16<BGradH, EpiL1B>(a, wt, d.C2, d.C1, tiled, e, st))) return rc;
    }
  }
  {  // P[b,c] = sum over the vertex groups, Q[n,c] = sum over the sample groups (fixed order)
    const float* pp = ws2 + v.Pp;
    int prow = l1_prow;
    if ((rc = pre_reduce<float>(pp, prow, d.B * d.ld1, ws2 + v.Ppre, st))) return rc;
    *q_sum = launch_l1_reduce2(pp, ws2 + v.Qp, d.ld1, d.B, d.N, d.C1, prow, l1_groups, ws2 + v.P, ws2 + v.Q, st);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}

bool params_ok(const obman_pointgen_params* p) {
  return p && p->B > 0 && p->N > 0 && p->C1 >= 8 && p->C1 / 4 <= 128 && p->grid && p->feat && p->w1 && p->w2 && p->w3 && p->w4 &&
         !(p->grid_per_sample && p->mfma_bf16);  // the random-sphere path is exact fp32 only
}

}  // namespace

extern "C" {

long obman_pointgen_ws_floats(const obman_pointgen_params* p, int backward) {
  if (!params_ok(p)) return -1;
  const Dims d = dims_of(p);
  return backward ? bwd_ws(d).total : fwd_ws(d).total;
}

int obman_pointgen_fwd(const obman_pointgen_params* p, float* out, float* ws, obman_stream_t stream) {
  if (!params_ok(p) || !out || !ws) return -1;
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(p);
  const FwdWs w = fwd_ws(d);
  const int tr = p->training;
  ObmanProfScope prof(OBMAN_K_DECODER_FWD, st);
  {
    const size_t sm = sizeof(float) * ((size_t)PREP_COLS * (d.C1 + d.B) + PREP_COLS * 4);
    const size_t sm_prep = sizeof(float) * ((size_t)PREP_SCOLS * (d.C1 + d.B) + PREP_SCOLS * 4) + sizeof(double) * (PREP_NT / 64) * 2;
    if (d.ps)
      prep_ps_kernel<<<obman_cdiv(d.C1, PREP_COLS), 256, sm, st>>>(p->w1, p->b1, p->grid, p->feat, d.B, d.N, d.C1, d.ld1, tr, p->eps,
                                                                    p->momentum, p->bn_rm[0], p->bn_rv[0], ws + w.Gx, ws + w.Fx,
                                                                    ws + w.mean1, ws + w.rstd1);
    else
    {  // shared template grid: channel constants + the sample factor, then the vertex factor (and both pre-scaled images) row-wise
      if ((long)d.N * (d.ld1 / 4) + (long)L1F_NT * L1F_ITEMS >= (1L << 32)) return -1;
      prep_kernel<PREP_SCOLS><<<obman_cdiv(d.ld1, PREP_SCOLS), PREP_NT, sm_prep, st>>>(
          p->w1, p->b1, p->grid, p->feat, d.B, d.N, d.C1, d.ld1, tr, p->eps, p->momentum, p->bn_rm[0], p->bn_rv[0], p->bn_w[0], p->bn_b[0],
          ws + w.Fx, ws + w.Fy, ws + w.mean1, ws + w.rstd1, ws + w.gm1);
      OBMAN_LAUNCH_CHECK();
      const long quads = (long)d.N * (d.ld1 / 4);
      const int items = (int)std::min<long>(L1F_ITEMS, std::max<long>(1, quads / (L1F_NT * 1024L)));
      l1_fill_kernel<<<obman_cdiv(quads, L1F_NT * items), L1F_NT, sizeof(float) * 6 * d.ld1, st>>>(
          p->w1, p->grid, ws + w.gm1, ws + w.rstd1, p->bn_w[0], d.N, d.C1, d.ld1, items, ws + w.Gx, ws + w.Gy);
    }
    OBMAN_LAUNCH_CHECK();
  }
  if (d.bf16) return forward_bf16(p, d, w, out, ws, st);
  double* moments = reinterpret_cast<double*>(ws + w.moments);
  const bool f2 = use_rows2f(d);
  {  // h2 = W2 relu(bn1(h1)) + b2
    int rc, mrows = d.rb;
    if (f2) {
      F2GridFeatPre a{ws + w.Gy, ws + w.Fy, d.ld1, d.N};
      F2EpiStore<4> e4{ws + w.H2, p->b2, tr ? moments : nullptr, d.ld2, d.C2};
      F2EpiStore<2> e2{ws + w.H2, p->b2, tr ? moments : nullptr, d.ld2, d.C2};
      // 64-column blocks (K = 515) tile their rows as 8 samples x 4 vertices: the factor loads touch 4 + 8 rows, not 32 + 1
      if ((rc = launch_rows2f_auto<F2GridFeatPre, F2EpiStore>(d, a, p->w2, d.C1, 0, d.C1, d.C2, e4, e2, 2, &mrows, st))) return rc;
    } else {
    AGridFeat a{ws + w.Gx, ws + w.Fx, p->bn_w[0], p->bn_b[0], d.N, d.ld1, (int)d.R, d.C1, d.ps};
    EpiStoreImpl e;
    e.C = ws + w.H2; e.bias = p->b2; e.moments = tr ? moments : nullptr; e.ldc = d.ld2; e.R = (int)d.R; e.Nc = d.C2;
    e.mstride = d.C2;
    rc = launch_rows<AGridFeat, true, EpiStoreImpl>(a, p->w2, d.C1, d.C1, d.C2, d.R, e, st);
    if (rc) return rc;
    }
    const double* mom = moments;
    if (tr && (rc = pre_reduce<double>(mom, mrows, d.C2 * 2, reinterpret_cast<double*>(ws + w.mred), st))) return rc;
    bn_finalize_kernel<<<obman_cdiv(d.C2, 4), 256, 0, st>>>(mom, mrows, d.R, d.C2, tr, p->eps, p->momentum, p->bn_w[1], p->bn_b[1],
                                                               p->bn_rm[1], p->bn_rv[1], ws + w.mean2, ws + w.rstd2, ws + w.s2, ws + w.t2);
    OBMAN_LAUNCH_CHECK();
  }
  {  // h3 = W3 relu(bn2(h2)) + b3
    int rc, mrows = d.rb;
    if (f2) {
      F2BnRelu a{ws + w.H2, ws + w.s2, ws + w.t2, d.ld2, d.C2};
      F2EpiStore<4> e4{ws + w.H3, p->b3, tr ? moments : nullptr, d.ld3, d.C3};
      F2EpiStore<2> e2{ws + w.H3, p->b3, tr ? moments : nullptr, d.ld3, d.C3};
      if ((rc = launch_rows2f_auto<F2BnRelu, F2EpiStore>(d, a, p->w3, d.C2, 0, d.C2, d.C3, e4, e2, 0, &mrows, st))) return rc;
    } else {
    ABnRelu a{ws + w.H2, ws + w.s2, ws + w.t2, d.ld2, (int)d.R, d.C2};
    EpiStoreImpl e;
    e.C = ws + w.H3; e.bias = p->b3; e.moments = tr ? moments : nullptr; e.ldc = d.ld3; e.R = (int)d.R; e.Nc = d.C3;
    e.mstride = d.C3;
    rc = launch_rows<ABnRelu, true, EpiStoreImpl>(a, p->w3, d.C2, d.C2, d.C3, d.R, e, st);
    if (rc) return rc;
    }
    const double* mom = moments;
    if (tr && (rc = pre_reduce<double>(mom, mrows, d.C3 * 2, reinterpret_cast<double*>(ws + w.mred), st))) return rc;
    bn_finalize_kernel<<<obman_cdiv(d.C3, 4), 256, 0, st>>>(mom, mrows, d.R, d.C3, tr, p->eps, p->momentum, p->bn_w[2], p->bn_b[2],
                                                               p->bn_rm[2], p->bn_rv[2], ws + w.mean3, ws + w.rstd3, ws + w.s3, ws + w.t3);
    OBMAN_LAUNCH_CHECK();
  }
  if (l4_wide(d)) l4w_fwd_kernel<float><<<2048, 256, 0, st>>>(ws + w.H3, ws + w.s3, ws + w.t3, p->w4, p->b4, p->out_factor, d.R, out);
  else l4_fwd_kernel<float><<<2048, 256, 0, st>>>(ws + w.H3, d.ld3, ws + w.s3, ws + w.t3, p->w4, p->b4, p->out_factor, d.R, d.C3, out);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_pointgen_bwd(const obman_pointgen_params* p, const float* g_out, const float* ws, float* ws2,
                       const obman_pointgen_grads* g, obman_stream_t stream) {
  if (!params_ok(p) || !g_out || !ws || !ws2 || !g) return -1;
  hipStream_t st = (hipStream_t)stream;
  const Dims d = dims_of(p);
  const FwdWs w = fwd_ws(d);
  const BwdWs v = bwd_ws(d);
  const int tr = p->training;
  const int R = (int)d.R;
  ObmanProfScope prof(OBMAN_K_DECODER_BWD, st);
  int rc;
  const float* Qsum = ws2 + v.Q;  // Q[n,c] = sum_b gy1 (or the single partial it equals, see launch_l1_reduce2)
  if (d.bf16) {
    if ((rc = backward_bf16(p, d, w, v, g_out, ws, ws2, g, &Qsum, st))) return rc;
  } else {
    double* sums = reinterpret_cast<double*>(ws2 + v.sums);
    float *k1 = ws2 + v.k, *k2 = k1 + d.ld1, *k3 = k2 + d.ld1;
    const float f = p->out_factor;
    // ---- layer 4 + BN-3 statistics
    const int l4rows = l4_rows(d);
    const int l4b = obman_cdiv(d.R, l4rows);
    if (l4_wide(d))
      l4w_bwd_kernel<float><<<l4b, 256, 0, st>>>(g_out, ws + w.H3, ws + w.s3, ws + w.t3, ws + w.mean3, ws + w.rstd3, p->w4, f, d.R, l4rows, sums,
                                                  ws2 + v.l4p);
    else
      l4_bwd_kernel<float><<<l4b, 128, 0, st>>>(g_out, ws + w.H3, d.ld3, ws + w.s3, ws + w.t3, ws + w.mean3, ws + w.rstd3, p->w4, f, d.R, d.C3,
                                                 l4rows, sums, ws2 + v.l4p);
    OBMAN_LAUNCH_CHECK();
    {
      const float* lp = ws2 + v.l4p;
      int lrows = l4b;
      if ((rc = pre_reduce<float>(lp, lrows, 3 * d.C3 + 4, ws2 + v.l4red, st))) return rc;
      l4_bwd_finalize_kernel<<<obman_cdiv(3 * d.C3 + 3, 4), 256, 0, st>>>(lp, lrows, d.C3, g->w4, g->b4);
      OBMAN_LAUNCH_CHECK();
    }
    const double* sp = sums;
    int srows = l4b;
    if ((rc = pre_reduce<double>(sp, srows, d.C3 * 2, reinterpret_cast<double*>(ws2 + v.sred), st))) return rc;
    bn_bwd_finalize_kernel<<<obman_cdiv(d.C3, 4), 256, 0, st>>>(sp, srows, d.R, d.C3, tr, p->bn_w[2], ws + w.mean3, ws + w.rstd3, g->bn_w[2],
                                                                   g->bn_b[2], g->b3, k1, k2, k3);
    OBMAN_LAUNCH_CHECK();
    AGradH3 gh3{g_out, p->w4, ws + w.H3, ws + w.s3, ws + w.t3, k1, k2, k3, f, d.ld3, R, d.C3};
    if (use_rows2f(d) && tn3_ok(d.C3, d.C2, d.N)) {  // gW3[o,c] = sum_r gh3[r,o] a2[r,c]: second-generation kernel
      T3GradH3 ta{g_out, p->w4, ws + w.H3, ws + w.s3, ws + w.t3, k1, k2, k3, f, d.ld3, d.C3};
      T3BnRelu tb{ws + w.H2, ws + w.s2, ws + w.t2, d.ld2, d.C2};
      if ((rc = launch_tn3<T3GradH3, T3BnRelu>(ta, tb, d.C3, d.C2, d.R, d.N, ws2 + v.tn, g->w3, d.C2, 0, st))) return rc;
    } else {  // gW3[o,c] = sum_r gh3[r,o] a2[r,c]
      ABnRelu a2{ws + w.H2, ws + w.s2, ws + w.t2, d.ld2, R, d.C2};
      if ((rc = launch_tn<AGradH3, ABnRelu>(gh3, a2, d.C3, d.C2, d.R, TN_CHUNK_ROWS, ws2 + v.tn, g->w3, d.C2, 0, st))) return rc;
    }
    const bool f2 = use_rows2f(d);
    int f2_prow = 0, f2_groups = 0;  // > 0: the layer-2 data gradient wrote P / Q partials itself (F2EpiL1PQ)
    srows = d.rb;
    if (f2) {  // gy2 = (gh3 W3) * (y2 > 0), BN-2 sums: second-generation kernel
      F2GradH3 a{g_out, p->w4, ws + w.H3, ws + w.s3, ws + w.t3, k1, k2, k3, f, d.ld3, d.C3};
      F2EpiMask<4> e4{ws2 + v.GY2, ws + w.H2, sums, ws + w.s2, ws + w.t2, ws + w.mean2, ws + w.rstd2, d.ld2, d.C2};
      F2EpiMask<2> e2{ws2 + v.GY2, ws + w.H2, sums, ws + w.s2, ws + w.t2, ws + w.mean2, ws + w.rstd2, d.ld2, d.C2};
      // 64 columns per block: the gh3 generator (7 VALU operations and two constant vectors per element) plus this epilogue's
      // state spill with four accumulator tiles
      if ((rc = launch_rows2f_auto<F2GradH3, F2EpiMask, false>(d, a, p->w3, d.C2, 1, d.C3, d.C2, e4, e2, 0, &srows, st))) return rc;
    } else {  // gy2 = (gh3 W3) * (y2 > 0), BN-2 sums
      EpiMaskStatsImpl e;
      e.C = ws2 + v.GY2; e.sums = sums; e.ldc = d.ld2; e.R = R; e.Nc = d.C2; e.mode = 0;
      e.H = ws + w.H2; e.s = ws + w.s2; e.t = ws + w.t2; e.mean = ws + w.mean2; e.rstd = ws + w.rstd2;
      e.Gx = e.Fx = e.gamma = e.beta = nullptr; e.N = d.N; e.ps = 0; e.sstride = d.C2;
      if ((rc = launch_rows<AGradH3, false, EpiMaskStatsImpl>(gh3, p->w3, d.C2, d.C3, d.C2, d.R, e, st))) return rc;
    }
    sp = sums;
    if ((rc = pre_reduce<double>(sp, srows, d.C2 * 2, reinterpret_cast<double*>(ws2 + v.sred), st))) return rc;
    bn_bwd_finalize_kernel<<<obman_cdiv(d.C2, 4), 256, 0, st>>>(sp, srows, d.R, d.C2, tr, p->bn_w[1], ws + w.mean2, ws + w.rstd2, g->bn_w[1],
                                                                   g->bn_b[1], g->b2, k1, k2, k3);
    OBMAN_LAUNCH_CHECK();
    AGradH gh2{ws2 + v.GY2, ws + w.H2, k1, k2, k3, d.ld2, R, d.C2};
    AGridFeat a1{ws + w.Gx, ws + w.Fx, p->bn_w[0], p->bn_b[0], d.N, d.ld1, R, d.C1, d.ps};
    // gW2[o,c] = sum_r gh2[r,o] a1[r,c].  (Tried in r02 and dropped: [256 x 512] on the tile grid + the 257th row and the last
    // three columns on VALU edge kernels - the tile kernel went 244 -> 146 us, the extra passes over the regenerated operands
    // cost 180 us; profiles/r02_kernels.md.)
    if (f2 && tn3_ok(d.C2, d.C1, d.N)) {  // second-generation kernel, layer-1 activation from the pre-scaled factors
      T3GradH ta{ws2 + v.GY2, ws + w.H2, k1, k2, k3, d.ld2, d.C2};
      T3Pre tb{ws + w.Gy, ws + w.Fy, d.ld1, d.N};
      if ((rc = launch_tn3<T3GradH, T3Pre>(ta, tb, d.C2, d.C1, d.R, d.N, ws2 + v.tn, g->w2, d.C1, 0, st))) return rc;
    } else if ((rc = launch_tn<AGradH, AGridFeat>(gh2, a1, d.C2, d.C1, d.R, TN_CHUNK_ROWS, ws2 + v.tn, g->w2, d.C1, 0, st))) return rc;
    if (f2) {  // gy1 = (gh2 W2) * (y1 > 0): second-generation kernel, mask from the pre-scaled factors
      F2GradH a{ws2 + v.GY2, ws + w.H2, k1, k2, k3, d.ld2, d.C2};
      static const int pq_on = [] { const char* e = getenv("OBMAN_DEC_F2PQ"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = materialise gy1
      const bool wide = f2_wide<F2GradH, F2EpiL1PQ<4>, F2EpiL1PQ<2>>(d, d.C2, d.C1) &&
                        f2_geo(d, d.C1, 4, 2).tiles >= 4 * F2_WAVES * device_cus() / f2_geo(d, d.C1, 4, 2).ngroups;
      const F2Geo gq = f2_geo_pq(d, d.C1, wide ? 4 : 2);
      if (pq_on && gq.wpg > 0 && f2_lds_bytes<F2GradH, F2EpiL1PQ<2>, 2>(kpad8(d.C2), gq) <= F2_LDS_LIMIT) {
        // P / Q partials straight from the accumulators: gy1 is never written, l1_reduce_kernel is not needed
        if (wide) {
          F2EpiL1PQ<4> e{ws2 + v.Pp, ws2 + v.Qp, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
          if ((rc = launch_rows2f<F2GradH, F2EpiL1PQ<4>, 4>(a, p->w2, d.C1, 1, d.C2, d.C1, gq, e, st))) return rc;
        } else {
          F2EpiL1PQ<2> e{ws2 + v.Pp, ws2 + v.Qp, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
          if ((rc = launch_rows2f<F2GradH, F2EpiL1PQ<2>, 2>(a, p->w2, d.C1, 1, d.C2, d.C1, gq, e, st))) return rc;
        }
        f2_prow = gq.prow();
        f2_groups = (d.B + 7) / 8;
      } else {
      F2EpiL1<4> e4{ws2 + v.GY1, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
      F2EpiL1<2> e2{ws2 + v.GY1, ws + w.Gy, ws + w.Fy, d.ld1, d.C1};
      if ((rc = launch_rows2f_auto<F2GradH, F2EpiL1>(d, a, p->w2, d.C1, 1, d.C2, d.C1, e4, e2, 2, nullptr, st))) return rc;  // row mode 2: see F2EpiL1
      }
    } else {  // gy1 = (gh2 W2) * (y1 > 0)
      EpiMaskStatsImpl e;
      e.C = ws2 + v.GY1; e.sums = sums; e.ldc = d.ld1; e.R = R; e.Nc = d.C1; e.mode = 1;
      e.H = e.s = e.t = e.mean = e.rstd = nullptr;
      e.Gx = ws + w.Gx; e.Fx = ws + w.Fx; e.gamma = p->bn_w[0]; e.beta = p->bn_b[0]; e.N = d.N; e.ps = d.ps; e.sstride = d.C1;
      if ((rc = launch_rows<AGradH, false, EpiMaskStatsImpl>(gh2, p->w2, d.C1, d.C2, d.C1, d.R, e, st))) return rc;
    }
    if (d.ps) {  // per-sample grid: BN-1 backward coefficients from the epilogue's sums, then one pass over gy1 and Gx
      sp = sums;
      srows = d.rb;
      if ((rc = pre_reduce<double>(sp, srows, d.C1 * 2, reinterpret_cast<double*>(ws2 + v.sred), st))) return rc;
      l1ps_coef_kernel<<<obman_cdiv(d.C1, 4), 256, 0, st>>>(sp, srows, d.R, d.C1, tr, p->bn_w[0], ws + w.rstd1, g->bn_w[0], g->bn_b[0], k1, k2,
                                                             k3);
      OBMAN_LAUNCH_CHECK();
      const int ch = obman_cdiv(d.N, L1PS_ROWS);
      l1ps_reduce_kernel<<<dim3(ch, d.B, obman_cdiv(d.C1, 256)), 256, 0, st>>>(ws2 + v.GY1, ws + w.Gx, ws + w.Fx, p->grid, k1, k2, k3, d.ld1,
                                                                                 d.N, d.C1, ch, ws2 + v.Pp, ws2 + v.Qp);
      OBMAN_LAUNCH_CHECK();
      l1ps_finalize_kernel<<<obman_cdiv(d.C1, 256), 256, 0, st>>>(ws2 + v.Pp, ws2 + v.Qp, d.ld1, d.B, d.C1, ch, ws2 + v.dF, g->b1, g->w1);
      OBMAN_LAUNCH_CHECK();
    } else {
    // ---- layer 1 in factored form: P[b,c] = sum_n gy1, Q[n,c] = sum_b gy1 from one read of gy1 - or, with the second-generation
    // data-gradient kernel, from the partials its epilogue wrote (f2_prow > 0)
    const L1Geo lg = l1_geo(d);
    int prow = lg.tiles, groups = lg.groups;
    if (f2_prow > 0) {
      prow = f2_prow;
      groups = f2_groups;
    } else {
      l1_reduce_kernel<<<dim3(lg.tiles, lg.groups, obman_cdiv(d.C1, 256)), 256, 0, st>>>(ws2 + v.GY1, d.ld1, d.B, d.N, d.C1, lg.S, ws2 + v.Pp,
                                                                                           ws2 + v.Qp);
      OBMAN_LAUNCH_CHECK();
    }
    Qsum = launch_l1_reduce2(ws2 + v.Pp, ws2 + v.Qp, d.ld1, d.B, d.N, d.C1, prow, groups, ws2 + v.P, ws2 + v.Q, st);
    OBMAN_LAUNCH_CHECK();
    }
  }
  if (d.ps) {
    // dF, g_b1, gW1[:, 0:3], g_gamma1, g_beta1 are done (l1ps_* kernels above)
  } else if (d.N > L1_SPLIT_N) {
    const int nseg = obman_cdiv(d.N, L1_SEG_ROWS);
    double* seg = reinterpret_cast<double*>(ws2 + v.seg);
    const dim3 grid(obman_cdiv(d.C1, 64), nseg);
    l1_seg_stats_kernel<<<grid, 1024, 0, st>>>(Qsum, ws + w.Gx, p->grid, d.ld1, d.N, d.C1, seg);
    OBMAN_LAUNCH_CHECK();
    l1_seg_final_kernel<<<obman_cdiv(d.C1, 64), 1024, 0, st>>>(ws2 + v.P, ws + w.Fx, d.ld1, d.B, d.N, d.C1, tr, p->bn_w[0], ws + w.rstd1, p->grid, seg,
                                                                nseg, g->bn_w[0], g->bn_b[0], g->b1, ws2 + v.dF, g->w1);
    OBMAN_LAUNCH_CHECK();
  } else {
    l1_finalize_kernel<<<obman_cdiv(d.C1, L1F_CH), 1024, 0, st>>>(ws2 + v.P, Qsum, ws + w.Gx, ws + w.Fx, d.ld1, d.B, d.N, d.C1, tr, p->bn_w[0],
                                                               ws + w.rstd1, p->grid, g->bn_w[0], g->bn_b[0], g->b1, g->w1, ws2 + v.dF, ws2 + v.dG);
    OBMAN_LAUNCH_CHECK();
  }
  const int Cf = d.C1 - 3;
  {  // gW1[c, 3+k] = sum_b dF[b,c] feat[b,k]
    APlain adf{ws2 + v.dF, d.ld1, d.B, d.C1};
    APlain afe{p->feat, Cf, d.B, Cf};
    rc = launch_tn<APlain, APlain>(adf, afe, d.C1, Cf, d.B, TN_CHUNK_ROWS, ws2 + v.tn, g->w1, d.C1, 3, st);
    if (rc) return rc;
  }
  if (g->feat) {  // g_feat[b,k] = sum_c dF[b,c] W1[c,3+k]
    gfeat_kernel<<<dim3(obman_cdiv(Cf, 64), d.B), 256, 0, st>>>(ws2 + v.dF, d.ld1, p->w1, d.C1, d.B, g->feat);
    OBMAN_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
