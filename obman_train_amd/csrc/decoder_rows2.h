// K6, bf16-MFMA flavour, second generation of the "rows" GEMMs (h2, h3, gy2): included by decoder.hip inside namespace dec,
// after decoder_bf16.h (operand generators B*, pack8 / pair_exchange helpers, xcd_virtual_id).
//
// What was wrong with the first generation (rows_bf16_kernel): 13 % matrix-pipe busy (profiles/r02_kernels.md).  A 128 x 320
// block re-staged its 20 KB weight tile and its freshly generated 8 KB activation tile through LDS for every 32-deep k-tile,
// with a block-wide barrier per k-tile and ONE block per CU: load -> transform -> LDS write -> barrier -> LDS read -> MFMA ran
// one after the other, 17 times per block, 31 blocks per CU.
//
// These GEMMs are tall and skinny: 1 M rows x (128 .. 257) columns x (128 .. 515) deep.  So here the WEIGHTS are the stationary
// operand and the activations never touch LDS:
//   * a persistent block (one per CU) copies its 128-column slice of the bf16 weight image into LDS ONCE ([col][k], pitch
//     Kp + 8 elements: conflict-free ds_read_b128 fragments) and then walks over row tiles; 257 = 2 x 128 + 1 and
//     515 = 4 x 128 + 3: the last column group carries its 1 .. 3 leftover columns as VALU side products (no padding tile);
//   * a wave owns 64 rows (two 32-row MFMA fragments) x 128 columns (4 tiles): 128 accumulator registers.  Its A fragments
//     are GENERATED IN REGISTERS in exactly the MFMA operand layout: lane (row = lane & 31, half = lane >> 5) loads the 8
//     consecutive k of its row with 16-byte loads (the same "one row, 8 k" unit the B* generators already work in), applies
//     the fused BatchNorm / ReLU / BatchNorm-backward transform and packs to bf16 - no LDS write, no LDS read, no barrier;
//   * the k-loop has NO barrier at all: the 8 waves of a block drift freely, one wave's loads and VALU transform overlap the
//     other wave's MFMAs on the same SIMD (2 waves per SIMD);
//   * BatchNorm statistics are accumulated per lane across ALL row tiles a wave processes (fp32 inside a tile, fp64 across
//     tiles) and leave the block once, at the end (one cross-wave reduction per block instead of one per 128 rows).
// A wave tile is 1 sample x 64 consecutive template vertices (the 8 waves of a block = 8 samples over the SAME vertices, so
// the layer-1 grid factor rows are shared through L1 / L2), or - for the dA(gy1) GEMM - 4 samples x 16 vertices.
#pragma once

constexpr int R2_NT = 4;               // 32-column MFMA tiles per wave
constexpr int R2_COLS = 32 * R2_NT;    // columns per block (+ up to R2_SIDE side columns in the last group)
constexpr int R2_SIDE = 3;
constexpr int R2_THREADS = 512;
constexpr int R2_WAVES = 8;

inline int kpad16(int K) { return (K + 15) / 16 * 16; }  // k extent of the weight image the rows2 kernels use (one MFMA k-step)

struct R2Geo {
  int R, N, B;
  int mode;     // 0: wave tile = 1 sample x 64 vertices (block = 8 samples); 1: 4 samples x 16 vertices (block = 32 samples)
  int nvt, nbg; // vertex tiles, sample groups
  int ngroups;  // column groups of R2_COLS
  int slots;    // persistent blocks per column group = spb * nbg
  int spb;      // slots per sample group
  int chunk;    // vertex tiles per slot
  // row i (0..31) of fragment f of wave `wave` in block tile (bg, vt)
  __device__ __forceinline__ void row(int bg, int vt, int wave, int f, int i, int& b, int& n, long& r, bool& ok) const {
    if (mode == 0) {
      b = bg * 8 + wave;
      n = vt * 64 + f * 32 + i;
    } else {
      b = bg * 32 + wave * 4 + f * 2 + (i >> 4);
      n = vt * 16 + (i & 15);
    }
    ok = b < B && n < N;
    if (!ok) { b = 0; n = 0; }
    r = (long)b * N + n;
  }
};

// ------------------------------------------------------------------------------------------------ operand pairs
// Two fragments (rows r0, r1 of one lane) through an existing operand generator; BGridFeat shares the sample's Fx chunk.
template <class AOp>
struct R2Pair {
  struct Raw2 { typename AOp::Raw a, b; };
  static __device__ __forceinline__ void load(const AOp& op, Raw2& q, const typename AOp::Row& r0, const typename AOp::Row& r1, int k) {
    op.load(q.a, r0, k);
    op.load(q.b, r1, k);
  }
  static __device__ __forceinline__ void fin(const AOp& op, const typename AOp::Row& r0, const typename AOp::Row& r1, const float* kcs, int Kp,
                                             int k, const Raw2& q, float* o0, float* o1) {
    op.fin(r0, kcs, Kp, k, q.a, o0);
    op.fin(r1, kcs, Kp, k, q.b, o1);
  }
};
template <>
struct R2Pair<BGridFeat> {  // mode 0: both fragments belong to ONE sample -> one Fx chunk for both
  struct Raw2 { float4 g0, g1, h0, h1, f0, f1; };
  static __device__ __forceinline__ void load(const BGridFeat& op, Raw2& q, const BGridFeat::Row& r0, const BGridFeat::Row& r1, int k) {
    const int c = k <= op.ld - 8 ? k : op.ld - 8;
    q.g0 = *reinterpret_cast<const float4*>(r0.g + c); q.g1 = *reinterpret_cast<const float4*>(r0.g + c + 4);
    q.h0 = *reinterpret_cast<const float4*>(r1.g + c); q.h1 = *reinterpret_cast<const float4*>(r1.g + c + 4);
    q.f0 = *reinterpret_cast<const float4*>(r0.f + c); q.f1 = *reinterpret_cast<const float4*>(r0.f + c + 4);
  }
  static __device__ __forceinline__ void fin(const BGridFeat&, const BGridFeat::Row&, const BGridFeat::Row&, const float* kcs, int Kp, int k,
                                             const Raw2& q, float* o0, float* o1) {
    const float4 ga0 = *reinterpret_cast<const float4*>(kcs + k), ga1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 be0 = *reinterpret_cast<const float4*>(kcs + Kp + k), be1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float ga[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
    const float be[8] = {be0.x, be0.y, be0.z, be0.w, be1.x, be1.y, be1.z, be1.w};
    const float fx[8] = {q.f0.x, q.f0.y, q.f0.z, q.f0.w, q.f1.x, q.f1.y, q.f1.z, q.f1.w};
    const float x0[8] = {q.g0.x, q.g0.y, q.g0.z, q.g0.w, q.g1.x, q.g1.y, q.g1.z, q.g1.w};
    const float x1[8] = {q.h0.x, q.h0.y, q.h0.z, q.h0.w, q.h1.x, q.h1.y, q.h1.z, q.h1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = fmaxf(__fmaf_rn(ga[j], x0[j] + fx[j], be[j]), 0.f);
      o1[j] = fmaxf(__fmaf_rn(ga[j], x1[j] + fx[j], be[j]), 0.f);
    }
  }
};

// the per-channel constants are the same for both fragments (same k): read them from LDS once
template <>
struct R2Pair<BBnRelu> {
  struct Raw2 { u32x4 a, b; };
  static __device__ __forceinline__ void load(const BBnRelu& op, Raw2& q, const BBnRelu::Row& r0, const BBnRelu::Row& r1, int k) {
    const int c = k <= op.ld - 8 ? k : op.ld - 8;
    q.a = *reinterpret_cast<const u32x4*>(r0.p + c);
    q.b = *reinterpret_cast<const u32x4*>(r1.p + c);
  }
  static __device__ __forceinline__ void fin(const BBnRelu&, const BBnRelu::Row&, const BBnRelu::Row&, const float* kcs, int Kp, int k,
                                             const Raw2& q, float* o0, float* o1) {
    const float4 s0 = *reinterpret_cast<const float4*>(kcs + k), s1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(kcs + Kp + k), t1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float tc[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    float ha[8], hb[8];
    unpack8(q.a, ha);
    unpack8(q.b, hb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = fmaxf(__fmaf_rn(sc[j], ha[j], tc[j]), 0.f);
      o1[j] = fmaxf(__fmaf_rn(sc[j], hb[j], tc[j]), 0.f);
    }
  }
};
template <>
struct R2Pair<BGradH3> {
  struct Raw2 { u32x4 a, b; };
  static __device__ __forceinline__ void load(const BGradH3& op, Raw2& q, const BGradH3::Row& r0, const BGradH3::Row& r1, int k) {
    const int c = k <= op.ld - 8 ? k : op.ld - 8;
    q.a = *reinterpret_cast<const u32x4*>(r0.p + c);
    q.b = *reinterpret_cast<const u32x4*>(r1.p + c);
  }
  static __device__ __forceinline__ void fin(const BGradH3&, const BGradH3::Row& r0, const BGradH3::Row& r1, const float* kcs, int, int k,
                                             const Raw2& q, float* o0, float* o1) {
    float ha[8], hb[8];
    unpack8(q.a, ha);
    unpack8(q.b, hb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 c0 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8);      // s, t, kb, kc
      const float4 c1 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8 + 4);  // ka*w0, ka*w1, ka*w2
      // select instead of branch: the dot product is three FMAs, cheaper than a divergent skip
      const float da = __fmaf_rn(r0.g2, c1.z, __fmaf_rn(r0.g1, c1.y, r0.g0 * c1.x));
      const float db = __fmaf_rn(r1.g2, c1.z, __fmaf_rn(r1.g1, c1.y, r1.g0 * c1.x));
      const float ga = __fmaf_rn(c0.x, ha[j], c0.y) > 0.f ? da : 0.f;
      const float gb = __fmaf_rn(c0.x, hb[j], c0.y) > 0.f ? db : 0.f;
      o0[j] = ga + __fmaf_rn(c0.z, ha[j], c0.w);
      o1[j] = gb + __fmaf_rn(c0.z, hb[j], c0.w);
    }
  }
};

// ------------------------------------------------------------------------------------------------ epilogues
struct R2Ctx {
  int lane, wave, c0, nside, last_group, slot, bg, vt;
};

__device__ __forceinline__ double r2_pick(const double (&v)[R2_SIDE], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }

// fp64 sums held by (two lane halves) x (eight waves) -> dst[(slot * Nc + col) * 2 + {0,1}], fixed order.  `smem` is the
// (dead) weight slice; called by every thread of the block after its last tile.  Side-column sums arrive as one fp32 partial
// per lane (a lane of half 0 owns one row per fragment and tile: a few dozen addends).
__device__ __forceinline__ void r2_flush_cols(double (&d1)[R2_NT], double (&d2)[R2_NT], const float (&f1)[R2_SIDE], const float (&f2)[R2_SIDE],
                                              const R2Ctx& c, int Nc, double* __restrict__ dst, char* smem) {
  const int li = c.lane & 31;
#pragma unroll
  for (int j = 0; j < R2_NT; ++j) { d1[j] += __shfl_xor(d1[j], 32, 64); d2[j] += __shfl_xor(d2[j], 32, 64); }
  double e1[R2_SIDE], e2[R2_SIDE];
#pragma unroll
  for (int t = 0; t < R2_SIDE; ++t) {  // every lane ends up with the wave's total
    e1[t] = (double)f1[t];
    e2[t] = (double)f2[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e1[t] += __shfl_xor(e1[t], off, 64); e2[t] += __shfl_xor(e2[t], off, 64); }
  }
  double* red = reinterpret_cast<double*>(smem);  // [7 waves][R2_NT + 1][32][2]
  __syncthreads();                                // every wave is done with the weight slice
  if (c.wave > 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { double* q = red + ((((c.wave - 1) * (R2_NT + 1) + j) * 32) + li) * 2; q[0] = d1[j]; q[1] = d2[j]; }
    if (li < R2_SIDE) { double* q = red + ((((c.wave - 1) * (R2_NT + 1) + R2_NT) * 32) + li) * 2; q[0] = r2_pick(e1, li); q[1] = r2_pick(e2, li); }
  }
  __syncthreads();
  if (c.wave == 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      if (col < Nc) {
        double a = d1[j], b = d2[j];
#pragma unroll
        for (int w = 0; w < R2_WAVES - 1; ++w) { const double* q = red + (((w * (R2_NT + 1) + j) * 32) + li) * 2; a += q[0]; b += q[1]; }
        double* o = dst + ((size_t)c.slot * Nc + col) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
    if (li < c.nside) {
      double a = r2_pick(e1, li), b = r2_pick(e2, li);
#pragma unroll
      for (int w = 0; w < R2_WAVES - 1; ++w) { const double* q = red + (((w * (R2_NT + 1) + R2_NT) * 32) + li) * 2; a += q[0]; b += q[1]; }
      double* o = dst + ((size_t)c.slot * Nc + c.c0 + R2_COLS + li) * 2;
      o[0] = a;
      o[1] = b;
    }
  }
}

struct EpiStoreB2 {  // C[r,n] = bf16(acc + bias[n]); fp64 column moments (sum, sum of squares of the STORED values) per slot
  bfraw* C;
  const float* bias;
  double* moments;  // [slots][Nc][2] or null
  int ldc, Nc;
  struct State {  // only what must persist across the wave's row tiles (constants are re-read per tile: L1 hits, and registers are tight)
    double d1[R2_NT], d2[R2_NT];
    float e1[R2_SIDE], e2[R2_SIDE];
  };
  __device__ __forceinline__ void init(State& s, const R2Ctx&) const {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { s.d1[j] = 0.0; s.d2[j] = 0.0; }
#pragma unroll
    for (int t = 0; t < R2_SIDE; ++t) { s.e1[t] = 0.f; s.e2[t] = 0.f; }
  }
  __device__ __forceinline__ void tile(State& s, const f32x16 (&acc)[2][R2_NT], const float (&side)[2][R2_SIDE], const R2Ctx& c,
                                       const R2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const bool odd = c.lane & 1;
    float bv[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      bv[j] = (bias && col < Nc) ? bias[col] : 0.f;
    }
    float s1[R2_NT], s2[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int i0 = acc_row(2 * p, c.lane);
        int b, n; long r0; bool ok0;
        geo.row(c.bg, c.vt, c.wave, f, i0, b, n, r0, ok0);
        const bool ok1 = ok0 && n + 1 < geo.N;  // row i0 + 1: the next vertex of the same sample
        bfraw* dst = C + (size_t)(odd ? r0 + 1 : r0) * ldc;
        const bool okw = odd ? ok1 : ok0;
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) {
          const int cl = c.c0 + j * 32 + li;
          const bool cok = cl < Nc;
          const unsigned pk = pack_bf16(cok ? acc[f][j][2 * p] + bv[j] : 0.f, cok ? acc[f][j][2 * p + 1] + bv[j] : 0.f);
          const float v0 = ok0 ? bf_lo(pk) : 0.f, v1 = ok1 ? bf_hi(pk) : 0.f;
          s1[j] += v0 + v1;
          s2[j] = __fmaf_rn(v0, v0, __fmaf_rn(v1, v1, s2[j]));
          const unsigned w = pair_exchange(pk, odd);
          if (okw && (cl & ~1) < ldc) *reinterpret_cast<unsigned*>(dst + (cl & ~1)) = w;
        }
      }
      // side columns and the pitch padding behind the last real column: one row per lane (half 0 / half 1)
      if (c.last_group) {
        int b, n; long r; bool ok;
        geo.row(c.bg, c.vt, c.wave, f, li, b, n, r, ok);
        if (ok && h == 0) {
#pragma unroll
          for (int t = 0; t < R2_SIDE; ++t) {
            if (t < c.nside) {
              const int col = c.c0 + R2_COLS + t;
              const unsigned pk = pack_bf16(side[f][t] + (bias ? bias[col] : 0.f), 0.f);
              C[(size_t)r * ldc + col] = (bfraw)(pk & 0xffffu);
              const float v = bf_lo(pk);
              s.e1[t] += v;
              s.e2[t] = __fmaf_rn(v, v, s.e2[t]);
            }
          }
        }
        if (ok && h == 1) {
          for (int col = c.c0 + R2_COLS + c.nside; col < ldc; ++col) C[(size_t)r * ldc + col] = 0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { s.d1[j] += (double)s1[j]; s.d2[j] += (double)s2[j]; }
  }
  __device__ __forceinline__ void flush(State& s, const R2Ctx& c, const R2Geo&, char* smem) const {
    if (moments) r2_flush_cols(s.d1, s.d2, s.e1, s.e2, c, Nc, moments, smem);
  }
};

struct EpiMaskB2 {  // C = bf16(acc * (y > 0)), y = s*H+t; column sums S1 = sum C, S2 = sum C * xhat, xhat = (H - mean) * rstd
  bfraw* C;
  const bfraw* H;  // same pitch as C
  double* sums;    // [slots][Nc][2]
  const float *s, *t, *mean, *rstd;
  int ldc, Nc;
  struct State {
    double d1[R2_NT], d2[R2_NT];
    float e1[R2_SIDE], e2[R2_SIDE];
  };
  __device__ __forceinline__ void init(State& q, const R2Ctx&) const {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { q.d1[j] = 0.0; q.d2[j] = 0.0; }
#pragma unroll
    for (int u = 0; u < R2_SIDE; ++u) { q.e1[u] = 0.f; q.e2[u] = 0.f; }
  }
  __device__ __forceinline__ void tile(State& q, const f32x16 (&acc)[2][R2_NT], const float (&side)[2][R2_SIDE], const R2Ctx& c,
                                       const R2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const bool odd = c.lane & 1;
    float cs[R2_NT], ct[R2_NT], cm[R2_NT], cr[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      const bool cok = col < Nc;
      const int cc = cok ? col : 0;
      cs[j] = cok ? s[cc] : 0.f; ct[j] = cok ? t[cc] : 0.f; cm[j] = cok ? mean[cc] : 0.f; cr[j] = cok ? rstd[cc] : 0.f;
    }
    float p1[R2_NT], p2[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { p1[j] = 0.f; p2[j] = 0.f; }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int i0 = acc_row(2 * p, c.lane);
        int b, n; long r0; bool ok0;
        geo.row(c.bg, c.vt, c.wave, f, i0, b, n, r0, ok0);
        const bool ok1 = ok0 && n + 1 < geo.N;
        const size_t ro = (size_t)(odd ? r0 + 1 : r0) * ldc;
        const bool okw = odd ? ok1 : ok0;
        unsigned hw[R2_NT];
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) {
          const int cl = c.c0 + j * 32 + li;
          hw[j] = (okw && (cl & ~1) < ldc) ? *reinterpret_cast<const unsigned*>(H + ro + (cl & ~1)) : 0u;
        }
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) {
          const int cl = c.c0 + j * 32 + li;
          const bool cok = cl < Nc;
          float h0, h1;
          pair_unexchange(hw[j], odd, h0, h1);
          const float g0 = (cok && ok0 && __fmaf_rn(cs[j], h0, ct[j]) > 0.f) ? acc[f][j][2 * p] : 0.f;
          const float g1 = (cok && ok1 && __fmaf_rn(cs[j], h1, ct[j]) > 0.f) ? acc[f][j][2 * p + 1] : 0.f;
          const unsigned pk = pack_bf16(g0, g1);
          const float v0 = bf_lo(pk), v1 = bf_hi(pk);
          p1[j] += v0 + v1;
          p2[j] = __fmaf_rn(v0, (h0 - cm[j]) * cr[j], __fmaf_rn(v1, (h1 - cm[j]) * cr[j], p2[j]));
          const unsigned w = pair_exchange(pk, odd);
          if (okw && (cl & ~1) < ldc) *reinterpret_cast<unsigned*>(C + ro + (cl & ~1)) = w;
        }
      }
      if (c.last_group) {
        int b, n; long r; bool ok;
        geo.row(c.bg, c.vt, c.wave, f, li, b, n, r, ok);
        if (ok && h == 0) {
#pragma unroll
          for (int u = 0; u < R2_SIDE; ++u) {
            if (u < c.nside) {
              const int col = c.c0 + R2_COLS + u;
              const size_t o = (size_t)r * ldc + col;
              const float hv = __uint_as_float((unsigned)H[o] << 16);
              const float g = __fmaf_rn(s[col], hv, t[col]) > 0.f ? side[f][u] : 0.f;
              const unsigned pk = pack_bf16(g, 0.f);
              C[o] = (bfraw)(pk & 0xffffu);
              const float v = bf_lo(pk);
              q.e1[u] += v;
              q.e2[u] = __fmaf_rn(v, (hv - mean[col]) * rstd[col], q.e2[u]);
            }
          }
        }
        if (ok && h == 1) {
          for (int col = c.c0 + R2_COLS + c.nside; col < ldc; ++col) C[(size_t)r * ldc + col] = 0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { q.d1[j] += (double)p1[j]; q.d2[j] += (double)p2[j]; }
  }
  __device__ __forceinline__ void flush(State& q, const R2Ctx& c, const R2Geo&, char* smem) const {
    r2_flush_cols(q.d1, q.d2, q.e1, q.e2, c, Nc, sums, smem);
  }
};

// ------------------------------------------------------------------------------------------------ the kernel
// grid = ngroups * slots blocks (1-D, XCD-aware virtual ids: the column groups of one slot are neighbours on one XCD - they
// stream the same activation rows).  Dynamic LDS: weight slice [(R2_COLS + R2_SIDE)][Kp + 8] bf16, then the generator's
// per-channel constants [AOp::NC][Kp] fp32.
template <class AOp, class Epi>
__global__ __launch_bounds__(R2_THREADS) void rows2_bf16_kernel(AOp aop, const bfraw* __restrict__ Wb, int Kp, int Nc, Epi epi, R2Geo geo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef R2Pair<AOp> P;
  const int KP2 = Kp + 8;
  bfraw* Ws = reinterpret_cast<bfraw*>(smem);
  float* kcs = reinterpret_cast<float*>(Ws + (size_t)(R2_COLS + R2_SIDE) * KP2);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int vid = xcd_virtual_id(blockIdx.x, gridDim.x), cg = vid % geo.ngroups, slot = vid / geo.ngroups;
  const int c0 = cg * R2_COLS;
  const int last_group = cg == geo.ngroups - 1;
  const int gcols = last_group ? Nc - c0 : R2_COLS;  // the last group holds the remainder: up to R2_COLS + R2_SIDE columns
  const int nside = gcols > R2_COLS ? gcols - R2_COLS : 0;
  {
    const int chunks = Kp >> 3, total = (R2_COLS + R2_SIDE) * chunks;
    for (int i = tid; i < total; i += R2_THREADS) {
      const int cc = i / chunks, q = i - cc * chunks;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (cc < gcols) v = *reinterpret_cast<const u32x4*>(Wb + (size_t)(c0 + cc) * Kp + q * 8);
      *reinterpret_cast<u32x4*>(Ws + (size_t)cc * KP2 + q * 8) = v;
    }
  }
  aop.stage(kcs, Kp, tid);
  __syncthreads();

  R2Ctx ctx{lane, wave, c0, nside, last_group, slot, 0, 0};
  typename Epi::State est;
  epi.init(est, ctx);
  const int bg = slot / geo.spb, sq = slot - bg * geo.spb;
  const int vt_beg = sq * geo.chunk, vt_end = vt_beg + geo.chunk < geo.nvt ? vt_beg + geo.chunk : geo.nvt;
  const int nks = Kp >> 4;
  const bfraw* wlane = Ws + (size_t)li * KP2 + h * 8;
  ctx.bg = bg;
  for (int vt = vt_beg; vt < vt_end; ++vt) {
    ctx.vt = vt;
    typename AOp::Row row0, row1;
    bool ok0, ok1;
    {
      int b, n; long r;
      geo.row(bg, vt, wave, 0, li, b, n, r, ok0);
      row0 = aop.row(r, b, n, ok0);
      geo.row(bg, vt, wave, 1, li, b, n, r, ok1);
      row1 = aop.row(r, b, n, ok1);
    }
    f32x16 acc[2][R2_NT];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < R2_NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;
    float side[2][R2_SIDE];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) side[f][t] = 0.f;

    // One k-step = one 16-deep MFMA per (fragment, column tile).  The raw operand chunk of step s + 1 is requested as soon as
    // step s has turned its own chunk into the packed fragments (single register set: 128 accumulators leave no room for a
    // deeper queue); the 8 MFMAs of the step and the SIMD's other wave cover the round trip.  Requests past the last step are
    // clamped re-reads that are never consumed, so the loop is branch-free.
    typename P::Raw2 q;
    P::load(aop, q, row0, row1, h * 8);
    for (int s = 0; s < nks; ++s) {
      u32x4 a0, a1;
      {
        float o0[8], o1[8];
        P::fin(aop, row0, row1, kcs, Kp, s * 16 + h * 8, q, o0, o1);
        a0 = pack8(o0);
        a1 = pack8(o1);
      }
      P::load(aop, q, row0, row1, (s + 1) * 16 + h * 8);
      if (!ok0) a0 = u32x4{0u, 0u, 0u, 0u};
      if (!ok1) a1 = u32x4{0u, 0u, 0u, 0u};
      const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0), fa1 = __builtin_bit_cast(bf16x8, a1);
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + s * 16);
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb, acc[1][j], 0, 0, 0);
      }
      if (nside) {  // leftover columns of the last group on the VALU, from the SAME rounded operands the MFMAs consume
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t) {
          if (t < nside) {
            const u32x4 wv = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + s * 16 + h * 8);
            float s0 = side[0][t], s1 = side[1][t];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned ww = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
              const unsigned w0 = e == 0 ? a0.x : (e == 1 ? a0.y : (e == 2 ? a0.z : a0.w));
              const unsigned w1 = e == 0 ? a1.x : (e == 1 ? a1.y : (e == 2 ? a1.z : a1.w));
              s0 = __fmaf_rn(bf_hi(w0), bf_hi(ww), __fmaf_rn(bf_lo(w0), bf_lo(ww), s0));
              s1 = __fmaf_rn(bf_hi(w1), bf_hi(ww), __fmaf_rn(bf_lo(w1), bf_lo(ww), s1));
            }
            side[0][t] = s0;
            side[1][t] = s1;
          }
        }
      }
    }
    if (nside) {  // the two lane halves covered different k: combine
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t) side[f][t] += __shfl_xor(side[f][t], 32, 64);
    }
    epi.tile(est, acc, side, ctx, geo);
  }
  epi.flush(est, ctx, geo, smem);
}
