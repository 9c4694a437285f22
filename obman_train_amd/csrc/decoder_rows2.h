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
//   * a wave owns 32 rows (one MFMA fragment) x 128 columns (4 tiles): 64 accumulator registers.  Its A fragment is
//     GENERATED IN REGISTERS in exactly the MFMA operand layout: lane (row = lane & 31, half = lane >> 5) loads the 8
//     consecutive k of its row with 16-byte loads (the same "one row, 8 k" unit the B* generators already work in), applies
//     the fused BatchNorm / ReLU / BatchNorm-backward transform and packs to bf16 - no LDS write, no LDS read, no barrier.
//     (First version: two fragments = 128 accumulators per wave.  Parity-green but SLOWER than the kernels it replaced: with
//     the register file full the compiler spilled around every tile and, because scratch traffic shares the vmcnt counter, waited
//     vmcnt(0) at every k-step - one or two 1 KB requests in flight per wave, 65 % of the wave cycles in s_waitcnt, 2.1 TB/s
//     (PMC: gpurun_out/r03g_pmc_dec_rows2.txt, profiles/r03_kernels.md).  Bandwidth = bytes in flight / latency: one fragment
//     leaves room for an 8-deep request queue and no spills.)
//   * the k-loop has NO barrier at all: the 8 waves of a block drift freely, one wave's loads and VALU transform overlap the
//     other wave's MFMAs on the same SIMD (2 waves per SIMD);
//   * BatchNorm statistics are accumulated per lane across ALL row tiles a wave processes (fp32 inside a tile, fp64 across
//     tiles) and leave the block once, at the end (one cross-wave reduction per block instead of one per 128 rows).
// A wave tile is 1 sample x 32 consecutive template vertices (the 8 waves of a block = 8 samples over the SAME vertices, so
// the layer-1 grid factor rows are shared through L1 / L2).
#pragma once

constexpr int R2_NT = 4;               // 32-column MFMA tiles per wave
constexpr int R2_COLS = 32 * R2_NT;    // columns per block (+ up to R2_SIDE side columns in the last group)
constexpr int R2_SIDE = 3;
constexpr int R2_THREADS = 512;
constexpr int R2_WAVES = 8;

inline int kpad16(int K) { return (K + 15) / 16 * 16; }  // k extent of the weight image the rows2 kernels use (one MFMA k-step)

struct R2Geo {
  int R, N, B;
  int nvt, nbg; // vertex tiles (32 vertices), sample groups (8 samples)
  int ngroups;  // column groups of R2_COLS
  int slots;    // persistent blocks per column group = spb * nbg
  int spb;      // slots per sample group
  int chunk;    // vertex tiles per slot
  // row i (0..31) of the fragment of wave `wave` in block tile (bg, vt): sample bg*8 + wave, vertex vt*32 + i
  __device__ __forceinline__ void row(int bg, int vt, int wave, int i, int& b, int& n, long& r, bool& ok) const {
    b = bg * 8 + wave;
    n = vt * 32 + i;
    ok = b < B && n < N;
    if (!ok) { b = 0; n = 0; }
    r = (long)b * N + n;
  }
};

// request-queue depth per operand generator (k-steps in flight per wave): 16 registers per step for the fp32 layer-1 factors,
// 4 / 8 for the bf16-stored activations
template <class AOp> struct R2Depth { static constexpr int value = 8; };
template <> struct R2Depth<BGridFeat> { static constexpr int value = 4; };
template <> struct R2Depth<BGradH> { static constexpr int value = 6; };

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
  __device__ __forceinline__ void tile(State& s, const f32x16 (&acc)[R2_NT], const float (&side)[R2_SIDE], const R2Ctx& c,
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
    {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int i0 = acc_row(2 * p, c.lane);
        int b, n; long r0; bool ok0;
        geo.row(c.bg, c.vt, c.wave, i0, b, n, r0, ok0);
        const bool ok1 = ok0 && n + 1 < geo.N;  // row i0 + 1: the next vertex of the same sample
        bfraw* dst = C + (size_t)(odd ? r0 + 1 : r0) * ldc;
        const bool okw = odd ? ok1 : ok0;
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) {
          const int cl = c.c0 + j * 32 + li;
          const bool cok = cl < Nc;
          const unsigned pk = pack_bf16(cok ? acc[j][2 * p] + bv[j] : 0.f, cok ? acc[j][2 * p + 1] + bv[j] : 0.f);
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
        geo.row(c.bg, c.vt, c.wave, li, b, n, r, ok);
        if (ok && h == 0) {
#pragma unroll
          for (int t = 0; t < R2_SIDE; ++t) {
            if (t < c.nside) {
              const int col = c.c0 + R2_COLS + t;
              const unsigned pk = pack_bf16(side[t] + (bias ? bias[col] : 0.f), 0.f);
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
  __device__ __forceinline__ void tile(State& q, const f32x16 (&acc)[R2_NT], const float (&side)[R2_SIDE], const R2Ctx& c,
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
    {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int i0 = acc_row(2 * p, c.lane);
        int b, n; long r0; bool ok0;
        geo.row(c.bg, c.vt, c.wave, i0, b, n, r0, ok0);
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
          const float g0 = (cok && ok0 && __fmaf_rn(cs[j], h0, ct[j]) > 0.f) ? acc[j][2 * p] : 0.f;
          const float g1 = (cok && ok1 && __fmaf_rn(cs[j], h1, ct[j]) > 0.f) ? acc[j][2 * p + 1] : 0.f;
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
        geo.row(c.bg, c.vt, c.wave, li, b, n, r, ok);
        if (ok && h == 0) {
#pragma unroll
          for (int u = 0; u < R2_SIDE; ++u) {
            if (u < c.nside) {
              const int col = c.c0 + R2_COLS + u;
              const size_t o = (size_t)r * ldc + col;
              const float hv = __uint_as_float((unsigned)H[o] << 16);
              const float g = __fmaf_rn(s[col], hv, t[col]) > 0.f ? side[u] : 0.f;
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
    typename AOp::Row row;
    bool ok;
    {
      int b, n; long r;
      geo.row(bg, vt, wave, li, b, n, r, ok);
      row = aop.row(r, b, n, ok);
    }
    f32x16 acc[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float side[R2_SIDE];
#pragma unroll
    for (int t = 0; t < R2_SIDE; ++t) side[t] = 0.f;

    // One k-step = one 16-deep MFMA per column tile.  Raw operand chunks are requested DQ k-steps ahead into a register queue
    // with compile-time slots (the loop is unrolled by DQ); requests past the last step are clamped re-reads that are never
    // consumed, so the loop body is branch-free and the compiler's counted vmcnt waits leave the younger requests in flight.
    constexpr int DQ = R2Depth<AOp>::value;
    typename AOp::Raw q[DQ];
#pragma unroll
    for (int u = 0; u < DQ; ++u) aop.load(q[u], row, u * 16 + h * 8);
    auto step = [&](typename AOp::Raw& qs, int s) {
      u32x4 a0;
      {
        float o0[8];
        aop.fin(row, kcs, Kp, s * 16 + h * 8, qs, o0);
        a0 = pack8(o0);
      }
      aop.load(qs, row, (s + DQ) * 16 + h * 8);
      if (!ok) a0 = u32x4{0u, 0u, 0u, 0u};
      const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + s * 16);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb, acc[j], 0, 0, 0);
      }
      if (nside) {  // leftover columns of the last group on the VALU, from the SAME rounded operands the MFMAs consume
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t) {
          if (t < nside) {
            const u32x4 wv = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + s * 16 + h * 8);
            float s0 = side[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned ww = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
              const unsigned w0 = e == 0 ? a0.x : (e == 1 ? a0.y : (e == 2 ? a0.z : a0.w));
              s0 = __fmaf_rn(bf_hi(w0), bf_hi(ww), __fmaf_rn(bf_lo(w0), bf_lo(ww), s0));
            }
            side[t] = s0;
          }
        }
      }
    };
    int s = 0;
    for (; s + DQ <= nks; s += DQ) {
#pragma unroll
      for (int u = 0; u < DQ; ++u) step(q[u], s + u);
    }
#pragma unroll
    for (int u = 0; u < DQ; ++u)
      if (s + u < nks) step(q[u], s + u);
    if (nside) {  // the two lane halves covered different k: combine
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) side[t] += __shfl_xor(side[t], 32, 64);
    }
    epi.tile(est, acc, side, ctx, geo);
  }
  epi.flush(est, ctx, geo, smem);
}
