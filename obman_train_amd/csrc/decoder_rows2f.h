// K6, exact-fp32 flavour, second generation of the "rows" GEMMs (h2, h3, gy2, gy1): included by decoder.hip inside namespace
// dec after decoder_tn2.h.  Round 4: the fp32 flavour is the only own MFMA kernel inside the driver-timed
// configuration and ran at 24 .. 47 % of the fp32 matrix rate per GEMM.
//
// What the first generation (gemm_rows_kernel) pays per 128 x 64 x 32 k-tile: every operand element is generated, written to
// LDS, and read back as 4-byte fragments (3 ds_read_b32 per two MFMAs), the 8 KB weight tile is re-staged for every row block
// (321 times per column block at 64 x 642 points), one block-wide barrier per k-tile, and 1 284 blocks on 768 block slots run
// two rounds for 1.67 rounds of work.
//
// Here (the shape the bf16 flavour got in round 3, decoder_rows2.h, re-derived for v_mfma_f32_32x32x2_f32):
//   * a block (8 waves) owns 64 (+ up to 3 "side") output columns for its whole life: their fp32 weights sit in LDS ONCE
//     ([col][k], pitch Kp + 4 floats: conflict-free 16-byte fragment reads), 257 = 4 x 64 + 1 and 515 = 8 x 64 + 3;
//   * a wave owns 32 consecutive rows x 64 columns (two accumulator tiles) and GENERATES its A operand IN REGISTERS in MFMA
//     layout: lane (row = lane & 31, half = lane >> 5) loads the 4 consecutive k  8 s + 4 half .. + 3  of its row with one
//     16-byte load per source array and k-step, applies the fused BatchNorm / ReLU / BatchNorm-backward transform, and feeds
//     FOUR MFMAs per column tile from it - MFMA step t multiplies k = 8 s + t of the lower lane half with k = 8 s + 4 + t of
//     the upper one (any pairing of k indices is a valid 2-deep step as long as A and B agree), so neither operand is ever
//     shuffled; the matching weight fragment is one ds_read_b128 per column tile and k-step;
//   * no barrier in the k loop, no LDS traffic for A; rows are linear (r = b N + n): 64 x 642 = 1 284 x 32 exactly;
//   * BatchNorm moments / backward sums accumulate per lane across all tiles of a wave (fp32 inside a tile, fp64 across) and
//     leave the block once.
// Per k-step (8 k) a wave issues 1 - 2 sixteen-byte loads, 2 LDS reads, 4 - 40 VALU operations and 8 MFMAs (512 matrix-pipe
// cycles): the loop is matrix-pipe bound, which the bf16 flavour (32 cycles per 16 k) never was.
#pragma once

// Measured on MI355X (tools/ubench/mfma_f32_loop.hip, profiles/r04_mfma_f32_loop.txt): 8 MFMAs per k-step alone run at 97.7 % of
// the 157.3 TF peak, with the weight-fragment LDS reads 95.7 %, with the transform 90 % - and with two 16-byte global loads in which
// every lane reads its OWN row 48 %: a wave-load that touches 32 .. 64 cache lines occupies the CU's texture path for ~2 cycles
// per line, and waves queue at its issue with their MFMAs behind it.  Hence:
//   * NT (32-column tiles per wave) = 4 wherever the weight slice of 128 columns fits in LDS (K <= 264): one operand load per 16
//     MFMAs instead of per 8;
//   * the layer-2 GEMM (K = 515: 64 columns per block) tiles its rows as 8 samples x 4 vertices per wave (row mode 2): the
//     load of the grid factor touches 4 rows and the load of the feature factor 8, instead of 32 + 1.
constexpr int F2_SIDE = 3;
constexpr int F2_THREADS = 512;
constexpr int F2_WAVES = 8;
constexpr int F2_DQ = 4;                 // k-steps requested ahead per wave

inline int kpad8(int K) { return (K + 7) / 8 * 8; }

struct F2Geo {
  int R, N, B;
  int mode;     // 0: a tile = 32 consecutive rows r = b N + n;  2: a tile = 8 samples x 4 vertices (tile row i: sample i >> 2, vertex i & 3)
  int nvt;      // mode 2: vertex groups of 4 per sample group
  int tiles;    // wave tiles
  int ngroups;  // column groups of 32 NT
  int slots;    // blocks per column group (all groups but the last)
  int slots_last;  // blocks of the LAST column group: it also carries the side columns on the VALU and gets fewer tiles per block
  int wside;    // side-column rows of the weight slice actually present (0 .. F2_SIDE)
  int wpg;      // > 0 (row mode 2 only): every wave stays inside ONE sample group - wpg waves per group (wpg_last in the last
  int wpg_last; //   column group, which has more blocks), nvt / wpg consecutive vertex tiles each - so that sums over a sample's
                //   vertices can stay in registers across tiles (F2EpiL1PQ)
  __host__ __device__ int prow() const { return wpg > wpg_last ? wpg : wpg_last; }  // P partial rows per sample
  __host__ __device__ int mrows() const { return ngroups > 1 && slots > slots_last ? slots : slots_last; }  // rows of the per-block partials
  // row i (0 .. 31) of tile t
  __device__ __forceinline__ void rowof(int t, int i, long& r, int& b, int& n, bool& ok) const {
    if (mode == 0) {
      r = (long)t * 32 + i;
      ok = r < R;
      if (!ok) r = (long)R - 1;
      b = (int)(r / N);
      n = (int)(r - (long)b * N);
    } else {
      const int bg = t / nvt, vt = t - bg * nvt;
      b = bg * 8 + (i >> 2);
      n = vt * 4 + (i & 3);
      ok = b < B && n < N;
      if (!ok) { b = B - 1; n = N - 1; }
      r = (long)b * N + n;
    }
  }
};

// ------------------------------------------------------------------------------------------------ A operands
// row(): per tile and lane; load(): issues the loads of k-step s (nothing consumes a loaded value); fin(): 4 operand values
// a[t] = A[row, 8 s + 4 half + t].  Per-channel constants live in LDS (kcs), staged once per block.  Columns >= K: every source
// array holds zeros (or finite values under a relu with zero constants) in its pitch padding and the weight slice is zero there.
struct F2GridFeatPre {  // a1 = relu(Gy[n] + Fy[b]): the pre-scaled layer-1 factors (l1_fill_kernel / prep_kernel); row N of Gy = -3e38
  const float *Gy, *Fy;
  int ld, N;
  struct Row { const float *g, *f; };
  struct Raw { float4 g, f; };
  static int lds_floats(int) { return 0; }
  __device__ __forceinline__ void stage(float*, int, int) const {}
  __device__ __forceinline__ Row row(long, int b, int n, bool ok, int half) const {
    return Row{Gy + (size_t)(ok ? n : N) * ld + 4 * half, Fy + (size_t)b * ld + 4 * half};
  }
  __device__ __forceinline__ void load(Raw& q, const Row& w, int s) const {
    q.g = *reinterpret_cast<const float4*>(w.g + 8 * s);
    q.f = *reinterpret_cast<const float4*>(w.f + 8 * s);
  }
  __device__ __forceinline__ void fin(const Row&, const float*, int, int, const Raw& q, float (&a)[4]) const {
    a[0] = fmaxf(q.g.x + q.f.x, 0.f); a[1] = fmaxf(q.g.y + q.f.y, 0.f);
    a[2] = fmaxf(q.g.z + q.f.z, 0.f); a[3] = fmaxf(q.g.w + q.f.w, 0.f);
  }
};
struct F2BnRelu {  // a = relu(s[k] * H[r,k] + t[k])
  const float *H, *s, *t;
  int ld, K;
  struct Row { const float* h; bool ok; };
  struct Raw { float4 h; };
  static int lds_floats(int Kp) { return 2 * Kp; }
  __device__ __forceinline__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < 2 * Kp; i += F2_THREADS) {
      const int j = i / Kp, k = i - j * Kp;
      kcs[i] = k < K ? (j == 0 ? s : t)[k] : 0.f;
    }
  }
  __device__ __forceinline__ Row row(long r, int, int, bool ok, int half) const { return Row{H + (size_t)r * ld + 4 * half, ok}; }
  __device__ __forceinline__ void load(Raw& q, const Row& w, int s_) const { q.h = *reinterpret_cast<const float4*>(w.h + 8 * s_); }
  __device__ __forceinline__ void fin(const Row& w, const float* kcs, int Kp, int k, const Raw& q, float (&a)[4]) const {
    const float4 sc = *reinterpret_cast<const float4*>(kcs + k), tc = *reinterpret_cast<const float4*>(kcs + Kp + k);
    a[0] = fmaxf(__fmaf_rn(sc.x, q.h.x, tc.x), 0.f); a[1] = fmaxf(__fmaf_rn(sc.y, q.h.y, tc.y), 0.f);
    a[2] = fmaxf(__fmaf_rn(sc.z, q.h.z, tc.z), 0.f); a[3] = fmaxf(__fmaf_rn(sc.w, q.h.w, tc.w), 0.f);
    if (!w.ok) { a[0] = 0.f; a[1] = 0.f; a[2] = 0.f; a[3] = 0.f; }
  }
};
struct F2GradH {  // gh = ka * gy + kb * h + kc (BatchNorm backward folded per channel, see AGradH)
  const float *GY, *H, *ka, *kb, *kc_;
  int ld, K;
  struct Row { const float *gy, *h; bool ok; };
  struct Raw { float4 gy, h; };
  static int lds_floats(int Kp) { return 3 * Kp; }
  __device__ __forceinline__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < 3 * Kp; i += F2_THREADS) {
      const int j = i / Kp, k = i - j * Kp;
      kcs[i] = k < K ? (j == 0 ? ka : (j == 1 ? kb : kc_))[k] : 0.f;
    }
  }
  __device__ __forceinline__ Row row(long r, int, int, bool ok, int half) const {
    return Row{GY + (size_t)r * ld + 4 * half, H + (size_t)r * ld + 4 * half, ok};
  }
  __device__ __forceinline__ void load(Raw& q, const Row& w, int s) const {
    q.gy = *reinterpret_cast<const float4*>(w.gy + 8 * s);
    q.h = *reinterpret_cast<const float4*>(w.h + 8 * s);
  }
  __device__ __forceinline__ void fin(const Row& w, const float* kcs, int Kp, int k, const Raw& q, float (&a)[4]) const {
    const float4 ca = *reinterpret_cast<const float4*>(kcs + k), cb = *reinterpret_cast<const float4*>(kcs + Kp + k);
    const float4 cc = *reinterpret_cast<const float4*>(kcs + 2 * Kp + k);
    a[0] = __fmaf_rn(ca.x, q.gy.x, __fmaf_rn(cb.x, q.h.x, cc.x)); a[1] = __fmaf_rn(ca.y, q.gy.y, __fmaf_rn(cb.y, q.h.y, cc.y));
    a[2] = __fmaf_rn(ca.z, q.gy.z, __fmaf_rn(cb.z, q.h.z, cc.z)); a[3] = __fmaf_rn(ca.w, q.gy.w, __fmaf_rn(cb.w, q.h.w, cc.w));
    if (!w.ok) { a[0] = 0.f; a[1] = 0.f; a[2] = 0.f; a[3] = 0.f; }
  }
};
struct F2GradH3 {  // gh3 = (y3 > 0 ? f * g . (ka * W4) : 0) + kb * h + kc, gy3 regenerated from the 3-channel output gradient
  const float *G, *W4, *H, *s, *t, *ka, *kb, *kc_;
  float f;
  int ld, K;
  struct Row { const float* h; float g0, g1, g2; bool ok; };
  struct Raw { float4 h; };
  static int lds_floats(int Kp) { return 8 * Kp; }  // [k][8]: s, t, kb, kc | ka*w0, ka*w1, ka*w2, 0
  __device__ __forceinline__ void stage(float* kcs, int Kp, int tid) const {
    for (int i = tid; i < 8 * Kp; i += F2_THREADS) {
      const int k = i >> 3, j = i & 7;
      float v = 0.f;
      if (k < K) {
        switch (j) {
          case 0: v = s[k]; break;
          case 1: v = t[k]; break;
          case 2: v = kb[k]; break;
          case 3: v = kc_[k]; break;
          case 4: case 5: case 6: v = ka[k] * W4[(j - 4) * K + k]; break;
          default: break;
        }
      }
      kcs[i] = v;
    }
  }
  __device__ __forceinline__ Row row(long r, int, int, bool ok, int half) const {
    return Row{H + (size_t)r * ld + 4 * half, f * G[r * 3], f * G[r * 3 + 1], f * G[r * 3 + 2], ok};
  }
  __device__ __forceinline__ void load(Raw& q, const Row& w, int s_) const { q.h = *reinterpret_cast<const float4*>(w.h + 8 * s_); }
  __device__ __forceinline__ void fin(const Row& w, const float* kcs, int, int k, const Raw& q, float (&a)[4]) const {
    const float hv[4] = {q.h.x, q.h.y, q.h.z, q.h.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 c0 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8 + 4);
      const float d = __fmaf_rn(w.g2, c1.z, __fmaf_rn(w.g1, c1.y, w.g0 * c1.x));
      const float gy = __fmaf_rn(c0.x, hv[j], c0.y) > 0.f ? d : 0.f;
      a[j] = w.ok ? gy + __fmaf_rn(c0.z, hv[j], c0.w) : 0.f;
    }
  }
};

// ------------------------------------------------------------------------------------------------ epilogues
struct F2Ctx { int lane, wave, c0, nside, last_group, slot, my_slots, mrows, t; long r0; int b0, n0; int pq_wi, pq_wpg; };  // r0 (mode 0) / b0, n0 (mode 2): the tile's first row

// fp64 sums held by (two lane halves) x (eight waves) -> dst[(slot * Nc + col) * 2 + {0,1}], fixed order.  `smem` is the (dead)
// weight slice; called by every thread of the block after its last tile.
template <int NT>
__device__ __forceinline__ void f2_flush_cols(double (&d1)[NT], double (&d2)[NT], const float (&f1)[F2_SIDE], const float (&f2)[F2_SIDE],
                                              const F2Ctx& c, int Nc, double* __restrict__ dst, char* smem) {
  const int li = c.lane & 31;
#pragma unroll
  for (int j = 0; j < NT; ++j) { d1[j] += __shfl_xor(d1[j], 32, 64); d2[j] += __shfl_xor(d2[j], 32, 64); }
  double e1[F2_SIDE], e2[F2_SIDE];
#pragma unroll
  for (int t = 0; t < F2_SIDE; ++t) {  // side columns: one fp32 partial per lane (a lane of half 0 owns one row per tile)
    e1[t] = (double)f1[t];
    e2[t] = (double)f2[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e1[t] += __shfl_xor(e1[t], off, 64); e2[t] += __shfl_xor(e2[t], off, 64); }
  }
  double* red = reinterpret_cast<double*>(smem);  // [7 waves][NT + 1][32][2]
  __syncthreads();                                // every wave is done with the weight slice
  if (c.wave > 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < NT; ++j) { double* q = red + ((((c.wave - 1) * (NT + 1) + j) * 32) + li) * 2; q[0] = d1[j]; q[1] = d2[j]; }
    if (li < F2_SIDE) {
      double* q = red + ((((c.wave - 1) * (NT + 1) + NT) * 32) + li) * 2;
      q[0] = li == 0 ? e1[0] : (li == 1 ? e1[1] : e1[2]);
      q[1] = li == 0 ? e2[0] : (li == 1 ? e2[1] : e2[2]);
    }
  }
  __syncthreads();
  if (c.wave == 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      if (col < Nc) {
        double a = d1[j], b = d2[j];
#pragma unroll
        for (int w = 0; w < F2_WAVES - 1; ++w) { const double* q = red + (((w * (NT + 1) + j) * 32) + li) * 2; a += q[0]; b += q[1]; }
        double* o = dst + ((size_t)c.slot * Nc + col) * 2;
        o[0] = a;
        o[1] = b;
        if (c.slot == 0)  // column groups with fewer blocks than the widest one: their missing partial rows read as zeros
          for (int z = c.my_slots; z < c.mrows; ++z) { double* oz = dst + ((size_t)z * Nc + col) * 2; oz[0] = 0.0; oz[1] = 0.0; }
      }
    }
    if (li < c.nside) {
      double a = li == 0 ? e1[0] : (li == 1 ? e1[1] : e1[2]), b = li == 0 ? e2[0] : (li == 1 ? e2[1] : e2[2]);
#pragma unroll
      for (int w = 0; w < F2_WAVES - 1; ++w) { const double* q = red + (((w * (NT + 1) + NT) * 32) + li) * 2; a += q[0]; b += q[1]; }
      double* o = dst + ((size_t)c.slot * Nc + c.c0 + 32 * NT + li) * 2;
      o[0] = a;
      o[1] = b;
      if (c.slot == 0)
        for (int z = c.my_slots; z < c.mrows; ++z) { double* oz = dst + ((size_t)z * Nc + c.c0 + 32 * NT + li) * 2; oz[0] = 0.0; oz[1] = 0.0; }
    }
  }
}
constexpr size_t f2_flush_bytes(int NT) { return (size_t)(F2_WAVES - 1) * (NT + 1) * 32 * 2 * sizeof(double); }

// row i of the current tile from the per-tile bases (no division): r always; b, n in mode 2 only
__device__ __forceinline__ void f2_row(const F2Geo& geo, const F2Ctx& c, int i, long& r, bool& ok) {
  if (geo.mode == 0) {
    r = c.r0 + i;
    ok = r < geo.R;
    if (!ok) r = (long)geo.R - 1;
  } else {
    const int b = c.b0 + (i >> 2), n = c.n0 + (i & 3);
    ok = b < geo.B && n < geo.N;
    r = ok ? (long)b * geo.N + n : 0;
  }
}
// zeros behind the last real column up to the pitch (the next GEMM reads whole 16-byte chunks of a row): lane 32 + i does row i
__device__ __forceinline__ void f2_zero_pitch(float* C, int ldc, int Nc, const F2Geo& geo, const F2Ctx& c) {
  if (c.lane >= 32) {
    long r; bool ok;
    f2_row(geo, c, c.lane - 32, r, ok);
    if (ok)
      for (int col = Nc; col < ldc; ++col) C[(size_t)r * ldc + col] = 0.f;
  }
}
// Accumulator register q = 4 m + v of a lane holds tile row i = 8 m + 4 h + v (h = lane >> 5).  rows(): the element offset of
// row (m, v = 0) for m = 0 .. 3 - rows v = 1 .. 3 follow at + v * ld in BOTH row modes (mode 0: consecutive rows; mode 2: the
// consecutive vertices of one sample) - and whether the whole tile lies inside the problem (wave-uniform: the common case takes
// straight-line stores with no per-element predicate; the ISA of the predicated form was ~20 scalar / vector instructions and
// two branches per store).
struct F2TileRows { size_t base[4]; bool full; };
__device__ __forceinline__ F2TileRows f2_tile_rows(const F2Geo& geo, const F2Ctx& c, int ld) {
  F2TileRows o;
  const int h = c.lane >> 5;
  if (geo.mode == 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) o.base[m] = (size_t)(c.r0 + 8 * m + 4 * h) * ld;
    o.full = c.r0 + 32 <= geo.R;
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m) o.base[m] = ((size_t)(c.b0 + 2 * m + h) * geo.N + c.n0) * ld;
    o.full = c.b0 + 8 <= geo.B && c.n0 + 4 <= geo.N;
  }
  return o;
}
// row of accumulator register q of this lane (register q <-> tile row acc_row(q, lane)); linear-row kernels (mode 0) only
__device__ __forceinline__ long f2_lin_row(const F2Ctx& c, int q) { return c.r0 + acc_row(q, c.lane); }

template <int NT>
struct F2EpiStore {  // C[r,n] = acc + bias[n]; fp64 column moments (sum, sum of squares) per block
  float* C;
  const float* bias;
  double* moments;  // [slots][Nc][2] or null
  int ldc, Nc;
  static constexpr int LDS_FLOATS = 0;
  struct State { double d1[NT], d2[NT]; float e1[F2_SIDE], e2[F2_SIDE]; float bv[NT]; };
  struct Pre {};
  __device__ __forceinline__ void prefetch(Pre&, const State&, const F2Ctx&, const F2Geo&) const {}
  __device__ __forceinline__ void init(State& s, const F2Ctx& c) const {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s.d1[j] = 0.0; s.d2[j] = 0.0;
      const int col = c.c0 + j * 32 + (c.lane & 31);
      s.bv[j] = (bias && col < Nc) ? bias[col] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < F2_SIDE; ++t) { s.e1[t] = 0.f; s.e2[t] = 0.f; }
  }
  __device__ __forceinline__ void tile(State& s, const Pre&, const f32x16 (&acc)[NT], const float (&side)[F2_SIDE], const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31;
    float s1[NT], s2[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const F2TileRows tr = f2_tile_rows(geo, c, ldc);
    if (tr.full && c.c0 + 32 * NT <= Nc) {  // whole tile inside the problem: no predicates
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float* dst = C + tr.base[m] + c.c0 + li;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float val = acc[j][4 * m + v] + s.bv[j];
            dst[(size_t)v * ldc + j * 32] = val;
            s1[j] += val;
            s2[j] = __fmaf_rn(val, val, s2[j]);
          }
        }
      }
    } else
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // row-major over the registers: one row index per 32 x NT stored values
      long r; bool ok;
      f2_row(geo, c, acc_row(q, c.lane), r, ok);
      float* dst = C + (size_t)r * ldc + c.c0 + li;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = acc[j][q] + s.bv[j];
        if (ok && c.c0 + j * 32 + li < Nc) {
          dst[j * 32] = v;
          s1[j] += v;
          s2[j] = __fmaf_rn(v, v, s2[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) { s.d1[j] += (double)s1[j]; s.d2[j] += (double)s2[j]; }
    if (c.last_group) {
      long r; bool ok;
      f2_row(geo, c, li, r, ok);
      if (c.lane < 32 && ok) {
#pragma unroll
        for (int t = 0; t < F2_SIDE; ++t) {
          if (t < c.nside) {
            const int col = c.c0 + 32 * NT + t;
            const float v = side[t] + (bias ? bias[col] : 0.f);
            C[(size_t)r * ldc + col] = v;
            s.e1[t] += v;
            s.e2[t] = __fmaf_rn(v, v, s.e2[t]);
          }
        }
      }
      f2_zero_pitch(C, ldc, Nc, geo, c);
    }
  }
  __device__ __forceinline__ void flush(State& s, const F2Ctx& c, const F2Geo&, char* smem) const {
    if (moments) f2_flush_cols<NT>(s.d1, s.d2, s.e1, s.e2, c, Nc, moments, smem);
  }
};

template <int NT>
struct F2EpiMask {  // C = acc * (y > 0), y = s*H+t; column sums S1 = sum C, S2 = sum C * xhat, xhat = (H - mean) * rstd
  float* C;
  const float* H;  // same pitch as C
  double* sums;    // [slots][Nc][2]
  const float *s, *t, *mean, *rstd;
  int ldc, Nc;
  static constexpr int LDS_FLOATS = 0;
  static constexpr int NPRE = NT == 4 ? 1 : (NT < 2 ? NT : 2);  // column tiles whose H values are requested BEFORE the tile's k loop (K = 128: 16 steps)
  struct State { double d1[NT], d2[NT]; float e1[F2_SIDE], e2[F2_SIDE]; float cs[NT], ct[NT], cm[NT], cr[NT]; };
  struct Pre { float h[NPRE][16]; };
  __device__ __forceinline__ void init(State& q, const F2Ctx& c) const {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      q.d1[j] = 0.0; q.d2[j] = 0.0;
      const int col = c.c0 + j * 32 + (c.lane & 31);
      const bool cok = col < Nc;
      q.cs[j] = cok ? s[col] : 0.f; q.ct[j] = cok ? t[col] : 0.f; q.cm[j] = cok ? mean[col] : 0.f; q.cr[j] = cok ? rstd[col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < F2_SIDE; ++u) { q.e1[u] = 0.f; q.e2[u] = 0.f; }
  }
  __device__ __forceinline__ void prefetch(Pre& p, const State&, const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31;
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int col = c.c0 + j * 32 + li;
      const bool cok = col < Nc;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long r = f2_lin_row(c, q);
        p.h[j][q] = (r < geo.R && cok) ? H[(size_t)r * ldc + col] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void tile(State& q, const Pre& p, const f32x16 (&acc)[NT], const float (&side)[F2_SIDE], const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31;
    const F2TileRows tr = f2_tile_rows(geo, c, ldc);
    if (tr.full && c.c0 + 32 * NT <= Nc) {  // whole tile inside the problem: straight-line loads and stores
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float hv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (j < NPRE) hv[r] = p.h[j < NPRE ? j : 0][r];
          else hv[r] = H[tr.base[r >> 2] + (size_t)(r & 3) * ldc + c.c0 + j * 32 + li];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = __fmaf_rn(q.cs[j], hv[r], q.ct[j]) > 0.f ? acc[j][r] : 0.f;
          C[tr.base[r >> 2] + (size_t)(r & 3) * ldc + c.c0 + j * 32 + li] = v;
          s1 += v;
          s2 = __fmaf_rn(v, (hv[r] - q.cm[j]) * q.cr[j], s2);
        }
        q.d1[j] += (double)s1;
        q.d2[j] += (double)s2;
      }
    } else
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      const bool cok = col < Nc;
      float hv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long rr = f2_lin_row(c, r);
        if (j < NPRE) hv[r] = p.h[j < NPRE ? j : 0][r];
        else hv[r] = (rr < geo.R && cok) ? H[(size_t)rr * ldc + col] : 0.f;
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long rr = f2_lin_row(c, r);
        if (rr < geo.R && cok) {
          const float v = __fmaf_rn(q.cs[j], hv[r], q.ct[j]) > 0.f ? acc[j][r] : 0.f;
          C[(size_t)rr * ldc + col] = v;
          s1 += v;
          s2 = __fmaf_rn(v, (hv[r] - q.cm[j]) * q.cr[j], s2);
        }
      }
      q.d1[j] += (double)s1;
      q.d2[j] += (double)s2;
    }
    if (c.last_group) {
      long r; bool ok;
      f2_row(geo, c, li, r, ok);
      if (c.lane < 32 && ok) {
#pragma unroll
        for (int u = 0; u < F2_SIDE; ++u) {
          if (u < c.nside) {
            const int col = c.c0 + 32 * NT + u;
            const size_t o = (size_t)r * ldc + col;
            const float hv = H[o];
            const float v = __fmaf_rn(s[col], hv, t[col]) > 0.f ? side[u] : 0.f;
            C[o] = v;
            q.e1[u] += v;
            q.e2[u] = __fmaf_rn(v, (hv - mean[col]) * rstd[col], q.e2[u]);
          }
        }
      }
      f2_zero_pitch(C, ldc, Nc, geo, c);
    }
  }
  __device__ __forceinline__ void flush(State& q, const F2Ctx& c, const F2Geo&, char* smem) const { f2_flush_cols<NT>(q.d1, q.d2, q.e1, q.e2, c, Nc, sums, smem); }
};

template <int NT>
struct F2EpiL1 {  // gy1 = acc * (y1 > 0), y1 > 0 <=> Gy[n] + Fy[b] > 0 (the forward's own relu argument); stored for l1_reduce_kernel
  // Row mode 2 only (a tile = 8 samples x 4 vertices): a lane's 16 accumulator rows are 4 vertices x 4 samples, so the mask needs
  // 4 + 4 factor values per column tile instead of 16 + 16 - measured with linear rows the 128 loads + 64 stores of this
  // epilogue were 55 % of the kernel (s_memtime, profiles/r04_kernels.md).  The 8 values are requested before the tile's k loop.
  float* C;
  const float *Gy, *Fy;
  int ldc, Nc;
  static constexpr int LDS_FLOATS = 0;
  struct State {};
  struct Pre { float g[NT][4], f[NT][4]; };
  __device__ __forceinline__ void init(State&, const F2Ctx&) const {}
  __device__ __forceinline__ void prefetch(Pre& p, const State&, const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      const bool cok = col < Nc;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = c.n0 + v, b = c.b0 + 2 * v + h;
        p.g[j][v] = (cok && n < geo.N) ? Gy[(size_t)n * ldc + col] : 0.f;
        p.f[j][v] = (cok && b < geo.B) ? Fy[(size_t)b * ldc + col] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void tile(State&, const Pre& p, const f32x16 (&acc)[NT], const float (&side)[F2_SIDE], const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const F2TileRows tr = f2_tile_rows(geo, c, ldc);
    if (tr.full && c.c0 + 32 * NT <= Nc) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float* dst = C + tr.base[m] + c.c0 + li;
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int j = 0; j < NT; ++j) dst[(size_t)v * ldc + j * 32] = p.g[j][v] + p.f[j][m] > 0.f ? acc[j][4 * m + v] : 0.f;
      }
    } else
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // accumulator register q = 4 m + v: sample b0 + 2 m + h, vertex n0 + v
      const int b = c.b0 + 2 * m + h;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = c.n0 + v;
        if (b < geo.B && n < geo.N) {
          float* dst = C + ((size_t)b * geo.N + n) * ldc + c.c0 + li;
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if (c.c0 + j * 32 + li < Nc) dst[j * 32] = p.g[j][v] + p.f[j][m] > 0.f ? acc[j][4 * m + v] : 0.f;
        }
      }
    }
    if (c.last_group) {
      const int b = c.b0 + (li >> 2), n = c.n0 + (li & 3);
      if (c.lane < 32 && b < geo.B && n < geo.N) {
        const size_t r = (size_t)b * geo.N + n;
#pragma unroll
        for (int u = 0; u < F2_SIDE; ++u) {
          if (u < c.nside) {
            const int col = c.c0 + 32 * NT + u;
            const float yv = Gy[(size_t)n * ldc + col] + Fy[(size_t)b * ldc + col];
            C[r * ldc + col] = yv > 0.f ? side[u] : 0.f;
          }
        }
      }
      f2_zero_pitch(C, ldc, Nc, geo, c);
    }
  }
  __device__ __forceinline__ void flush(State&, const F2Ctx&, const F2Geo&, char*) const {}
};

template <int NT>
struct F2EpiL1PQ {  // dA(gy1) WITHOUT gy1: layer 1 only needs P[b,c] = sum_n gy1 and Q[n,c] = sum_b gy1 (the fp32 counterpart of EpiL1B2)
  // gy1 = acc * (Gy[n] + Fy[b] > 0).  Row mode 2 with wpg > 0: a wave's tiles all belong to ONE sample group.  Accumulator
  // register q = 4 m + v of lane (col, h): sample b0 + 2 m + h, vertex n0 + v.
  //   P: the sum over a tile's 4 vertices is in-lane; it keeps accumulating IN REGISTERS over all tiles of the wave and leaves
  //      once:  Pp[wave index inside the sample group][b][c]  (l1_reduce2_kernel adds the wpg partials in order).
  //   Q: the sum over the tile's 8 samples is in-lane over m plus one exchange with the other lane half:
  //      Qp[sample group][n][c]  (each (group, vertex tile) is written by exactly one wave).
  // Against the materialised form: no [R, 515] fp32 store (87 MB at 64 x 642 points, 2.2 GB at 16 050) and no l1_reduce pass.
  float *Pp, *Qp;
  const float *Gy, *Fy;
  int ldc, Nc;
  static constexpr int LDS_FLOATS = 0;
  struct State { float p[4][NT]; float ps[F2_SIDE]; };
  struct Pre { float g[NT][4], f[NT][4]; };
  __device__ __forceinline__ void init(State& s, const F2Ctx&) const {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < NT; ++j) s.p[m][j] = 0.f;
#pragma unroll
    for (int u = 0; u < F2_SIDE; ++u) s.ps[u] = 0.f;
  }
  __device__ __forceinline__ void prefetch(Pre& p, const State&, const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      const bool cok = col < Nc;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int n = c.n0 + v, b = c.b0 + 2 * v + h;
        p.g[j][v] = (cok && n < geo.N) ? Gy[(size_t)n * ldc + col] : -3.0e38f;  // rows / columns outside the problem: mask false
        p.f[j][v] = (cok && b < geo.B) ? Fy[(size_t)b * ldc + col] : -3.0e38f;
      }
    }
  }
  __device__ __forceinline__ void tile(State& s, const Pre& p, const f32x16 (&acc)[NT], const float (&side)[F2_SIDE], const F2Ctx& c, const F2Geo& geo) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const int bg = c.b0 >> 3;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float gv = p.g[j][v] + p.f[j][m] > 0.f ? acc[j][4 * m + v] : 0.f;
          q[v] += gv;
          s.p[m][j] += gv;
        }
      }
      const int col = c.c0 + j * 32 + li;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float tot = q[v] + __shfl_xor(q[v], 32, 64);  // the other four samples of the tile
        if (h == 0 && col < Nc && c.n0 + v < geo.N) Qp[((size_t)bg * geo.N + c.n0 + v) * ldc + col] = tot;
      }
    }
    if (c.nside) {  // side columns: lane i < 32 holds tile row i = sample i >> 2, vertex i & 3
      const int b = c.b0 + (li >> 2), n = c.n0 + (li & 3);
      const bool live = c.lane < 32 && b < geo.B && n < geo.N;
#pragma unroll
      for (int u = 0; u < F2_SIDE; ++u) {
        if (u < c.nside) {
          const int col = c.c0 + 32 * NT + u;
          const float yv = live ? Gy[(size_t)n * ldc + col] + Fy[(size_t)b * ldc + col] : 0.f;
          const float gv = (live && yv > 0.f) ? side[u] : 0.f;
          float pv = gv + __shfl_xor(gv, 1, 64);  // over the sample's 4 vertices
          pv += __shfl_xor(pv, 2, 64);
          s.ps[u] += pv;                          // lanes with (li & 3) == 0 carry sample li >> 2
          float qs = gv + __shfl_xor(gv, 4, 64);  // over the tile's 8 samples
          qs += __shfl_xor(qs, 8, 64);
          qs += __shfl_xor(qs, 16, 64);
          if (c.lane < 4 && c.n0 + c.lane < geo.N) Qp[((size_t)bg * geo.N + c.n0 + c.lane) * ldc + col] = qs;
        }
      }
    }
  }
  __device__ __forceinline__ void flush(State& s, const F2Ctx& c, const F2Geo& geo, char*) const {
    // every wave that owns tiles writes the P partial of its 8 samples (waves without tiles own no partial slot: geo.wpg counts
    // only waves with work, see f2_geo_pq)
    const int li = c.lane & 31, h = c.lane >> 5;
    if (c.t < 0) return;  // no tile processed
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int b = c.b0 + 2 * m + h;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = c.c0 + j * 32 + li;
        if (b < geo.B && col < Nc) Pp[((size_t)c.pq_wi * geo.B + b) * ldc + col] = s.p[m][j];
      }
    }
    if (c.nside && c.lane < 32 && (li & 3) == 0 && c.b0 + (li >> 2) < geo.B) {
#pragma unroll
      for (int u = 0; u < F2_SIDE; ++u)
        if (u < c.nside) Pp[((size_t)c.pq_wi * geo.B + c.b0 + (li >> 2)) * ldc + c.c0 + 32 * NT + u] = s.ps[u];
    }
    if (c.pq_wi == 0) {  // a column group with fewer waves per sample group than the widest one: its missing partial rows read as zeros
      for (int z = c.pq_wpg; z < geo.prow(); ++z) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int b = c.b0 + 2 * m + h;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int col = c.c0 + j * 32 + li;
            if (b < geo.B && col < Nc) Pp[((size_t)z * geo.B + b) * ldc + col] = 0.f;
          }
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------ the kernel
// grid = ngroups * slots blocks (1-D, XCD-aware virtual ids: the column groups of one slot are neighbours on one XCD - they
// stream the same activation rows).  Dynamic LDS: weight slice [(32 NT + wside)][Kp + 4] fp32, then the generator's constants.
// W is the fp32 weight matrix as the module holds it: w_kn == 0: W[n][k] (row stride ldw), w_kn == 1: W[k][n].
#ifdef OBMAN_F2_TIMING  // measurement build (tools/archive/r04/dec_dbg.sh): s_memtime per wave around staging / k loops / epilogues
__device__ unsigned long long f2_dbg[2048 * 8];
#define F2_TICK() __builtin_readcyclecounter()
#else
#define F2_TICK() 0ull
#endif
template <class AOp, class Epi, int NT>
__global__ __launch_bounds__(F2_THREADS) void rows2f_kernel(AOp aop, const float* __restrict__ W, int ldw, int w_kn, int K, int Kp, int Nc, Epi epi,
                                                            F2Geo geo, int lds_aop_floats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int COLS = 32 * NT;
  const int KP2 = Kp + 4;
  float* Ws = reinterpret_cast<float*>(smem);
  float* kcs = Ws + (size_t)(COLS + geo.wside) * KP2;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  // block -> (column group, slot): the first (ngroups - 1) * slots virtual ids interleave the full groups, the rest are the last group's
  const int vid = xcd_virtual_id(blockIdx.x, gridDim.x), nmain = (geo.ngroups - 1) * geo.slots;
  const int cg = vid < nmain ? vid % (geo.ngroups - 1) : geo.ngroups - 1, slot = vid < nmain ? vid / (geo.ngroups - 1) : vid - nmain;
  const int c0 = cg * COLS;
  const int last_group = cg == geo.ngroups - 1;
  const int my_slots = last_group ? geo.slots_last : geo.slots;
  const int gcols = last_group ? Nc - c0 : COLS;  // the last group holds the remainder: up to COLS + F2_SIDE columns
  const int nside = gcols > COLS ? gcols - COLS : 0;
  {
    // 16 independent loads in flight per thread and round (a plain loop exposes one global latency per element)
    const int nrows = COLS + geo.wside, total = nrows * Kp;
    constexpr int SB = 16;
#pragma unroll 1
    for (int base = tid; base < total; base += SB * F2_THREADS) {
      float v[SB];
      int dst[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int i = base + u * F2_THREADS;
        int cc, k;
        if (w_kn) { k = i / nrows; cc = i - k * nrows; }  // W[k][n]: consecutive threads along n (coalesced reads, one-time strided LDS writes)
        else { cc = i / Kp; k = i - cc * Kp; }            // W[n][k]: consecutive threads along k
        const bool live = i < total && cc < gcols && k < K;
        const size_t src = w_kn ? (size_t)k * ldw + c0 + cc : (size_t)(c0 + cc) * ldw + k;
        v[u] = live ? W[src] : 0.f;
        dst[u] = i < total ? cc * KP2 + k : -1;
      }
#pragma unroll
      for (int u = 0; u < SB; ++u)
        if (dst[u] >= 0) Ws[dst[u]] = v[u];
    }
  }
  const unsigned long long T0 = F2_TICK();
  aop.stage(kcs, Kp, tid);
  __syncthreads();
  const unsigned long long T1 = F2_TICK();
  unsigned long long tk = 0, te = 0, ntile = 0;

  F2Ctx ctx{lane, wave, c0, nside, last_group, slot, my_slots, geo.mrows(), -1, 0, 0, 0, 0, 0};
  typename Epi::State est;
  epi.init(est, ctx);
  const int nks = Kp >> 3;
  const float* wlane = Ws + (size_t)li * KP2 + 4 * h;
  // Tiles are dealt to "virtual" wave ids that number waves 0..3 of every block first and waves 4..7 after them: when the tile
  // count is not a multiple of the wave count the extra tiles go to ONE wave of every SIMD (waves w and w + 4 share a SIMD)
  // instead of to both waves of the SIMDs of some CUs - the matrix pipe is per SIMD.
  const int stride = my_slots * F2_WAVES;
  const int vwave = wave < 4 ? slot * 4 + wave : my_slots * 4 + slot * 4 + (wave - 4);
  // wpg > 0: wave vwave owns vertex tiles [wi * chunk, (wi + 1) * chunk) of sample group vwave / wpg (contiguous, one group)
  const int my_wpg = last_group ? geo.wpg_last : geo.wpg;
  const int pq_bg = my_wpg > 0 ? vwave / my_wpg : 0, pq_wi = my_wpg > 0 ? vwave - pq_bg * my_wpg : 0;
  const int nbg = geo.nvt > 0 ? geo.tiles / geo.nvt : 0;
  int t_beg = vwave, t_end = geo.tiles, t_step = stride;
  if (my_wpg > 0) {  // nvt tiles over wpg waves as evenly as they go: the first nvt % wpg waves take one more (nvt / wpg >= 1)
    const int chunk = geo.nvt / my_wpg, rem = geo.nvt - chunk * my_wpg;
    const int v0 = pq_wi * chunk + (pq_wi < rem ? pq_wi : rem), v1 = v0 + chunk + (pq_wi < rem ? 1 : 0);
    t_beg = pq_bg * geo.nvt + v0;
    t_end = pq_bg < nbg && v0 < v1 ? pq_bg * geo.nvt + v1 : t_beg;
    t_step = 1;
  }
  ctx.pq_wi = pq_wi;
  ctx.pq_wpg = my_wpg;
#pragma unroll 1
  for (int t = t_beg; t < t_end; t += t_step) {
    ctx.t = t;
    ctx.r0 = (long)t * 32;
    if (geo.mode != 0) { const int bg = t / geo.nvt; ctx.b0 = bg * 8; ctx.n0 = (t - bg * geo.nvt) * 4; }
    typename AOp::Row row;
    {
      long r; int b, n; bool ok;
      geo.rowof(t, li, r, b, n, ok);
      row = aop.row(r, b, n, ok, h);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    float side[F2_SIDE];
#pragma unroll
    for (int u = 0; u < F2_SIDE; ++u) side[u] = 0.f;

    // Raw operand chunks are requested F2_DQ k-steps ahead into a register queue with compile-time slots (the loop is unrolled
    // by F2_DQ); requests past the last step read the following bytes of the arena (never consumed; WS_TAIL_FLOATS of padding
    // close the workspaces), so the loop body is branch-free and the counted vmcnt waits leave the younger requests in flight.
    const unsigned long long TA = F2_TICK();
    typename AOp::Raw q[F2_DQ];
#pragma unroll
    for (int u = 0; u < F2_DQ; ++u) aop.load(q[u], row, u);
    typename Epi::Pre pre;
    epi.prefetch(pre, est, ctx, geo);
    // Weight fragments are double-buffered in registers: the LDS reads of k-step s + 1 are issued before the transform and the
    // MFMAs of step s (two waves per SIMD run this loop in lockstep: when one waits for LDS, the other waits too).
    struct Frag { float4 w[NT]; };
    auto fetch = [&](Frag& fr, int ks) {
#pragma unroll
      for (int j = 0; j < NT; ++j) fr.w[j] = *reinterpret_cast<const float4*>(wlane + (size_t)j * 32 * KP2 + ks * 8);
    };
    auto step = [&](typename AOp::Raw& qs, int ks, Frag& cur, Frag& nxt) {
      fetch(nxt, ks + 1 < nks ? ks + 1 : ks);
      float4 sw[F2_SIDE];
      if (nside) {  // side-column weights of THIS step: requested here, consumed after the MFMAs (block-uniform branch)
#pragma unroll
        for (int u = 0; u < F2_SIDE; ++u)
          if (u < nside) sw[u] = *reinterpret_cast<const float4*>(Ws + (size_t)(COLS + u) * KP2 + ks * 8 + 4 * h);
      }
      float a[4];
      aop.fin(row, kcs, Kp, ks * 8 + 4 * h, qs, a);
      aop.load(qs, row, ks + F2_DQ);
      // t-major: consecutive MFMAs go round the accumulator tiles.  The scheduling barriers pin that order: left alone the
      // compiler chained the four MFMAs of one accumulator back to back (ISA of the first version), and a dependent 16-pass MFMA
      // cannot issue until its predecessor has drained - the h2 GEMM ran at 61 % of the matrix rate where the same instruction
      // mix in round-robin order reaches 88 - 90 % (tools/ubench/mfma_f32_loop.hip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], cur.w[j].x, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], cur.w[j].y, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], cur.w[j].z, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], cur.w[j].w, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (nside) {  // leftover columns of the last group on the VALU, from the same operand values
#pragma unroll
        for (int u = 0; u < F2_SIDE; ++u)
          if (u < nside) side[u] = __fmaf_rn(a[3], sw[u].w, __fmaf_rn(a[2], sw[u].z, __fmaf_rn(a[1], sw[u].y, __fmaf_rn(a[0], sw[u].x, side[u]))));
      }
    };
    static_assert(F2_DQ % 2 == 0, "the two fragment sets alternate with compile-time parity");
    Frag fa, fb;
    fetch(fa, 0);
    int s = 0;
#pragma unroll 1
    for (; s + F2_DQ <= nks; s += F2_DQ) {
#pragma unroll
      for (int u = 0; u < F2_DQ; ++u) {
        if (u & 1) step(q[u], s + u, fb, fa);
        else step(q[u], s + u, fa, fb);
      }
    }
#pragma unroll
    for (int u = 0; u < F2_DQ; ++u)
      if (s + u < nks) {
        if (u & 1) step(q[u], s + u, fb, fa);
        else step(q[u], s + u, fa, fb);
      }
    if (nside) {  // the two lane halves covered different k: combine
#pragma unroll
      for (int u = 0; u < F2_SIDE; ++u) side[u] += __shfl_xor(side[u], 32, 64);
    }
    const unsigned long long TB = F2_TICK();
    epi.tile(est, pre, acc, side, ctx, geo);
    tk += TB - TA; te += F2_TICK() - TB; ++ntile;
  }
  const unsigned long long T2 = F2_TICK();
  epi.flush(est, ctx, geo, smem);
#ifdef OBMAN_F2_TIMING
  if (lane == 0 && blockIdx.x < 256) { unsigned long long* o = f2_dbg + ((size_t)blockIdx.x * 8 + wave) * 8; o[0] = T1 - T0; o[1] = tk; o[2] = te; o[3] = ntile; o[4] = T2 - T1; o[5] = F2_TICK() - T0; o[6] = cg; o[7] = nks; }
#else
  (void)T0; (void)T1; (void)T2; (void)tk; (void)te; (void)ntile;
#endif
}
