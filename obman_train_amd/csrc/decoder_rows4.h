// K6, bf16-MFMA flavour, round 6: the h2 GEMM with ONE wave per SIMD and several 32-row fragments per wave - the measured
// experiment VERDICT r05 asked for (task 1a), included by decoder.hip inside namespace dec after decoder_rows2.h.
//
// rows2_bf16_kernel runs two waves per SIMD, each owning one 32-row fragment x 128 columns (64 accumulator registers): per 16-deep
// k-step a wave issues 4 MFMAs and ~20 other instructions (4 weight-fragment reads, 2 + 2 operand requests / constant reads, 12
// transform instructions) - 5 per MFMA before the epilogue - and the SIMD's issue port, not the matrix pipe (25 % busy), is what
// the two waves fight over (profiles/r05_kernels.md section 2).  The round-5 note dismissed wider wave tiles on paper ("one wave per
// SIMD has nobody to hide its waits").  Here it is built:
//   * 256-thread blocks = four waves, ONE per SIMD, up to 512 registers each;
//   * a wave owns MF fragments of (8 samples) x (4 vertices) - the SAME 8 samples in every fragment - x 128 columns: 16 MF
//     accumulator registers x 4.  The weight slice stays stationary in LDS exactly as in rows2 (257 x 528 bf16 is 271 KB: 256
//     columns per block cannot be held, so the column width per wave cannot grow; the row count can);
//   * per k-step a wave reads its 4 weight fragments and its 2 feature-factor vectors ONCE for all MF fragments (the fragments
//     share the samples), and generates MF operand fragments: (6 + 14 MF) instructions per 4 MF MFMAs - 3.9 per MFMA at MF = 4
//     against 5.0 - with 4 x the MFMAs between two LDS round trips;
//   * the waits a second wave used to hide are covered by distance instead: raw operand chunks are requested DQ k-steps ahead
//     (register queue, as rows2), weight fragments and feature-factor vectors of step s + 1 are read while step s multiplies.
// Epilogue, moments, side column, geometry (mode 3 = mode 2 with a wider block tile) and the weight image are rows2's.
#pragma once

constexpr int R4_THREADS = 256, R4_WAVES = 4;

// fp64 sums held by (two lane halves) x (four waves) -> dst[(slot * Nc + col) * 2 + {0,1}], fixed order: r2_flush_cols for 4 waves
__device__ __forceinline__ void r4_flush_cols(double (&d1)[R2_NT], double (&d2)[R2_NT], const float (&f1)[R2_SIDE], const float (&f2)[R2_SIDE],
                                              const R2Ctx& c, int rwave, int Nc, double* __restrict__ dst, char* smem) {
  const int li = c.lane & 31;
#pragma unroll
  for (int j = 0; j < R2_NT; ++j) { d1[j] += __shfl_xor(d1[j], 32, 64); d2[j] += __shfl_xor(d2[j], 32, 64); }
  double e1[R2_SIDE], e2[R2_SIDE];
#pragma unroll
  for (int t = 0; t < R2_SIDE; ++t) {
    e1[t] = (double)f1[t];
    e2[t] = (double)f2[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { e1[t] += __shfl_xor(e1[t], off, 64); e2[t] += __shfl_xor(e2[t], off, 64); }
  }
  double* red = reinterpret_cast<double*>(smem);  // [3 waves][R2_NT + 1][32][2]
  __syncthreads();                                // every wave is done with the weight slice
  if (rwave > 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { double* q = red + ((((rwave - 1) * (R2_NT + 1) + j) * 32) + li) * 2; q[0] = d1[j]; q[1] = d2[j]; }
    if (li < R2_SIDE) { double* q = red + ((((rwave - 1) * (R2_NT + 1) + R2_NT) * 32) + li) * 2; q[0] = r2_pick(e1, li); q[1] = r2_pick(e2, li); }
  }
  __syncthreads();
  if (rwave == 0 && c.lane < 32) {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int col = c.c0 + j * 32 + li;
      if (col < Nc) {
        double a = d1[j], b = d2[j];
#pragma unroll
        for (int w = 0; w < R4_WAVES - 1; ++w) { const double* q = red + (((w * (R2_NT + 1) + j) * 32) + li) * 2; a += q[0]; b += q[1]; }
        double* o = dst + ((size_t)c.slot * Nc + col) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
    if (li < c.nside) {
      double a = r2_pick(e1, li), b = r2_pick(e2, li);
#pragma unroll
      for (int w = 0; w < R4_WAVES - 1; ++w) { const double* q = red + (((w * (R2_NT + 1) + R2_NT) * 32) + li) * 2; a += q[0]; b += q[1]; }
      double* o = dst + ((size_t)c.slot * Nc + c.c0 + R2_COLS + li) * 2;
      o[0] = a;
      o[1] = b;
    }
  }
}

// grid = ngroups * slots blocks as rows2.  Dynamic LDS: weight slice [(R2_COLS + geo.wside)][Kp + 8] bf16, the block's 8 rows of
// Fy [8][Kp + 4] fp32, then R4_WAVES x 1 KB of epilogue scratch.
// TIMING: s_memtime stamps per wave (k loop / epilogue / whole), summed into r4_dbg[block][wave][4] (measurement build only).
#ifdef OBMAN_R4_TIMING
__device__ unsigned long long r4_dbg[1024 * R4_WAVES * 4];
#endif
template <int MF, int DQ, int PF>  // PF: LDS operands of step s + 1 read while step s multiplies (second register set)
__global__ __launch_bounds__(R4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void rows4_h2_kernel(BGridFeatPre aop, const bfraw* __restrict__ Wb, int Kp, int Nc, EpiStoreB2 epi, R2Geo geo, int lds_aop_floats) {
  typedef BGridFeatPre AOp;
  typedef R2Fin<AOp> Fin;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KP2 = Kp + 8;
  bfraw* Ws = reinterpret_cast<bfraw*>(smem);
  float* kcs = reinterpret_cast<float*>(Ws + (size_t)(R2_COLS + geo.wside) * KP2);
  float* red = kcs + lds_aop_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int vid = xcd_virtual_id(blockIdx.x, gridDim.x), cg = vid % geo.ngroups, slot = vid / geo.ngroups;
  const int c0 = cg * R2_COLS;
  const int last_group = cg == geo.ngroups - 1;
  const int gcols = last_group ? Nc - c0 : R2_COLS;
  const int nside = gcols > R2_COLS ? gcols - R2_COLS : 0;
  const int bg = slot / geo.spb, sq = slot - bg * geo.spb;
  {
    const int chunks = Kp >> 3, total = (R2_COLS + geo.wside) * chunks;
    for (int i = tid; i < total; i += R4_THREADS) {
      const int cc = i / chunks, q = i - cc * chunks;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (cc < gcols) v = *reinterpret_cast<const u32x4*>(Wb + (size_t)(c0 + cc) * Kp + q * 8);
      *reinterpret_cast<u32x4*>(Ws + (size_t)cc * KP2 + q * 8) = v;
    }
    const int pitch = Kp + AOp::FPITCH_PAD, fch = Kp >> 2;  // the block's 8 rows of Fy (R2Lds<BGridFeatPre>::stage for 256 threads)
    for (int i = tid; i < 8 * fch; i += R4_THREADS) {
      const int sl = i / fch, c = (i - sl * fch) * 4, b = bg * 8 + sl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < geo.B && c + 4 <= aop.ld) v = *reinterpret_cast<const float4*>(aop.Fy + (size_t)b * aop.ld + c);
      *reinterpret_cast<float4*>(kcs + (size_t)sl * pitch + c) = v;
    }
  }
  __syncthreads();

  R2Src<AOp> src;
  src.init(aop, geo);
  R2Ctx ctx{lane, wave * MF, c0, nside, last_group, slot, bg, 0, wave};
  EpiStoreB2::State est;
  epi.init(est, ctx);
  const int vt_beg = sq * geo.chunk, vt_end = vt_beg + geo.chunk < geo.nvt ? vt_beg + geo.chunk : geo.nvt;
  const int nks = Kp >> 4;
  const bfraw* wlane = Ws + (size_t)li * KP2 + h * 8;
  const AOp::Row rowl = aop.row(0, bg * 8 + (li >> 2), 0, true);  // sample slot li >> 2: the same in every fragment
#ifdef OBMAN_R4_TIMING
  unsigned long long t_loop = 0, t_epi = 0;
  const unsigned long long t_begin = __builtin_readcyclecounter();
#endif
  for (int vt = vt_beg; vt < vt_end; ++vt) {
    ctx.vt = vt;
    typename R2Src<AOp>::Off roff[MF];
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      int b, n; long r; bool ok;
      geo.row(bg, vt, wave * MF + f, li, b, n, r, ok);
      if (!ok) n = geo.N;  // the sentinel row of Gy: relu gives exact zeros
      roff[f] = src.off(aop, r, b, n, h);
    }
    f32x16 acc[MF][R2_NT];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int j = 0; j < R2_NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][j][r] = 0.f;
    float side[MF][R2_SIDE];
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) side[f][t] = 0.f;
#ifdef OBMAN_R4_TIMING
    const unsigned long long T0 = __builtin_readcyclecounter();
#endif
    AOp::Raw q[DQ][MF];
#pragma unroll
    for (int u = 0; u < DQ; ++u)
#pragma unroll
      for (int f = 0; f < MF; ++f) src.load(q[u][f], roff[f], u);
    // LDS operands of a k-step (weight fragments, feature-factor vectors, side-column weights): read one step ahead
    struct Stage { bf16x8 fb[R2_NT]; Fin::Pref c; u32x4 wv[R2_SIDE]; };
    auto fetch = [&](Stage& st, int ks) {
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) st.fb[j] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + ks * 16);
      st.c = Fin::pre(aop, rowl, kcs, Kp, ks * 16 + h * 8);
      if (nside) {
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t)
          if (t < nside) st.wv[t] = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + ks * 16 + h * 8);
      }
    };
    auto step = [&](AOp::Raw (&qs)[MF], int ks, Stage& cur, Stage& nxt) {
      if constexpr (PF) fetch(nxt, ks + 1 < nks ? ks + 1 : ks);
      else { fetch(cur, ks); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const u32x4 a0 = Fin::finq(cur.c, qs[f]);
        src.load(qs[f], roff[f], ks + DQ);
        const bf16x8 fa = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cur.fb[j], acc[f][j], 0, 0, 0);
        if (nside) {
#pragma unroll
          for (int t = 0; t < R2_SIDE; ++t) {
            if (t < nside) {
              float s0 = side[f][t];
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.x), "v"(cur.wv[t].x));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.y), "v"(cur.wv[t].y));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.z), "v"(cur.wv[t].z));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.w), "v"(cur.wv[t].w));
              side[f][t] = s0;
            }
          }
        }
      }
    };
    static_assert(DQ % 2 == 0, "the two LDS operand sets alternate with compile-time parity");
    Stage sa, sb;
    if constexpr (PF) fetch(sa, 0);
    int s = 0;
    for (; s + DQ <= nks; s += DQ) {
#pragma unroll
      for (int u = 0; u < DQ; ++u) {
        if (PF && (u & 1)) step(q[u], s + u, sb, sa);
        else step(q[u], s + u, sa, sb);
      }
    }
#pragma unroll
    for (int u = 0; u < DQ; ++u)
      if (s + u < nks) {
        if (PF && (u & 1)) step(q[u], s + u, sb, sa);
        else step(q[u], s + u, sa, sb);
      }
    if (nside) {
#pragma unroll
      for (int f = 0; f < MF; ++f)
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t) side[f][t] += __shfl_xor(side[f][t], 32, 64);
    }
#ifdef OBMAN_R4_TIMING
    const unsigned long long T1 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      ctx.wave = wave * MF + f;
      __builtin_amdgcn_sched_barrier(0);  // one fragment's epilogue at a time: interleaved, their temporaries spill
      epi.tile(est, acc[f], side[f], ctx, geo, red);
    }
    __builtin_amdgcn_sched_barrier(0);
#ifdef OBMAN_R4_TIMING
    const unsigned long long T2 = __builtin_readcyclecounter();
    t_loop += T1 - T0;
    t_epi += T2 - T1;
#endif
  }
#ifdef OBMAN_R4_TIMING
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* o = r4_dbg + ((size_t)blockIdx.x * R4_WAVES + wave) * 4;
    o[0] = t_loop; o[1] = t_epi; o[2] = __builtin_readcyclecounter() - t_begin; o[3] = (unsigned long long)(vt_end - vt_beg);
  }
#endif
  if (epi.moments) r4_flush_cols(est.d1, est.d2, est.e1, est.e2, ctx, wave, epi.Nc, epi.moments, smem);
}
