// K6, bf16-MFMA flavour, third generation of a "rows" GEMM whose A operand is ONE bf16 array read as it is stored - in the product:
// h3 (A from H2, BBnRelu + EpiStoreB2).  Included by decoder.hip inside namespace dec, after decoder_rows2.h, whose geometry
// (R2Geo), operand transforms (R2Fin), weight-stationary layout and epilogues it re-uses unchanged.
// Measured on the other two single-array kernels and NOT taken there (profiles/r05_kernels.md section 2): gy2 (BGradH3 + EpiMaskB2)
// 508 -> 496 .. 516 us (noise; that kernel is VALU-issue bound in its generator and epilogue), dA (BPlain + EpiL1B2, which only has
// LDS for 64-byte row pieces) 611 -> 640 .. 655 us.  h3: 236 -> 190 .. 210 us.
//
// What is wrong with the second generation (rows2_bf16_kernel) for h3: its A fragment is loaded "one row per lane" -
// lane (row i, k half h) pulls the 16 bytes of ITS row for each 16-deep k-step, so one wave-level load touches 32 - 40 different
// 128-byte lines for 1 KB of data, four consecutive k-steps touch the same lines again, and the branch-free loop requests 8 more
// k-steps per tile than the tile has (profiles/r04_kernels.md section 4: the CU's one texture path, not the matrix pipe, is
// what these kernels sit on - 5.5 .. 20 % matrix-busy).
//
// Here the wave's A tile is brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`: 1 KB per wave-instruction, no VGPR round trip),
// one instruction = 8 rows x 128 contiguous bytes (= 64 k = four k-steps), into a wave-private ring of NB blocks of 4 KB
// (32 rows x 128 B) that runs ACROSS the wave's row tiles: the first block of the next tile is in flight while the current
// tile's last block and its epilogue run, and nothing is requested that is never consumed except the tail of a row's last
// block (272 = 4 x 64 + 16).  The MFMA fragment of a k-step is one ds_read_b128 per lane from the ring.
//   * A DMA writes LDS linearly (lane l -> base + 16 l), so the LAYOUT is made on the source side: slot (instruction q,
//     slot-row rho = l >> 3, position p = l & 7) holds 16-byte chunk p ^ (i & 7) of tile row i = 16 (q >> 1) + 8 (rho & 1) +
//     4 (q & 1) + (rho >> 1).  A fragment read of (row i, chunk c) then lands on 16-byte bank slot 8 ((i >> 3) & 1) + (c ^ (i & 7))
//     - distinct for the 16 rows of each ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, same for the upper
//     half): conflict-free.
//   * The DMA and its waits are inline asm (hipcc knows nothing about them): the compiler would otherwise wait for EVERY
//     outstanding LDS-DMA before any LDS read that it cannot prove disjoint - i.e. before every weight-fragment read.  Counted
//     waits: before block m is read, `s_waitcnt vmcnt(4 (NB - 1))` - the NB - 1 younger blocks (4 instructions each) may stay
//     in flight; vector-memory operations return in order, so operations the compiler issued in between (the epilogue's
//     loads and stores) only make the wait stricter, never wrong.  A slot is re-armed only after `s_waitcnt lgkmcnt(0)`: every
//     fragment read of its previous content has returned.
//   * Rows outside the problem carry an out-of-range per-lane offset (the hardware range check) and their fragment is zeroed by
//     a select on the consuming side, as in rows2.
// Everything else - persistent block per CU, 128 (+ up to 3 side) weight columns stationary in LDS, wave = 32 rows x 128 columns,
// no barrier in the k loop, per-lane BatchNorm moments - is rows2's.
#pragma once

constexpr int R3_BLOCK_BYTES = 4096;  // 32 rows x 128 bytes = four k-steps

template <class AOp> struct R3Src;    // the one bf16 array of the generator and its pitch
template <> struct R3Src<BBnRelu> { static __device__ __forceinline__ const bfraw* ptr(const BBnRelu& a) { return a.H; } };
template <> struct R3Src<BGradH3> { static __device__ __forceinline__ const bfraw* ptr(const BGradH3& a) { return a.H; } };
template <> struct R3Src<BPlain> { static __device__ __forceinline__ const bfraw* ptr(const BPlain& a) { return a.A; } };
template <class AOp> struct R3Raw { static __device__ __forceinline__ typename AOp::Raw make(u32x4 v) { return typename AOp::Raw{v}; } };

__device__ __forceinline__ void r3_dma16(__amdgpu_buffer_rsrc_t r, unsigned lds_base, unsigned voff, int soff) {
  // m0 = LDS base of the wave-instruction; lane l writes [m0 + 16 l, + 16).  m0 is declared clobbered: gfx9 DS
  // instructions do not read it, but the compiler may use it for s_movrel / v_readlane / LDS-direct and must not keep a value
  // of its own live across this statement
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}

template <class AOp, class Epi, int NB>  // NB = 2 or 3 ring blocks per wave
__global__ __launch_bounds__(R2_THREADS) void rows3_bf16_kernel(AOp aop, const bfraw* __restrict__ Wb, int Kp, int Nc, Epi epi, R2Geo geo,
                                                                 int lds_aop_floats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KP2 = Kp + 8;
  bfraw* Ws = reinterpret_cast<bfraw*>(smem);
  float* kcs = reinterpret_cast<float*>(Ws + (size_t)(R2_COLS + geo.wside) * KP2);
  float* red = kcs + lds_aop_floats;        // Epi::LDS_FLOATS floats of epilogue scratch
  char* ring0 = reinterpret_cast<char*>(red + Epi::LDS_FLOATS);  // [R2_WAVES][NB][R3_BLOCK_BYTES], 16-byte aligned by construction
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int vid = xcd_virtual_id(blockIdx.x, gridDim.x), cg = vid % geo.ngroups, slot = vid / geo.ngroups;
  const int c0 = cg * R2_COLS;
  const int last_group = cg == geo.ngroups - 1;
  const int gcols = last_group ? Nc - c0 : R2_COLS;
  const int nside = gcols > R2_COLS ? gcols - R2_COLS : 0;
  {
    const int chunks = Kp >> 3, total = (R2_COLS + geo.wside) * chunks;
    for (int i = tid; i < total; i += R2_THREADS) {
      const int cc = i / chunks, q = i - cc * chunks;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (cc < gcols) v = *reinterpret_cast<const u32x4*>(Wb + (size_t)(c0 + cc) * Kp + q * 8);
      *reinterpret_cast<u32x4*>(Ws + (size_t)cc * KP2 + q * 8) = v;
    }
  }
  R2Lds<AOp>::stage(aop, kcs, Kp, tid, (vid / geo.ngroups) / geo.spb, geo);
  __syncthreads();

  R2Ctx ctx{lane, wave, c0, nside, last_group, slot, 0, 0};
  typename Epi::State est;
  epi.init(est, ctx);
  const int bg = slot / geo.spb, sq = slot - bg * geo.spb;
  const int vt_beg = sq * geo.chunk, vt_end = vt_beg + geo.chunk < geo.nvt ? vt_beg + geo.chunk : geo.nvt;
  constexpr int BKS = 4;  // k-steps per ring block
  const int nks = Kp >> 4, nb = (nks + BKS - 1) / BKS;  // k-steps and ring blocks per tile
  const bfraw* wlane = Ws + (size_t)li * KP2 + h * 8;
  ctx.bg = bg;

  // ---- the wave's ring
  const int ld = aop.ld;
  const __amdgpu_buffer_rsrc_t rs = r2_rsrc(R3Src<AOp>::ptr(aop), (size_t)geo.R * ld * 2);
  char* ring = ring0 + (size_t)wave * (NB * R3_BLOCK_BYTES);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)ring);
  // DMA side: instruction q of a block, this lane = slot-row rho, position p.  Fragment side: where (row li, chunk c) lives.
  const int rho = lane >> 3, pos = lane & 7;
  const unsigned frag_base = (unsigned)((2 * (li >> 4) + ((li >> 2) & 1)) * 1024 + (2 * (li & 3) + ((li >> 3) & 1)) * 128);
  unsigned frag_off[BKS];
#pragma unroll
  for (int s4 = 0; s4 < BKS; ++s4) frag_off[s4] = frag_base + (unsigned)((((2 * s4 + h) ^ (li & (2 * BKS - 1)))) << 4);

  // per-lane byte offsets of the BKS DMA instructions of the tile being REQUESTED (vt_req), and the request cursor
  unsigned voff[BKS];
  int vt_req = vt_beg, blk_req = 0, m_req = 0;
  auto tile_offsets = [&](int vt) {
#pragma unroll
    for (int q = 0; q < BKS; ++q) {
      const int i = 16 * (q >> 1) + 8 * (rho & 1) + 4 * (q & 1) + (rho >> 1);
      int b, n; long r; bool ok;
      geo.row(bg, vt, wave, i, b, n, r, ok);
      ok = ok && vt < vt_end;
      voff[q] = ok ? (unsigned)(((size_t)r * ld) * 2 + (unsigned)((pos ^ (i & (2 * BKS - 1))) << 4)) : 0xffffff00u;
    }
  };
  auto request = [&]() {  // block (vt_req, blk_req) -> ring slot m_req % NB; past the last tile: out-of-range requests (keep the counts uniform)
    const unsigned dst = ring_lds + (unsigned)((m_req % NB) * R3_BLOCK_BYTES);
    const int soff = blk_req * (32 * BKS);
#pragma unroll
    for (int q = 0; q < BKS; ++q) r3_dma16(rs, dst + q * 1024, voff[q], soff);
    ++m_req;
    if (++blk_req == nb) { blk_req = 0; ++vt_req; tile_offsets(vt_req); }
  };
  tile_offsets(vt_req);
#pragma unroll
  for (int u = 0; u < NB; ++u) request();

  int m = 0;  // block being consumed
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

    for (int blk = 0; blk < nb; ++blk, ++m) {
      // block m has landed once at most the NB - 1 younger blocks (BKS instructions each) are outstanding
      if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      const char* blkp = ring + (size_t)(m % NB) * R3_BLOCK_BYTES;
      const int s0 = blk * BKS;
      const int ns = nks - s0 < BKS ? nks - s0 : BKS;
#pragma unroll
      for (int s4 = 0; s4 < BKS; ++s4) {
        if (s4 < ns) {
          const int s = s0 + s4;
          bf16x8 fb[R2_NT];
#pragma unroll
          for (int j = 0; j < R2_NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + s * 16);
          const typename AOp::Raw qs = R3Raw<AOp>::make(*reinterpret_cast<const u32x4*>(blkp + frag_off[s4]));
          __builtin_amdgcn_sched_barrier(0);
          u32x4 a0;
          const int kf = s * 16 + h * 8;
          if constexpr (R2Packed<R2Fin<AOp>>::value) {
            a0 = R2Fin<AOp>::finp(aop, row, kcs, Kp, kf, qs);
          } else {
            float o0[8];
            R2Fin<AOp>::fin(aop, row, kcs, Kp, kf, qs, o0);
            a0 = pack8(o0);
          }
          if (!ok) a0 = u32x4{0u, 0u, 0u, 0u};
          const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
          for (int j = 0; j < R2_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[j], acc[j], 0, 0, 0);
          if (nside) {  // leftover columns of the last group on the VALU, from the SAME rounded operands the MFMAs consume
#pragma unroll
            for (int t = 0; t < R2_SIDE; ++t) {
              if (t < nside) {
                const u32x4 wv = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + s * 16 + h * 8);
                float sd = side[t];
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(sd) : "v"(a0.x), "v"(wv.x));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(sd) : "v"(a0.y), "v"(wv.y));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(sd) : "v"(a0.z), "v"(wv.z));
                asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(sd) : "v"(a0.w), "v"(wv.w));
                side[t] = sd;
              }
            }
          }
        }
      }
      // every fragment read of this slot has returned -> re-arm it with the block NB ahead
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      request();
    }
    if (nside) {
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) side[t] += __shfl_xor(side[t], 32, 64);
    }
    epi.tile(est, acc, side, ctx, geo, red);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the out-of-range tail requests: nothing may land after the ring's LDS is re-used
  epi.flush(est, ctx, geo, smem);
}
