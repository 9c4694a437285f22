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

#ifndef R2_EPI_ABL
#define R2_EPI_ABL 0  // measurement-only epilogue ablations (wrong results): 1 no global stores, 2 no transposition and no stores, 3 no moments
#endif
constexpr int R2_NT = 4;               // 32-column MFMA tiles per wave
constexpr int R2_COLS = 32 * R2_NT;    // columns per block (+ up to R2_SIDE side columns in the last group)
constexpr int R2_SIDE = 3;
constexpr int R2_THREADS = 512;
constexpr int R2_WAVES = 8;

inline int kpad16(int K) { return (K + 15) / 16 * 16; }  // k extent of the weight image the rows2 kernels use (one MFMA k-step)

struct R2Geo {
  int R, N, B;
  int mode;     // 0: wave = 1 sample x 32 vertices, block = 8 samples (h2, h3, gy2)
                // 1: wave = 8 samples x 4 vertices, block = 64 samples over the SAME 4 vertices (dA: sums over samples stay in the block)
                // 2: wave = 8 samples x 4 vertices, block = the SAME 8 samples over 32 vertices (h2: the block's 8 feature-factor rows sit in LDS)
  int nvt, nbg; // vertex tiles, sample groups
  int ngroups;  // column groups of R2_COLS
  int slots;    // persistent blocks per column group = spb * nbg
  int spb;      // slots per sample group
  int wside;    // side-column rows of the weight slice actually present (0 .. R2_SIDE): LDS rows = R2_COLS + wside
  int chunk;    // vertex tiles per slot
  // row i (0..31) of the fragment of wave `wave` in block tile (bg, vt)
  __device__ __forceinline__ void row(int bg, int vt, int wave, int i, int& b, int& n, long& r, bool& ok) const {
    if (mode == 0) {
      b = bg * 8 + wave;
      n = vt * 32 + i;
    } else if (mode == 1) {
      b = bg * 64 + wave * 8 + (i >> 2);
      n = vt * 4 + (i & 3);
    } else {
      b = bg * 8 + (i >> 2);
      n = vt * 32 + wave * 4 + (i & 3);
    }
    ok = b < B && n < N;
    if (!ok) { b = 0; n = 0; }
    r = (long)b * N + n;
  }
};

// Layer-1 activation from PRE-SCALED factors (h2 only):  a1 = relu(Gy[n,k] + Fy[b,k]),  Gy = gamma * Gx,  Fy = gamma * Fx + beta
// (l1_fill_kernel / prep_kernel).  Two VALU operations per element instead of three and no per-channel constants; with geometry mode 2 the
// block's 8 rows of Fy sit in LDS for the block's whole life, so a k-step fetches 32 bytes per lane instead of 64.
struct BGridFeatPre {
  const float *Gy, *Fy;
  int ld, K;
  static constexpr int NC = 0;
  static constexpr bool SENTINEL = true;  // rows outside the problem read Gy's row N (-3e38 everywhere): relu gives exact zeros, no select
  static constexpr int FPITCH_PAD = 4;  // row pitch Kp + 4 floats: the 8 sample rows start 20 banks apart (conflict-free 16-byte reads)
  struct Row { int sl; bool ok; };      // sample slot 0..7 inside the block
  struct Raw { u32x4 g0, g1; };
  __device__ Row row(long, int b, int, bool ok) const { return Row{b & 7, ok}; }
};

// A operand stored as it is consumed: bf16 [R, ld] (gh2 after gh2_inplace_kernel).  Rows outside the problem are addressed beyond
// the buffer's range - the hardware bounds check returns zeros, so there is neither a transform nor a select.
constexpr size_t R2_PLAIN_MAX_BYTES = 0xffff0000ull;  // BPlain addresses its operand with 32-bit byte offsets and masks rows with 0xffffff00
struct BPlain {
  const bfraw* A;
  int ld, K;
  static constexpr int NC = 0;
  static constexpr bool SELFMASK = true;
  struct Row {};
  struct Raw { u32x4 v; };
  __device__ Row row(long, int, int, bool) const { return Row{}; }
  __device__ void stage(float*, int, int) const {}
};

// request-queue depth per operand generator (k-steps in flight per wave): 16 registers per step for the fp32 layer-1 factors,
// 4 / 8 for the bf16-stored activations
template <class AOp> struct R2Depth { static constexpr int value = 8; };
template <> struct R2Depth<BGridFeat> { static constexpr int value = 4; };
template <> struct R2Depth<BGradH> { static constexpr int value = 2; };  // layers too wide for a materialised gh2 (ld2 > 512): EpiL1B2 keeps 32 registers of state
template <> struct R2Depth<BGridFeatPre> { static constexpr int value = 4; };

// LDS floats an operand generator needs besides the weight slice (per-channel constants; the 8 Fy rows), and how it fills them
template <class AOp> struct R2Lds {
  static int floats(int Kp) { return AOp::NC * Kp; }
  static __device__ __forceinline__ void stage(const AOp& op, float* kcs, int Kp, int tid, int, const R2Geo&) { op.stage(kcs, Kp, tid); }
};
template <> struct R2Lds<BGridFeatPre> {
  static int floats(int Kp) { return 8 * (Kp + BGridFeatPre::FPITCH_PAD); }
  static __device__ __forceinline__ void stage(const BGridFeatPre& op, float* kcs, int Kp, int tid, int bg, const R2Geo& geo) {
    const int pitch = Kp + BGridFeatPre::FPITCH_PAD, chunks = Kp >> 2;
    for (int i = tid; i < 8 * chunks; i += R2_THREADS) {
      const int sl = i / chunks, c = (i - sl * chunks) * 4, b = bg * 8 + sl;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < geo.B && c + 4 <= op.ld) v = *reinterpret_cast<const float4*>(op.Fy + (size_t)b * op.ld + c);
      *reinterpret_cast<float4*>(kcs + (size_t)sl * pitch + c) = v;
    }
  }
};

// ------------------------------------------------------------------------------------------------ operand sources
// The raw chunks of a lane's row are fetched with BUFFER loads: a wave-uniform 128-bit descriptor per source array, a 32-bit
// per-lane byte offset fixed for the whole row tile (row start + the lane half's 8 k), and the k-step as the instruction's
// SCALAR offset - no per-load 64-bit address arithmetic and no clamping on the VALU (the flat-load version spent ~16 of its
// ~100 instructions per k-step on it).  Reads past the last k-step of a row land in the next row (finite values, never
// consumed: only issued to keep the loop branch-free).  The hardware range check covers the PER-LANE offset only (not the
// scalar k offset): the requests past the last row of an array land in whatever follows it in the workspace arena - never
// consumed either; fwd_ws / bwd_ws (decoder.hip) end with WS_TAIL_FLOATS of padding so that this is always inside the arena.
// The transforms are the B* generators' own fin().
__device__ __forceinline__ __amdgpu_buffer_rsrc_t r2_rsrc(const void* p, size_t bytes) {
  // uniformity made PROVABLE (readfirstlane on the inputs): under SGPR pressure hipcc parks the descriptor in VGPRs and then
  // wraps every buffer load in a waterfall loop (~12 instructions and a serialisation point per load)
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)(bytes > 0xfffffffcull ? 0xfffffffcull : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, (int)n, 0x00020000);
}
__device__ __forceinline__ u32x4 r2_ld16(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
__device__ __forceinline__ float4 r2_f4(u32x4 v) { return __builtin_bit_cast(float4, v); }

template <class AOp> struct R2Src;
template <>
struct R2Src<BGridFeat> {
  __amdgpu_buffer_rsrc_t rg, rf;
  struct Off { unsigned g, f; };
  __device__ __forceinline__ void init(const BGridFeat& op, const R2Geo& geo) {
    rg = r2_rsrc(op.Gx, (size_t)geo.N * op.ld * 4);
    rf = r2_rsrc(op.Fx, (size_t)geo.B * op.ld * 4);
  }
  __device__ __forceinline__ Off off(const BGridFeat& op, long, int b, int n, int h) const {
    return Off{(unsigned)(((size_t)n * op.ld + h * 8) * 4), (unsigned)(((size_t)b * op.ld + h * 8) * 4)};
  }
  __device__ __forceinline__ void load(BGridFeat::Raw& q, const Off& o, int s) const {
    q.g0 = r2_f4(r2_ld16(rg, o.g, s * 64)); q.g1 = r2_f4(r2_ld16(rg, o.g + 16, s * 64));
    q.f0 = r2_f4(r2_ld16(rf, o.f, s * 64)); q.f1 = r2_f4(r2_ld16(rf, o.f + 16, s * 64));
  }
};
template <>
struct R2Src<BGridFeatPre> {
  __amdgpu_buffer_rsrc_t rg;
  struct Off { unsigned g; };
  __device__ __forceinline__ void init(const BGridFeatPre& op, const R2Geo& geo) { rg = r2_rsrc(op.Gy, (size_t)(geo.N + 1) * op.ld * 4); }
  __device__ __forceinline__ Off off(const BGridFeatPre& op, long, int, int n, int h) const { return Off{(unsigned)(((size_t)n * op.ld + h * 8) * 4)}; }
  __device__ __forceinline__ void load(BGridFeatPre::Raw& q, const Off& o, int s) const {
    q.g0 = r2_ld16(rg, o.g, s * 64);
    q.g1 = r2_ld16(rg, o.g + 16, s * 64);
  }
};
template <>
struct R2Src<BBnRelu> {
  __amdgpu_buffer_rsrc_t rh;
  struct Off { unsigned h; };
  __device__ __forceinline__ void init(const BBnRelu& op, const R2Geo& geo) { rh = r2_rsrc(op.H, (size_t)geo.R * op.ld * 2); }
  __device__ __forceinline__ Off off(const BBnRelu& op, long r, int, int, int h) const { return Off{(unsigned)(((size_t)r * op.ld + h * 8) * 2)}; }
  __device__ __forceinline__ void load(BBnRelu::Raw& q, const Off& o, int s) const { q.h = r2_ld16(rh, o.h, s * 32); }
};
template <>
struct R2Src<BGradH3> {
  __amdgpu_buffer_rsrc_t rh;
  struct Off { unsigned h; };
  __device__ __forceinline__ void init(const BGradH3& op, const R2Geo& geo) { rh = r2_rsrc(op.H, (size_t)geo.R * op.ld * 2); }
  __device__ __forceinline__ Off off(const BGradH3& op, long r, int, int, int h) const { return Off{(unsigned)(((size_t)r * op.ld + h * 8) * 2)}; }
  __device__ __forceinline__ void load(BGradH3::Raw& q, const Off& o, int s) const { q.h = r2_ld16(rh, o.h, s * 32); }
};
template <>
struct R2Src<BPlain> {
  __amdgpu_buffer_rsrc_t ra;
  struct Off { unsigned o; };
  __device__ __forceinline__ void init(const BPlain& op, const R2Geo& geo) { ra = r2_rsrc(op.A, (size_t)geo.R * op.ld * 2); }
  __device__ __forceinline__ Off off(const BPlain& op, long r, int, int, int h) const { return Off{(unsigned)(((size_t)r * op.ld + h * 8) * 2)}; }
  // The per-lane offset alone decides the range check of a raw buffer load (the scalar k offset is not part of it): 0xffffff00 is
  // beyond every buffer this path accepts (R2_PLAIN_MAX_BYTES, checked by the host - the configs[4] operand is 2.2 GB).
  __device__ __forceinline__ Off off_masked(const BPlain& op, long r, int h, bool ok) const {
    return Off{ok ? (unsigned)(((size_t)r * op.ld + h * 8) * 2) : 0xffffff00u};
  }
  __device__ __forceinline__ void load(BPlain::Raw& q, const Off& o, int s) const { q.v = r2_ld16(ra, o.o, s * 32); }
};
template <>
struct R2Src<BGradH> {
  __amdgpu_buffer_rsrc_t rgy, rh;
  struct Off { unsigned o; };
  __device__ __forceinline__ void init(const BGradH& op, const R2Geo& geo) {
    rgy = r2_rsrc(op.GY, (size_t)geo.R * op.ld * 2);
    rh = r2_rsrc(op.H, (size_t)geo.R * op.ld * 2);
  }
  __device__ __forceinline__ Off off(const BGradH& op, long r, int, int, int h) const { return Off{(unsigned)(((size_t)r * op.ld + h * 8) * 2)}; }
  __device__ __forceinline__ void load(BGradH::Raw& q, const Off& o, int s) const {
    q.gy = r2_ld16(rgy, o.o, s * 32);
    q.h = r2_ld16(rh, o.o, s * 32);
  }
};

// transforms: the generators' own fin(), with the per-channel constants read from LDS as whole 16-byte vectors up front (the
// element-indexed form compiled to eight dependent ds_read2_b32 + wait pairs per k-step)
template <class AOp>
struct R2Fin {
  static __device__ __forceinline__ void fin(const AOp& op, const typename AOp::Row& row, const float* kcs, int Kp, int k,
                                             const typename AOp::Raw& q, float* o) {
    op.fin(row, kcs, Kp, k, q, o);
  }
};
// does the generator produce the packed bf16 fragment itself (finp) / zero the rows outside the problem itself?
template <class F, class = void> struct R2Packed { static constexpr bool value = false; };
template <class F> struct R2Packed<F, std::enable_if_t<F::PACKED>> { static constexpr bool value = true; };
template <class AOp, class = void> struct R2SelfMask { static constexpr bool value = false; };
template <class AOp> struct R2SelfMask<AOp, std::enable_if_t<AOp::SELFMASK>> { static constexpr bool value = true; };
template <class AOp, class = void> struct R2Sentinel { static constexpr bool value = false; };
template <class AOp> struct R2Sentinel<AOp, std::enable_if_t<AOp::SENTINEL>> { static constexpr bool value = true; };
typedef short s16x2v __attribute__((ext_vector_type(2)));
// relu on a PACKED bf16 pair: as signed 16-bit integers every negative value (sign bit) is < 0 and every non-negative one is
// >= 0 and unchanged by max(., 0) - one v_pk_max_i16 per two elements; rounding first and clamping second gives the same
// bits as the other order (round-to-nearest keeps the sign)
__device__ __forceinline__ unsigned relu_bf16x2(unsigned w) {
  const s16x2v z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2v, w), z));
}
template <>
struct R2Fin<BGridFeatPre> {
  // 4 v_pk_add_f32 + 4 v_cvt_pk_bf16_f32 + 4 v_pk_max_i16 per 8 elements (scalar form: 8 adds, 8 max, 4 converts)
  static constexpr bool PACKED = true;
  struct Pref { float4 f0, f1; };  // the k-step's LDS constants, requested one k-step ahead by the pipelined loop
  static __device__ __forceinline__ Pref pre(const BGridFeatPre&, const BGridFeatPre::Row& w, const float* fy, int Kp, int k) {
    const float* f = fy + (size_t)w.sl * (Kp + BGridFeatPre::FPITCH_PAD) + k;
    return Pref{*reinterpret_cast<const float4*>(f), *reinterpret_cast<const float4*>(f + 4)};
  }
  static __device__ __forceinline__ u32x4 finp(const BGridFeatPre& op, const BGridFeatPre::Row& w, const float* fy, int Kp, int k, const BGridFeatPre::Raw& q) {
    return finq(pre(op, w, fy, Kp, k), q);
  }
  static __device__ __forceinline__ u32x4 finq(const Pref& c, const BGridFeatPre::Raw& q) {
    const float4 f0 = c.f0, f1 = c.f1;
    const float4 g0 = r2_f4(q.g0), g1 = r2_f4(q.g1);
    const f32x2v s0 = f32x2v{g0.x, g0.y} + f32x2v{f0.x, f0.y}, s1 = f32x2v{g0.z, g0.w} + f32x2v{f0.z, f0.w};
    const f32x2v s2 = f32x2v{g1.x, g1.y} + f32x2v{f1.x, f1.y}, s3 = f32x2v{g1.z, g1.w} + f32x2v{f1.z, f1.w};
    u32x4 o;
    o.x = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(s0, bf16x2)));
    o.y = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(s1, bf16x2)));
    o.z = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(s2, bf16x2)));
    o.w = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(s3, bf16x2)));
    return o;
  }
  static __device__ __forceinline__ void fin(const BGridFeatPre&, const BGridFeatPre::Row& w, const float* fy, int Kp, int k, const BGridFeatPre::Raw& q,
                                             float* o) {
    const float* f = fy + (size_t)w.sl * (Kp + BGridFeatPre::FPITCH_PAD) + k;
    const float4 f0 = *reinterpret_cast<const float4*>(f), f1 = *reinterpret_cast<const float4*>(f + 4);
    const float4 g0 = r2_f4(q.g0), g1 = r2_f4(q.g1);
    o[0] = fmaxf(g0.x + f0.x, 0.f); o[1] = fmaxf(g0.y + f0.y, 0.f); o[2] = fmaxf(g0.z + f0.z, 0.f); o[3] = fmaxf(g0.w + f0.w, 0.f);
    o[4] = fmaxf(g1.x + f1.x, 0.f); o[5] = fmaxf(g1.y + f1.y, 0.f); o[6] = fmaxf(g1.z + f1.z, 0.f); o[7] = fmaxf(g1.w + f1.w, 0.f);
  }
};
template <>
struct R2Fin<BPlain> {
  static constexpr bool PACKED = true;
  static __device__ __forceinline__ u32x4 finp(const BPlain&, const BPlain::Row&, const float*, int, int, const BPlain::Raw& q) { return q.v; }
  struct Pref {};  // no LDS constants (the timing / pipelined measurement variants of the kernel ask for them)
  static __device__ __forceinline__ Pref pre(const BPlain&, const BPlain::Row&, const float*, int, int) { return Pref{}; }
  static __device__ __forceinline__ u32x4 finq(const Pref&, const BPlain::Raw& q) { return q.v; }
};
template <>
struct R2Fin<BBnRelu> {
  static __device__ __forceinline__ void fin(const BBnRelu&, const BBnRelu::Row&, const float* kcs, int Kp, int k, const BBnRelu::Raw& q, float* o) {
    const float4 s0 = *reinterpret_cast<const float4*>(kcs + k), s1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(kcs + Kp + k), t1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float tc[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    float h[8];
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaxf(__fmaf_rn(sc[j], h[j], tc[j]), 0.f);
  }
  // packed: 4 v_pk_fma_f32 + 4 converts + 4 v_pk_max_i16 after the 8 unpacks (scalar: 8 FMAs, 8 max, 4 converts)
  static constexpr bool PACKED = true;
  static __device__ __forceinline__ u32x4 finp(const BBnRelu&, const BBnRelu::Row&, const float* kcs, int Kp, int k, const BBnRelu::Raw& q) {
    const float4 s0 = *reinterpret_cast<const float4*>(kcs + k), s1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(kcs + Kp + k), t1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    float h[8];
    unpack8(q.h, h);
    const f32x2v y0 = __builtin_elementwise_fma(f32x2v{s0.x, s0.y}, f32x2v{h[0], h[1]}, f32x2v{t0.x, t0.y});
    const f32x2v y1 = __builtin_elementwise_fma(f32x2v{s0.z, s0.w}, f32x2v{h[2], h[3]}, f32x2v{t0.z, t0.w});
    const f32x2v y2 = __builtin_elementwise_fma(f32x2v{s1.x, s1.y}, f32x2v{h[4], h[5]}, f32x2v{t1.x, t1.y});
    const f32x2v y3 = __builtin_elementwise_fma(f32x2v{s1.z, s1.w}, f32x2v{h[6], h[7]}, f32x2v{t1.z, t1.w});
    u32x4 o;
    o.x = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(y0, bf16x2)));
    o.y = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(y1, bf16x2)));
    o.z = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(y2, bf16x2)));
    o.w = relu_bf16x2(__builtin_bit_cast(unsigned, __builtin_convertvector(y3, bf16x2)));
    return o;
  }
};
template <>
struct R2Fin<BGradH> {
  static __device__ __forceinline__ void fin(const BGradH&, const BGradH::Row&, const float* kcs, int Kp, int k, const BGradH::Raw& q, float* o) {
    const float4 a0 = *reinterpret_cast<const float4*>(kcs + k), a1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(kcs + Kp + k), b1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(kcs + 2 * Kp + k), c1 = *reinterpret_cast<const float4*>(kcs + 2 * Kp + k + 4);
    const float ka[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const float kb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float kc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    float gy[8], h[8];
    unpack8(q.gy, gy);
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = __fmaf_rn(ka[j], gy[j], __fmaf_rn(kb[j], h[j], kc[j]));
  }
  static constexpr bool PACKED = true;  // 8 v_pk_fma_f32 instead of 16 FMAs
  static __device__ __forceinline__ u32x4 finp(const BGradH&, const BGradH::Row&, const float* kcs, int Kp, int k, const BGradH::Raw& q) {
    const float4 a0 = *reinterpret_cast<const float4*>(kcs + k), a1 = *reinterpret_cast<const float4*>(kcs + k + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(kcs + Kp + k), b1 = *reinterpret_cast<const float4*>(kcs + Kp + k + 4);
    const float4 c0 = *reinterpret_cast<const float4*>(kcs + 2 * Kp + k), c1 = *reinterpret_cast<const float4*>(kcs + 2 * Kp + k + 4);
    float gy[8], h[8];
    unpack8(q.gy, gy);
    unpack8(q.h, h);
    auto one = [](float ax, float ay, float gx, float gy_, float bx, float by, float hx, float hy, float cx, float cy) {
      const f32x2v t = __builtin_elementwise_fma(f32x2v{bx, by}, f32x2v{hx, hy}, f32x2v{cx, cy});
      const f32x2v y = __builtin_elementwise_fma(f32x2v{ax, ay}, f32x2v{gx, gy_}, t);
      return __builtin_bit_cast(unsigned, __builtin_convertvector(y, bf16x2));
    };
    u32x4 o;
    o.x = one(a0.x, a0.y, gy[0], gy[1], b0.x, b0.y, h[0], h[1], c0.x, c0.y);
    o.y = one(a0.z, a0.w, gy[2], gy[3], b0.z, b0.w, h[2], h[3], c0.z, c0.w);
    o.z = one(a1.x, a1.y, gy[4], gy[5], b1.x, b1.y, h[4], h[5], c1.x, c1.y);
    o.w = one(a1.z, a1.w, gy[6], gy[7], b1.z, b1.w, h[6], h[7], c1.z, c1.w);
    return o;
  }
};
template <>
struct R2Fin<BGradH3> {
  static __device__ __forceinline__ void fin(const BGradH3&, const BGradH3::Row& w, const float* kcs, int, int k, const BGradH3::Raw& q, float* o) {
    float h[8];
    unpack8(q.h, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#ifdef OBMAN_ABL_GH3CONST  // measurement only (wrong results): ONE channel's constants for all eight elements - an eighth of the LDS reads
      const float4 c0 = *reinterpret_cast<const float4*>(kcs + (size_t)k * 8);
      const float4 c1 = *reinterpret_cast<const float4*>(kcs + (size_t)k * 8 + 4);
#else
      const float4 c0 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8);      // s, t, kb, kc
      const float4 c1 = *reinterpret_cast<const float4*>(kcs + (size_t)(k + j) * 8 + 4);  // ka*w0, ka*w1, ka*w2
#endif
      // select instead of a divergent skip: the dot product is three FMAs
      const float d = __fmaf_rn(w.g2, c1.z, __fmaf_rn(w.g1, c1.y, w.g0 * c1.x));
      const float gy = __fmaf_rn(c0.x, h[j], c0.y) > 0.f ? d : 0.f;
      o[j] = gy + __fmaf_rn(c0.z, h[j], c0.w);
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
  // Stores are ISSUE-bound on this chip (a wave's store instruction costs the same whether it carries 4 or 16 bytes per lane;
  // the 32 dword stores per tile of the first version were 30 % of the h2 kernel, tools/archive/r03/r03_abl.sh): every 16 x 32 piece of
  // the tile is transposed through 1 KB of LDS per wave (4 ds_write_b32 + 1 ds_read_b128 per lane) and leaves as ONE
  // 16-byte store per lane - 8 store instructions per tile instead of 32.
  static constexpr int LDS_FLOATS = R2_WAVES * 256;
  bfraw* C;
  const float* bias;
  double* moments;  // [slots][Nc][2] or null
  int ldc, Nc;
  struct State {  // only what must persist across the wave's row tiles (the bias is re-read per tile: the h2 kernel has no register left)
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
                                       const R2Geo& geo, float* red) const {
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
    unsigned* tb = reinterpret_cast<unsigned*>(red) + c.wave * 256;  // [16 rows][16 words = 32 columns]
    bool interior;
    {
      int b, n; long r; bool okl;
      geo.row(c.bg, c.vt, c.wave, li, b, n, r, okl);
      interior = __all(okl);  // wave-uniform: every row of the tile is inside the problem
    }
#if R2_EPI_ABL == 0
    if (interior) {
      // Round 5: these kernels are bound by VALU ISSUE (profiles/r05_kernels.md section 2: 47 - 73 k vector instructions per wave
      // against 1 - 4 k MFMAs), and this epilogue was half of h2's / h3's.  For a tile without rows outside the problem a pair
      // of outputs now costs 8 vector instructions instead of ~19: the two moments are ONE v_dot2c_f32_bf16 each on the packed
      // pair (sum = pair . (1, 1), sum of squares = pair . pair; the products of bf16 values are exact in fp32), and the pair
      // exchange with lane ^ 1 is two v_perm_b32 around the DPP move instead of shifts, masks and selects.
      const unsigned sel_send = odd ? 0x0c0c0100u : 0x0c0c0302u;  // the half the neighbour stores: own row 2p (odd) / row 2p + 1 (even)
      const unsigned sel_out = odd ? 0x03020504u : 0x05040100u;   // odd: (neighbour's, own hi)   even: (own lo, neighbour's)
      const unsigned ones = 0x3f803f80u;
      bfraw* dstg[2];
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        int b, n; long r; bool ok;
        geo.row(c.bg, c.vt, c.wave, 16 * G + (c.lane >> 2), b, n, r, ok);
        dstg[G] = C + (size_t)r * ldc + c.c0 + (c.lane & 3) * 8;
      }
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const bool cok = c.c0 + j * 32 + (c.lane & 3) * 8 < ldc;
#pragma unroll
        for (int G = 0; G < 2; ++G) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int p = 4 * G + q;
            const unsigned pk = pack_bf16(acc[j][2 * p] + bv[j], acc[j][2 * p + 1] + bv[j]);
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s1[j]) : "v"(pk), "v"(ones));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s2[j]) : "v"(pk), "v"(pk));
            const int rr = ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * h + (odd ? 1 : 0);
            const unsigned recv = lane_swap1(__builtin_amdgcn_perm(pk, pk, sel_send));
            tb[rr * 16 + (li >> 1)] = __builtin_amdgcn_perm(recv, pk, sel_out);
          }
          const u32x4 v = *reinterpret_cast<const u32x4*>(tb + (c.lane >> 2) * 16 + (c.lane & 3) * 4);
          if (cok) *reinterpret_cast<u32x4*>(dstg[G] + j * 32) = v;
        }
      }
    } else
#endif
    {
      // which rows of the tile this lane's registers hold (validity for the moments) ...
      bool ok0[8], ok1[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int b, n; long r0;
        geo.row(c.bg, c.vt, c.wave, acc_row(2 * p, c.lane), b, n, r0, ok0[p]);
        ok1[p] = ok0[p] && n + 1 < geo.N;  // row i0 + 1: the next vertex of the same sample
      }
      // ... and which rows it stores: row 16 G + (lane >> 2) of the tile, columns 8 (lane & 3) .. + 7 of each 32-column piece
      bfraw* dstg[2];
      bool okg[2];
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        int b, n; long r;
        geo.row(c.bg, c.vt, c.wave, 16 * G + (c.lane >> 2), b, n, r, okg[G]);
        dstg[G] = C + (size_t)r * ldc + c.c0 + (c.lane & 3) * 8;
      }
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const bool cok = c.c0 + j * 32 + (c.lane & 3) * 8 < ldc;  // columns >= Nc inside the pitch hold zeros (zero weights, no bias)
#pragma unroll
        for (int G = 0; G < 2; ++G) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int p = 4 * G + q;
            const unsigned pk = pack_bf16(acc[j][2 * p] + bv[j], acc[j][2 * p + 1] + bv[j]);
            const float v0 = ok0[p] ? bf_lo(pk) : 0.f, v1 = ok1[p] ? bf_hi(pk) : 0.f;
#if R2_EPI_ABL != 3
            s1[j] += v0 + v1;
            s2[j] = __fmaf_rn(v0, v0, __fmaf_rn(v1, v1, s2[j]));
#endif
            // even lane: row 2p, columns (li, li + 1); odd lane: row 2p + 1, columns (li - 1, li)
            const int rr = ((2 * q) & 3) + 8 * ((2 * q) >> 2) + 4 * h + (odd ? 1 : 0);
#if R2_EPI_ABL != 2
            tb[rr * 16 + (li >> 1)] = pair_exchange(pk, odd);
#endif
          }
#if R2_EPI_ABL != 2
          const u32x4 v = *reinterpret_cast<const u32x4*>(tb + (c.lane >> 2) * 16 + (c.lane & 3) * 4);
#if R2_EPI_ABL == 1
          asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
          if (okg[G] && cok) *reinterpret_cast<u32x4*>(dstg[G] + j * 32) = v;
#endif
#endif
        }
      }
    }
    {
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
  // H in and C out as 16-byte row pieces through 1 KB of LDS per wave (see EpiStoreB2): 8 loads + 8 stores per tile instead of
  // 32 + 32 four-byte ones - with only 8 k-steps per tile (K = 128) this epilogue IS the kernel.
  // (Round 5, measured and dropped: the EpiStoreB2 treatment here - v_perm_b32 pair exchange, both sums as v_dot2c_f32_bf16 on packed
  // pairs with xhat applied per column at the flush, a fast path for tiles without rows / columns outside the problem: ~17
  // instead of ~30 vector instructions per output pair, parity-green, 511 -> 516 us: the second code path cost 176 bytes of
  // scratch per lane and the kernel is not bound by this epilogue's instruction count alone; profiles/r05_kernels.md section 2.)
  static constexpr int LDS_FLOATS = R2_WAVES * 256;
  bfraw* C;
  const bfraw* H;  // same pitch as C
  double* sums;    // [slots][Nc][2]
  const float *s, *t, *mean, *rstd;
  int ldc, Nc;
  struct State {
    double d1[R2_NT], d2[R2_NT];
    float e1[R2_SIDE], e2[R2_SIDE];
    float cs[R2_NT], ct[R2_NT], cm[R2_NT], cr[R2_NT];  // the lane's per-column constants, fetched once per block
  };
  __device__ __forceinline__ void init(State& q, const R2Ctx& c) const {
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { q.d1[j] = 0.0; q.d2[j] = 0.0; }
#pragma unroll
    for (int u = 0; u < R2_SIDE; ++u) { q.e1[u] = 0.f; q.e2[u] = 0.f; }
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int col = c.c0 + j * 32 + (c.lane & 31);
      const bool cok = col < Nc;
      const int cc = cok ? col : 0;
      q.cs[j] = cok ? s[cc] : 0.f; q.ct[j] = cok ? t[cc] : 0.f; q.cm[j] = cok ? mean[cc] : 0.f; q.cr[j] = cok ? rstd[cc] : 0.f;
    }
  }
  __device__ __forceinline__ void tile(State& q, const f32x16 (&acc)[R2_NT], const float (&side)[R2_SIDE], const R2Ctx& c,
                                       const R2Geo& geo, float* red) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const bool odd = c.lane & 1;
    const float (&cs)[R2_NT] = q.cs;
    const float (&ct)[R2_NT] = q.ct;
    const float (&cm)[R2_NT] = q.cm;
    const float (&cr)[R2_NT] = q.cr;
    float p1[R2_NT], p2[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) { p1[j] = 0.f; p2[j] = 0.f; }
    {
      unsigned* tb = reinterpret_cast<unsigned*>(red) + c.wave * 256;  // [16 rows][16 words = 32 columns]
      // the rows this lane moves: row 16 G + (lane >> 2) of the tile, columns 8 (lane & 3) .. + 7 of each 32-column piece
      size_t og[2];
      bool okg[2];
#pragma unroll
      for (int G = 0; G < 2; ++G) {
        int b, n; long r;
        geo.row(c.bg, c.vt, c.wave, 16 * G + (c.lane >> 2), b, n, r, okg[G]);
        og[G] = (size_t)r * ldc + c.c0 + (c.lane & 3) * 8;
      }
      u32x4 hq[R2_NT][2];  // all eight pieces of H requested up front
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const bool cokp = c.c0 + j * 32 + (c.lane & 3) * 8 < ldc;
#pragma unroll
        for (int G = 0; G < 2; ++G) {
          hq[j][G] = u32x4{0u, 0u, 0u, 0u};
          if (okg[G] && cokp) hq[j][G] = *reinterpret_cast<const u32x4*>(H + og[G] + j * 32);
        }
      }
      bool ok0[8], ok1[8];  // which rows of the tile this lane's registers hold
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        int b, n; long r0;
        geo.row(c.bg, c.vt, c.wave, acc_row(2 * p, c.lane), b, n, r0, ok0[p]);
        ok1[p] = ok0[p] && n + 1 < geo.N;
      }
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const int cl = c.c0 + j * 32 + li;
        const bool cok = cl < Nc;
        const bool cokp = c.c0 + j * 32 + (c.lane & 3) * 8 < ldc;
#pragma unroll
        for (int G = 0; G < 2; ++G) {
          *reinterpret_cast<u32x4*>(tb + (c.lane >> 2) * 16 + (c.lane & 3) * 4) = hq[j][G];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int p = 4 * G + qq;
            // even lane: row 2p, columns (li, li + 1); odd lane: row 2p + 1, columns (li - 1, li)
            const int rr = ((2 * qq) & 3) + 8 * ((2 * qq) >> 2) + 4 * h + (odd ? 1 : 0);
            float h0, h1;
            pair_unexchange(tb[rr * 16 + (li >> 1)], odd, h0, h1);
            const float g0 = (cok && ok0[p] && __fmaf_rn(cs[j], h0, ct[j]) > 0.f) ? acc[j][2 * p] : 0.f;
            const float g1 = (cok && ok1[p] && __fmaf_rn(cs[j], h1, ct[j]) > 0.f) ? acc[j][2 * p + 1] : 0.f;
            const unsigned pk = pack_bf16(g0, g1);
            const float v0 = bf_lo(pk), v1 = bf_hi(pk);
            p1[j] += v0 + v1;
            p2[j] = __fmaf_rn(v0, (h0 - cm[j]) * cr[j], __fmaf_rn(v1, (h1 - cm[j]) * cr[j], p2[j]));
            tb[rr * 16 + (li >> 1)] = pair_exchange(pk, odd);
          }
          const u32x4 v = *reinterpret_cast<const u32x4*>(tb + (c.lane >> 2) * 16 + (c.lane & 3) * 4);
          if (okg[G] && cokp) *reinterpret_cast<u32x4*>(C + og[G] + j * 32) = v;
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

// dA(gy1) without gy1 (geometry mode 1).  gy1 = acc * (y1 > 0), y1 = gamma * (Gx[n] + Fx[b]) + beta; layer 1 only needs
//   P[b,c] = sum_n gy1   and   Q[n,c] = sum_b gy1.
// Accumulator register r of lane (li, h): vertex v = r & 3, sample h + 2 (r >> 2) of the wave's eight.
//   P: the sum over a tile's 4 vertices is in-lane (r & 3); it keeps accumulating IN REGISTERS over all vertex tiles of the
//      block's range and leaves once per block:  Pp[slot of the sample group][b][c].
//   Q: the sum over the wave's 8 samples is in-lane (r >> 2) plus one exchange with the other lane half; the 8 waves of the
//      block (= all 64 samples of the group) meet in a double-buffered LDS array, one barrier per tile, and 512 threads add the
//      eight partials in wave order:  Qp[sample group][n][c].
// Leftover (side) columns of the last column group: one row per lane; their vertex / sample sums go through lane exchanges.
struct EpiL1B2 {
  float *Pp, *Qp;  // [spb][B][ld], [nbg][N][ld]
  // the PRE-SCALED layer-1 factors of the forward pass (l1_fill_kernel / prep_kernel): the mask is the forward's own relu argument,
  // Gy[n] + Fy[b] > 0.  A block's 64 samples and its columns are fixed for its whole life, so its Fy values sit in registers
  // (16 per lane) and a tile only fetches the 4 x 4 Gy values of its vertices: 16 load instructions per tile instead of 40.
  // Vertices beyond N read Gy's sentinel row (-3e38) and samples beyond B cache -3e38: no selects.
  const float *Gy, *Fy;
  int ld, Nc;
  static constexpr int LDS_FLOATS = 2 * R2_WAVES * (R2_NT * 4 * 32 + R2_SIDE * 4);  // two buffers of per-wave Q partials
  struct State {
    float p[4][R2_NT];   // samples h, h+2, h+4, h+6 of the wave x column tiles
    float fy[4][R2_NT];  // Fy of those samples at the lane's columns
    float ps[R2_SIDE];   // side columns: lanes (li % 4 == 0, h == 0) hold sample li >> 2
    int parity, ready;
  };
  __device__ __forceinline__ void init(State& s, const R2Ctx&) const {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) s.p[m][j] = 0.f;
#pragma unroll
    for (int t = 0; t < R2_SIDE; ++t) s.ps[t] = 0.f;
    s.parity = 0;
    s.ready = 0;
  }
  __device__ __forceinline__ void tile(State& s, const f32x16 (&acc)[R2_NT], const float (&side)[R2_SIDE], const R2Ctx& c, const R2Geo& geo,
                                       float* red) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    float* buf = red + (size_t)s.parity * (LDS_FLOATS / 2);
    const int b0 = c.bg * 64 + c.wave * 8, n0 = c.vt * 4;
    if (!s.ready) {  // first tile of the block (wave-uniform)
      s.ready = 1;
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const int cl = c.c0 + j * 32 + li, cc = cl < ld ? cl : ld - 1;
#pragma unroll
        for (int m = 0; m < 4; ++m) s.fy[m][j] = b0 + h + 2 * m < geo.B ? Fy[(size_t)(b0 + h + 2 * m) * ld + cc] : -3.0e38f;
      }
    }
    float q[R2_NT][4];
    float gy[R2_NT][4];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
      const int cl = c.c0 + j * 32 + li, cc = cl < ld ? cl : ld - 1;
#pragma unroll
      for (int v = 0; v < 4; ++v) gy[j][v] = Gy[(size_t)(n0 + v < geo.N ? n0 + v : geo.N) * ld + cc];
    }
#pragma unroll
    for (int j = 0; j < R2_NT; ++j) {
#pragma unroll
      for (int v = 0; v < 4; ++v) q[j][v] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int v = r & 3, m = r >> 2;
        const float g = gy[j][v] + s.fy[m][j] > 0.f ? acc[j][r] : 0.f;  // columns beyond Nc: acc is exactly zero (zero weights)
        q[j][v] += g;
        s.p[m][j] += g;
      }
#pragma unroll
      for (int v = 0; v < 4; ++v) q[j][v] += __shfl_xor(q[j][v], 32, 64);  // the other four samples of the wave
      if (h == 0) {
#pragma unroll
        for (int v = 0; v < 4; ++v) buf[((c.wave * R2_NT + j) * 4 + v) * 32 + li] = q[j][v];
      }
    }
    float* sbuf = buf + R2_WAVES * R2_NT * 4 * 32;  // [wave][side column][vertex]
    if (c.nside) {
      const int v = li & 3, sm = li >> 2;  // this lane's row: vertex v of sample b0 + sm
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) {
        if (t < c.nside) {
          const int col = c.c0 + R2_COLS + t;
          const bool live = n0 + v < geo.N && b0 + sm < geo.B;
          const float y = live ? Gy[(size_t)(n0 + v) * ld + col] + Fy[(size_t)(b0 + sm) * ld + col] : 0.f;
          const float g = (live && y > 0.f) ? side[t] : 0.f;
          float pv = g + __shfl_xor(g, 1, 64);  // over the sample's 4 vertices
          pv += __shfl_xor(pv, 2, 64);
          s.ps[t] += pv;
          float qs = g + __shfl_xor(g, 4, 64);  // over the wave's 8 samples
          qs += __shfl_xor(qs, 8, 64);
          qs += __shfl_xor(qs, 16, 64);
          if (c.lane < 4) sbuf[(c.wave * R2_SIDE + t) * 4 + c.lane] = qs;
        }
      }
    }
    __syncthreads();  // one barrier per tile: the buffer written two tiles ago was consumed before the previous barrier
    {
      const int o = c.wave * 64 + c.lane;  // 512 outputs: (column tile, vertex, lane)
      const int j = o >> 7, v = (o >> 5) & 3, l = o & 31;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < R2_WAVES; ++w) a += buf[((w * R2_NT + j) * 4 + v) * 32 + l];
      const int col = c.c0 + j * 32 + l, n = n0 + v;
      if (col < Nc && col < c.c0 + R2_COLS && n < geo.N) Qp[((size_t)c.bg * geo.N + n) * ld + col] = a;
      if (o < c.nside * 4) {
        const int t = o >> 2, vv = o & 3;
        float e = 0.f;
#pragma unroll
        for (int w = 0; w < R2_WAVES; ++w) e += sbuf[(w * R2_SIDE + t) * 4 + vv];
        if (n0 + vv < geo.N) Qp[((size_t)c.bg * geo.N + n0 + vv) * ld + c.c0 + R2_COLS + t] = e;
      }
    }
    s.parity ^= 1;
  }
  __device__ __forceinline__ void flush(State& s, const R2Ctx& c, const R2Geo& geo, char*) const {
    const int li = c.lane & 31, h = c.lane >> 5;
    const int sq = c.slot % geo.spb;  // the block's slot inside its sample group = its partial's index
    const int b0 = c.bg * 64 + c.wave * 8;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int b = b0 + h + 2 * m;
#pragma unroll
      for (int j = 0; j < R2_NT; ++j) {
        const int col = c.c0 + j * 32 + li;
        if (b < geo.B && col < Nc) Pp[((size_t)sq * geo.B + b) * ld + col] = s.p[m][j];
      }
    }
    if (c.nside && h == 0 && (li & 3) == 0 && b0 + (li >> 2) < geo.B) {
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t)
        if (t < c.nside) Pp[((size_t)sq * geo.B + b0 + (li >> 2)) * ld + c.c0 + R2_COLS + t] = s.ps[t];
    }
  }
};

// ------------------------------------------------------------------------------------------------ the kernel
// grid = ngroups * slots blocks (1-D, XCD-aware virtual ids: the column groups of one slot are neighbours on one XCD - they
// stream the same activation rows).  Dynamic LDS: weight slice [(R2_COLS + geo.wside)][Kp + 8] bf16, then the generator's
// per-channel constants [AOp::NC][Kp] fp32.
// (Round 5, measured and dropped: a PING-PONG k loop - waves 0-3 and 4-7 held half a k-step apart by raw s_barriers so that one
// group's MFMAs run beside the other group's fragment reads / transform / requests - was 30 - 40 % SLOWER on all four kernels
// (h2 539 -> 726 us, dA 611 -> 872 us): these kernels are bound by VALU ISSUE, not by idle matrix pipes waiting for operands, and
// a barrier interval costs the slower of the two phases; profiles/r05_kernels.md section 2.)
// ABL != 0: measurement-only variants, instantiated only with -DOBMAN_ABLATION (tools/ablate_gemm.sh build, then tools/archive/r03/r03_abl.sh /
// tools/archive/r03/r03_dbg.sh with OBMAN_R2_ABL; wrong results): 1 no operand requests inside the k loop,
// 2 weight fragments read once, 3 no epilogue, 4 no MFMAs, 5 = 2 + the generator's LDS constants read once
// ABL == 8: s_memtime stamps inside the k-step of every wave (LDS phase / transform + operand wait / MFMA issue / rest / epilogue),
// summed per wave into r2_dbg[block][wave][8] - printed by launch_rows2 (measurement only; the stamps serialise the phases)
#ifdef OBMAN_ABLATION
__device__ unsigned long long r2_dbg[1024 * R2_WAVES * 8];
#else
__device__ unsigned long long* const r2_dbg = nullptr;  // the ABL == 8 variant is never instantiated in the product build
#endif
template <class AOp, class Epi, int ABL = 0>
__global__ __launch_bounds__(R2_THREADS) void rows2_bf16_kernel(AOp aop, const bfraw* __restrict__ Wb, int Kp, int Nc, Epi epi, R2Geo geo,
                                                                 int lds_aop_floats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KP2 = Kp + 8;
  bfraw* Ws = reinterpret_cast<bfraw*>(smem);
  float* kcs = reinterpret_cast<float*>(Ws + (size_t)(R2_COLS + geo.wside) * KP2);
  float* red = kcs + lds_aop_floats;        // Epi::LDS_FLOATS floats of epilogue scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 31, h = lane >> 5;
  const int vid = xcd_virtual_id(blockIdx.x, gridDim.x), cg = vid % geo.ngroups, slot = vid / geo.ngroups;
  const int c0 = cg * R2_COLS;
  const int last_group = cg == geo.ngroups - 1;
  const int gcols = last_group ? Nc - c0 : R2_COLS;  // the last group holds the remainder: up to R2_COLS + R2_SIDE columns
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

  R2Src<AOp> src;
  src.init(aop, geo);
  R2Ctx ctx{lane, wave, c0, nside, last_group, slot, 0, 0};
  typename Epi::State est;
  epi.init(est, ctx);
  const int bg = slot / geo.spb, sq = slot - bg * geo.spb;
  const int vt_beg = sq * geo.chunk, vt_end = vt_beg + geo.chunk < geo.nvt ? vt_beg + geo.chunk : geo.nvt;
  const int nks = Kp >> 4;
  const bfraw* wlane = Ws + (size_t)li * KP2 + h * 8;
  ctx.bg = bg;
  unsigned long long dbg_l = 0, dbg_v = 0, dbg_m = 0, dbg_g = 0, dbg_e = 0, dbg_steps = 0, dbg_last = 0;
  const unsigned long long dbg_t0 = ABL == 8 ? __builtin_readcyclecounter() : 0;
  for (int vt = vt_beg; vt < vt_end; ++vt) {
    ctx.vt = vt;
    typename AOp::Row row;
    typename R2Src<AOp>::Off roff;
    bool ok;
    {
      int b, n; long r;
      geo.row(bg, vt, wave, li, b, n, r, ok);
      if constexpr (R2Sentinel<AOp>::value) { if (!ok) n = geo.N; }
      row = aop.row(r, b, n, ok);
      if constexpr (R2SelfMask<AOp>::value) roff = src.off_masked(aop, r, h, ok);
      else roff = src.off(aop, r, b, n, h);
    }
#ifdef OBMAN_ABLATION
    if constexpr (ABL == 9 && std::is_same<AOp, BPlain>::value) {
      // measurement only (wrong results): the A operand AS IF stored fragment-major - a wave-level load = one contiguous 1 KB
      roff.o = (unsigned)(((size_t)(bg * geo.nvt + vt) * R2_WAVES + wave) * (size_t)nks * 1024u) + (unsigned)lane * 16u;
    }
#endif
    f32x16 acc[R2_NT];
#pragma unroll
    for (int j = 0; j < R2_NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float side[R2_SIDE];
#pragma unroll
    for (int t = 0; t < R2_SIDE; ++t) side[t] = 0.f;

    // One k-step = one 16-deep MFMA per column tile.  Raw operand chunks are requested DQ k-steps ahead into a register queue
    // with compile-time slots (the loop is unrolled by DQ); requests past the last step are re-reads that are never consumed, so
    // the loop body is branch-free and the compiler's counted vmcnt waits leave the younger requests in flight.
    // (Measured and dropped: software-pipelining the loop inside the wave - next step's fragment reads and transform between
    // this step's MFMAs via sched_group_barrier - was 4 .. 10 % SLOWER: the SIMD's other wave already fills those slots, and
    // the second fragment set costs registers the request queue needs.)
    constexpr int DQ = R2Depth<AOp>::value;
    typename AOp::Raw q[DQ];
#pragma unroll
    for (int u = 0; u < DQ; ++u) src.load(q[u], roff, ABL == 9 ? u * 32 : u);
    auto step = [&](typename AOp::Raw& qs, int s) {
      // the four weight fragments of this k-step first: the transform below covers their LDS latency (left to itself the
      // compiler sank each read next to its MFMA: read -> wait -> MFMA, four exposed LDS round trips per k-step)
      bf16x8 fb[R2_NT];
#pragma unroll
      for (int j = 0; j < R2_NT; ++j)
        fb[j] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + ((ABL == 2 || ABL == 5) ? 0 : s * 16));
      __builtin_amdgcn_sched_barrier(0);
      u32x4 a0;
      const int kf = (ABL == 5 ? 0 : s * 16) + h * 8;
      if constexpr (R2Packed<R2Fin<AOp>>::value) {
        a0 = R2Fin<AOp>::finp(aop, row, kcs, Kp, kf, qs);
      } else {
        float o0[8];
        R2Fin<AOp>::fin(aop, row, kcs, Kp, kf, qs, o0);
        a0 = pack8(o0);
      }
      if constexpr (ABL != 1) src.load(qs, roff, ABL == 9 ? (s + DQ) * 32 : s + DQ);
      if constexpr (!R2Sentinel<AOp>::value && !R2SelfMask<AOp>::value) { if (!ok) a0 = u32x4{0u, 0u, 0u, 0u}; }
      const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0);
      if constexpr (ABL == 4) {
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) {
          const u32x4 fw = __builtin_bit_cast(u32x4, fb[j]);
          asm volatile("" ::"v"(fw.x), "v"(fw.y), "v"(fw.z), "v"(fw.w), "v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
        }
      } else {
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[j], acc[j], 0, 0, 0);
      }
      if (nside) {  // leftover columns of the last group on the VALU, from the SAME rounded operands the MFMAs consume
#pragma unroll
        for (int t = 0; t < R2_SIDE; ++t) {
          if (t < nside) {
            const u32x4 wv = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + s * 16 + h * 8);
            // v_dot2c_f32_bf16: two bf16 products accumulated in fp32 per instruction (4 per side column and k-step instead of
            // 8 unpacks + 8 FMAs).  Inline asm: with __builtin_amdgcn_fdot2_f32_bf16 on components of the two u32x4 values this
            // compiler (ROCm 7.2 clang) emitted all four instructions with the FIRST component's registers.
            float s0 = side[t];
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.x), "v"(wv.x));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.y), "v"(wv.y));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.z), "v"(wv.z));
            asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.w), "v"(wv.w));
            side[t] = s0;
          }
        }
      }
    };
    int s = 0;
    if constexpr (ABL == 8) {
      typedef typename R2Fin<AOp>::Pref Pref;
      for (; s < nks; ++s) {
        const unsigned long long T0 = __builtin_readcyclecounter();
        bf16x8 fb[R2_NT];
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + s * 16);
        const Pref pc = R2Fin<AOp>::pre(aop, row, kcs, Kp, s * 16 + h * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long T1 = __builtin_readcyclecounter();
        typename AOp::Raw& qs = q[0];
        u32x4 a0 = R2Fin<AOp>::finq(pc, qs);
        asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w));
        const unsigned long long T2 = __builtin_readcyclecounter();
        // rotate the queue by hand (runtime s): q[0] <- q[1] ... and request step s + DQ into the last slot
#pragma unroll
        for (int u = 0; u + 1 < DQ; ++u) q[u] = q[u + 1];
        src.load(q[DQ - 1], roff, s + DQ);
        const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb[j], acc[j], 0, 0, 0);
        const unsigned long long T3 = __builtin_readcyclecounter();
        dbg_l += T1 - T0; dbg_v += T2 - T1; dbg_m += T3 - T2; dbg_steps += 1;
        if (dbg_last) dbg_g += T0 - dbg_last;
        dbg_last = T3;
      }
    } else if constexpr (ABL == 6) {
      // Pipelined inside the wave: the LDS reads of k-step s + 1 (weight fragments, generator constants, side-column weights) are
      // issued BEFORE the transform and the MFMAs of step s into the other of two register sets (compile-time parity: DQ is even),
      // so a wave never sits on an LDS round trip with an idle matrix pipe.
      typedef typename R2Fin<AOp>::Pref Pref;
      struct Stage { bf16x8 fb[R2_NT]; Pref c; u32x4 wv[R2_SIDE]; };
      auto fetch = [&](Stage& st, int ks) {
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) st.fb[j] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)j * 32 * KP2 + ks * 16);
        st.c = R2Fin<AOp>::pre(aop, row, kcs, Kp, ks * 16 + h * 8);
        if (nside) {
#pragma unroll
          for (int t = 0; t < R2_SIDE; ++t)
            if (t < nside) st.wv[t] = *reinterpret_cast<const u32x4*>(Ws + (size_t)(R2_COLS + t) * KP2 + ks * 16 + h * 8);
        }
      };
      auto stepp = [&](typename AOp::Raw& qs, int ks, Stage& cur, Stage& nxt) {
        fetch(nxt, ks + 1 < nks ? ks + 1 : ks);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 a0 = R2Fin<AOp>::finq(cur.c, qs);
        src.load(qs, roff, ks + DQ);
        if constexpr (!R2Sentinel<AOp>::value && !R2SelfMask<AOp>::value) { if (!ok) a0 = u32x4{0u, 0u, 0u, 0u}; }
        const bf16x8 fa0 = __builtin_bit_cast(bf16x8, a0);
#pragma unroll
        for (int j = 0; j < R2_NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, cur.fb[j], acc[j], 0, 0, 0);
        if (nside) {
#pragma unroll
          for (int t = 0; t < R2_SIDE; ++t) {
            if (t < nside) {
              float s0 = side[t];
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.x), "v"(cur.wv[t].x));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.y), "v"(cur.wv[t].y));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.z), "v"(cur.wv[t].z));
              asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s0) : "v"(a0.w), "v"(cur.wv[t].w));
              side[t] = s0;
            }
          }
        }
      };
      static_assert(DQ % 2 == 0, "the two register sets alternate with compile-time parity");
      Stage sa, sb;
      fetch(sa, 0);
      for (; s + DQ <= nks; s += DQ) {
#pragma unroll
        for (int u = 0; u < DQ; ++u) {
          if (u & 1) stepp(q[u], s + u, sb, sa);
          else stepp(q[u], s + u, sa, sb);
        }
      }
#pragma unroll
      for (int u = 0; u < DQ; ++u)
        if (s + u < nks) {
          if (u & 1) stepp(q[u], s + u, sb, sa);
          else stepp(q[u], s + u, sa, sb);
        }
    } else {
    for (; s + DQ <= nks; s += DQ) {
#pragma unroll
      for (int u = 0; u < DQ; ++u) step(q[u], s + u);
    }
#pragma unroll
    for (int u = 0; u < DQ; ++u)
      if (s + u < nks) step(q[u], s + u);
    }
    if (nside) {  // the two lane halves covered different k: combine
#pragma unroll
      for (int t = 0; t < R2_SIDE; ++t) side[t] += __shfl_xor(side[t], 32, 64);
    }
    if constexpr (ABL == 3) {
#pragma unroll
      for (int j = 0; j < R2_NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[j][r]));
    } else if constexpr (ABL == 8) {
      const unsigned long long E0 = __builtin_readcyclecounter();
      epi.tile(est, acc, side, ctx, geo, red);
      const unsigned long long E1 = __builtin_readcyclecounter();
      dbg_e += E1 - E0;
      dbg_last = 0;
    } else {
      epi.tile(est, acc, side, ctx, geo, red);
    }
  }
  if constexpr (ABL == 8) {
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long* o = r2_dbg + ((size_t)blockIdx.x * R2_WAVES + wave) * 8;
      o[0] = dbg_l; o[1] = dbg_v; o[2] = dbg_m; o[3] = dbg_g; o[4] = dbg_e; o[5] = dbg_steps;
      o[6] = __builtin_readcyclecounter() - dbg_t0; o[7] = (unsigned long long)(vt_end - vt_beg);
    }
  }
  epi.flush(est, ctx, geo, smem);
}
