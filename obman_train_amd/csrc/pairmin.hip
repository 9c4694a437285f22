// K1/K2/K3 - brute-force bidirectional pair-min (Chamfer / hand<->object closest vertex), gfx950.
//
// Replaces the reference's materialised N x M matrix (3 bmm + diag gather + broadcast add + two
// torch.min, atlasutils.py:11-39 / contactloss.py:60-79,164-166) by one fused sweep:
//   * each lane keeps QPT query points in VGPRs, the reference set is staged once per block in LDS
//     as float4 and read back with broadcast ds_read_b128 (all lanes same address: conflict-free),
//   * distances in the direct-difference form (3 sub, 1 mul, 2 fma - no cancellation),
//   * arg-min tracked per 4-reference group (1 compare + 2 selects per 4 pairs instead of per pair);
//     the exact index inside the winning group is recovered afterwards by re-evaluating 4 distances
//     with the same pinned instruction sequence (bit-identical), first index wins ties,
//   * both directions run in ONE launch (blockIdx.z), the role of query/reference swapped,
//   * long reference sets with few queries are split over blocks (rsplit) and merged with a 64-bit
//     atomicMin on (dist_bits << 32 | idx): order independent => deterministic, ties -> lowest idx.
// Bound: fp32 VALU (~7.5 lane-ops/pair); algorithmic HBM bytes 20*(Nx+Ny) per sample.
#include <cstdlib>

#include "common.h"
#include "prof.h"
#include "../../include/obman_hip.h"

namespace {

constexpr int PM_THREADS = 256;
constexpr int PM_REF_TILE = 2048;  // float4 -> 32 KiB LDS per block
constexpr int PM_CHUNK = 8;        // references per arg-min chunk of the general kernel
constexpr float PM_BIG = 1.0e18f;  // padding coordinate: d = 3e36 < FLT_MAX, never wins
typedef unsigned long long u64;

typedef float f2 __attribute__((ext_vector_type(2)));

// Two queries per instruction: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32.  Element-wise identical rounding to
// obman_dist2 (sub, mul, fma, fma), so the scalar index-resolution pass reproduces these values bit for bit.
__device__ __forceinline__ f2 pm_dist2_pk(f2 qx, f2 qy, f2 qz, float rx, float ry, float rz) {
  const f2 dx = qx - rx, dy = qy - ry, dz = qz - rz;
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}

struct PmDir {
  const float* q;  // queries  [B,nq,3]
  const float* r;  // references [B,nr,3]
  float* omin;     // [B,nq] or null (direction skipped)
  int* oidx;       // [B,nq] or null
  u64* ws;         // packed scratch when rsplit > 1
  int nq, nr, qtiles, rsplit, rchunk, tile;
  u64* rs_ws;      // round 6 (fused sweep): [B,nr] keys (distance bits << 32 | query tile) of the REFERENCE-side minima, or null
};

// Block = 4 waves.  All 4 waves hold the SAME 64*QPT queries (lane l owns queries l, l+64, ..) and each
// wave sweeps its own quarter of the staged reference tile, so a query's serial chain is 4x shorter and
// 4x more waves are in flight to hide the LDS latency (the r01a kernel ran ~1.2 waves/SIMD and was
// latency-bound).  The four partial (min, group) pairs are merged through LDS lexicographically
// (value, then lower group start => first index), then 64*QPT lanes resolve the exact index.
struct PmGrid { int gx, gz, B, xcd; };  // query tiles (x reference splits) per direction, directions in this launch, samples

// Wave-wide minima of EIGHT values at once, valid in lane 63: four DPP steps inside the 16-lane rows, then row_bcast15 / row_bcast31
// across the rows (rows outside the row mask keep their value).  One v_min_f32_dpp per value and step, as inline asm: through
// __builtin_amdgcn_update_dpp + fminf hipcc emitted a copy, a v_mov_b32_dpp and a v_min_f32 per step (144 instead of 48 instructions
// per chunk of 8 references).  The eight chains are interleaved, so a DPP read is 8 instructions behind the write it depends on
// (the 2 wait states a VALU write -> DPP read needs); the s_nop covers the compiler's own last write before the block.
#define PM_DPP8(ctrl)                                  \
  "v_min_f32_dpp %0, %0, %0 " ctrl "\n\t"              \
  "v_min_f32_dpp %1, %1, %1 " ctrl "\n\t"              \
  "v_min_f32_dpp %2, %2, %2 " ctrl "\n\t"              \
  "v_min_f32_dpp %3, %3, %3 " ctrl "\n\t"              \
  "v_min_f32_dpp %4, %4, %4 " ctrl "\n\t"              \
  "v_min_f32_dpp %5, %5, %5 " ctrl "\n\t"              \
  "v_min_f32_dpp %6, %6, %6 " ctrl "\n\t"              \
  "v_min_f32_dpp %7, %7, %7 " ctrl "\n\t"
__device__ __forceinline__ void pm_wave_min8_lane63(float (&v)[8]) {
  asm volatile("s_nop 1\n\t"
               PM_DPP8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
               PM_DPP8("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
               PM_DPP8("row_half_mirror row_mask:0xf bank_mask:0xf")
               PM_DPP8("row_mirror row_mask:0xf bank_mask:0xf")
               PM_DPP8("row_bcast:15 row_mask:0xa bank_mask:0xf")
               PM_DPP8("row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}
#undef PM_DPP8

// RS (round 6, VERDICT r05 task 4: "evaluate every pair once"): the sweep ALSO produces the reference-side minima from the same
// distances.  A ChamferLoss at 16 050 x 600 used to be two independent all-pairs sweeps (predicted -> ground truth with the
// predictions as queries, ground truth -> predicted with the roles swapped and the long reference set split over blocks): every
// pair evaluated twice.  With RS the long side is the query side only: per chunk of 8 references a lane folds its QPT distances to
// each reference with v_min3 (QPT / 2 instructions per reference), one DPP wave-min per reference leaves the minimum over the
// block's 64 QPT queries in lane 63, which parks it in LDS (a reference belongs to ONE wave of the block); at the end of the block
// the references' minima go to global memory as 64-bit atomicMin keys (distance bits << 32 | query tile): order-independent, and
// the lowest tile among equal minima is the one that holds the first index.  pairmin_resolve_kernel then re-evaluates the winning
// tile's queries with the same pinned instruction sequence and takes the first equal one - values, arg-mins and tie rule are bit-identical to the
// swapped-role sweep (tests/test_pairmin_gpu.py keeps that path as the checker: OBMAN_PM_FUSED=0).
//
// QS = false (RS only; the contact term's hand -> object direction, 778 x 16 050 / 64 050): ONLY the reference-side minima are
// wanted and the wanted side is the short one.  The same sweep with the long side in the registers, without the per-query
// bookkeeping (chunk minimum, compare, two selects per query and chunk) and without its merge buffers in LDS; before, this
// direction ran as pairmin_fwd_kernel<2> with the 16 050 references split over three blocks and merged through 64-bit atomics.
template <int QPT, bool RS, bool QS = true>
__global__ __launch_bounds__(PM_THREADS) void pairmin_fwd_kernel(PmDir d0, PmDir d1, PmGrid pg) {
  static_assert(RS || QS, "a sweep without outputs");
  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  All blocks
  // of one sample (every query tile, both directions: they read the same two point sets) take consecutive slots on ONE XCD,
  // so a sample's points are fetched from HBM by one L2 instead of by all eight (profiles/r01_chamfer_pmc.md: 3.5x the
  // algorithmic bytes at 642 x 600 before).
  int b, tile, dir;
  if (pg.xcd) {
    const int id = blockIdx.x, members = pg.gx * pg.gz, slot = id >> 3, m = slot % members;
    b = (slot / members) * 8 + (id & 7);
    dir = m / pg.gx;
    tile = m - dir * pg.gx;
  } else {
    const int id = blockIdx.x, m = id % (pg.gx * pg.gz);
    b = id / (pg.gx * pg.gz);
    dir = m / pg.gx;
    tile = m - dir * pg.gx;
  }
  if (b >= pg.B) return;
  const PmDir d = dir == 0 ? d0 : d1;
  if constexpr (QS) {
    if (d.omin == nullptr) return;
  }
  if (tile >= d.qtiles * d.rsplit) return;
  const int qt = tile % d.qtiles, rs = tile / d.qtiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ qb = d.q + (size_t)b * d.nq * 3;
  const float* __restrict__ rb = d.r + (size_t)b * d.nr * 3;

  // dynamic LDS sized to the reference tile actually used (642 refs -> 10 KiB: 8 blocks / 32 waves per CU)
  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  float4* sref = reinterpret_cast<float4*>(pm_smem);
  float(*s_val)[64 * QPT] = reinterpret_cast<float(*)[64 * QPT]>(pm_smem + (size_t)d.tile * sizeof(float4));
  int(*s_grp)[64 * QPT] = reinterpret_cast<int(*)[64 * QPT]>(pm_smem + (size_t)d.tile * sizeof(float4) + 4 * 64 * QPT * sizeof(float));
  float* s_rmin = reinterpret_cast<float*>(pm_smem + (size_t)d.tile * sizeof(float4) + (QS ? 8 * 64 * QPT * sizeof(float) : 0));  // RS: [tile] reference-side minima

  static_assert(QPT % 2 == 0, "queries are processed as packed pairs");
  constexpr int QP = QPT / 2;
  f2 qx[QP], qy[QP], qz[QP];
  float best[QPT];
  int bestj[QPT];
#pragma unroll
  for (int k = 0; k < QPT; ++k) {
    const int qi = qt * (64 * QPT) + k * 64 + lane;
    const int qc = qi < d.nq ? qi : d.nq - 1;  // clamp: idle lanes redo the last point, never stored
    qx[k >> 1][k & 1] = qb[(size_t)qc * 3 + 0];
    qy[k >> 1][k & 1] = qb[(size_t)qc * 3 + 1];
    qz[k >> 1][k & 1] = qb[(size_t)qc * 3 + 2];
    best[k] = __builtin_inff();
    bestj[k] = 0x7fffffff;
  }

  const int rbeg = rs * d.rchunk;
  const int rend = min(d.nr, rbeg + d.rchunk);
  const bool single = rend - rbeg <= d.tile;  // whole reference range stays in LDS: resolve indices from it
  for (int base = rbeg; base < rend; base += d.tile) {
    const int cnt = min(d.tile, rend - base);
    const int slice = ((cnt + 31) >> 5) << 3;  // references per wave, multiple of PM_CHUNK
    const int padded = slice * 4;
    for (int i = tid; i < padded; i += PM_THREADS) {
      float4 v = make_float4(PM_BIG, PM_BIG, PM_BIG, 0.f);
      if (i < cnt) {
        const float* p = rb + (size_t)(base + i) * 3;
        v = make_float4(p[0], p[1], p[2], 0.f);
      }
      sref[i] = v;
    }
    __syncthreads();
    const int jbeg = wave * slice, jend = jbeg + slice;
    // One iteration = a chunk of PM_CHUNK = 8 references, all eight LDS reads issued first (the compiler refused to unroll the
    // 4-reference form: "loop not unrolled" x 3 in r03 - one exposed LDS round trip per 4 references).  Chunk minimum with
    // v_min3 (3 + 1 instructions for 8 values instead of 2 x 3), ONE compare + two selects per chunk and query; the exact index
    // inside the winning chunk is resolved afterwards (first index wins ties, as before).
    for (int j = jbeg; j < jend; j += PM_CHUNK) {
      float4 r[PM_CHUNK];
#pragma unroll
      for (int u = 0; u < PM_CHUNK; ++u) r[u] = sref[j + u];
#pragma unroll
      for (int u = 0; u < PM_CHUNK; ++u) asm volatile("" ::"v"(r[u].w));  // keep the reads ds_read_b128 (b96 is 2x the LDS cycles)
      float rm[PM_CHUNK];
      if constexpr (RS) {
#pragma unroll
        for (int u = 0; u < PM_CHUNK; ++u) rm[u] = __builtin_inff();
      }
#pragma unroll
      for (int p = 0; p < QP; ++p) {
        f2 e[PM_CHUNK];
#pragma unroll
        for (int u = 0; u < PM_CHUNK; ++u) e[u] = pm_dist2_pk(qx[p], qy[p], qz[p], r[u].x, r[u].y, r[u].z);
        if constexpr (RS) {
#pragma unroll
          for (int u = 0; u < PM_CHUNK; ++u) rm[u] = __builtin_fminf(__builtin_fminf(rm[u], e[u][0]), e[u][1]);  // v_min3_f32
        }
        if constexpr (QS) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float m0 = __builtin_fminf(__builtin_fminf(e[0][h], e[1][h]), e[2][h]);  // v_min3_f32
          const float m1 = __builtin_fminf(__builtin_fminf(e[3][h], e[4][h]), e[5][h]);
          const float m = __builtin_fminf(__builtin_fminf(m0, m1), __builtin_fminf(e[6][h], e[7][h]));
          const bool better = m < best[2 * p + h];
          best[2 * p + h] = better ? m : best[2 * p + h];
          bestj[2 * p + h] = better ? base + j : bestj[2 * p + h];
        }
        }
      }
      if constexpr (RS) {
        static_assert(PM_CHUNK == 8, "pm_wave_min8_lane63 reduces eight values");
        pm_wave_min8_lane63(rm);
        if (lane == 63) {
#pragma unroll
          for (int u = 0; u < PM_CHUNK; u += 4) *reinterpret_cast<float4*>(s_rmin + j + u) = make_float4(rm[u], rm[u + 1], rm[u + 2], rm[u + 3]);
        }
      }
    }
    if (!single) __syncthreads();
  }
  if constexpr (QS) {
#pragma unroll
  for (int k = 0; k < QPT; ++k) {
    s_val[wave][k * 64 + lane] = best[k];
    s_grp[wave][k * 64 + lane] = bestj[k];
  }
  }
  __syncthreads();  // (RS: also orders the waves' s_rmin stores before the block-wide read below)
  for (int t = tid; QS && t < 64 * QPT; t += PM_THREADS) {
    const int qi = qt * (64 * QPT) + t;
    if (qi >= d.nq) continue;
    float bv = s_val[0][t];
    int bj = s_grp[0][t];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float v = s_val[w][t];
      const int j = s_grp[w][t];
      if (v < bv || (v == bv && j < bj)) { bv = v; bj = j; }
    }
    // exact index inside the winning PM_CHUNK-reference chunk, re-evaluated with the same instruction sequence
    const float x = qb[(size_t)qi * 3], y = qb[(size_t)qi * 3 + 1], z = qb[(size_t)qi * 3 + 2];
    if (bj == 0x7fffffff) bj = rbeg;  // NaN inputs: nothing ever compared smaller
    int idx = bj;
    float e[PM_CHUNK];
#pragma unroll
    for (int u = 0; u < PM_CHUNK; ++u) {
      const int j = min(bj + u, rend - 1);
      if (single) {
        const float4 r = sref[j - rbeg];
        e[u] = obman_dist2(x, y, z, r.x, r.y, r.z);
      } else {
        const float* p = rb + (size_t)j * 3;
        e[u] = obman_dist2(x, y, z, p[0], p[1], p[2]);
      }
    }
#pragma unroll
    for (int u = PM_CHUNK - 1; u >= 0; --u)  // descending so the FIRST matching index survives
      if (bj + u < rend && e[u] == bv) idx = bj + u;
    const size_t o = (size_t)b * d.nq + qi;
    if (d.rsplit == 1) {
      d.omin[o] = bv;
      if (d.oidx) d.oidx[o] = idx;
    } else {
      const u64 packed = ((u64)__float_as_uint(bv) << 32) | (unsigned)idx;
      atomicMin(&d.ws[o], packed);
    }
  }
  if constexpr (RS) {  // single reference tile (the launcher guarantees it): s_rmin holds the block's minimum for every reference
    if (d.rs_ws) {
      const int cnt = rend - rbeg;
      for (int j = tid; j < cnt; j += PM_THREADS) {
        float m = s_rmin[j];
        if (!(m < __builtin_inff())) m = __builtin_inff();  // all-NaN / overflowed column: the swapped-role sweep reports (inf, first index)
        // a key can only go down: a relaxed device-scope look first spares the atomic (and its write towards memory) wherever this
        // block cannot improve the key - all but ~ln(tiles) of a reference's tiles (first version: 16.2 MB written per launch at
        // 64 x 16 050 x 600 for 8.5 MB of results)
        u64* dst = &d.rs_ws[(size_t)b * d.nr + rbeg + j];
        const u64 key = ((u64)__float_as_uint(m) << 32) | (unsigned)qt;
        if (key < __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(dst, key);
      }
    }
  }
}

// Reference-side arg-mins of the fused sweep: one wave per (sample, reference).  key = (minimum bits << 32 | query tile); the
// queries of that tile are re-evaluated with obman_dist2 (the sweep's own rounding) 64 at a time and the first one that equals the
// minimum wins - "first index on ties", as every other path of this file.
__global__ __launch_bounds__(256) void pairmin_resolve_kernel(const u64* __restrict__ keys, const float* __restrict__ q, const float* __restrict__ r,
                                                              int B, int nq, int nr, int qtile, float* __restrict__ omin, int* __restrict__ oidx) {
  // XCD-aware order (as the sweep's): the blocks of one sample take consecutive slots of ONE XCD, so a sample's query points are
  // fetched by one L2 (first version, sample-major linear order: 75.6 MB fetched per launch at 64 x 16 050 x 600 for 12.3 MB of points)
  const int bps = (nr + 3) >> 2;  // blocks per sample
  const int id = blockIdx.x, slot = id >> 3, b = (slot / bps) * 8 + (id & 7), j = (slot % bps) * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B || j >= nr) return;
  const long w = (long)b * nr + j;
  const u64 key = keys[w];
  const float val = __uint_as_float((unsigned)(key >> 32));
  int idx = 0;
  if (val < __builtin_inff()) {
    const int t0 = (int)(unsigned)(key & 0xffffffffu) * qtile, t1 = min(nq, t0 + qtile);
    const float* rp = r + (size_t)w * 3;
    const float rx = rp[0], ry = rp[1], rz = rp[2];
    const float* qb = q + (size_t)b * nq * 3;
    idx = t0;
    for (int i0 = t0; i0 < t1; i0 += 64) {
      const int i = i0 + lane;
      bool hit = false;
      if (i < t1) hit = obman_dist2(qb[(size_t)i * 3], qb[(size_t)i * 3 + 1], qb[(size_t)i * 3 + 2], rx, ry, rz) == val;
      const unsigned long long m = __ballot(hit);
      if (m) { idx = i0 + __builtin_ctzll(m); break; }
    }
  }
  if (lane == 0) {
    omin[w] = val;
    if (oidx) oidx[w] = idx;
  }
}

__global__ __launch_bounds__(256) void pairmin_unpack_kernel(const u64* ws, float* omin, int* oidx, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const u64 p = ws[i];
  omin[i] = __uint_as_float((unsigned)(p >> 32));
  if (oidx) oidx[i] = (int)(unsigned)(p & 0xffffffffu);
}

// loss[b] = mean over n of mins[b,:] - fixed reduction tree (deterministic).  Block = 256 threads, or 1024 where a row is long
// (r06: one 256-thread block walked the 16 050 / 64 050 minima of a sample in 63 / 250 dependent rounds: 18.7 / ~60 us per launch).
__global__ __launch_bounds__(1024) void rowmean2_kernel(const float* a, int na, float* out_a, const float* c, int nc,
                                                        float* out_c) {
  const float* src = blockIdx.y == 0 ? a : c;
  const int n = blockIdx.y == 0 ? na : nc;
  float* dst = blockIdx.y == 0 ? out_a : out_c;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, nw = nt >> 6;
  float s = 0.f;
  for (int i = tid; i < n; i += nt) s += src[(size_t)b * n + i];
  s = obman_wave_sum(s);
  __shared__ float part[16];
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < nw; w += 4) t += (part[w] + part[w + 1]) + (part[w + 2] + part[w + 3]);  // nw = 4 or 16: fixed order
    dst[b] = t / (float)n;
  }
}


// ---- K1 small-set path ("S5"): every reference of a sample fits LDS and the batch alone fills the chip --------------------
// 642 x 600 x 64 samples (BASELINE configs[1]) spent 16 % of its lanes on tile padding (6 + 5 tiles of 128 query slots for
// 642 + 600 queries), three dependent global round trips (query loads, a staging loop that waited per iteration, the query
// re-load of the resolve pass) and two more launches for the per-sample means.  This path:
//   * block = 512 threads = 8 waves over the SAME 320 queries (5 per lane in VGPRs), each wave sweeping an eighth of the staged
//     references: 642 = 2 x 320 + 2 and 600 = 2 x 300, so a sample is 4 equal blocks = exactly one per CU at bs 64;
//   * a remainder of <= S5_LEFT_MAX queries (the 2 of 642) is not padded to another tile: the direction's last block evaluates
//     it TRANSPOSED after its own tile (lanes over references, u64 (distance bits, index) min across the block);
//   * the launch asks for more than half of a CU's LDS, so two blocks never share a CU: 256 equal blocks land on 256 CUs
//     (the first version let the dispatcher double some CUs up: 21 us instead of 13 for the same work);
//   * arg-min bookkeeping per 16-reference chunk instead of per 4-reference group: inside a chunk only the running minimum is
//     kept (two v_min3_f32 per 4 pairs), one compare + two selects per chunk and query; the exact index is recovered by
//     re-evaluating the winning chunk with the same pinned instruction sequence (bit-identical, first index wins);
//   * all staging loads are issued before the first wait, the resolve pass takes its query from registers;
//   * the per-sample means (ChamferLoss: loss_1[b], loss_2[b]) are produced by the LAST block of the sample to finish: every
//     block publishes the fixed-order sum of its minima with a system-scope (write-through) store, waits for the write
//     acknowledgement, then draws a ticket from a per-sample counter with a relaxed agent-scope atomic; the block that draws
//     the last ticket reads the partial sums back with system-scope loads and adds them in member order (deterministic for any
//     dispatch order or XCD placement; no fence, no second launch).  It also resets the counter, so the caller's
//     zero-initialised `sync` buffer stays zero between calls.
constexpr int S5_THREADS = 512, S5_WAVES = 8, S5_QPT = 5, S5_TILE = 64 * S5_QPT, S5_CHUNK_GROUPS = 4;
constexpr int S5_LEFT_MAX = 8, S5_MAX_REFS = 1024;

struct S5Dir {
  const float* q;  // queries [B,nq,3]
  const float* r;  // references [B,nr,3]
  float* omin;     // [B,nq] (null: direction skipped, no members)
  int* oidx;       // [B,nq] or null
  int nq, nr, tiles, left, slice;  // main tiles of 320 queries, leftover queries (transposed role), references per wave (multiple of 4)
};
struct S5Args {
  S5Dir d[2];
  int B, members, m0;   // blocks per sample; of which direction 0's (its tiles)
  int pairwise;         // every direction has <= 2 tiles: partial sums exchanged inside one atomicOr (see the kernel's epilogue)
  float* part;          // [B, members] published partial sums (null: no fused means)
  unsigned* ticket;     // [B] arrival counters, zero on entry, zero on return
  float* loss[2];       // [B] per direction
};

__device__ __forceinline__ float s5_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

__global__ __launch_bounds__(S5_THREADS) void pairmin_s5_kernel(S5Args a) {
  // XCD-aware order: the blocks of one sample take consecutive slots of ONE XCD (they read the same two point sets)
  const int id = blockIdx.x, slot = id >> 3, member = slot % a.members;
  const int b = (slot / a.members) * 8 + (id & 7);
  if (b >= a.B) return;
  const int dir = member >= a.m0 ? 1 : 0;
  const S5Dir d = a.d[dir];
  const int mem = member - (dir ? a.m0 : 0);
  const bool leftover = d.left > 0 && mem == d.tiles - 1;  // this block also evaluates the direction's remainder queries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ qb = d.q + (size_t)b * d.nq * 3;
  const float* __restrict__ rb = d.r + (size_t)b * d.nr * 3;
  const int nmain = d.nq - d.left;  // queries covered by the main tiles

  extern __shared__ __attribute__((aligned(16))) char pm_smem[];
  float4* sref = reinterpret_cast<float4*>(pm_smem);
  const int padded = d.slice * S5_WAVES;
  float(*s_val)[S5_TILE] = reinterpret_cast<float(*)[S5_TILE]>(pm_smem + (size_t)S5_MAX_REFS * sizeof(float4));
  int(*s_grp)[S5_TILE] = reinterpret_cast<int(*)[S5_TILE]>(pm_smem + (size_t)S5_MAX_REFS * sizeof(float4) + sizeof(float) * S5_WAVES * S5_TILE);
  __shared__ u64 s_key[S5_LEFT_MAX];  // remainder role: one packed (distance, index) key per remainder query
  __shared__ float s_sum[S5_WAVES];

  // queries first (main role), then every staging load, one wait for all of them
  float qx[S5_QPT], qy[S5_QPT], qz[S5_QPT];
  {
#pragma unroll
    for (int k = 0; k < S5_QPT; ++k) {
      const int qi = mem * S5_TILE + k * 64 + lane;
      const int qc = qi < nmain ? qi : nmain - 1;  // idle lanes redo the last point, never stored
      qx[k] = qb[(size_t)qc * 3 + 0];
      qy[k] = qb[(size_t)qc * 3 + 1];
      qz[k] = qb[(size_t)qc * 3 + 2];
    }
  }
  {
    constexpr int PER = S5_MAX_REFS / S5_THREADS;  // 2 references per thread
    float rx[PER], ry[PER], rz[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * S5_THREADS;
      const int ic = i < d.nr ? i : d.nr - 1;
      rx[u] = rb[(size_t)ic * 3 + 0];
      ry[u] = rb[(size_t)ic * 3 + 1];
      rz[u] = rb[(size_t)ic * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + u * S5_THREADS;
      if (i < padded) sref[i] = i < d.nr ? make_float4(rx[u], ry[u], rz[u], 0.f) : make_float4(PM_BIG, PM_BIG, PM_BIG, 0.f);
    }
  }
  if (leftover && tid < S5_LEFT_MAX) s_key[tid] = ~0ull;
  __syncthreads();

  float block_sum = 0.f;  // valid in thread 0 after the epilogues
  {
    float best[S5_QPT], cm[S5_QPT];
    int bestc[S5_QPT];
#pragma unroll
    for (int k = 0; k < S5_QPT; ++k) { best[k] = __builtin_inff(); bestc[k] = 0x7fffffff; }
    const int jbeg = wave * d.slice, groups = d.slice >> 2;
    // queries 0..3 as two packed pairs (v_pk_add / v_pk_mul / v_pk_fma: 1.8x the time of the scalar op for 2x the work,
    // tools/ubench/valu_rate.hip), query 4 scalar; element-wise the same rounding as obman_dist2
    const f2 px0 = {qx[0], qx[1]}, py0 = {qy[0], qy[1]}, pz0 = {qz[0], qz[1]};
    const f2 px1 = {qx[2], qx[3]}, py1 = {qy[2], qy[3]}, pz1 = {qz[2], qz[3]};
    auto sweep4 = [&](const float4 (&r)[4]) {  // four references against the lane's five queries: running chunk minima only
      // the .w lanes are "used" HERE, at consumption: keeps the reads ds_read_b128 (b96 is 2x the LDS cycles) without forcing a
      // wait right behind the prefetch
      asm volatile("" ::"v"(r[0].w), "v"(r[1].w), "v"(r[2].w), "v"(r[3].w));
      const f2 a0 = pm_dist2_pk(px0, py0, pz0, r[0].x, r[0].y, r[0].z), a1 = pm_dist2_pk(px0, py0, pz0, r[1].x, r[1].y, r[1].z);
      const f2 a2 = pm_dist2_pk(px0, py0, pz0, r[2].x, r[2].y, r[2].z), a3 = pm_dist2_pk(px0, py0, pz0, r[3].x, r[3].y, r[3].z);
      cm[0] = s5_min3(s5_min3(cm[0], a0[0], a1[0]), a2[0], a3[0]);
      cm[1] = s5_min3(s5_min3(cm[1], a0[1], a1[1]), a2[1], a3[1]);
      const f2 b0 = pm_dist2_pk(px1, py1, pz1, r[0].x, r[0].y, r[0].z), b1 = pm_dist2_pk(px1, py1, pz1, r[1].x, r[1].y, r[1].z);
      const f2 b2 = pm_dist2_pk(px1, py1, pz1, r[2].x, r[2].y, r[2].z), b3 = pm_dist2_pk(px1, py1, pz1, r[3].x, r[3].y, r[3].z);
      cm[2] = s5_min3(s5_min3(cm[2], b0[0], b1[0]), b2[0], b3[0]);
      cm[3] = s5_min3(s5_min3(cm[3], b0[1], b1[1]), b2[1], b3[1]);
      const float e0 = obman_dist2(qx[4], qy[4], qz[4], r[0].x, r[0].y, r[0].z), e1 = obman_dist2(qx[4], qy[4], qz[4], r[1].x, r[1].y, r[1].z);
      const float e2 = obman_dist2(qx[4], qy[4], qz[4], r[2].x, r[2].y, r[2].z), e3 = obman_dist2(qx[4], qy[4], qz[4], r[3].x, r[3].y, r[3].z);
      cm[4] = s5_min3(s5_min3(cm[4], e0, e1), e2, e3);
    };
    auto fetch4 = [&](float4 (&r)[4], int j) {
      r[0] = sref[j]; r[1] = sref[j + 1]; r[2] = sref[j + 2]; r[3] = sref[j + 3];
    };
    auto close_chunk = [&](int first) {
#pragma unroll
      for (int k = 0; k < S5_QPT; ++k) {
        const bool better = cm[k] < best[k];
        best[k] = better ? cm[k] : best[k];
        bestc[k] = better ? first : bestc[k];
      }
    };
    // the next group's references are requested before the current group is evaluated (two register sets): the wave never sits
    // on an LDS round trip with nothing to issue.  The last prefetch of a slice re-reads its final group (never evaluated).
    const int jlast = jbeg + (groups - 1) * 4;
    float4 ra[4], rb4[4];
    fetch4(ra, jbeg);
    int g = 0;
    for (; g + S5_CHUNK_GROUPS <= groups; g += S5_CHUNK_GROUPS) {  // full chunks of 16 references
#pragma unroll
      for (int k = 0; k < S5_QPT; ++k) cm[k] = __builtin_inff();
      const int j = jbeg + g * 4;
      fetch4(rb4, j + 4);
      sweep4(ra);
      fetch4(ra, j + 8);
      sweep4(rb4);
      fetch4(rb4, j + 12);
      sweep4(ra);
      fetch4(ra, min(j + 16, jlast));
      sweep4(rb4);
      close_chunk(j);
    }
    for (; g < groups; ++g) {  // remaining groups of the slice: chunks of 4 references
#pragma unroll
      for (int k = 0; k < S5_QPT; ++k) cm[k] = __builtin_inff();
      const int j = jbeg + g * 4;
      sweep4(ra);
      fetch4(ra, min(j + 4, jlast));
      close_chunk(j);
    }
#pragma unroll
    for (int k = 0; k < S5_QPT; ++k) {
      s_val[wave][k * 64 + lane] = best[k];
      s_grp[wave][k * 64 + lane] = bestc[k];
    }
    __syncthreads();
    float mine = 0.f;
    if (wave < S5_QPT) {  // thread (wave k, lane l) resolves query k*64 + l: the lane's own k-th query, already in registers
      const int t = wave * 64 + lane, qi = mem * S5_TILE + t;
      float x = qx[0], y = qy[0], z = qz[0];
#pragma unroll
      for (int k = 1; k < S5_QPT; ++k)
        if (wave == k) { x = qx[k]; y = qy[k]; z = qz[k]; }
      float bv = s_val[0][t];
      int bc = s_grp[0][t];
#pragma unroll
      for (int w = 1; w < S5_WAVES; ++w) {
        const float v = s_val[w][t];
        const int c = s_grp[w][t];
        if (v < bv || (v == bv && c < bc)) { bv = v; bc = c; }
      }
      if (bc == 0x7fffffff) bc = 0;  // NaN inputs: nothing ever compared smaller
      // exact index inside the winning chunk (<= 16 references of one wave's slice), re-evaluated with the same sequence;
      // a later chunk cannot hold an equal value at a lower index, so the first match from the chunk start is the arg-min
      const int cend = min(min(bc + 4 * S5_CHUNK_GROUPS, (bc / d.slice + 1) * d.slice), d.nr);
      int idx = bc;
      for (int j = cend - 1; j >= bc; --j) {  // descending so the FIRST matching index survives
        const float4 r = sref[j];
        if (obman_dist2(x, y, z, r.x, r.y, r.z) == bv) idx = j;
      }
      if (qi < nmain) {
        const size_t o = (size_t)b * d.nq + qi;
        d.omin[o] = bv;
        if (d.oidx) d.oidx[o] = idx;
        mine = bv;
      }
    }
    if (a.part) {  // fixed-order block sum of the tile's minima: wave sums by DPP, then waves 0..4 in order
      const float ws = obman_wave_sum(mine);
      if (lane == 0) s_sum[wave] = ws;
      __syncthreads();
      if (tid == 0) block_sum = (((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3])) + s_sum[4]);
    }
  }
  if (leftover) {
    // transposed role: each remainder query against all references, lanes over references
    for (int lq = 0; lq < d.left; ++lq) {
      const int qi = nmain + lq;
      const float x = qb[(size_t)qi * 3], y = qb[(size_t)qi * 3 + 1], z = qb[(size_t)qi * 3 + 2];
      u64 key = ~0ull;
      for (int j = tid; j < d.nr; j += S5_THREADS) {
        const float4 r = sref[j];
        const float e = obman_dist2(x, y, z, r.x, r.y, r.z);
        // e >= +0: its bit pattern orders like the value; NaN (0x7fc...) sorts above every finite distance and +inf
        const u64 k2 = ((u64)__float_as_uint(e) << 32) | (unsigned)j;
        key = k2 < key ? k2 : key;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(key, off, 64);
        key = o < key ? o : key;
      }
      if (lane == 0) atomicMin(&s_key[lq], key);  // LDS ds_min_u64: order independent
    }
    __syncthreads();
    if (tid == 0) {
      for (int lq = 0; lq < d.left; ++lq) {
        const u64 key = s_key[lq];
        const float bv = __uint_as_float((unsigned)(key >> 32));
        const size_t o = (size_t)b * d.nq + nmain + lq;
        d.omin[o] = bv;
        if (d.oidx) d.oidx[o] = (int)(unsigned)(key & 0xffffffffu);
        block_sum += bv;
      }
    }
  }

  if (a.part == nullptr || tid != 0) return;
  if (a.pairwise) {
    // Every direction has at most two blocks per sample (642 x 600: two tiles each): the partial sum travels INSIDE the one
    // atomic.  A 64-bit word per (sample, direction), zero on entry, one 32-bit field per tile holding the bits of the partial
    // sum with the sign bit set (sums are >= +0, so a field is non-zero exactly when its tile has arrived).  atomicOr returns the
    // other tile's field: whoever sees it set has both sums, adds them tile 0 first, writes the mean and clears the word.  One
    // round trip instead of store -> acknowledgement -> ticket -> read-back; still order- and placement-independent.
    float* out = a.loss[dir];
    if (d.tiles == 1) {
      if (out) out[b] = block_sum / (float)d.nq;
      return;
    }
    u64* word = reinterpret_cast<u64*>(a.ticket) + (size_t)b * 2 + dir;
    const u64 mine_field = (u64)(__float_as_uint(block_sum) | 0x80000000u) << (32 * mem);
    const u64 old = __hip_atomic_fetch_or(word, mine_field, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned other = (unsigned)(old >> (32 * (1 - mem)));
    if (!other) return;  // first of the two
    const float po = __uint_as_float(other & 0x7fffffffu);
    const float total = mem == 0 ? block_sum + po : po + block_sum;
    if (out) out[b] = total / (float)d.nq;
    __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning for the next call
    return;
  }
  // general case: publish, draw a ticket, and - for the last arriver of this sample - add the partial sums in member order
  float* mine_slot = a.part + (size_t)b * a.members + member;
  __hip_atomic_store(mine_slot, block_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through store is acknowledged before the ticket is drawn
  const unsigned old = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old != (unsigned)a.members - 1u) return;
  float s0 = 0.f, s1 = 0.f;
  for (int m = 0; m < a.members; ++m) {
    const float v = m == member ? block_sum : __hip_atomic_load(a.part + (size_t)b * a.members + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (m < a.m0) s0 += v; else s1 += v;
  }
  if (a.loss[0] && a.d[0].omin) a.loss[0][b] = s0 / (float)a.d[0].nq;
  if (a.loss[1] && a.d[1].omin) a.loss[1][b] = s1 / (float)a.d[1].nq;
  __hip_atomic_store(a.ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-cleaning for the next call
}

struct PmBwdSide {
  const float* p;        // own points   [B,n,3]
  const float* o;        // other points [B,m,3]
  const int* idx_own;    // [B,n] nearest other of each own point
  const int* idx_other;  // [B,m] nearest own of each other point
  const float* g_own;    // upstream grads of own minima (null = 0)
  const float* g_other;  // upstream grads of other minima (null = 0)
  float* grad;           // [B,n,3] or null (side skipped)
  int n, m, per_sample;  // per_sample: g_* are [B] loss grads, scaled by 1/n resp. 1/m (fused Chamfer mean)
};

constexpr int PB_TILE = 1024;

// Block = 4 waves over the same 64 own points.  Scatter side as an OWNER SCAN without float atomics: the other side's points
// whose arg-min falls into this block's 64 own points are first compacted, in ascending index order, into LDS (ballot +
// prefix counts: ~m*64/n entries instead of m - 60 of 600 at 642 x 600), then each wave scans a quarter of that short list
// and the four partial sums are added in fixed wave order => deterministic.  (The r01/r02 kernel scanned all m arg-mins
// per own point: 10 instructions x 150 iterations per wave, as much VALU work as the forward sweep.)
__global__ __launch_bounds__(256) void pairmin_bwd_kernel(PmBwdSide s0, PmBwdSide s1) {
  const PmBwdSide s = blockIdx.z == 0 ? s0 : s1;
  if (s.grad == nullptr) return;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x * 64 >= s.n) return;
  const int i0 = blockIdx.x * 64, i = i0 + lane;
  const float* __restrict__ pb = s.p + (size_t)b * s.n * 3;
  const float* __restrict__ ob = s.o + (size_t)b * s.m * 3;
  const int ic = i < s.n ? i : s.n - 1;
  const float px = pb[(size_t)ic * 3], py = pb[(size_t)ic * 3 + 1], pz = pb[(size_t)ic * 3 + 2];
  // own term: issue its loads now, they are consumed at the very end
  float own_w = 0.f, ox = px, oy = py, oz = pz;
  if (s.g_own && wave == 0 && i < s.n) {
    own_w = 2.f * (s.per_sample ? s.g_own[b] / (float)s.n : s.g_own[(size_t)b * s.n + i]);
    const int j = s.idx_own[(size_t)b * s.n + i];
    ox = ob[(size_t)j * 3];
    oy = ob[(size_t)j * 3 + 1];
    oz = ob[(size_t)j * 3 + 2];
  }
  float gx = 0.f, gy = 0.f, gz = 0.f;
  __shared__ float4 so[PB_TILE];
  __shared__ int sidx[PB_TILE];
  __shared__ int s_cnt[(PB_TILE / 256) * 4 + 1];
  __shared__ float s_acc[3][3][64];
  if (s.g_other) {
    constexpr int ROUNDS = PB_TILE / 256;
    const float gs = s.per_sample ? 2.f * s.g_other[b] / (float)s.m : 0.f;
    const int* __restrict__ idx_o = s.idx_other + (size_t)b * s.m;
    for (int base = 0; base < s.m; base += PB_TILE) {
      const int cnt = min(PB_TILE, s.m - base);
      int rel[ROUNDS], rank[ROUNDS];
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {  // coalesced sweep over the arg-mins, round-major = ascending index
        const int t = r * 256 + tid;
        rel[r] = t < cnt ? idx_o[base + t] - i0 : -1;
        const bool hit = (unsigned)rel[r] < 64u;
        const u64 bal = __ballot(hit);
        rank[r] = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (lane == 0) s_cnt[r * 4 + wave] = __popcll(bal);
      }
      __syncthreads();
      int total = 0;
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          if (w == wave && (unsigned)rel[r] < 64u) {
            const int j = base + r * 256 + tid, pos = total + rank[r];
            const float wgt = s.per_sample ? gs : 2.f * s.g_other[(size_t)b * s.m + j];
            so[pos] = make_float4(ob[(size_t)j * 3], ob[(size_t)j * 3 + 1], ob[(size_t)j * 3 + 2], wgt);
            sidx[pos] = rel[r];
          }
          total += s_cnt[r * 4 + w];
        }
      }
      __syncthreads();
      const int slice = (total + 3) >> 2;
      const int tend = min(total, (wave + 1) * slice);
      for (int t = wave * slice; t < tend; ++t) {
        const float4 v = so[t];
        const float w = sidx[t] == lane ? v.w : 0.f;
        gx = __fmaf_rn(w, px - v.x, gx);
        gy = __fmaf_rn(w, py - v.y, gy);
        gz = __fmaf_rn(w, pz - v.z, gz);
      }
      __syncthreads();
    }
    if (wave > 0) { s_acc[wave - 1][0][lane] = gx; s_acc[wave - 1][1][lane] = gy; s_acc[wave - 1][2][lane] = gz; }
    __syncthreads();
    if (wave == 0) {
      gx = ((gx + s_acc[0][0][lane]) + s_acc[1][0][lane]) + s_acc[2][0][lane];
      gy = ((gy + s_acc[0][1][lane]) + s_acc[1][1][lane]) + s_acc[2][1][lane];
      gz = ((gz + s_acc[0][2][lane]) + s_acc[1][2][lane]) + s_acc[2][2][lane];
    }
  }
  if (wave != 0 || i >= s.n) return;
  if (s.g_own) {
    gx = __fmaf_rn(own_w, px - ox, gx);
    gy = __fmaf_rn(own_w, py - oy, gy);
    gz = __fmaf_rn(own_w, pz - oz, gz);
  }
  float* g = s.grad + ((size_t)b * s.n + i) * 3;
  g[0] = gx;
  g[1] = gy;
  g[2] = gz;
}

// A block covers 64*QPT queries (QPT in {2,4,10}; packed pairs).  More queries per lane = fewer LDS reads per pair and,
// for the short side of an asymmetric problem (600 GT points vs 16 050 vertices), QPT = 10 puts every query of a sample
// into ONE block column so the long reference set is swept once instead of once per 64-query tile
// (profiles/r01_chamfer_pmc.md: 10x re-read).  OBMAN_PM_QPT overrides (tuning only).
int choose_qpt(int B, int nq, int nr) {
  static const int forced = [] { const char* e = getenv("OBMAN_PM_QPT"); return e ? atoi(e) : 0; }();
  if (forced == 2 || forced == 4 || forced == 10) return forced;
  if (nq <= 640 && nr >= 8192) return 10;  // below that the extra memset + unpack launches of the split path cost more than the re-reads
  if ((long)B * obman_cdiv(nq, 256) >= 512) return 4;
  return 2;
}

template <int QPT, bool RS = false, bool QS = true>
void launch_fwd_t(dim3 grid, size_t smem, hipStream_t st, const PmDir& a, const PmDir& b) {
  static const int xcd = [] { const char* e = getenv("OBMAN_PM_XCD"); return e ? atoi(e) : 1; }();  // A/B knob
  const PmGrid pg{(int)grid.x, (int)grid.z, (int)grid.y, xcd};
  const unsigned blocks = (unsigned)(((pg.B + 7) / 8) * 8) * grid.x * grid.z;  // samples padded to a multiple of 8 (one group per XCD slot)
  pairmin_fwd_kernel<QPT, RS, QS><<<dim3(blocks), PM_THREADS, smem, st>>>(a, b, pg);
}
void launch_fwd(int qpt, dim3 grid, size_t smem, hipStream_t st, const PmDir& a, const PmDir& b, int kid) {
  ObmanProfScope prof(kid, st);
  switch (qpt) {
    case 10: launch_fwd_t<10>(grid, smem, st, a, b); break;
    case 4: launch_fwd_t<4>(grid, smem, st, a, b); break;
    default: launch_fwd_t<2>(grid, smem, st, a, b); break;
  }
}

void plan_dir(PmDir& d, int B, int qpt, bool have_ws) {
  d.qtiles = obman_cdiv(d.nq, 64 * qpt);
  d.rsplit = 1;
  d.rchunk = d.nr;
  d.tile = PM_REF_TILE;
  if (have_ws && d.omin) {
    const long blocks = (long)B * d.qtiles;
    if (blocks < 512 && d.nr >= 8192) {
      int want = (int)((1024 + blocks - 1) / blocks);
      int maxsplit = obman_cdiv(d.nr, 256);  // at least 64 references per wave
      d.rsplit = want < maxsplit ? want : maxsplit;
      d.rchunk = ((obman_cdiv(d.nr, d.rsplit) + 3) / 4) * 4;
      d.rsplit = obman_cdiv(d.nr, d.rchunk);
    }
  }
}

// ---- S5 planning (shared by the launcher and obman_chamfer_sync_bytes)
struct S5Plan { S5Dir d0, d1; int members, m0; bool ok; };
void s5_plan_dir(S5Dir& d) {
  const int rem = d.nq % S5_TILE;
  d.left = (d.nq > S5_TILE && rem > 0 && rem <= S5_LEFT_MAX) ? rem : 0;
  d.tiles = d.left ? d.nq / S5_TILE : obman_cdiv(d.nq, S5_TILE);
  d.slice = ((obman_cdiv(d.nr, S5_WAVES) + 3) / 4) * 4;
}
S5Plan s5_plan(const float* x, const float* y, int Nx, int Ny, float* min_x, int* idx_x, float* min_y, int* idx_y) {
  static const int enabled = [] { const char* e = getenv("OBMAN_PM_S5"); return e ? atoi(e) : 1; }();  // A/B knob
  S5Plan p{};
  p.d0 = S5Dir{x, y, min_x, idx_x, Nx, Ny, 0, 0, 0};
  p.d1 = S5Dir{y, x, min_y, idx_y, Ny, Nx, 0, 0, 0};
  p.ok = enabled && (min_x || min_y) && (!min_x || Ny <= S5_MAX_REFS) && (!min_y || Nx <= S5_MAX_REFS);
  if (!p.ok) return p;
  s5_plan_dir(p.d0);
  s5_plan_dir(p.d1);
  p.m0 = min_x ? p.d0.tiles : 0;
  p.members = p.m0 + (min_y ? p.d1.tiles : 0);
  return p;
}
// what the kernel uses (reference tile + merge buffers = 36 KB), raised to just over half of a CU's 160 KB so that the dispatcher
// cannot put two of these blocks on one CU while another CU idles (equal blocks, one round: the slowest CU is the kernel time)
constexpr size_t S5_LDS_USED = (size_t)S5_MAX_REFS * sizeof(float4) + (size_t)2 * S5_WAVES * S5_TILE * sizeof(float);
constexpr size_t S5_LDS = S5_LDS_USED > 82 * 1024 ? S5_LDS_USED : 82 * 1024;
// `sync` layout: [S5_MAX_TICKETS arrival counters][B x members partial sums].  The counter region has a FIXED size so that calls
// with different batch sizes sharing one buffer never see another call's (non-zero) partial sums where they expect zero counters.
constexpr int S5_MAX_TICKETS = 4096;
long s5_sync_bytes(int B, int members) {
  return B <= S5_MAX_TICKETS ? (long)sizeof(unsigned) * S5_MAX_TICKETS + (long)sizeof(float) * B * members : 0;
}

// One launch: minima + arg-mins of both directions and (with `sync` and loss pointers) the per-sample means.
int launch_s5(const S5Plan& p, int B, float* loss_1, float* loss_2, void* sync, long sync_bytes, hipStream_t st, bool* fused, int kid) {
  S5Args a{};
  a.d[0] = p.d0;
  a.d[1] = p.d1;
  a.B = B;
  a.members = p.members;
  a.m0 = p.m0;
  const bool want_mean = loss_1 || loss_2;
  const long need = s5_sync_bytes(B, p.members);
  *fused = want_mean && sync && need > 0 && sync_bytes >= need;
  if (*fused) {
    a.pairwise = p.d0.tiles <= 2 && p.d1.tiles <= 2 && (size_t)B * 2 * sizeof(u64) <= (size_t)S5_MAX_TICKETS * sizeof(unsigned);
    a.ticket = reinterpret_cast<unsigned*>(sync);
    a.part = reinterpret_cast<float*>(a.ticket + S5_MAX_TICKETS);
    a.loss[0] = loss_1;
    a.loss[1] = loss_2;
  }
  static std::atomic<int> granted[MAX_DEVICES];
  const int dev = current_device();
  if (!granted[dev].load(std::memory_order_relaxed)) {
    const hipError_t err = hipFuncSetAttribute((const void*)pairmin_s5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S5_LDS);
    if (err != hipSuccess) return (int)err;
    granted[dev].store(1, std::memory_order_relaxed);
  }
  const unsigned blocks = (unsigned)(((B + 7) / 8) * 8) * (unsigned)p.members;
  {
    ObmanProfScope prof(kid, st);
    pairmin_s5_kernel<<<dim3(blocks), S5_THREADS, S5_LDS, st>>>(a);
  }
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int launch_pairmin(const float* x, const float* y, int B, int Nx, int Ny, float* min_x, int* idx_x, float* min_y,
                   int* idx_y, void* ws, long ws_bytes, hipStream_t st, float* loss_x = nullptr, float* loss_y = nullptr,
                   void* sync = nullptr, long sync_bytes = 0, bool* means_done = nullptr, int kid = OBMAN_K_PAIRMIN_FWD) {
  if (B < 0 || Nx < 0 || Ny < 0) return -1;
  if (B == 0) return 0;
  if (Nx == 0 || Ny == 0) return -2;  // torch.min over an empty dim raises in the reference
  if (means_done) *means_done = false;
  {
    const S5Plan p = s5_plan(x, y, Nx, Ny, min_x, idx_x, min_y, idx_y);
    if (p.ok) {
      bool fused = false;
      const int rc = launch_s5(p, B, loss_x, loss_y, sync, sync_bytes, st, &fused, kid);
      if (means_done) *means_done = fused;
      return rc;
    }
  }
  PmDir d0{x, y, min_x, idx_x, nullptr, Nx, Ny, 0, 1, Ny, PM_REF_TILE, nullptr};
  PmDir d1{y, x, min_y, idx_y, nullptr, Ny, Nx, 0, 1, Nx, PM_REF_TILE, nullptr};
  {
    // Round 6, the fused sweep: the SHORT side's minima wanted (with or without the long side's), one side long (>= 8192 points:
    // where the swapped-role direction needs the split path) and the other a single LDS tile.  The long side is the query side (10 per lane); the short side's minima come out
    // of the same distances (pairmin_fwd_kernel<10, true>) and pairmin_resolve_kernel finds their arg-mins.  OBMAN_PM_FUSED=0: the
    // two independent sweeps (the checker).
    const char* fe = getenv("OBMAN_PM_FUSED");  // read per call: tests flip it inside one process to get the checker
    const int fused_on = fe ? atoi(fe) : 1;
    const bool x_long = Nx >= Ny;
    const int nl = x_long ? Nx : Ny, ns = x_long ? Ny : Nx;
    const long need = (long)sizeof(u64) * B * ns;
    float* min_s = x_long ? min_y : min_x;  // minima of the short side = the reference side of this sweep
    float* min_l = x_long ? min_x : min_y;
    if (fused_on && min_s && nl >= 8192 && ns <= PM_REF_TILE && ws && ws_bytes >= need) {
      PmDir d = x_long ? d0 : d1;
      d.qtiles = obman_cdiv(d.nq, 640);
      d.tile = ((ns + 31) / 32) * 32;
      d.rs_ws = (u64*)ws;
      (void)obman_fill_u32(ws, 0xffffffffu, (size_t)2 * B * ns, st);
      // only the short side wanted (the contact term's hand -> object minima): the sweep without the per-query bookkeeping
      const bool qs = min_l != nullptr;
      const size_t smem = (size_t)d.tile * sizeof(float4) + (qs ? (size_t)8 * 64 * 10 * sizeof(float) : 0) + (size_t)d.tile * sizeof(float);
      {
        ObmanProfScope prof(kid, st);
        if (qs) launch_fwd_t<10, true>(dim3(d.qtiles, B, 1), smem, st, d, d);
        else launch_fwd_t<10, true, false>(dim3(d.qtiles, B, 1), smem, st, d, d);
      }
      OBMAN_LAUNCH_CHECK();
      pairmin_resolve_kernel<<<((B + 7) / 8) * 8 * obman_cdiv(ns, 4), 256, 0, st>>>((const u64*)ws, d.q, d.r, B, d.nq, d.nr, 640, min_s,
                                                                                    x_long ? idx_y : idx_x);
      OBMAN_LAUNCH_CHECK();
      return 0;
    }
  }
  int q0 = choose_qpt(B, Nx, Ny), q1 = choose_qpt(B, Ny, Nx);
  // no split path involved: one launch for both directions beats two (2562 x 600, bs 64: 53 us merged at QPT 2, 61 us as
  // two launches at QPT 4 / 2, 69 us merged at QPT 4)
  if (q0 != 10 && q1 != 10) q0 = q1 = (q0 < q1 ? q0 : q1);
  const bool have_ws = ws && ws_bytes >= (long)sizeof(u64) * B * ((long)Nx + Ny);
  plan_dir(d0, B, q0, have_ws);
  plan_dir(d1, B, q1, have_ws);
  if (d0.rsplit > 1) d0.ws = (u64*)ws;
  if (d1.rsplit > 1) d1.ws = (u64*)ws + (size_t)B * Nx;
  if (d0.rsplit > 1) (void)obman_fill_u32(d0.ws, 0xffffffffu, (size_t)2 * B * Nx, st);
  if (d1.rsplit > 1) (void)obman_fill_u32(d1.ws, 0xffffffffu, (size_t)2 * B * Ny, st);
  // LDS reference tile (points per staging pass); OBMAN_PM_TILE overrides the 2048-point cap for the tile sweep of
  // BASELINE.json configs[4] (multiples of 32, <= 3072 so that tile + merge buffers stay under 64 KiB)
  static const int tile_cap = [] {
    const char* e = getenv("OBMAN_PM_TILE");
    int t = e ? atoi(e) : PM_REF_TILE;
    t = (t / 32) * 32;
    return t < 64 ? 64 : (t > 3072 ? 3072 : t);
  }();
  auto tile_of = [](const PmDir& d) {
    int t = ((d.rchunk + 31) / 32) * 32;  // 4 waves x whole chunks of PM_CHUNK references
    return t > tile_cap ? tile_cap : t;
  };
  auto smem_of = [](int tile, int qpt) { return (size_t)tile * sizeof(float4) + (size_t)8 * 64 * qpt * sizeof(float); };
  if (min_x && min_y && q0 == q1) {  // symmetric sizes: both directions in one launch
    const int gx0 = d0.qtiles * d0.rsplit, gx1 = d1.qtiles * d1.rsplit;
    const int tile = tile_of(d0) > tile_of(d1) ? tile_of(d0) : tile_of(d1);
    d0.tile = d1.tile = tile;
    launch_fwd(q0, dim3(gx0 > gx1 ? gx0 : gx1, B, 2), smem_of(tile, q0), st, d0, d1, kid);
  } else {  // one launch per direction, each with its own query tiling
    if (min_x) {
      d0.tile = tile_of(d0);
      launch_fwd(q0, dim3(d0.qtiles * d0.rsplit, B, 1), smem_of(d0.tile, q0), st, d0, d0, kid);
    }
    if (min_y) {
      d1.tile = tile_of(d1);
      // the second direction's own launch has its own profiler id: a launch is priced against ITS bytes (bench.py roofline)
      launch_fwd(q1, dim3(d1.qtiles * d1.rsplit, B, 1), smem_of(d1.tile, q1), st, d1, d1,
                 kid == OBMAN_K_CHAMFER_FWD ? OBMAN_K_CHAMFER_FWD_Y : (kid == OBMAN_K_PAIRMIN_FWD ? OBMAN_K_PAIRMIN_FWD_Y : kid));
    }
  }
  OBMAN_LAUNCH_CHECK();
  if (d0.rsplit > 1) {
    const long n = (long)B * Nx;
    pairmin_unpack_kernel<<<obman_cdiv(n, 256), 256, 0, st>>>(d0.ws, min_x, idx_x, n);
  }
  if (d1.rsplit > 1) {
    const long n = (long)B * Ny;
    pairmin_unpack_kernel<<<obman_cdiv(n, 256), 256, 0, st>>>(d1.ws, min_y, idx_y, n);
  }
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int launch_pairmin_bwd(const float* x, const float* y, int B, int Nx, int Ny, const int* idx_x, const int* idx_y,
                       const float* g_x, const float* g_y, float* grad_x, float* grad_y, int per_sample,
                       hipStream_t st, int kid = OBMAN_K_PAIRMIN_BWD) {
  if (B < 0 || Nx <= 0 || Ny <= 0) return B == 0 ? 0 : -1;
  if (B == 0) return 0;
  if ((g_x && !idx_x) || (g_y && !idx_y)) return -3;
  PmBwdSide s0{x, y, idx_x, idx_y, g_x, g_y, grad_x, Nx, Ny, per_sample};
  PmBwdSide s1{y, x, idx_y, idx_x, g_y, g_x, grad_y, Ny, Nx, per_sample};
  const int n_max = (grad_x ? Nx : 0) > (grad_y ? Ny : 0) ? Nx : Ny;
  if (!grad_x && !grad_y) return 0;
  dim3 grid(obman_cdiv(n_max, 64), B, 2);
  {
    ObmanProfScope prof(kid, st);
    pairmin_bwd_kernel<<<grid, 256, 0, st>>>(s0, s1);
  }
  OBMAN_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" {

int obman_abi_version(void) { return OBMAN_ABI_VERSION; }

long obman_pairmin_ws_bytes(int B, int Nx, int Ny) { return (long)sizeof(u64) * B * ((long)Nx + Ny); }

int obman_pairmin_fwd(const float* x, const float* y, int B, int Nx, int Ny, float* min_x, int* idx_x, float* min_y,
                      int* idx_y, void* ws, long ws_bytes, obman_stream_t stream) {
  return launch_pairmin(x, y, B, Nx, Ny, min_x, idx_x, min_y, idx_y, ws, ws_bytes, (hipStream_t)stream);
}

int obman_pairmin_bwd(const float* x, const float* y, int B, int Nx, int Ny, const int* idx_x, const int* idx_y,
                      const float* g_min_x, const float* g_min_y, float* grad_x, float* grad_y,
                      obman_stream_t stream) {
  return launch_pairmin_bwd(x, y, B, Nx, Ny, idx_x, idx_y, g_min_x, g_min_y, grad_x, grad_y, 0, (hipStream_t)stream);
}

long obman_chamfer_sync_bytes(int B, int Np, int Ng) {
  if (B <= 0 || Np <= 0 || Ng <= 0) return 0;
  float dummy = 0.f;  // only the pointers' null-ness matters to the plan
  const S5Plan p = s5_plan(nullptr, nullptr, Np, Ng, &dummy, nullptr, &dummy, nullptr);
  return p.ok ? s5_sync_bytes(B, p.members) : 0;
}

int obman_chamfer_fwd(const float* preds, const float* gts, int B, int Np, int Ng, float* loss_1, float* loss_2,
                      float* min_pred, int* idx_pred, float* min_gt, int* idx_gt, void* ws, long ws_bytes,
                      void* sync, long sync_bytes, obman_stream_t stream) {
  if (!loss_1 || !loss_2 || !min_pred || !min_gt) return -4;
  hipStream_t st = (hipStream_t)stream;
  bool means_done = false;
  const int rc = launch_pairmin(preds, gts, B, Np, Ng, min_pred, idx_pred, min_gt, idx_gt, ws, ws_bytes, st, loss_1, loss_2, sync,
                                sync_bytes, &means_done, OBMAN_K_CHAMFER_FWD);
  if (rc != 0 || B == 0 || means_done) return rc;
  rowmean2_kernel<<<dim3(B, 2), (Np > 4096 || Ng > 4096) ? 1024 : 256, 0, st>>>(min_pred, Np, loss_1, min_gt, Ng, loss_2);
  OBMAN_LAUNCH_CHECK();
  return 0;
}

int obman_chamfer_bwd(const float* preds, const float* gts, int B, int Np, int Ng, const int* idx_pred,
                      const int* idx_gt, const float* g_loss_1, const float* g_loss_2, float* grad_preds,
                      float* grad_gts, obman_stream_t stream) {
  return launch_pairmin_bwd(preds, gts, B, Np, Ng, idx_pred, idx_gt, g_loss_1, g_loss_2, grad_preds, grad_gts, 1,
                            (hipStream_t)stream, OBMAN_K_CHAMFER_BWD);
}

}  // extern "C"
