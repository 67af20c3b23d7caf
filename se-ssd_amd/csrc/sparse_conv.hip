// Sparse 3-D convolution (SubMConv3d / SparseConv3d) + folded BatchNorm1d + ReLU on gfx950.
// Replaces spconv's indice_conv (per-offset gather -> cuBLAS mm -> scatter-add) as called from
//   det3d/models/backbones/scn.py:106-148,183 (SpMiddleFHD.middle_conv) and the BN/ReLU modules
//   that SparseSequential applies to .features; optionally the `.dense()` of scn.py:184-187.
//
// Output-stationary implicit GEMM on the exact-f32 matrix cores: one wave owns a tile of 16 output
// sites x all Cout channels and walks the kernel offsets. For an offset with at least one neighbour
// in the tile (wave-uniform test on the rulebook's tile bitmask) it does
//     acc[16 x Cout] += A[16 x Cin] * W[k][Cin x Cout]      v_mfma_f32_16x16x4_f32
// A is gathered straight into registers: lane (i, kq) = (lane&15, lane>>4) reads the CONTIGUOUS
// quarter row in[nbr[k][i]][kq*Cin/4 .. +Cin/4) with 16-byte loads (four lanes cover one feature
// row = whole cache lines; no LDS round trip is needed because MFMA step s may use any K order,
// here cin = kq*Cin/4 + s). W is pre-packed in exactly that fragment order so B operands are
// coalesced 16-byte loads that stay L2/L1 resident (<= 442 KB per layer). Results are bit-for-bit
// an fmaf chain in (offset, cin) order; every output row is written once, fused with
// y = max(0, acc*scale + shift). HBM traffic ~ features in (once per neighbour hit, L2-absorbed)
// + features out once + 4*27 B/site of rulebook.
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x4v = __attribute__((ext_vector_type(4))) int;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bufload1(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f32x4 bufload4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#define SESSD_OOB 0x80000000u

// NTW = 16-wide cout tiles per wave; blockIdx.y selects the cout group (COUT/16/NTW groups): splitting Cout over
// more waves fills the 1024 SIMDs when a level has fewer than 1024 site tiles (batch 1).
// DEPTH = operand register sets: the gathered rows and weights of DEPTH-1 offsets are in flight while one is multiplied.
//
// Everything that steers the walk over the tile's active offsets is WAVE-UNIFORM and kept in SGPRs (tile index and tile
// mask through readfirstlane): the offset loop is SALU + scalar branches, and the weight loads take their per-offset base
// as an SGPR offset. (Round 1 kept the mask in a VGPR: hipcc then wrapped every weight load in a waterfall loop and
// ended each iteration with s_waitcnt vmcnt(0) for the rotated index registers, i.e. no load ever overlapped an MFMA of
// the next step -- 25 us per 64->64 layer whatever the number of sites; rocprofv3 counters in profiles/r2_sparse_pmc.txt.)
// The tile's neighbour table (<= 27 x 16 row indices) is staged once in LDS, so that the per-step index fetch is a
// ds_read on its own counter (lgkmcnt) and never forces a wait on the operand loads (vmcnt counts in order).
//
// KSPLIT (offset split, for levels with fewer 16-site tiles than SIMDs): the four waves of a workgroup share ONE tile, wave w
// walks the active offsets k with k % 4 == w (at most 7 of 27 instead of up to 27 in sequence), the four partial tiles are added
// through LDS in wave order and wave w finishes rows 4 q + w. The per-site summation order is then
// ((S0 + S1) + S2) + S3 with S_w = the fmaf chain over the offsets of class w -- fixed by the offset index alone, so it does
// not depend on how sites fall into tiles, but it is NOT the single chain of the unsplit kernel (last-bit differences).
template <int CIN, int COUT, int NTW, int DEPTH, bool DENSE_OUT, bool KSPLIT = false>
__global__ __launch_bounds__(256) void sparse_conv_kernel(const float* __restrict__ in_feat,
                                                           const int* __restrict__ nbr,
                                                           const uint32_t* __restrict__ tile_mask, int kv,
                                                           const int* __restrict__ n_dev, int n_cap,
                                                           const float* __restrict__ wpk,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu,
                                                           float* __restrict__ out_feat,
                                                           const int* __restrict__ out_indices,
                                                           float* __restrict__ dense_out, int dD, int dH, int dW,
                                                           const uint8_t* __restrict__ perm) {
  // perm != NULL: OFFSET-PATTERN TILES (sessd_sparse_chain_rulebooks: sessd_rulebook_job_t.perm / tile_mask_sorted). Tile t holds
  // the sorted positions 16 t .. 16 t + 15 of its 256-row group; position p is row (t >> 4) * 256 + perm[16 t + p]. `tile_mask`
  // is then the mask array of THAT order. Everything per site is unchanged (same offsets, same fmaf chain): the same bits.
  constexpr int STEPS = CIN / 4;          // MFMA k-steps per offset
  constexpr int NTILE = NTW;              // 16-wide cout tiles handled by this wave
  constexpr int NTALL = COUT / 16;        // ... of all groups
  constexpr int G = STEPS < 4 ? STEPS : 4;  // floats per vector load
  constexpr int SG = STEPS / G;
  __shared__ int s_nbr[4][32][16];        // per wave: row index of (offset, site) -- 8 KB per workgroup
  // 1-D grid. Workgroup b runs on XCD b % 8 (observed dispatch order; only speed depends on it). The live 64-site tile
  // groups are dealt to the XCDs in runs of C consecutive groups: rows are numbered in (b,z,y,x) order, so a run is a
  // spatial neighbourhood whose gathered rows are mostly its own -- each XCD's L2 fetches its share of the feature table
  // instead of most of it (FETCH_SIZE of a 64->64 layer: 4.7 -> 2.9 MB at batch 1, profiles/) -- while runs, not whole
  // slabs, per XCD keep the dense regions of a scene from landing on one XCD (contiguous eighths: +25 % time at batch 1).
  // C grows with the level (1 below 512 groups: plain round robin, 8 from 4096 groups). Inside an XCD the cout groups of
  // a tile group are adjacent workgroups (they share the gathered rows) and the live workgroups come first, so the
  // dispatcher spreads them over all CUs. (With the cout group in blockIdx.y the live workgroups came as separate
  // bursts, each landing on the same few CUs: 4 waves per SIMD there, most CUs idle.)
  constexpr int NGRP = COUT / 16 / NTW;
  const int n = min(n_dev[0], n_cap);
  const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
  const int groups = KSPLIT ? ((n + 15) >> 4) : ((n + 63) >> 6);  // unit of the XCD mapping: a workgroup's sites
  const int lg = groups >= 4096 ? 3 : (groups >= 2048 ? 2 : (groups >= 512 ? 1 : 0));  // C = 1 << lg
  const int t_local = j / NGRP;
  const int group = ((((t_local >> lg) << 3) + xcd) << lg) + (t_local & ((1 << lg) - 1));
  if (group >= groups) return;
  const int tbase = (j - t_local * NGRP) * NTW;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile = KSPLIT ? group : group * 4 + wv;   // wave-uniform
  if (tile * 16 >= n) return;             // scalar branch; without KSPLIT every wave is independent (no workgroup barrier
                                          // below), with it the four waves of a workgroup leave together
  const int i = lane & 15, kq = lane >> 4;
  const uint32_t tmask_raw = tile_mask[tile];  // used (readfirstlane) after the neighbour rows below have been requested
  const int gbase = (tile >> 4) << 8;          // first row of the tile's 256-row group
  // row of position p of this tile (p < 16; positions beyond the live count map to the tile's first site: read, never written)
#define SESSD_ROW(P) (perm ? gbase + (int)perm[(size_t)tile * 16 + (P)] : tile * 16 + (P))

  f32x4 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // what only the epilogue needs is fetched NOW: loaded at their use, the BatchNorm constants (and the output coordinates of the
  // dense scatter) cost one more memory round trip at the end of a kernel that lasts 5 .. 20 us
  float scv[NTILE], shv[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    scv[t] = scale ? scale[co] : 1.f;
    shv[t] = shift ? shift[co] : 0.f;
  }
  int4 ocoord[DENSE_OUT ? 4 : 1];
  if constexpr (DENSE_OUT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = kq * 4 + r;
      ocoord[r] = *reinterpret_cast<const int4*>(out_indices + (size_t)(tile * 16 + p < n ? SESSD_ROW(p) : SESSD_ROW(0)) * 4);
    }
  }

  // stage the tile's neighbour table: lane (i, kq) fetches offsets kq, kq+4, ... of site i (64-byte segments).
  // lanes of the last tile whose site is >= n read the tile's first site instead (their results are discarded below):
  // with a capacity that is not a multiple of 16 their own column would lie past the end of the last rulebook row
  {
    const int* nb = nbr + (tile * 16 + i < n ? SESSD_ROW(i) : SESSD_ROW(0));
    int r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int k = p * 4 + kq;
      // every (offset, site < n) entry of the rulebook is written (-1 = no neighbour): the fetch does not wait for the tile mask
      r[p] = (k < kv && (!KSPLIT || (k & 3) == wv)) ? nb[(size_t)k * n_cap] : -1;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 8; ++p) s_nbr[wv][p * 4 + kq][i] = r[p];
  }
  const uint32_t tmask_all = __builtin_amdgcn_readfirstlane(tmask_raw);
  const uint32_t tmask = KSPLIT ? (tmask_all & (0x11111111u << wv)) : tmask_all;
  __builtin_amdgcn_wave_barrier();

  // Software pipeline over the ACTIVE offsets of this tile (bits of tmask). All operand loads are unconditional
  // (an exhausted list re-loads its last offset, a missing neighbour gets an out-of-range buffer offset and the hardware
  // returns zeros for its row), so the vmcnt of every wait is a compile-time constant. Buffer loads (SGPR resource +
  // 32-bit lane offset + SGPR offset): no 64-bit VALU address arithmetic in the loop (the f32 MFMA shares the SIMD
  // lanes with the VALU).
  const rsrc_t fr = make_rsrc(in_feat, 0x7FFFFFFFu);
  const rsrc_t wrs = make_rsrc(wpk, (unsigned)kv * NTALL * STEPS * 64u * 4u);
  float a[DEPTH][STEPS], bw[DEPTH][NTILE][STEPS];
  uint32_t rest = tmask;                        // SGPR
  const int remaining = __builtin_popcount(tmask);
  int kn = 0, rn = -1;
  // next active offset and this lane's input row for it; past the end of the list: the last offset again with "no
  // neighbour" (zeros), which the counted loop below multiplies harmlessly
#define SESSD_FETCH()                                        \
  {                                                          \
    const bool more = rest != 0u;                            \
    if (more) {                                              \
      kn = __builtin_ctz(rest);                              \
      rest &= rest - 1;                                      \
    }                                                        \
    const int rr = s_nbr[wv][kn][i];                         \
    rn = more ? rr : -1;                                     \
  }
#define SESSD_LOADAB(SET, K, ROW)                                                                  \
  {                                                                                                \
    const unsigned ao = (ROW) >= 0 ? (unsigned)(((ROW)*CIN + kq * STEPS) * 4) : SESSD_OOB;          \
    const unsigned ws = (unsigned)(K) * (NTALL * STEPS * 64 * 4);                                  \
    if (G == 4) {                                                                                  \
      _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                             \
        const f32x4 v = bufload4(fr, ao + 16u * g, 0);                                             \
        a[SET][4 * g] = v.x; a[SET][4 * g + 1] = v.y; a[SET][4 * g + 2] = v.z; a[SET][4 * g + 3] = v.w; \
      }                                                                                            \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                           \
          const f32x4 v = bufload4(wrs, (unsigned)lane * 16u + (unsigned)((tbase + t) * SG + g) * 1024u, ws); \
          bw[SET][t][4 * g] = v.x; bw[SET][t][4 * g + 1] = v.y; bw[SET][t][4 * g + 2] = v.z; bw[SET][t][4 * g + 3] = v.w; \
        }                                                                                          \
    } else {                                                                                       \
      _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2) a[SET][s2] = bufload1(fr, ao + 4u * s2, 0); \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                       \
          bw[SET][t][s2] = bufload1(wrs, ((unsigned)((tbase + t) * SG) * 64u + lane) * (G * 4u) + 4u * s2, ws); \
    }                                                                                              \
  }
#define SESSD_MMA(SET)                                                                             \
  {                                                                                                \
    _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                           \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SET][s2], bw[SET][t][s2], acc[t], 0, 0, 0); \
  }
  if (remaining > 0) {
    // fill DEPTH-1 sets, keep the (offset, row) of the next refill at hand
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) {
      SESSD_FETCH()
      SESSD_LOADAB(d, kn, rn)
      __builtin_amdgcn_sched_barrier(0);  // keep the sets in issue order: the loop's wait counts assume it
    }
    SESSD_FETCH()
    __builtin_amdgcn_sched_barrier(0);
    // one step: refill the set multiplied in the previous step with the offset at hand, fetch the next (offset, row),
    // multiply set CUR. The loop is COUNTED in whole rotations (the list is padded with zero rows to a multiple of DEPTH):
    // a straight-line body with one back edge is what keeps hipcc's s_waitcnt placement exact -- with early exits inside
    // the rotation it either waited for vmcnt(0) at the loop head or re-ordered the steps.
#define SESSD_STEP(CUR)                                                                            \
  {                                                                                                \
    SESSD_LOADAB((CUR + DEPTH - 1) % DEPTH, kn, rn)                                                \
    SESSD_FETCH()                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_MMA(CUR)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
    const int iters = (remaining + DEPTH - 1) / DEPTH;
    for (int it = 0; it < iters; ++it) {
      SESSD_STEP(0)
      SESSD_STEP(1)
      if constexpr (DEPTH >= 3) SESSD_STEP(2)
      if constexpr (DEPTH >= 4) SESSD_STEP(3)
    }
#undef SESSD_STEP
  }
#undef SESSD_FETCH
#undef SESSD_LOADAB
#undef SESSD_MMA

  // C/D layout: column (cout) = lane & 15, rows (sites) = (lane >> 4) * 4 + r
  constexpr int RS = NTILE * 16 + 4;      // row stride of the partial tiles in LDS: the four row groups of a wave on distinct banks
  __shared__ float s_red[KSPLIT ? 4 * 16 * RS : 1];
  if constexpr (KSPLIT) {
#pragma unroll
    for (int t = 0; t < NTILE; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_red[(wv * 16 + kq * 4 + r) * RS + t * 16 + i] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const float* p = &s_red[(kq * 4 + wv) * RS + t * 16 + i];
      acc[t][0] = ((p[0] + p[16 * RS]) + p[32 * RS]) + p[48 * RS];
    }
  }
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    const float sc = scv[t], sh = shv[t];
#pragma unroll
    for (int r = 0; r < (KSPLIT ? 1 : 4); ++r) {
      const int pos = kq * 4 + (KSPLIT ? wv : r);
      if (tile * 16 + pos >= n) continue;
      float v = fmaf(acc[t][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      if (DENSE_OUT) {
        // .dense() + view(N, C*D, H, W): channel = c*D + z  (scn.py:184-187)
        const int4 c = ocoord[DENSE_OUT ? r : 0];
        dense_out[(((size_t)c.x * COUT + co) * dD + c.y) * dH * dW + (size_t)c.z * dW + c.w] = v;
      } else {
        out_feat[(size_t)SESSD_ROW(pos) * COUT + co] = v;
      }
    }
  }
#undef SESSD_ROW
}

// Variant for levels with many tiles (tuning bits 20-21: TPW = 2 or 4 tiles per wave). Round 4 measured that a sparse layer's time
// does not follow its MFMA steps (offset-pattern tiles: 18 - 25 % fewer steps, +2 ... +8 % time, profiles/r4_sorted_tiles_probe.json):
// a wave's life is the dependent chain  tile mask + neighbour rows -> operand rows -> MFMAs -> stores,  paid once per 16-site tile.
// Here a wave takes TPW tiles (the same position of TPW consecutive 64-site groups) and fetches the NEXT tile's mask and neighbour
// rows right after publishing the current tile's table, i.e. under the current tile's whole MFMA walk; the BatchNorm constants are
// loaded once per wave. Per site the arithmetic is the plain kernel's: identical bits. Only for levels with many more tiles than
// wave slots (the dense-scene batch): at batch 1 it would thin out the already scarce waves.
template <int CIN, int COUT, int NTW, int DEPTH, int TPW>
__global__ __launch_bounds__(256) void sparse_conv_mt_kernel(const float* __restrict__ in_feat, const int* __restrict__ nbr,
                                                              const uint32_t* __restrict__ tile_mask, int kv,
                                                              const int* __restrict__ n_dev, int n_cap, const float* __restrict__ wpk,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              int relu, float* __restrict__ out_feat) {
  constexpr int STEPS = CIN / 4, NTILE = NTW, NTALL = COUT / 16;
  constexpr int G = STEPS < 4 ? STEPS : 4;
  constexpr int SG = STEPS / G;
  constexpr int NGRP = COUT / 16 / NTW;
  __shared__ int s_nbr[4][32][16];
  const int n = min(n_dev[0], n_cap);
  const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
  const int groups64 = (n + 63) >> 6;
  const int groups = (groups64 + TPW - 1) / TPW;   // unit of the XCD mapping: TPW consecutive 64-site groups
  const int lg = groups >= 4096 ? 3 : (groups >= 2048 ? 2 : (groups >= 512 ? 1 : 0));
  const int t_local = j / NGRP;
  const int group = ((((t_local >> lg) << 3) + xcd) << lg) + (t_local & ((1 << lg) - 1));
  if (group >= groups) return;
  const int tbase = (j - t_local * NGRP) * NTW;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int tile = (group * TPW) * 4 + wv;       // wave-uniform; the wave's later tiles are tile + 4, + 8, ...
  if (tile * 16 >= n) return;
  const int i = lane & 15, kq = lane >> 4;
  float scv[NTILE], shv[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    scv[t] = scale ? scale[co] : 1.f;
    shv[t] = shift ? shift[co] : 0.f;
  }
  const rsrc_t fr = make_rsrc(in_feat, 0x7FFFFFFFu);
  const rsrc_t wrs = make_rsrc(wpk, (unsigned)kv * NTALL * STEPS * 64u * 4u);
  // neighbour rows + mask of the first tile
  uint32_t mcur = tile_mask[tile];
  int rcur[8];
  {
    const int* nb = nbr + tile * 16 + (tile * 16 + i < n ? i : 0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int k = p * 4 + kq;
      rcur[p] = k < kv ? nb[(size_t)k * n_cap] : -1;
    }
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) s_nbr[wv][p * 4 + kq][i] = rcur[p];
  uint32_t tmask = __builtin_amdgcn_readfirstlane(mcur);
#pragma unroll 1
  for (int tt = 0; tt < TPW; ++tt) {
    __builtin_amdgcn_wave_barrier();
    // the NEXT tile's table: requested now, published (LDS) after this tile's MFMA walk and BEFORE its output stores -- a wait
    // placed after the stores would be a wait for the stores (unconditional loads: a wave without a next tile re-reads its own)
    const int tile_n = tile + 4;
    const bool more = (tt + 1 < TPW) && (tile_n * 16 < n);   // wave-uniform
    const int tile_f = more ? tile_n : tile;
    uint32_t mnext = tile_mask[tile_f];
    int rnext[8];
    {
      const int* nb = nbr + tile_f * 16 + (tile_f * 16 + i < n ? i : 0);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int k = p * 4 + kq;
        rnext[p] = k < kv ? nb[(size_t)k * n_cap] : -1;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[NTILE];
#pragma unroll
    for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[DEPTH][STEPS], bw[DEPTH][NTILE][STEPS];
    uint32_t rest = tmask;
    const int remaining = __builtin_popcount(tmask);
    int kn = 0, rn = -1;
#define SESSD_FETCH()                                        \
  {                                                          \
    const bool more_ = rest != 0u;                           \
    if (more_) {                                             \
      kn = __builtin_ctz(rest);                              \
      rest &= rest - 1;                                      \
    }                                                        \
    const int rr = s_nbr[wv][kn][i];                         \
    rn = more_ ? rr : -1;                                    \
  }
#define SESSD_LOADAB(SET, K, ROW)                                                                  \
  {                                                                                                \
    const unsigned ao = (ROW) >= 0 ? (unsigned)(((ROW)*CIN + kq * STEPS) * 4) : SESSD_OOB;          \
    const unsigned ws = (unsigned)(K) * (NTALL * STEPS * 64 * 4);                                  \
    if (G == 4) {                                                                                  \
      _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                             \
        const f32x4 v = bufload4(fr, ao + 16u * g, 0);                                             \
        a[SET][4 * g] = v.x; a[SET][4 * g + 1] = v.y; a[SET][4 * g + 2] = v.z; a[SET][4 * g + 3] = v.w; \
      }                                                                                            \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                           \
          const f32x4 v = bufload4(wrs, (unsigned)lane * 16u + (unsigned)((tbase + t) * SG + g) * 1024u, ws); \
          bw[SET][t][4 * g] = v.x; bw[SET][t][4 * g + 1] = v.y; bw[SET][t][4 * g + 2] = v.z; bw[SET][t][4 * g + 3] = v.w; \
        }                                                                                          \
    } else {                                                                                       \
      _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2) a[SET][s2] = bufload1(fr, ao + 4u * s2, 0); \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                       \
          bw[SET][t][s2] = bufload1(wrs, ((unsigned)((tbase + t) * SG) * 64u + lane) * (G * 4u) + 4u * s2, ws); \
    }                                                                                              \
  }
#define SESSD_MMA(SET)                                                                             \
  {                                                                                                \
    _Pragma("unroll") for (int s2 = 0; s2 < STEPS; ++s2)                                           \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SET][s2], bw[SET][t][s2], acc[t], 0, 0, 0); \
  }
    if (remaining > 0) {
#pragma unroll
      for (int d = 0; d < DEPTH - 1; ++d) {
        SESSD_FETCH()
        SESSD_LOADAB(d, kn, rn)
        __builtin_amdgcn_sched_barrier(0);
      }
      SESSD_FETCH()
      __builtin_amdgcn_sched_barrier(0);
#define SESSD_STEP(CUR)                                                                            \
  {                                                                                                \
    SESSD_LOADAB((CUR + DEPTH - 1) % DEPTH, kn, rn)                                                \
    SESSD_FETCH()                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_MMA(CUR)                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  }
      const int iters = (remaining + DEPTH - 1) / DEPTH;
      for (int it = 0; it < iters; ++it) {
        SESSD_STEP(0)
        SESSD_STEP(1)
        if constexpr (DEPTH >= 3) SESSD_STEP(2)
        if constexpr (DEPTH >= 4) SESSD_STEP(3)
      }
#undef SESSD_STEP
    }
#undef SESSD_FETCH
#undef SESSD_LOADAB
#undef SESSD_MMA
    // this tile's table has been read for the last time: publish the next one
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 8; ++p) s_nbr[wv][p * 4 + kq][i] = rnext[p];
    const uint32_t tmask_n = __builtin_amdgcn_readfirstlane(mnext);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
      const int co = (tbase + t) * 16 + i;
      const float sc = scv[t], sh = shv[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int site = tile * 16 + kq * 4 + r;
        if (site >= n) continue;
        float v = fmaf(acc[t][r], sc, sh);
        if (relu) v = fmaxf(v, 0.f);
        out_feat[(size_t)site * COUT + co] = v;
      }
    }
    if (!more) break;
    tmask = tmask_n;
    tile = tile_n;
  }
}

// Variant for levels with many tiles (tuning bit 17): the four site tiles of a workgroup walk the UNION of their active offsets in
// step and share W[k] through LDS. Without it every wave streams its own copy of W[k] (Cin x Cout/split floats per 16-site tile
// and offset: 2/3 of the operand bytes), and at scale the kernel is bound by that L1/L2 traffic, not by the matrix cores
// (45 % of the f32 MFMA peak on the dense-scene batch). Per step: all threads fetch W[k_next] (16-byte loads, the packed fragment
// order is kept) and each wave the rows of its own tile for k_next, the MFMAs of offset k take B from the LDS copy staged during
// the previous step, then the fetched W goes to the other LDS buffer; one barrier per offset. A wave whose tile has no
// neighbour at offset k multiplies zero rows in that step (it waits for the others at the barrier anyway; a branch around the
// MFMAs makes hipcc shuttle the accumulators between AGPRs and VGPRs). Same (offset, cin) fmaf chain per site as the plain
// kernel: identical bits.
template <int CIN, int COUT, int NTW>
__global__ __launch_bounds__(256) void sparse_conv_wshare_kernel(const float* __restrict__ in_feat, const int* __restrict__ nbr,
                                                                  const uint32_t* __restrict__ tile_mask, int kv,
                                                                  const int* __restrict__ n_dev, int n_cap,
                                                                  const float* __restrict__ wpk, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int relu,
                                                                  float* __restrict__ out_feat, const uint8_t* __restrict__ perm) {
  constexpr int STEPS = CIN / 4, NTILE = NTW, NTALL = COUT / 16, SG = STEPS / 4;
  static_assert(CIN % 16 == 0, "16-byte operand loads");
  constexpr int WFLOATS = NTILE * CIN * 16;   // W[k] of this workgroup's couts, in fragment order
  constexpr int WVEC = WFLOATS / 1024;        // 16-byte loads per thread and offset
  static_assert(WVEC >= 1, "at least one 16-byte load per thread");
  constexpr int NGRP = COUT / 16 / NTW;
  __shared__ __attribute__((aligned(16))) float s_w[2][WFLOATS];
  __shared__ int s_nbr[4][32][16];
  const int n = min(n_dev[0], n_cap);
  const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
  const int groups = (n + 63) >> 6;
  const int lg = groups >= 4096 ? 3 : (groups >= 2048 ? 2 : (groups >= 512 ? 1 : 0));
  const int t_local = j / NGRP;
  const int group = ((((t_local >> lg) << 3) + xcd) << lg) + (t_local & ((1 << lg) - 1));
  if (group >= groups) return;
  const int tbase = (j - t_local * NGRP) * NTW;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile = group * 4 + wv;
  const bool live = tile * 16 < n;          // wave-uniform; dead waves still take part in the staging and the barriers
  const int i = lane & 15, kq = lane >> 4;
  const int ntiles = (n + 15) >> 4;
  uint32_t un = 0, mine = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t m = (group * 4 + q < ntiles) ? __builtin_amdgcn_readfirstlane(tile_mask[group * 4 + q]) : 0u;
    un |= m;
    if (q == wv) mine = m;
  }
  mine = __builtin_amdgcn_readfirstlane(mine);
  float scv[NTILE], shv[NTILE];  // fetched now, used by the epilogue (see sparse_conv_kernel)
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    scv[t] = scale ? scale[co] : 1.f;
    shv[t] = shift ? shift[co] : 0.f;
  }
  f32x4 acc[NTILE];
#pragma unroll
  for (int t = 0; t < NTILE; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int gbase = (tile >> 4) << 8;   // offset-pattern tiles: see sparse_conv_kernel
#define SESSD_ROW(P) (perm ? gbase + (int)perm[(size_t)tile * 16 + (P)] : tile * 16 + (P))
  if (live) {
    const int* nb = nbr + (tile * 16 + i < n ? SESSD_ROW(i) : SESSD_ROW(0));
    int r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int k = p * 4 + kq;
      r[p] = (k < kv && live) ? nb[(size_t)k * n_cap] : -1;  // does not wait for the tile masks (see sparse_conv_kernel)
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) s_nbr[wv][p * 4 + kq][i] = r[p];
  }
  __builtin_amdgcn_wave_barrier();
  const rsrc_t fr = make_rsrc(in_feat, 0x7FFFFFFFu);
  const rsrc_t wrs = make_rsrc(wpk, (unsigned)kv * NTALL * STEPS * 64u * 4u);
  const unsigned wlane = (unsigned)tid * 16u + (unsigned)(tbase * SG) * 1024u;  // this thread's 16 bytes of the W[k] block
  f32x4 wreg[WVEC];
  float a[2][STEPS];
#define SESSD_WS_LOADW(K)                                                                          \
  _Pragma("unroll") for (int p = 0; p < WVEC; ++p)                                                 \
    wreg[p] = bufload4(wrs, wlane + 4096u * p, (unsigned)(K) * (NTALL * STEPS * 64 * 4));
#define SESSD_WS_LOADA(SET, K)                                                                     \
  {                                                                                                \
    const int row = (live && ((mine >> (K)) & 1u)) ? s_nbr[wv][(K)][i] : -1;                       \
    const unsigned ao = row >= 0 ? (unsigned)((row * CIN + kq * STEPS) * 4) : SESSD_OOB;           \
    _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                               \
      const f32x4 v = bufload4(fr, ao + 16u * g, 0);                                               \
      a[SET][4 * g] = v.x; a[SET][4 * g + 1] = v.y; a[SET][4 * g + 2] = v.z; a[SET][4 * g + 3] = v.w; \
    }                                                                                              \
  }
#define SESSD_WS_STAGE(BUF)                                                                        \
  _Pragma("unroll") for (int p = 0; p < WVEC; ++p)                                                 \
    *reinterpret_cast<f32x4*>(&s_w[BUF][(p * 256 + tid) * 4]) = wreg[p];
#define SESSD_WS_MMA(SET, BUF, K)                                                                  \
  {                                                                                                \
    _Pragma("unroll") for (int g = 0; g < SG; ++g) {                                               \
      f32x4 b[NTILE];                                                                              \
      _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                            \
        b[t] = *reinterpret_cast<const f32x4*>(&s_w[BUF][((t * SG + g) * 64 + lane) * 4]);         \
      /* cout tiles innermost: consecutive MFMAs go to different accumulators (same order per accumulator) */ \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                \
        _Pragma("unroll") for (int t = 0; t < NTILE; ++t)                                          \
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[SET][4 * g + e], b[t][e], acc[t], 0, 0, 0); \
    }                                                                                              \
  }
  // one step: fetch the operands of the next offset, multiply the current one, publish the fetched W
#define SESSD_WS_STEP(SET)                                                                         \
  {                                                                                                \
    const bool more = rest != 0u;                                                                  \
    const int kn = more ? __builtin_ctz(rest) : k;                                                 \
    rest &= rest - 1;                                                                              \
    SESSD_WS_LOADW(kn)                                                                             \
    SESSD_WS_LOADA((SET) ^ 1, kn)                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WS_MMA(SET, SET, k)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    SESSD_WS_STAGE((SET) ^ 1)                                                                      \
    __syncthreads();                                                                               \
    k = kn;                                                                                        \
  }
  if (un != 0u) {
    uint32_t rest = un;
    int k = __builtin_ctz(rest);
    rest &= rest - 1;
    const int steps = __builtin_popcount(un);
    SESSD_WS_LOADW(k)
    SESSD_WS_LOADA(0, k)
    SESSD_WS_STAGE(0)
    __syncthreads();
    for (int it = 0; it < steps; it += 2) {
      SESSD_WS_STEP(0)
      if (it + 1 >= steps) break;
      SESSD_WS_STEP(1)
    }
  }
#undef SESSD_WS_LOADW
#undef SESSD_WS_LOADA
#undef SESSD_WS_STAGE
#undef SESSD_WS_MMA
#undef SESSD_WS_STEP
  if (!live) return;
#pragma unroll
  for (int t = 0; t < NTILE; ++t) {
    const int co = (tbase + t) * 16 + i;
    const float sc = scv[t], sh = shv[t];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int pos = kq * 4 + r;
      if (tile * 16 + pos >= n) continue;
      float v = fmaf(acc[t][r], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      out_feat[(size_t)SESSD_ROW(pos) * COUT + co] = v;
    }
  }
#undef SESSD_ROW
}

// W (kv, cin, cout) row-major [the flattened spconv layout (kz,ky,kx,Cin,Cout)] -> fragment order
//   wpk[(((k*NTILE + t)*SG + g)*64 + lane)*G + e] = W[k][ (lane>>4)*STEPS + g*G + e ][ t*16 + (lane&15) ]
// (kv, cin, cout) are those of the PACKED conv. adjoint = 0: w is that conv's own (kv, cin, cout) tensor. adjoint = 1: w is the
// (kv, cout, cin) tensor of the layer whose DATA GRADIENT the packed conv computes (its input channels are that layer's output
// channels): element (k, ci, co) = w[k'][co][ci], k' = kv - 1 - k with reverse_k (a submanifold layer's gradient runs on the
// forward tables with the offsets reversed), else k' = k (strided layers: on the transposed rulebook).
__device__ __forceinline__ void pack_weight_body(const float* __restrict__ w, int kv, int cin, int cout, int adjoint, int reverse_k,
                                                 float* __restrict__ wpk, size_t idx) {
  const int steps = cin / 4, ntile = cout / 16, G = steps < 4 ? steps : 4, SG = steps / G;
  const size_t total = (size_t)kv * cin * cout;
  if (idx >= total) return;
  int e = idx % G;
  size_t r = idx / G;
  int lane = r % 64; r /= 64;
  int g = r % SG; r /= SG;
  int t = r % ntile;
  int k = (int)(r / ntile);
  int ci = (lane >> 4) * steps + g * G + e, co = t * 16 + (lane & 15);
  const int ks = reverse_k ? kv - 1 - k : k;
  wpk[idx] = adjoint ? w[((size_t)ks * cout + co) * cin + ci] : w[((size_t)ks * cin + ci) * cout + co];
}
__global__ void pack_weight_kernel(const float* __restrict__ w, int kv, int cin, int cout, int adjoint, int reverse_k,
                                   float* __restrict__ wpk) {
  pack_weight_body(w, kv, cin, cout, adjoint, reverse_k, wpk, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// every sparse weight packing of an iteration in one launch (jobs: sessd_hip_types.h)
__global__ __launch_bounds__(256) void sparse_pack_batch_kernel(const sessd_sparse_pack_job_t* __restrict__ jobs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const sessd_sparse_pack_job_t* J = jobs + lo;
  const size_t idx = (size_t)((int)blockIdx.x - J->block_start) * 256 + threadIdx.x;
  // the packed conv's (cin, cout): swapped for the data-gradient layer
  if (J->adjoint)
    pack_weight_body(J->w, J->kernel_volume, J->cout, J->cin, 1, J->reverse_k, J->out, idx);
  else
    pack_weight_body(J->w, J->kernel_volume, J->cin, J->cout, 0, 0, J->out, idx);
}

template <int CIN, int COUT, int NTW, int DEPTH, bool KS = false>
int launch_depth(bool dense, const float* in_feat, const int* nbr, const uint32_t* tile_mask, int kv, const int* n_dev,
                 int n_cap, const float* wpk, const float* scale, const float* shift, int relu, float* out_feat,
                 const int* out_indices, float* dense_out, const int* dd, const uint8_t* perm, hipStream_t stream) {
  const int tiles = sessd_divup(n_cap, 16);
  // per XCD: ceil(groups / 8C) runs of C groups; C <= 8, so ceil(groups / 8) + 8 positions always suffice
  dim3 grid(8 * (sessd_divup(KS ? tiles : sessd_divup(tiles, 4), 8) + 8) * (COUT / 16 / NTW)), block(256);
  if constexpr (KS) {
    SESSD_LAUNCH((sparse_conv_kernel<CIN, COUT, NTW, DEPTH, false, true>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv,
                       n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, 0, 0, 0, perm);
    SESSD_CHECK_LAUNCH();
    return SESSD_OK;
  }
  if (dense)
    SESSD_LAUNCH((sparse_conv_kernel<CIN, COUT, NTW, DEPTH, true>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv,
                       n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, dd[0], dd[1], dd[2], perm);
  else
    SESSD_LAUNCH((sparse_conv_kernel<CIN, COUT, NTW, DEPTH, false>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv,
                       n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, 0, 0, 0, perm);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// operand sets (A: CIN/4 registers, B: NTW * CIN/4) of `depth` offsets must fit the 512-entry register file with room
// for the accumulators and addresses; deeper than that is clamped
template <int CIN, int COUT, int NTW, bool KS = false>
int launch_ntw(int depth, bool dense, const float* in_feat, const int* nbr, const uint32_t* tile_mask, int kv,
               const int* n_dev, int n_cap, const float* wpk, const float* scale, const float* shift, int relu,
               float* out_feat, const int* out_indices, float* dense_out, const int* dd, const uint8_t* perm, hipStream_t stream) {
  constexpr int SET = (CIN / 4) * (1 + NTW);
  constexpr int DMAX = SET * 4 <= 400 ? 4 : (SET * 3 <= 400 ? 3 : 2);
  if (depth <= 0) depth = 3;
  if (depth > DMAX) depth = DMAX;
#define SESSD_ARGS2 dense, in_feat, nbr, tile_mask, kv, n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, dd, perm, stream
  if constexpr (KS) {  // offset split: at most 7 offsets per wave -- two or three operand sets
    if constexpr (DMAX >= 3) {
      if (depth >= 3) return launch_depth<CIN, COUT, NTW, 3, true>(SESSD_ARGS2);
    }
    return launch_depth<CIN, COUT, NTW, 2, true>(SESSD_ARGS2);
  }
  if constexpr (DMAX >= 4) {
    if (depth >= 4) return launch_depth<CIN, COUT, NTW, 4>(SESSD_ARGS2);
  }
  if constexpr (DMAX >= 3) {
    if (depth == 3) return launch_depth<CIN, COUT, NTW, 3>(SESSD_ARGS2);
  }
  return launch_depth<CIN, COUT, NTW, 2>(SESSD_ARGS2);
#undef SESSD_ARGS2
}

template <int CIN, int COUT>
int launch(int tuning, bool dense, const float* in_feat, const int* nbr, const uint32_t* tile_mask, int kv, const int* n_dev,
           int n_cap, const float* wpk, const float* scale, const float* shift, int relu, float* out_feat,
           const int* out_indices, float* dense_out, const int* dd, const uint8_t* perm, hipStream_t stream) {
  constexpr int NT = COUT / 16;
  int split = tuning & 0xFF;
  const int depth = (tuning >> 8) & 0xFF;
  if (split <= 0) split = (n_cap / 16 < 4096) ? (NT >= 4 ? 4 : (NT >= 2 ? 2 : 1)) : 1;  // fill 1024 SIMDs on small levels
  const bool ksplit = ((tuning >> 16) & 1) && !dense;
  const int tpw_sel = (tuning >> 20) & 3;   // 1: two, 2: four tiles per wave (sparse_conv_mt_kernel); plain tiles, not dense / split / shared-W
  if (tpw_sel && !dense && !ksplit && !((tuning >> 17) & 1) && !perm) {
    const int tiles = sessd_divup(n_cap, 16);
    int depth_mt = depth <= 0 ? 3 : depth;
#define SESSD_MT(NTW_, DEPTH_, TPW_)                                                                                        \
    {                                                                                                                       \
      dim3 grid(8 * (sessd_divup(sessd_divup(sessd_divup(tiles, 4), (TPW_)), 8) + 8) * (NT / (NTW_))), block(256);           \
      SESSD_LAUNCH((sparse_conv_mt_kernel<CIN, COUT, (NTW_), (DEPTH_), (TPW_)>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv, \
                   n_dev, n_cap, wpk, scale, shift, relu, out_feat);                                                        \
      SESSD_CHECK_LAUNCH();                                                                                                 \
      return SESSD_OK;                                                                                                      \
    }
#define SESSD_MT_NTW(NTW_)                                                                   \
    {                                                                                        \
      constexpr int SET_ = (CIN / 4) * (1 + (NTW_));                                         \
      if (depth_mt >= 3 && SET_ * 3 <= 400) {                                                \
        if (tpw_sel == 1) SESSD_MT(NTW_, 3, 2) else SESSD_MT(NTW_, 3, 4)                     \
      } else {                                                                               \
        if (tpw_sel == 1) SESSD_MT(NTW_, 2, 2) else SESSD_MT(NTW_, 2, 4)                     \
      }                                                                                      \
    }
    if constexpr (NT % 4 == 0) {
      if (split >= 4) SESSD_MT_NTW(NT / 4)
    }
    if constexpr (NT % 2 == 0) {
      if (split >= 2) SESSD_MT_NTW(NT / 2)
    }
    SESSD_MT_NTW(NT)
#undef SESSD_MT_NTW
#undef SESSD_MT
  }
  if (((tuning >> 17) & 1) && !dense) {
    // shared-W variant: needs 16-byte A loads and at least one 16-byte W load per thread
    const int tiles = sessd_divup(n_cap, 16);
#define SESSD_WSHARE(NTW_)                                                                                            \
    if constexpr (CIN % 16 == 0 && (NTW_) * CIN >= 64) {                                                              \
      dim3 grid(8 * (sessd_divup(sessd_divup(tiles, 4), 8) + 8) * (NT / (NTW_))), block(256);                          \
      SESSD_LAUNCH((sparse_conv_wshare_kernel<CIN, COUT, (NTW_)>), grid, block, 0, stream, in_feat, nbr, tile_mask, kv, \
                   n_dev, n_cap, wpk, scale, shift, relu, out_feat, perm);                                            \
      SESSD_CHECK_LAUNCH();                                                                                           \
      return SESSD_OK;                                                                                                \
    }
    if constexpr (NT % 4 == 0) {
      if (split >= 4) { SESSD_WSHARE(NT / 4) }
    }
    if constexpr (NT % 2 == 0) {
      if (split >= 2) { SESSD_WSHARE(NT / 2) }
    }
    { SESSD_WSHARE(NT) }
#undef SESSD_WSHARE
    // not instantiable for this channel pair / split: the plain kernel (same bits)
  }
#define SESSD_ARGS depth, dense, in_feat, nbr, tile_mask, kv, n_dev, n_cap, wpk, scale, shift, relu, out_feat, out_indices, dense_out, dd, perm, stream
  if (ksplit) {
    if constexpr (NT % 4 == 0) {
      if (split >= 4) return launch_ntw<CIN, COUT, NT / 4, true>(SESSD_ARGS);
    }
    if constexpr (NT % 2 == 0) {
      if (split >= 2) return launch_ntw<CIN, COUT, NT / 2, true>(SESSD_ARGS);
    }
    return launch_ntw<CIN, COUT, NT, true>(SESSD_ARGS);
  }
  if constexpr (NT % 4 == 0) {
    if (split >= 4) return launch_ntw<CIN, COUT, NT / 4>(SESSD_ARGS);
  }
  if constexpr (NT % 2 == 0) {
    if (split >= 2) return launch_ntw<CIN, COUT, NT / 2>(SESSD_ARGS);
  }
  return launch_ntw<CIN, COUT, NT>(SESSD_ARGS);
#undef SESSD_ARGS
}

}  // namespace

extern "C" {

int sessd_sparse_pack_weight(const float* weight, int kernel_volume, int cin, int cout, float* packed,
                             hipStream_t stream) {
  if (cin % 4 || cout % 16 || kernel_volume <= 0) return SESSD_EINVAL;
  const int steps = cin / 4;
  if (steps > 4 && steps % 4) return SESSD_EINVAL;
  size_t total = (size_t)kernel_volume * cin * cout;
  SESSD_LAUNCH(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, weight,
                     kernel_volume, cin, cout, 0, 0, packed);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_sparse_pack_batch(const sessd_sparse_pack_job_t* jobs_dev, int n_jobs, int total_blocks, hipStream_t stream) {
  if (!jobs_dev || n_jobs < 1 || total_blocks < 1) return SESSD_EINVAL;
  SESSD_LAUNCH(sparse_pack_batch_kernel, dim3(total_blocks), dim3(256), 0, stream, jobs_dev, n_jobs);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// The packed weight of the conv that computes a layer's DATA GRADIENT, straight from the layer's own weight (kernel_volume, cin,
// cout): a (cout -> cin) conv with W'[k] = W[k']^T, k' = kernel_volume - 1 - k when reverse_offsets (submanifold layers: the
// forward tables read with the offsets reversed), else k (strided layers: on the transposed rulebook). Replaces a flip, a
// transpose copy and a pack per layer and iteration. cout % 4 == 0, cin % 16 == 0.
int sessd_sparse_pack_weight_adjoint(const float* weight, int kernel_volume, int cin, int cout, int reverse_offsets, float* packed,
                                     hipStream_t stream) {
  if (cout % 4 || cin % 16 || kernel_volume <= 0) return SESSD_EINVAL;
  const int steps = cout / 4;
  if (steps > 4 && steps % 4) return SESSD_EINVAL;
  size_t total = (size_t)kernel_volume * cin * cout;
  SESSD_LAUNCH(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, weight, kernel_volume, cout, cin, 1,
               reverse_offsets ? 1 : 0, packed);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// tuning = cout_split + 256 * depth + 65536 * offset_split. cout_split: 0 = heuristic, 1/2/4 = number of waves that share one
// 16-site tile (each takes Cout/split channels); depth: 0 = default (3), 2..4 = operand register sets (offsets in flight + the
// one multiplied). Results do not depend on either. offset_split = 1 (ignored with dense_out): the four waves of a workgroup
// split the kernel offsets of one tile by k % 4 and add their partial tiles -- for levels with fewer tiles than SIMDs; its
// results are the same for every cout_split / depth but differ in the last bits from offset_split = 0 (four partial chains
// added instead of one chain). Bit 17 (131072; ignored with dense_out and where the shape does not allow it): the four tiles of a
// workgroup walk the union of their offsets and share W[k] through LDS -- for levels with many tiles; same bits as the plain kernel.
// out[o] = act( (sum_k W[k]^T in[nbr[k][o]]) * scale + shift ). If dense_out != NULL the result is
// scattered instead into the dense BEV tensor (B, cout*D, H, W) with dense_dims3 = (D,H,W) (pre-zeroed
// by the caller) and out_feat may be NULL.
// perm != NULL: offset-pattern tiles -- `tile_mask` is the job's tile_mask_sorted, `perm` its position -> row table
// (sessd_rulebook_job_t); NULL: tiles of 16 consecutive rows. Same results either way.
int sessd_sparse_conv_sorted(const float* in_feat, int cin, const int* nbr, const uint32_t* tile_mask, int kernel_volume,
                             const int* n_out_dev, int n_out_cap, const float* packed_weight, const float* scale,
                             const float* shift, int relu, float* out_feat, int cout, const int* out_indices,
                             float* dense_out, const int* dense_dims3, int tuning, const uint8_t* perm, hipStream_t stream) {
  if (n_out_cap <= 0 || kernel_volume <= 0 || kernel_volume > 32) return SESSD_EINVAL;
  const bool dense = dense_out != nullptr;
  if (dense && (!out_indices || !dense_dims3)) return SESSD_EINVAL;
  if (!dense && !out_feat) return SESSD_EINVAL;
#define SESSD_SC(CI, CO)                                                                                              \
  if (cin == CI && cout == CO)                                                                                        \
    return launch<CI, CO>(tuning, dense, in_feat, nbr, tile_mask, kernel_volume, n_out_dev, n_out_cap, packed_weight, scale,  \
                          shift, relu, out_feat, out_indices, dense_out, dense_dims3, perm, stream);
  SESSD_SC(4, 16)
  SESSD_SC(16, 16)
  SESSD_SC(16, 32)
  SESSD_SC(32, 32)
  SESSD_SC(32, 64)
  SESSD_SC(64, 64)
  SESSD_SC(8, 16)
  SESSD_SC(16, 64)
  SESSD_SC(64, 128)
  SESSD_SC(128, 128)
  SESSD_SC(32, 16)  // data-gradient shapes of the 16->32 and 32->64 strided convs
  SESSD_SC(64, 32)
#undef SESSD_SC
  return SESSD_EINVAL;  // channel pair not instantiated
}

int sessd_sparse_conv(const float* in_feat, int cin, const int* nbr, const uint32_t* tile_mask, int kernel_volume,
                      const int* n_out_dev, int n_out_cap, const float* packed_weight, const float* scale,
                      const float* shift, int relu, float* out_feat, int cout, const int* out_indices,
                      float* dense_out, const int* dense_dims3, int tuning, hipStream_t stream) {
  return sessd_sparse_conv_sorted(in_feat, cin, nbr, tile_mask, kernel_volume, n_out_dev, n_out_cap, packed_weight, scale, shift, relu,
                                  out_feat, cout, out_indices, dense_out, dense_dims3, tuning, nullptr, stream);
}

}  // extern "C"
