// Active sites and rulebooks of the WHOLE strided chain of SpMiddleFHD: 3 + levels small launches (gfx950).
// Replaces what the reference gets from spconv's get_indice_pairs, once per SparseConv3d / per indice_key:
//   det3d/models/backbones/scn.py:106-148 (four SparseConv3d, four groups of SubMConv3d), :179-183.
// Semantics (restated in oracle/sparse_conv.py, pinned against F.conv3d): an output cell of SparseConv3d(k, s, p) is
// active iff an active input lies in its receptive field; SubMConv3d keeps the site set.
//
// The site set of every level depends on the level-0 sites only (geometry, no features). Every level >= 1 gets an
// occupancy BIT MAP over its dense grid (cell index ((b*D+z)*H+y)*W+x), stored interleaved with the rank of each word's
// first cell, occ[w] = {bits, rank}:
//   mark      level 1 from the level-0 sites: <= 2x2x2 cells per site, the words are loaded first (one memory latency),
//             then only missing bits are set with atomicOr
//   gather    level l >= 2 from the MAP of level l-1 (one launch per level), no atomics on contended words: one thread per
//             <= 32 output cells of a grid row ORs the <= kz*ky input rows of its receptive field (128-bit windows,
//             shift-OR over the kernel width, every s-th bit sampled). Measured alternatives: marking every level from
//             the level-0 sites in one kernel serialises on same-address atomics (hundreds of sites hit each word of the
//             coarse levels: 33 us for 15 k sites, 440 us for the dense-scene batch); gathering all deeper levels
//             from level 1 in one launch makes threads walk 49 / 105 rows each (57 us).
//   emit      exclusive prefix of the popcounts = row number of every active cell; rows are numbered in ascending (b,z,y,x)
//             order: deterministic (no atomic decides a number), and 16 consecutive rows are spatial neighbours -- what the
//             16-site MFMA tiles of sparse_conv.hip want (fewer distinct offsets per tile). A popcount pass gives the cells per
//             block (count) and every emit block sums its level's earlier blocks itself (round 2: a separate scan launch).
//             Measured and dropped: counting the bits where they are set (atomicOr returns the old word, the fresh bits go to
//             the block's counter with an atomicAdd) -- exact, one launch less, but ~10^5 atomics on a few hundred counters:
//             chain_mark 6.8 -> 45.7 us, the gathers 8 -> 14 us
//   rulebooks ALL neighbour tables of the chain in one launch (submanifold table of each level + the strided table
//             into the next), nbr[k][o] + per-16-site tile masks exactly as sessd_sparse_rulebook builds them. A lookup
//             "cell -> row or -1" is ONE 8-byte load + popcount and the x-neighbours of a window share a word; all loads
//             of a lane are issued before the first is used (round 1: a hash probe per neighbour -- two dependent random
//             loads each, nine in sequence; site generation by CAS insertion, one launch per level).
// Level 0 keeps the voxelizer's hash (its rows are the voxels, in the reference's first-come order).
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

constexpr int NT = 256;
constexpr int WPT = 4;              // words per thread in the scan
constexpr int SPAN = NT * WPT;      // words per scan block; level ranges are padded to multiples of it
constexpr int MAXLEV = 6;
constexpr int MAXJOB = 12;
typedef unsigned __int128 u128;

struct LevelDev {
  int ks[3], st[3], pd[3];
  int dims[3];       // spatial shape of this level
  int cap;
  int blk_off;       // first scan block of this level
  int n_blk;
  int* indices;
  int* n_dev;
  // gather geometry (levels >= 1 of the chain array, i.e. level numbers >= 2)
  int src;           // chain index of the level whose map is gathered from
  int seg_len;       // output cells per thread (<= 32)
  int nseg;          // segments per grid row
  int thr_off;       // first gather thread of this level
};
struct ChainDev {
  int nlev, batch;
  LevelDev L[MAXLEV];
};

__device__ __forceinline__ void reach(int lo_in, int hi_in, int k, int s, int p, int dim, int& lo, int& hi) {
  // outputs q with 0 <= i + p - q*s <= k-1 for some i in [lo_in, hi_in]
  const int a = lo_in + p - k + 1;
  lo = a <= 0 ? 0 : (a + s - 1) / s;
  hi = (hi_in + p) / s;  // hi_in + p >= 0
  if (hi > dim - 1) hi = dim - 1;
}

// level 1 (chain index 0) from the level-0 sites
__global__ __launch_bounds__(NT) void chain_mark_kernel(const int* __restrict__ indices0, const int* __restrict__ n0_dev,
                                                         int n0_cap, ChainDev C, uint2* __restrict__ occ) {
  const int i = blockIdx.x * NT + threadIdx.x;
  const int n = min(n0_dev[0], n0_cap);
  if (i >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(indices0 + (size_t)i * 4);
  const LevelDev& L = C.L[0];
  int lo[3], hi[3];
  reach(c.y, c.y, L.ks[0], L.st[0], L.pd[0], L.dims[0], lo[0], hi[0]);
  reach(c.z, c.z, L.ks[1], L.st[1], L.pd[1], L.dims[1], lo[1], hi[1]);
  reach(c.w, c.w, L.ks[2], L.st[2], L.pd[2], L.dims[2], lo[2], hi[2]);
  if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) return;
  uint2* o = occ + (size_t)L.blk_off * SPAN;
  const int nz = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nx = hi[2] - lo[2] + 1;
  if (nz <= 3 && ny <= 3 && nx <= 3) {
    unsigned word[9][2], mask[9][2], have[9][2];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int dz = r / 3, dy = r - dz * 3;
      const bool ok = dz < nz && dy < ny;
      const unsigned cell = (unsigned)(((c.x * L.dims[0] + lo[0] + dz) * L.dims[1] + lo[1] + dy) * L.dims[2] + lo[2]);
      const unsigned w0 = cell >> 5, b0 = cell & 31u;
      const unsigned long long m = (unsigned long long)((1u << nx) - 1u) << b0;  // nx <= 3 bits, may straddle two words
      word[r][0] = w0; word[r][1] = w0 + 1;
      mask[r][0] = ok ? (unsigned)m : 0u;
      mask[r][1] = ok ? (unsigned)(m >> 32) : 0u;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      have[r][0] = mask[r][0] ? o[word[r][0]].x : 0xFFFFFFFFu;
      have[r][1] = mask[r][1] ? o[word[r][1]].x : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {  // a stale read only costs a redundant atomic
      if (mask[r][0] & ~have[r][0]) atomicOr(&o[word[r][0]].x, mask[r][0]);
      if (mask[r][1] & ~have[r][1]) atomicOr(&o[word[r][1]].x, mask[r][1]);
    }
    return;
  }
  for (int z = lo[0]; z <= hi[0]; ++z)
    for (int y = lo[1]; y <= hi[1]; ++y) {
      const unsigned base = (unsigned)(((c.x * L.dims[0] + z) * L.dims[1] + y) * L.dims[2]);
      for (int x = lo[2]; x <= hi[2]; ++x) {
        const unsigned cell = base + (unsigned)x;
        const unsigned m = 1u << (cell & 31u);
        if (!(o[cell >> 5].x & m)) atomicOr(&o[cell >> 5].x, m);
      }
    }
}

// levels l_first..l_last (chain indices >= 1) from the map of their `src` level
__global__ __launch_bounds__(NT) void chain_gather_kernel(ChainDev C, int l_first, int l_last, int thr_base,
                                                           uint2* __restrict__ occ) {
  const int t = thr_base + blockIdx.x * NT + threadIdx.x;
  int l = l_first;
#pragma unroll
  for (int q = 1; q < MAXLEV; ++q)
    if (q > l_first && q <= l_last && t >= C.L[q].thr_off) l = q;
  // static-index copies of the levels this thread needs (dynamic indexing of the kernel argument would go through scratch)
  int dims_o[3] = {1, 1, 1}, seg_len = 1, nseg = 1, thr_off = 0, src = 0, blk_o = 0;
#pragma unroll
  for (int q = 1; q < MAXLEV; ++q)
    if (q == l) {
      dims_o[0] = C.L[q].dims[0]; dims_o[1] = C.L[q].dims[1]; dims_o[2] = C.L[q].dims[2];
      seg_len = C.L[q].seg_len; nseg = C.L[q].nseg; thr_off = C.L[q].thr_off; src = C.L[q].src; blk_o = C.L[q].blk_off;
    }
  int u = t - thr_off;
  const int seg = u % nseg; u /= nseg;
  const int y = u % dims_o[1]; u /= dims_o[1];
  const int z = u % dims_o[0];
  const int b = u / dims_o[0];
  if (b >= C.batch) return;
  const int x0 = seg * seg_len;
  const int ncell = min(seg_len, dims_o[2] - x0);
  // intervals of the receptive field, level by level down to src; S / P / K = composite stride / pad / width along x
  int zl = z, zh = z, yl = y, yh = y, xl = x0, xh = x0 + ncell - 1;
  int S = 1, xstart = x0, K = 1;  // window of output cell x0 + i at src: [xstart + i*S, xstart + i*S + K - 1] before clipping
  int dims_s[3] = {dims_o[0], dims_o[1], dims_o[2]}, blk_s = blk_o;
#pragma unroll
  for (int q = MAXLEV - 1; q >= 1; --q)
    if (q <= l && q > src) {  // window of level q over level q - 1
      const LevelDev& Lq = C.L[q];
      const LevelDev& Lp = C.L[q - 1];
      zl = zl * Lq.st[0] - Lq.pd[0]; zh = zh * Lq.st[0] - Lq.pd[0] + Lq.ks[0] - 1;
      yl = yl * Lq.st[1] - Lq.pd[1]; yh = yh * Lq.st[1] - Lq.pd[1] + Lq.ks[1] - 1;
      xl = xl * Lq.st[2] - Lq.pd[2]; xh = xh * Lq.st[2] - Lq.pd[2] + Lq.ks[2] - 1;
      xstart = xstart * Lq.st[2] - Lq.pd[2];
      K = (K - 1) * Lq.st[2] + Lq.ks[2];
      S *= Lq.st[2];
      zl = max(zl, 0); zh = min(zh, Lp.dims[0] - 1);
      yl = max(yl, 0); yh = min(yh, Lp.dims[1] - 1);
      xl = max(xl, 0); xh = min(xh, Lp.dims[2] - 1);
      dims_s[0] = Lp.dims[0]; dims_s[1] = Lp.dims[1]; dims_s[2] = Lp.dims[2];
      blk_s = Lp.blk_off;
    }
  if (zl > zh || yl > yh || xl > xh) return;
  const uint2* in = occ + (size_t)blk_s * SPAN;
  const int nbits = xh - xl + 1;  // <= 128 by the choice of seg_len
  const u128 lenmask = nbits >= 128 ? ~(u128)0 : (((u128)1 << nbits) - 1);
  u128 R = 0;
  const int nzr = zh - zl + 1, nyr = yh - yl + 1;
  const bool wide = nbits > 97;  // sh + nbits may exceed 128 bits: fifth word needed
  auto window = [&](unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned w4, unsigned sh) {
    u128 v = ((u128)w0) | ((u128)w1 << 32) | ((u128)w2 << 64) | ((u128)w3 << 96);
    v >>= sh;
    if (sh) v |= (u128)w4 << (128 - sh);
    return v & lenmask;
  };
  if (nzr * nyr <= 9) {
    // the usual 3 x 3 (kz, ky) window: all row loads are issued before the first is used (one memory latency per thread)
    unsigned wd[9][5], shv[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int dz = r / nyr, dy = r - dz * nyr;
      const bool ok = r < nzr * nyr;
      const unsigned cell = (unsigned)(((b * dims_s[0] + zl + dz) * dims_s[1] + yl + dy) * dims_s[2] + xl);
      const unsigned w = cell >> 5;
      shv[r] = cell & 31u;
#pragma unroll
      for (int e = 0; e < 4; ++e) wd[r][e] = ok ? in[w + e].x : 0u;
      wd[r][4] = (ok && wide) ? in[w + 4].x : 0u;
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) R |= window(wd[r][0], wd[r][1], wd[r][2], wd[r][3], wd[r][4], shv[r]);
  } else {
    for (int zz = zl; zz <= zh; ++zz)
      for (int yy = yl; yy <= yh; ++yy) {
        const unsigned cell = (unsigned)(((b * dims_s[0] + zz) * dims_s[1] + yy) * dims_s[2] + xl);
        const unsigned w = cell >> 5;
        // 160 bits from word w on (the slack behind the last level is part of the workspace), shifted down to the cell
        R |= window(in[w].x, in[w + 1].x, in[w + 2].x, in[w + 3].x, in[w + 4].x, cell & 31u);
      }
  }
  if (R == 0) return;
  // bit u of R' <-> src cell x = xstart + u (xstart <= xl: cells left of the grid read as empty)
  R <<= (xl - xstart);
  u128 T = R;
  for (int j = 1; j < K; ++j) T |= R >> j;
  unsigned out = 0;
  for (int i = 0; i < ncell; ++i) out |= (unsigned)((T >> (i * S)) & 1) << i;
  if (!out) return;
  uint2* o = occ + (size_t)blk_o * SPAN;
  const unsigned cell = (unsigned)(((b * dims_o[0] + z) * dims_o[1] + y) * dims_o[2] + x0);
  const unsigned sh = cell & 31u;
  atomicOr(&o[cell >> 5].x, out << sh);  // a word is shared by at most the few segments of neighbouring rows
  if (sh && (out >> (32 - sh))) atomicOr(&o[(cell >> 5) + 1].x, out >> (32 - sh));
}

__global__ __launch_bounds__(NT) void chain_count_kernel(const uint2* __restrict__ occ, int* __restrict__ blk_cnt) {
  __shared__ int sm[NT / 64];
  const uint4* o = reinterpret_cast<const uint4*>(occ + (size_t)blockIdx.x * SPAN + threadIdx.x * WPT);
  int s = 0;
#pragma unroll
  for (int j = 0; j < WPT / 2; ++j) {
    const uint4 v = o[j];
    s += __popc(v.x) + __popc(v.z);
  }
  s = sessd_wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < NT / 64; ++w) t += sm[w];
    blk_cnt[blockIdx.x] = t;
  }
}

// ranks + site table of one SPAN-word block. blk_cnt = live cells per block (chain_count_kernel); the block's
// first rank = the sum over the level's earlier blocks, taken by the block itself (<= a few hundred integers: round 2 ran a
// separate scan launch for it); the level's last block also publishes the level's row count and the overflow flag.
__global__ __launch_bounds__(NT) void chain_emit_kernel(uint2* __restrict__ occ, const int* __restrict__ blk_cnt, ChainDev C,
                                                         int* __restrict__ err_flag) {
  __shared__ int sm[NT / 64];
  __shared__ unsigned s_bits[SPAN];
  __shared__ int s_rank[SPAN];
  __shared__ unsigned s_bzy[SPAN];  // first cell of the word: b << 24 | z << 16 | y   (limits checked on the host)
  __shared__ int s_x[SPAN];
  int cap = C.L[0].cap, blk_off = 0, n_blk = C.L[0].n_blk, D = C.L[0].dims[0], H = C.L[0].dims[1], W = C.L[0].dims[2];
  int* indices = C.L[0].indices;
  int* n_dev = C.L[0].n_dev;
#pragma unroll
  for (int l = 1; l < MAXLEV; ++l)
    if (l < C.nlev && (int)blockIdx.x >= C.L[l].blk_off) {
      cap = C.L[l].cap; blk_off = C.L[l].blk_off; n_blk = C.L[l].n_blk;
      D = C.L[l].dims[0]; H = C.L[l].dims[1]; W = C.L[l].dims[2];
      indices = C.L[l].indices; n_dev = C.L[l].n_dev;
    }
  const int b_in_level = (int)blockIdx.x - blk_off;
  int base;
  {
    int part = 0;
    for (int q = threadIdx.x; q < b_in_level; q += NT) part += blk_cnt[blk_off + q];
    int total;
    sessd_block_exscan<NT>(part, sm, &total);
    base = total;
    __syncthreads();  // sm is reused below
  }
  if (b_in_level == n_blk - 1 && threadIdx.x == 0) {
    const int total = base + blk_cnt[blockIdx.x];
    if (total > cap) atomicOr(err_flag, 1);  // capacity overflow: reported, never silent; rows >= cap are dropped
    n_dev[0] = total < cap ? total : cap;
  }
  uint2* o = occ + (size_t)blockIdx.x * SPAN + threadIdx.x * WPT;
  unsigned bits[WPT];
  int mine = 0;
#pragma unroll
  for (int j = 0; j < WPT / 2; ++j) {
    const uint4 v = reinterpret_cast<const uint4*>(o)[j];
    bits[2 * j] = v.x; bits[2 * j + 1] = v.z;
    mine += __popc(v.x) + __popc(v.z);
  }
  int tot;
  int rank = base + sessd_block_exscan<NT>(mine, sm, &tot);
  if (tot == 0) return;  // ranks of an empty block's words are never read (a lookup reads the rank only of a set bit)
  int nzw = 0;
#pragma unroll
  for (int j = 0; j < WPT; ++j) nzw += bits[j] ? 1 : 0;
  int nnz;
  int slot = sessd_block_exscan<NT>(nzw, sm, &nnz);
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const int w = threadIdx.x * WPT + j;
    o[j].y = (unsigned)rank;
    if (bits[j]) {  // decode the word's first cell once (integer divisions are slow: keep them out of the row loop)
      const unsigned cell = ((unsigned)b_in_level * SPAN + (unsigned)w) << 5;
      const unsigned zy = cell / (unsigned)W, bz = zy / (unsigned)H, bb = bz / (unsigned)D;
      s_bits[slot] = bits[j];
      s_rank[slot] = rank;
      s_bzy[slot] = (bb << 24) | ((bz - bb * (unsigned)D) << 16) | (zy - bz * (unsigned)H);
      s_x[slot] = (int)(cell - zy * (unsigned)W);
      ++slot;
    }
    rank += __popc(bits[j]);
  }
  __syncthreads();
  // one thread per (non-empty word, bit): no data-dependent trip counts (emitting a word per thread, a wave waited for its
  // fullest word: a BEV-dense block of the coarse levels took 14 us)
  for (int p = threadIdx.x; p < nnz * 32; p += NT) {
    const int e = p >> 5, bit = p & 31;
    const unsigned m = s_bits[e];
    if (!((m >> bit) & 1u)) continue;
    const int row = s_rank[e] + __popc(m & ((1u << bit) - 1u));
    if (row >= cap) continue;
    const unsigned bzy = s_bzy[e];
    int x = s_x[e] + bit, y = (int)(bzy & 0xFFFFu), z = (int)((bzy >> 16) & 0xFFu), bb = (int)(bzy >> 24);
    while (x >= W) {  // a word may straddle grid rows when W is not a multiple of 32
      x -= W;
      if (++y == H) {
        y = 0;
        if (++z == D) { z = 0; ++bb; }
      }
    }
    *reinterpret_cast<int4*>(indices + (size_t)row * 4) = make_int4(bb, z, y, x);
  }
}

// ---- all rulebooks of the chain --------------------------------------------------------------------------------
struct JobDev {
  int ks[3], st[3], pd[3];
  int in_dims[3];
  const uint2* occ;        // input level's occupancy map, or NULL: hash lookup
  const uint32_t* keys;    // level-0 hash
  const int* vals;
  uint32_t mask;
  int in_cap;
  const int* out_indices;
  const int* n_out_dev;
  int out_cap;
  int* nbr;
  uint32_t* tile_mask;
  uint32_t* site_mask;     // optional: per-site offset pattern (input of chain_tile_sort_kernel)
  int blk_off;             // first block of this job (4 tiles per block)
};
struct JobsDev {
  int njobs;
  JobDev J[MAXJOB];
};

// One wave = 16 output sites x 4 lanes; lane (i, q) handles the (kz, ky) rows q, q+4, q+8 of site i's window (<= 3 rows
// for a 3x3 (kz,ky) plane) with <= 3 kx each. All loads of a lane are issued before the first is used.
__global__ __launch_bounds__(NT) void chain_rulebook_kernel(JobsDev Q) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < MAXJOB; ++q)
    if (q < Q.njobs && (int)blockIdx.x >= Q.J[q].blk_off) j = q;
  // static-index copy of the job (see chain_gather_kernel)
  JobDev J = Q.J[0];
#pragma unroll
  for (int q = 1; q < MAXJOB; ++q)
    if (q == j) J = Q.J[q];
  const int lane = threadIdx.x & 63;
  const int tile = ((int)blockIdx.x - J.blk_off) * (NT / 64) + (threadIdx.x >> 6);
  const int n_cap = J.out_cap;
  if (tile * 16 >= n_cap) return;
  const int n = min(J.n_out_dev[0], n_cap);
  if (tile * 16 >= n) {  // beyond the live sites: keep the table defined
    if (lane == 0) J.tile_mask[tile] = 0u;
    return;
  }
  const int i = lane & 15, q = lane >> 4;
  const int o = tile * 16 + i;
  const bool live = o < n;
  int4 c = make_int4(0, 0, 0, 0);
  if (live) c = *reinterpret_cast<const int4*>(J.out_indices + (size_t)o * 4);
  const int ksx = J.ks[2], npairs = J.ks[0] * J.ks[1];
  const int x0 = c.w * J.st[2] - J.pd[2];
  // phase 1: addresses
  bool pv[3], rowok[3];
  unsigned cell0[3];
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int pp = pass * 4 + q;
    pv[pass] = pp < npairs;
    const int kz = pp / J.ks[1], ky = pp - kz * J.ks[1];
    const int z = c.y * J.st[0] - J.pd[0] + kz, y = c.z * J.st[1] - J.pd[1] + ky;
    rowok[pass] = pv[pass] && live && z >= 0 && z < J.in_dims[0] && y >= 0 && y < J.in_dims[1];
    cell0[pass] = (unsigned)(((c.x * J.in_dims[0] + z) * J.in_dims[1] + y) * J.in_dims[2] + x0);  // cell of kx = 0 (x0 may be -1)
  }
  int found[3][3];
  if (J.occ) {
    // phase 2: the one or two map words that hold the row's <= 3 x-neighbours
    uint2 ua[3], ub[3];
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const unsigned ca = cell0[pass] + (x0 < 0 ? (unsigned)(-x0) : 0u);  // first in-range cell
      const unsigned wa = ca >> 5, wb = (cell0[pass] + (unsigned)(ksx - 1)) >> 5;
      ua[pass] = rowok[pass] ? J.occ[wa] : make_uint2(0u, 0u);
      ub[pass] = (rowok[pass] && wb != wa) ? J.occ[wb] : ua[pass];
    }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
      const unsigned ca = cell0[pass] + (x0 < 0 ? (unsigned)(-x0) : 0u);
      const unsigned wa = ca >> 5;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int x = x0 + kx;
        int f = -1;
        if (kx < ksx && rowok[pass] && x >= 0 && x < J.in_dims[2]) {
          const unsigned cell = cell0[pass] + (unsigned)kx;
          const uint2 u = (cell >> 5) == wa ? ua[pass] : ub[pass];
          const unsigned bit = cell & 31u;
          if ((u.x >> bit) & 1u) {
            const int row = (int)u.y + __popc(u.x & ((1u << bit) - 1u));
            f = row < J.in_cap ? row : -1;
          }
        }
        found[pass][kx] = f;
      }
    }
  } else {
    // phase 2: the home slot's key of every neighbour; phase 3: the value of every hit; collisions (load factor <= 0.5:
    // about one probe in four) fall back to the sequential probe. (Also fetching the successor slot and both values in the
    // same round was measured: no gain at batch 1, 455 -> 600 us on the dense-scene batch -- the kernel is bound by the
    // number of random memory transactions, not by their latency.)
    unsigned key[3][3], slot[3][3], k0[3][3];
    bool want[3][3];
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int x = x0 + kx;
        want[pass][kx] = kx < ksx && rowok[pass] && x >= 0 && x < J.in_dims[2];
        key[pass][kx] = cell0[pass] + (unsigned)kx;
        slot[pass][kx] = sessd_hash_home(key[pass][kx], J.mask);
      }
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) k0[pass][kx] = want[pass][kx] ? J.keys[slot[pass][kx]] : SESSD_HASH_EMPTY;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        found[pass][kx] = (want[pass][kx] && k0[pass][kx] == key[pass][kx]) ? J.vals[slot[pass][kx]] : -1;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        if (want[pass][kx] && k0[pass][kx] == key[pass][kx]) {
          if (found[pass][kx] == SESSD_SENT) found[pass][kx] = -1;
        } else if (want[pass][kx] && k0[pass][kx] != SESSD_HASH_EMPTY) {
          found[pass][kx] = sessd_hash_find(J.keys, J.vals, J.mask, key[pass][kx]);
        }
      }
  }
  // phase 4: table + tile mask (+ the site's own offset pattern for the tile sort)
  uint32_t tm = 0, sm = 0;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      if (kx < ksx) {  // wave-uniform
        const int pp = pass * 4 + q;
        if (pv[pass] && o < n_cap) J.nbr[(size_t)(pp * ksx + kx) * n_cap + o] = found[pass][kx];
        if (pv[pass] && found[pass][kx] >= 0) sm |= 1u << (pp * ksx + kx);
        const unsigned long long bal = __ballot(found[pass][kx] >= 0);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if ((bal >> (16 * qq)) & 0xFFFFull) tm |= 1u << ((pass * 4 + qq) * ksx + kx);
      }
    }
  if (lane == 0) J.tile_mask[tile] = tm;
  if (J.site_mask) {   // the four lanes (i, q = 0..3) of a site hold its (kz, ky) rows q, q + 4, q + 8
    sm |= (uint32_t)__shfl_xor((int)sm, 16, 64);
    sm |= (uint32_t)__shfl_xor((int)sm, 32, 64);
    if (q == 0 && live) J.site_mask[o] = sm;
  }
}

// ---- offset-pattern tiles ------------------------------------------------------------------------------------------
// The sparse conv multiplies a 16-row MFMA tile once per kernel offset that ANY of the tile's sites has a neighbour at: with the
// sites in (b, z, y, x) order only 57 % (20 k-point frame) to 63 % (dense scene) of the executed rows carry a pair. Sites with the
// same offset pattern in one tile waste nothing. Renumbering the rows by pattern would cost every neighbour lookup an extra
// indirection; instead the ROWS KEEP THEIR NUMBERS and only the grouping of rows into tiles changes: inside every group of 256
// consecutive rows (the gathers of a group still hit the same cache lines) the live sites are sorted by pattern, tile t of the
// group takes the sorted positions 16 t .. 16 t + 15, and the conv kernel reads "position -> row" from a byte table. Results
// are the same bits per site (a site's sum does not depend on its tile mates). Useful rows on the oracle's rulebooks
// (scripts/tile_occupancy_probe.py): 57 -> 73 % at batch 1, 63 -> 80 % on the dense scene.
struct SortJob {
  const uint32_t* site_mask;
  const int* n_out_dev;
  int out_cap;
  uint8_t* perm;
  uint32_t* tile_mask_sorted;
  int blk_off;   // first block (one block = one 256-row group)
};
struct SortJobs {
  int njobs;
  SortJob J[MAXJOB];
};

__global__ __launch_bounds__(256) void chain_tile_sort_kernel(SortJobs Q) {
  __shared__ unsigned long long s_key[256];
  int j = 0;
#pragma unroll
  for (int q = 1; q < MAXJOB; ++q)
    if (q < Q.njobs && (int)blockIdx.x >= Q.J[q].blk_off) j = q;
  SortJob J = Q.J[0];
#pragma unroll
  for (int q = 1; q < MAXJOB; ++q)
    if (q == j) J = Q.J[q];
  const int tid = threadIdx.x;
  const int g = (int)blockIdx.x - J.blk_off;
  const int n = min(J.n_out_dev[0], J.out_cap);
  const int o = g * 256 + tid;
  const bool live = o < n;
  // live sites first, by pattern then by row (a stable order: the result does not depend on the sort network's tie handling)
  s_key[tid] = live ? (((unsigned long long)J.site_mask[o] << 8) | (unsigned long long)tid) : (0xFFFFFFFFFFFFFF00ull | (unsigned long long)tid);
  __syncthreads();
  for (int k = 2; k <= 256; k <<= 1)
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      const int partner = tid ^ jj;
      if (partner > tid) {
        const unsigned long long a = s_key[tid], b = s_key[partner];
        const bool up = (tid & k) == 0;
        if ((a > b) == up) { s_key[tid] = b; s_key[partner] = a; }
      }
      __syncthreads();
    }
  const unsigned long long key = s_key[tid];
  J.perm[(size_t)g * 256 + tid] = (uint8_t)(key & 0xFFull);
  uint32_t m = (g * 256 + tid < n) ? (uint32_t)(key >> 8) : 0u;   // sorted position tid is live iff it is below the live count
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) m |= (uint32_t)__shfl_xor((int)m, d, 64);
  if ((tid & 15) == 0) J.tile_mask_sorted[(size_t)g * 16 + (tid >> 4)] = m;
}

struct Layout {
  size_t occ_words;   // padded total (+ slack for the gather's 5-word windows)
  size_t blk_cnt_off; // bytes
  size_t total;
  int nblk;
  bool composite;     // levels 2.. gathered from level 1 in one launch
  int gather_threads; // of that one launch
};

// geometry along x of the window of level l (chain index) over level src: stride S, width K
void composite_x(const sessd_chain_level_t* lv, int l, int src, int& S, int& K) {
  S = 1; K = 1;
  for (int q = l; q > src; --q) {
    K = (K - 1) * lv[q].stride[2] + lv[q].ksize[2];
    S *= lv[q].stride[2];
  }
}

Layout chain_layout(int batch, int nlev, const sessd_chain_level_t* lv, ChainDev* C) {
  Layout Y;
  int blk = 0;
  // Every level is gathered from the one above it (<= kz * ky rows per thread). Gathering all deeper levels from level 1 in
  // one launch is possible (the kernel takes any src) but its threads walk 49 / 105 rows each: 57 us against 3 x 4 us.
  Y.composite = false;
  int thr = 0;
  for (int l = 0; l < nlev; ++l) {
    const long long cells = (long long)batch * lv[l].out_dims[0] * lv[l].out_dims[1] * lv[l].out_dims[2];
    const int words = (int)((cells + 31) / 32);
    const int nb = sessd_divup(words, SPAN);
    const int src = (l == 0 || Y.composite) ? 0 : l - 1;
    int S = 1, K = 1, seg_len = 32;
    if (l > 0) {
      composite_x(lv, l, src, S, K);
      while (seg_len > 1 && (seg_len - 1) * S + K > 128) seg_len >>= 1;
    }
    const int nseg = sessd_divup(lv[l].out_dims[2], seg_len);
    if (C) {
      LevelDev& L = C->L[l];
      for (int d = 0; d < 3; ++d) {
        L.ks[d] = lv[l].ksize[d]; L.st[d] = lv[l].stride[d]; L.pd[d] = lv[l].pad[d]; L.dims[d] = lv[l].out_dims[d];
      }
      L.cap = lv[l].cap; L.blk_off = blk; L.n_blk = nb; L.indices = lv[l].indices; L.n_dev = lv[l].n_dev;
      L.src = src; L.seg_len = seg_len; L.nseg = nseg;
      L.thr_off = Y.composite ? thr : 0;
    }
    if (l > 0) thr += batch * lv[l].out_dims[0] * lv[l].out_dims[1] * nseg;
    blk += nb;
  }
  Y.gather_threads = thr;
  Y.nblk = blk;
  Y.occ_words = (size_t)blk * SPAN + 8;
  Y.blk_cnt_off = sessd_align(Y.occ_words * sizeof(uint2), 256);
  Y.total = Y.blk_cnt_off + sessd_align((size_t)blk * 4 + 4, 256);
  if (C) {
    C->nlev = nlev;
    C->batch = batch;
  }
  return Y;
}

bool chain_valid(int batch, int nlev, const sessd_chain_level_t* lv) {
  if (batch <= 0 || nlev <= 0 || nlev > MAXLEV || !lv) return false;
  for (int l = 0; l < nlev; ++l) {
    long long cells = batch;
    for (int d = 0; d < 3; ++d) {
      if (lv[l].ksize[d] <= 0 || lv[l].stride[d] <= 0 || lv[l].pad[d] < 0 || lv[l].out_dims[d] <= 0) return false;
      if (lv[l].ksize[d] < lv[l].stride[d]) return false;  // receptive fields of neighbouring inputs must touch (see top)
      cells *= lv[l].out_dims[d];
    }
    if (cells >= 0x7F000000ll || lv[l].cap <= 0) return false;
    if (batch > 256 || lv[l].out_dims[0] > 256 || lv[l].out_dims[1] > 65536) return false;  // packed coordinates in chain_emit_kernel
    if (lv[l].stride[2] + lv[l].ksize[2] > 128) return false;  // a one-cell segment's window must fit 128 bits
  }
  return true;
}

}  // namespace

extern "C" {

size_t sessd_sparse_chain_workspace_bytes(int batch, int n_levels, const sessd_chain_level_t* levels) {
  if (!chain_valid(batch, n_levels, levels)) return 0;
  return chain_layout(batch, n_levels, levels, nullptr).total;
}

int sessd_sparse_chain_sites(const int32_t* indices0, const int32_t* n0_dev, int n0_cap, int batch, int n_levels,
                             const sessd_chain_level_t* levels, void* workspace, size_t workspace_bytes, int clear,
                             int32_t* err_flag, hipStream_t stream) {
  if (n0_cap <= 0 || !chain_valid(batch, n_levels, levels) || !workspace) return SESSD_EINVAL;
  ChainDev C;
  const Layout Y = chain_layout(batch, n_levels, levels, &C);
  if (Y.total > workspace_bytes) return SESSD_EWORKSPACE;
  uint2* occ = (uint2*)workspace;
  int* blk_cnt = (int*)((char*)workspace + Y.blk_cnt_off);
  if (clear) SESSD_FILL(occ, 0u, Y.occ_words * 2, stream);
  SESSD_LAUNCH(chain_mark_kernel, dim3(sessd_divup(n0_cap, NT)), dim3(NT), 0, stream, indices0, n0_dev, n0_cap, C, occ);
  SESSD_CHECK_LAUNCH();
  if (n_levels > 1) {
    if (Y.composite) {
      SESSD_LAUNCH(chain_gather_kernel, dim3(sessd_divup(Y.gather_threads, NT)), dim3(NT), 0, stream, C, 1, n_levels - 1, 0, occ);
      SESSD_CHECK_LAUNCH();
    } else {
      for (int l = 1; l < n_levels; ++l) {
        const int thr = batch * C.L[l].dims[0] * C.L[l].dims[1] * C.L[l].nseg;
        SESSD_LAUNCH(chain_gather_kernel, dim3(sessd_divup(thr, NT)), dim3(NT), 0, stream, C, l, l, 0, occ);
        SESSD_CHECK_LAUNCH();
      }
    }
  }
  SESSD_LAUNCH(chain_count_kernel, dim3(Y.nblk), dim3(NT), 0, stream, occ, blk_cnt);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(chain_emit_kernel, dim3(Y.nblk), dim3(NT), 0, stream, occ, blk_cnt, C, err_flag);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

int sessd_sparse_chain_rulebooks(const int32_t* indices0, const int32_t* n0_dev, int n0_cap, const uint32_t* keys0,
                                 const int32_t* vals0, uint32_t capacity0, const int32_t* dims0, int batch, int n_levels,
                                 const sessd_chain_level_t* levels, const void* workspace, int n_jobs,
                                 const sessd_rulebook_job_t* jobs, hipStream_t stream) {
  if (n0_cap <= 0 || !chain_valid(batch, n_levels, levels) || !workspace || n_jobs <= 0 || n_jobs > MAXJOB || !jobs)
    return SESSD_EINVAL;
  if ((capacity0 & (capacity0 - 1)) != 0) return SESSD_EINVAL;
  ChainDev C;
  chain_layout(batch, n_levels, levels, &C);
  const uint2* occ = (const uint2*)workspace;
  JobsDev Q;
  Q.njobs = n_jobs;
  int blk = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const sessd_rulebook_job_t& S = jobs[j];
    JobDev& J = Q.J[j];
    if (S.in_level < 0 || S.in_level > n_levels || S.out_level < 0 || S.out_level > n_levels || !S.nbr || !S.tile_mask)
      return SESSD_EINVAL;
    if (S.ksize[0] * S.ksize[1] * S.ksize[2] > 32) return SESSD_EINVAL;
    // chain_rulebook_kernel walks the (kz, ky) pairs in three passes of four lane groups and kx in registers: kernels beyond
    // 12 (kz, ky) pairs or 3 taps in x would leave table rows unwritten
    if (S.ksize[0] * S.ksize[1] > 12 || S.ksize[2] > 3 || S.ksize[0] <= 0 || S.ksize[1] <= 0 || S.ksize[2] <= 0) return SESSD_EINVAL;
    for (int d = 0; d < 3; ++d) {
      J.ks[d] = S.ksize[d]; J.st[d] = S.stride[d]; J.pd[d] = S.pad[d];
      J.in_dims[d] = S.in_level == 0 ? dims0[d] : C.L[S.in_level - 1].dims[d];
    }
    if (S.in_level == 0) {
      J.occ = nullptr; J.keys = keys0; J.vals = vals0; J.mask = capacity0 - 1; J.in_cap = n0_cap;
    } else {
      J.occ = occ + (size_t)C.L[S.in_level - 1].blk_off * SPAN;
      J.keys = nullptr; J.vals = nullptr; J.mask = 0; J.in_cap = C.L[S.in_level - 1].cap;
    }
    if (S.out_level == 0) {
      J.out_indices = indices0; J.n_out_dev = n0_dev; J.out_cap = n0_cap;
    } else {
      const LevelDev& L = C.L[S.out_level - 1];
      J.out_indices = L.indices; J.n_out_dev = L.n_dev; J.out_cap = L.cap;
    }
    const bool sorted = S.site_mask || S.perm || S.tile_mask_sorted;
    if (sorted && !(S.site_mask && S.perm && S.tile_mask_sorted)) return SESSD_EINVAL;
    J.nbr = S.nbr; J.tile_mask = S.tile_mask; J.site_mask = S.site_mask; J.blk_off = blk;
    blk += sessd_divup(sessd_divup(J.out_cap, 16), NT / 64);
  }
  SESSD_LAUNCH(chain_rulebook_kernel, dim3(blk), dim3(NT), 0, stream, Q);
  SESSD_CHECK_LAUNCH();
  // offset-pattern tiles of the jobs that ask for them: one more launch for all of them
  SortJobs T;
  T.njobs = 0;
  int sblk = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!jobs[j].perm) continue;
    SortJob& K = T.J[T.njobs++];
    K.site_mask = jobs[j].site_mask; K.n_out_dev = Q.J[j].n_out_dev; K.out_cap = Q.J[j].out_cap; K.perm = jobs[j].perm;
    K.tile_mask_sorted = jobs[j].tile_mask_sorted; K.blk_off = sblk;
    sblk += sessd_divup(K.out_cap, 256);
  }
  if (T.njobs) {
    SESSD_LAUNCH(chain_tile_sort_kernel, dim3(sblk), dim3(256), 0, stream, T);
    SESSD_CHECK_LAUNCH();
  }
  return SESSD_OK;
}

}  // extern "C"
