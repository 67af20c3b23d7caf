// Which tiles of the SSFA layers in front of conv_0 / conv_1 have anything to compute (gfx950; round 4).
//
// The BEV map that enters the neck (det3d/models/backbones/scn.py:179-183: `.dense()` of the last sparse level, 64 channels x 2
// z-slices per pixel) is ZERO outside the sparse backbone's sites -- 7 % of the 200 x 176 pixels of a KITTI-shaped scan hold one.
// rpn_v1.py:135-160 then runs 3x3 conv + BatchNorm + ReLU layers over the whole map. Away from the sites that is arithmetic on
// constants: conv(0) = 0 -> BatchNorm -> ReLU gives the same value c1[co] in every pixel whose 3x3 window holds no site, the next
// layer maps a window of c1 to c2[co], and so on (the image border, where zero padding enters the window, is not constant from
// the second layer on). A 2x2-output Winograd tile has something to compute iff its 4x4 input patch touches a non-constant pixel of
// the layer's input; on 20 k-point scans that is 12 - 14 % / 23 - 25 % / 32 - 36 % of the tiles of b0.0 / b0.1 / b0.2.
//   bev_tile_activity   per image: site pixels -> bit rows in LDS; then a STEP PROGRAM, one slot (tile mask + ordered tile list +
//                       device count) per step that takes one:
//                         0  3x3 stride-1 layer: tile bits = 4x4 patch test (OR of four pixel rows, shifted-OR over the patch
//                            columns, every second bit; + the border ring once the input constant is not zero); the next step's
//                            non-constant rows = the tile rows with every bit doubled
//                         1  3x3 stride-2 layer computed over the whole map (rpn_v1.py:150-152): an output pixel is constant iff
//                            its 3x3 stride-2 window is, zero padding enters at the top / left border only; the map halves
//                         2  the same layer over a list of its own: the 2x2 tiles of its OUTPUT that hold a non-constant pixel
//                         3  stride-2 transposed conv on the current map (rpn_v1.py:175-199) whose output also receives a map of
//                            the resolution before the halving as a residual (:224): 2x2 tiles of its INPUT = 4x4 output blocks
//                         4  (round 6) no layer: the map becomes the OUTPUT of the step-3 transposed conv in front of it (twice the
//                            resolution; non-constant pixels = the 4x4 blocks of that step's tiles), so that a following step 0 is
//                            a 3x3 layer over the transposed conv's output -- conv_0 / conv_1 (rpn_v1.py:200-210), whose input is
//                            constant PER OUTPUT PARITY CLASS outside those blocks, and so is its output outside its own tiles
//                       rpn_v1.py:135-160 + 224 is {0, 0, 0, 2, 0, 0, 3} (+ {4, 0} for conv_0 / conv_1); the 1x1 trans layers take
//                       the list of the layer that produced their input. Lists are ascending in (image, tile): deterministic, no atomics. One launch at batch
//                       1; with more images a second launch numbers the lists (an image's base is the sum of the earlier images'
//                       counts). (First version: byte maps walked by one workgroup -- 104 us; now 25 - 28 us, issue-bound.)
//   fill_inactive_tiles the layers' constants into the tiles nobody computes, all layers in one launch (2x2-pixel tiles with one
//                       constant per channel; 4x4-pixel tiles with one constant per channel and output parity class behind the
//                       transposed convs).
// The convolutions themselves: conv3x3s1_winograd_sk_kernel<.., LIST = true> (dense_wino_sk.hip), conv2d_sk_kernel<LIST = true>
// (dense_conv_sk.hip) and conv_body<.., LIST = true> (dense_conv.hip) over the lists.
// The constants come from the host (float64 over the folded weights: sessd_hip.engine.active_tile_constants); a computed tile and
// a filled tile agree to float32 rounding of that chain (1e-7 relative), bit-exactly for the first layer (0 * U = 0). The rule and
// the constants are held to torch's own convolutions in tests/test_active_rule_cpu.py, the kernels to the rule and to the dense
// kernels in tests/test_dense_active_gpu.py.
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

// 1024 threads: a thread pair per tile row needs 256 of them, but the list emission (a wave per mask word, a lane per bit) and the
// site walk use all sixteen waves. MEASURED in round 5 (review item "four waves instead of sixteen"): with NT = 256 / LPT = 16
// the kernel took 46.8 us instead of 29 (profiles/r5_dense_pmc_whole_unit_shares.txt was taken with that build): the steps'
// arithmetic is not what the kernel waits for, its ~200 mask words per slot emitted four at a time are.
constexpr int NT = 1024;
constexpr int LPT = 8;    // site rows a thread fetches per round (all in flight together)
constexpr int MAX_H = 256, MAX_W = 192;   // one image's pixel rows as 3 x 64-bit words in LDS; a thread pair per tile row
constexpr int MAX_SLOTS = 8, MAX_STEPS = 10, MAX_FILL_JOBS = 12;
constexpr int FILL_CG = 16;   // channels per fill block
typedef unsigned long long u64;

struct ActArgs {
  const int* indices;   // (n, 4) rows (image, z, y, x) of the last sparse level
  const int* n_dev;
  int n_cap, batch, h, w;
  int n_steps, n_slots;
  int kind[MAX_STEPS];        // 0: 3x3 stride-1 layer (takes the next slot), 1: 3x3 stride-2 transition (halves the map),
                              // 2: the same transition, itself computed over a tile list (takes the next slot),
                              // 3: a stride-2 TRANSPOSED conv on the current map, its 2x2 INPUT tiles (takes the next slot; the map
                              //    and the resolution stay)
  int slot_th[MAX_SLOTS], slot_tw[MAX_SLOTS];
  int list_cap, mask_th;      // rows of a (slot, image) block of tile_mask = h / 2 (the first resolution's)
  u64* tile_mask;       // [n_slots][batch][mask_th][2]: bit tx of the row's 128-bit word = tile (ty, tx) is computed
  int* tile_list;       // [n_slots][list_cap]
  int* n_list;          // [n_slots]
  int* counts;          // [n_slots][batch]
};

// bits 0, 2, 4, .. of x -> bits 0 .. 31
__device__ __forceinline__ unsigned even_bits(u64 x) {
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
  return (unsigned)x;
}
// every bit of x twice: bit i -> bits 2i, 2i + 1
__device__ __forceinline__ u64 double_bits(unsigned v) {
  u64 x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x << 2)) & 0x3333333333333333ull;
  x = (x | (x << 1)) & 0x5555555555555555ull;
  return x | (x << 1);
}
// the valid bits of a row of `n` bits over two words
__device__ __forceinline__ void valid_bits(int n, u64& v0, u64& v1) {
  v0 = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
  v1 = n > 64 ? (n >= 128 ? ~0ull : ((1ull << (n - 64)) - 1ull)) : 0ull;
}

// __syncthreads() is a workgroup-scope FENCE as well: it waits until every global store the wave has issued is acknowledged
// (~0.7 us each time; the masks / lists / counts this kernel writes are read by nobody before it ends). Four of them per step
// were two thirds of a step's 3 us. The LDS traffic of the workgroup only needs its own counter drained.
#define ACT_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// sessd_block_exscan (common.hpp) with LDS-only barriers
__device__ __forceinline__ int act_exscan(int v, int* smem, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) smem[wid] = incl;
  ACT_LDS_BARRIER();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int sv = smem[w];
    if (w < wid) wbase += sv;
    tot += sv;
  }
  ACT_LDS_BARRIER();
  *total = tot;
  return wbase + incl - v;
}

// The set bits of the mask words s_m[0 .. n_words) (word wd = tile row wd / 2, half wd % 2; s_pre[wd] = entries before the word) as
// list entries first + row * TW + 64 * half + bit, ascending from list[0]: a wave per word, a lane per bit -- the entries of a word
// are consecutive addresses (first version: every thread walked its own word bit by bit, up to 64 dependent iterations and
// scattered 4-byte stores; that loop was a third of this kernel's 34 us).
__device__ __forceinline__ void emit_words(const u64* s_m, const int* s_pre, int n_words, int TW, int first, int* list, int list_cap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // n_words <= MAX_H = 16 words per wave: all of a wave's LDS reads are issued before the first is used
  u64 m[MAX_H / (NT / 64)];
  int pre[MAX_H / (NT / 64)];
#pragma unroll
  for (int k = 0; k < MAX_H / (NT / 64); ++k) {
    const int wd = wave + k * (NT / 64);
    m[k] = wd < n_words ? s_m[wd] : 0ull;
    pre[k] = wd < n_words ? s_pre[wd] : 0;
  }
#pragma unroll
  for (int k = 0; k < MAX_H / (NT / 64); ++k) {
    const int wd = wave + k * (NT / 64);
    if ((m[k] >> lane) & 1ull) {
      const int at = pre[k] + __popcll(m[k] & ((1ull << lane) - 1ull));
      if (at < list_cap) list[at] = first + (wd >> 1) * TW + 64 * (wd & 1) + lane;
    }
  }
}

// A slot's tile rows -> its mask words (written by the caller), its ordered list (batch 1) or its per-image count. Every thread of
// the workgroup calls this (block scan + barriers); thread order = (row, word) = ascending tile order.
__device__ __forceinline__ void emit_slot(const ActArgs& A, int slot, int b, int row, int half, int TH, int TW, u64 mine, int* s_scan,
                                          u64* s_m, int* s_pre) {
  int total;
  const int at = act_exscan(__popcll(mine), s_scan, &total);
  if (row < TH) { s_m[threadIdx.x] = mine; s_pre[threadIdx.x] = at; }   // (s_m is also what a later transition keeps)
  if (A.batch == 1) {
    ACT_LDS_BARRIER();
    emit_words(s_m, s_pre, 2 * TH, TW, 0, A.tile_list + (size_t)slot * A.list_cap, A.list_cap);
    if (threadIdx.x == 0) A.n_list[slot] = total;
  } else if (threadIdx.x == 0) {
    A.counts[slot * A.batch + b] = total;
  }
}

// One workgroup per image. Pixel rows and tile rows are bit vectors: a tile row is OR of four pixel rows, shifted-OR over the four
// patch columns, every second bit kept -- a thread pair per tile row and layer.
__global__ __launch_bounds__(NT) void bev_tile_activity_kernel(ActArgs A) {
  __shared__ u64 nc[MAX_H][3];            // non-constant pixels of the current step's input
  __shared__ u64 tmb[MAX_H / 2][2];       // computed tiles of the current layer / rows of a transition
  __shared__ int s_scan[NT / 64];
  __shared__ u64 s_m[MAX_H];              // a slot's mask words / entries before each word (emit_slot)
  __shared__ int s_pre[MAX_H];
  __shared__ u64 s_keep[MAX_H];           // mask words of the last LAYER slot before the most recent transition (step 3's residual input)
  const int b = blockIdx.x;
  int H = A.h, W = A.w;
  // the site rows of ALL images are walked by every workgroup (they are few); eight independent loads per thread and round. The
  // first round is fetched up to the CAPACITY, together with the count (one memory latency instead of two), and masked afterwards.
  int4 c[LPT];
#pragma unroll
  for (int k = 0; k < LPT; ++k) {
    const int i = k * NT + (int)threadIdx.x;
    c[k] = i < A.n_cap ? *reinterpret_cast<const int4*>(A.indices + (size_t)i * 4) : make_int4(-1, 0, 0, 0);
  }
  const int n = min(A.n_dev[0], A.n_cap);
  for (int i = threadIdx.x; i < H * 3; i += NT) (&nc[0][0])[i] = 0ull;
  ACT_LDS_BARRIER();
  for (int base = 0; base < n; base += NT * LPT) {
    if (base > 0) {
#pragma unroll
      for (int k = 0; k < LPT; ++k) {
        const int i = base + k * NT + (int)threadIdx.x;
        c[k] = i < n ? *reinterpret_cast<const int4*>(A.indices + (size_t)i * 4) : make_int4(-1, 0, 0, 0);
      }
    }
#pragma unroll
    for (int k = 0; k < LPT; ++k)
      if (base + k * NT + (int)threadIdx.x < n && c[k].x == b && c[k].z >= 0 && c[k].z < H && c[k].w >= 0 && c[k].w < W)
        atomicOr(&nc[c[k].z][c[k].w >> 6], 1ull << (c[k].w & 63));
  }
  ACT_LDS_BARRIER();
  bool zero_input = true;   // the map's constant is zero: zero padding does not show at the border
  bool have_layer = false, have_keep = false;   // s_m holds a layer slot of the current resolution / s_keep one of twice the resolution
  int slot = 0;
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  for (int st = 0; st < A.n_steps; ++st) {
    if (A.kind[st] == 0) {
      // ---- 3x3 stride-1 layer: tile (ty, tx) is computed iff its 4x4 input patch rows 2ty-1 .. 2ty+2, cols 2tx-1 .. 2tx+2 holds a
      // non-constant pixel (or, with a non-zero constant, lies on the border ring)
      const int TH = H >> 1, TW = W >> 1;
      u64 v0, v1;
      valid_bits(TW, v0, v1);
      u64 mine = 0ull;
      if (row < TH) {
        u64 r0 = 0ull, r1 = 0ull, r2 = 0ull;
        for (int y = max(2 * row - 1, 0); y <= min(2 * row + 2, H - 1); ++y) { r0 |= nc[y][0]; r1 |= nc[y][1]; r2 |= nc[y][2]; }
        // r | r << 1 | r >> 1 | r >> 2 over the 192-bit row, read at the even bits
        const u64 h0 = r0 | (r0 << 1) | (r0 >> 1) | (r1 << 63) | (r0 >> 2) | (r1 << 62);
        const u64 h1 = r1 | (r1 << 1) | (r0 >> 63) | (r1 >> 1) | (r2 << 63) | (r1 >> 2) | (r2 << 62);
        const u64 h2 = r2 | (r2 << 1) | (r1 >> 63) | (r2 >> 1) | (r2 >> 2);
        u64 t0 = (u64)even_bits(h0) | ((u64)even_bits(h1) << 32), t1 = (u64)even_bits(h2);
        if (!zero_input) {
          if (row == 0 || row == TH - 1) { t0 = ~0ull; t1 = ~0ull; }
          t0 |= 1ull;
          if (TW - 1 < 64) t0 |= 1ull << (TW - 1); else t1 |= 1ull << (TW - 1 - 64);
        }
        t0 &= v0; t1 &= v1;
        mine = half ? t1 : t0;
        A.tile_mask[(((size_t)slot * A.batch + b) * A.mask_th + row) * 2 + half] = mine;
      }
      emit_slot(A, slot, b, row, half, TH, TW, mine, s_scan, s_m, s_pre);   // (its barriers also separate this step's reads of nc from the writes below)
      // the layer's output is non-constant exactly in its computed tiles: pixel rows 2 ty, 2 ty + 1 = the tile row, every bit twice
      // (each thread of the pair writes the words that come from its own half)
      if (row < TH) {
        if (half == 0) {
          const u64 p0 = double_bits((unsigned)mine), p1 = double_bits((unsigned)(mine >> 32));
          nc[2 * row][0] = p0; nc[2 * row][1] = p1;
          nc[2 * row + 1][0] = p0; nc[2 * row + 1][1] = p1;
        } else {
          const u64 p2 = double_bits((unsigned)mine);
          nc[2 * row][2] = p2;
          nc[2 * row + 1][2] = p2;
        }
      }
      ACT_LDS_BARRIER();
      zero_input = false;   // (a layer's own constant relu(shift) is not zero in general)
      have_layer = true;
      ++slot;
    } else if (A.kind[st] == 3) {
      // ---- ConvTranspose2d(3, stride 2, padding 1, output_padding 1) on the current map, computed over 2x2 tiles of its INPUT
      // (= 4x4 blocks of its output): out(2y+py, 2x+px) reads in(y .. y+py, x .. x+px), so tile (ty, tx) has something to compute
      // iff rows 2ty .. 2ty+2, cols 2tx .. 2tx+2 hold a non-constant pixel; the last tile row / column always (beyond the map the
      // sum loses terms: not the constant); and wherever the RESIDUAL added to the output is not constant -- the kept layer slot
      // of twice the resolution, 2x2 of its tiles per tile of this one (rpn_v1.py:224: deconv_block_0(x_trans_1) + x_trans_0).
      const int TH = H >> 1, TW = W >> 1;   // TW <= 64 (checked by the host)
      u64 v0, v1;
      valid_bits(TW, v0, v1);
      u64 mine = 0ull;
      if (row < TH) {
        if (half == 0) {
          u64 r0 = 0ull, r1 = 0ull;
          for (int y = 2 * row; y <= min(2 * row + 2, H - 1); ++y) { r0 |= nc[y][0]; r1 |= nc[y][1]; }
          const u64 h0 = r0 | (r0 >> 1) | (r1 << 63) | (r0 >> 2) | (r1 << 62);
          const u64 h1 = r1 | (r1 >> 1) | (r1 >> 2);
          u64 t0 = (u64)even_bits(h0) | ((u64)even_bits(h1) << 32);
          if (!zero_input) {
            if (row == TH - 1) t0 = ~0ull;
            t0 |= 1ull << (TW - 1);
          }
          if (have_keep) {
            const u64 k0 = s_keep[4 * row] | s_keep[4 * row + 2], k1 = s_keep[4 * row + 1] | s_keep[4 * row + 3];
            t0 |= (u64)even_bits(k0 | (k0 >> 1)) | ((u64)even_bits(k1 | (k1 >> 1)) << 32);
          }
          mine = t0 & v0;
        }
        A.tile_mask[(((size_t)slot * A.batch + b) * A.mask_th + row) * 2 + half] = mine;
      }
      emit_slot(A, slot, b, row, half, TH, TW, mine, s_scan, s_m, s_pre);
      ACT_LDS_BARRIER();
      have_layer = false;   // (s_m no longer holds a layer slot)
      ++slot;
    } else if (A.kind[st] == 4) {
      // ---- the map becomes the output of the transposed conv of the step before (a step 3: its tile rows are in s_m, one word
      // each, TW <= 64; checked by the host): pixel (y, x) of the 2H x 2W map is non-constant iff tile (y >> 2, x >> 2) was
      // computed. Nothing is emitted; the resolution doubles; a constant that depends on the output parity class is "a constant"
      // for every 3x3 stride-1 layer that follows on 2x2 tiles (the period is one tile).
      const int H2 = H << 1, W2 = W << 1;
      u64 v0, v1, v2 = 0ull;
      valid_bits(min(W2, 128), v0, v1);
      if (W2 > 128) { u64 d; valid_bits(W2 - 128, v2, d); }
      // the pair's mask words leave s_m through registers: nc and s_m are different arrays, but every thread reads before any writes
      const int y = (int)threadIdx.x;
      u64 m = 0ull;
      if (y < H2) m = s_m[2 * (y >> 2)];
      ACT_LDS_BARRIER();
      if (y < H2) {
        // every bit four times: bits 0 .. 15 -> word 0, 16 .. 31 -> word 1, 32 .. 47 -> word 2
        nc[y][0] = double_bits((unsigned)double_bits((unsigned)(m & 0xFFFFull))) & v0;
        nc[y][1] = double_bits((unsigned)double_bits((unsigned)((m >> 16) & 0xFFFFull))) & v1;
        nc[y][2] = double_bits((unsigned)double_bits((unsigned)((m >> 32) & 0xFFFFull))) & v2;
      }
      ACT_LDS_BARRIER();
      H = H2; W = W2;
      zero_input = false;
      have_layer = false; have_keep = false;
    } else {
      if (have_layer && threadIdx.x < 2 * (H >> 1)) s_keep[threadIdx.x] = s_m[threadIdx.x];   // (the thread's own words)
      have_keep = have_layer;
      have_layer = false;
      // ---- 3x3 stride-2 conv, padding 1, computed everywhere: output pixel (Y, X) is constant iff rows 2Y-1 .. 2Y+1, cols 2X-1 ..
      // 2X+1 are; the padding enters at Y = 0 / X = 0 only (2Y+1 <= H-1 for even H)
      const int H2 = H >> 1, W2 = W >> 1;
      u64 v0, v1;
      valid_bits(W2, v0, v1);
      if (row < H2 && half == 0) {
        u64 r0 = 0ull, r1 = 0ull, r2 = 0ull;
        for (int y = max(2 * row - 1, 0); y <= min(2 * row + 1, H - 1); ++y) { r0 |= nc[y][0]; r1 |= nc[y][1]; r2 |= nc[y][2]; }
        const u64 h0 = r0 | (r0 << 1) | (r0 >> 1) | (r1 << 63);
        const u64 h1 = r1 | (r1 << 1) | (r0 >> 63) | (r1 >> 1) | (r2 << 63);
        const u64 h2 = r2 | (r2 << 1) | (r1 >> 63) | (r2 >> 1);
        u64 t0 = (u64)even_bits(h0) | ((u64)even_bits(h1) << 32), t1 = (u64)even_bits(h2);
        if (!zero_input) {
          if (row == 0) { t0 = ~0ull; t1 = ~0ull; }
          t0 |= 1ull;
        }
        tmb[row][0] = t0 & v0; tmb[row][1] = t1 & v1;
      }
      ACT_LDS_BARRIER();
      if (A.kind[st] == 2) {
        // the transition itself over a tile list: its 2x2-output tile (ty, tx) is computed iff one of its four output pixels is
        // not constant (W2 <= 96: the tile row is one word). The NEXT layer still sees the pixel rows: a pixel that is computed
        // inside such a tile although its window is constant holds the constant to float32 rounding, as it does when the whole
        // map is computed.
        const int TH = H2 >> 1, TW = W2 >> 1;
        u64 mine = 0ull;
        if (row < TH) {
          if (half == 0) {
            const u64 p0 = tmb[2 * row][0] | tmb[2 * row + 1][0], p1 = tmb[2 * row][1] | tmb[2 * row + 1][1];
            mine = (u64)even_bits(p0 | (p0 >> 1)) | ((u64)even_bits(p1 | (p1 >> 1)) << 32);
          }
          A.tile_mask[(((size_t)slot * A.batch + b) * A.mask_th + row) * 2 + half] = mine;
        }
        emit_slot(A, slot, b, row, half, TH, TW, mine, s_scan, s_m, s_pre);
        ++slot;
      }
      if (row < H2 && half == 0) { nc[row][0] = tmb[row][0]; nc[row][1] = tmb[row][1]; nc[row][2] = 0ull; }
      ACT_LDS_BARRIER();
      H = H2; W = W2;
      zero_input = false;   // (conv + BatchNorm + ReLU of a zero map is relu(shift))
    }
  }
}

// batch > 1: number the lists (grid = (batch, n_slots)); an image's entries follow those of the earlier images
__global__ __launch_bounds__(NT) void bev_tile_list_kernel(ActArgs A) {
  __shared__ int s_scan[NT / 64];
  __shared__ u64 s_m[MAX_H];
  __shared__ int s_pre[MAX_H];
  const int b = blockIdx.x, l = blockIdx.y, TH = A.slot_th[l < MAX_SLOTS ? l : 0], TW = A.slot_tw[l < MAX_SLOTS ? l : 0], tiles = TH * TW;
  int base = 0;
  for (int q = 0; q < b; ++q) base += A.counts[l * A.batch + q];
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  const u64 mine = row < TH ? A.tile_mask[(((size_t)l * A.batch + b) * A.mask_th + row) * 2 + half] : 0ull;
  int total;
  const int at = base + sessd_block_exscan<NT>(__popcll(mine), s_scan, &total);
  if (row < TH) { s_m[threadIdx.x] = mine; s_pre[threadIdx.x] = at; }
  __syncthreads();
  emit_words(s_m, s_pre, 2 * TH, TW, b * tiles, A.tile_list + (size_t)l * A.list_cap, A.list_cap);
  if (b == A.batch - 1 && threadIdx.x == 0) A.n_list[l] = base + total;
}

struct FillJobs {
  int njobs;
  int blk_off[MAX_FILL_JOBS + 1];   // first block of every job (a job = tile-pair chunks of 256 x channel groups of FILL_CG x batch blocks)
  sessd_fill_tiles_job_t J[MAX_FILL_JOBS];
};

// One block = 256 threads = 256 pairs of adjacent tiles (2 tx, 2 tx + 1) of one (job, image, group of FILL_CG channels): a thread
// tests its pair's mask bits once and writes up to two 16-byte rows-of-four-pixels twice per channel.
__global__ __launch_bounds__(256) void fill_inactive_tiles_kernel(FillJobs Q, int batch) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < MAX_FILL_JOBS; ++q)
    if (q < Q.njobs && (int)blockIdx.x >= Q.blk_off[q]) j = q;
  // static-index copies (a dynamically indexed kernel-argument array goes to scratch)
  sessd_fill_tiles_job_t J = Q.J[0];
  int off = Q.blk_off[0];
#pragma unroll
  for (int q = 1; q < MAX_FILL_JOBS; ++q)
    if (q == j) { J = Q.J[q]; off = Q.blk_off[q]; }
  const int h = J.h, w = J.w;
  const int cgroups = sessd_divup(J.cout, FILL_CG);
  if (J.tile == 4) {
    // 4x4-pixel tiles (the output of a transposed conv over 2x2 tiles of its input): a thread per tile, four 16-byte rows per
    // channel; the constant depends on the output parity class: value[(py * 2 + px) * cout + co]
    const int th = h >> 2, tw = w >> 2, tiles = th * tw;
    const int chunks = sessd_divup(tiles, 256);
    int r = (int)blockIdx.x - off;
    const int chunk = r % chunks; r /= chunks;
    const int cg = r % cgroups, b = r / cgroups;
    const int t = chunk * 256 + threadIdx.x;
    if (t >= tiles) return;
    const int ty = t / tw, tx = t - ty * tw;
    const u64 word = J.tile_mask[((size_t)b * J.mask_th + ty) * 2 + (tx >> 6)];
    if ((word >> (tx & 63)) & 1ull) return;
    const size_t plane = (size_t)h * w;
    float* o = J.out + ((size_t)b * J.cout + (size_t)cg * FILL_CG) * plane + (size_t)(4 * ty) * w + 4 * tx;
    const int nco = min(FILL_CG, J.cout - cg * FILL_CG);
    for (int cc = 0; cc < nco; ++cc, o += plane) {
      const int co = cg * FILL_CG + cc;
      const float c00 = J.value[co], c01 = J.value[J.cout + co], c10 = J.value[2 * J.cout + co], c11 = J.value[3 * J.cout + co];
      const float4 e = make_float4(c00, c01, c00, c01), d = make_float4(c10, c11, c10, c11);
      *reinterpret_cast<float4*>(o) = e;
      *reinterpret_cast<float4*>(o + w) = d;
      *reinterpret_cast<float4*>(o + 2 * w) = e;
      *reinterpret_cast<float4*>(o + 3 * w) = d;
    }
    return;
  }
  const int th = h >> 1, tw = w >> 1, pairs = th * (tw >> 1);
  const int chunks = sessd_divup(pairs, 256);
  int r = (int)blockIdx.x - off;
  const int chunk = r % chunks; r /= chunks;
  const int cg = r % cgroups, b = r / cgroups;
  const int p = chunk * 256 + threadIdx.x;
  if (p >= pairs) return;
  const int ty = p / (tw >> 1), tx = 2 * (p - ty * (tw >> 1));
  const u64 word = J.tile_mask[((size_t)b * J.mask_th + ty) * 2 + (tx >> 6)];   // tx even: both bits in one word
  bool on0 = (word >> (tx & 63)) & 1ull, on1 = (word >> ((tx & 63) + 1)) & 1ull;
  if (on0 && on1) return;
  if (J.near_mask && J.near_kind != 0) {
    // the reader runs over a tile list on a grid TWICE AS COARSE (round 5): near_kind 1 = it touches this map only inside its listed
    // tiles (the residual of the transposed pair: pair tile (Ty, Tx) = 4x4 output pixels = tiles (2Ty .. 2Ty+1, 2Tx .. 2Tx+1) here);
    // near_kind 2 = a 3x3 stride-2 layer whose listed 2x2-OUTPUT tile (Ty, Tx) reads input pixel rows 4Ty-1 .. 4Ty+3, i.e. tiles
    // 2Ty-1 .. 2Ty+1 of this map: tile ty is reached from Ty = ty / 2 and, for odd ty, Ty + 1. tx is even, tx + 1 odd.
    const int cth = th >> 1, ctw = tw >> 1, Tx = tx >> 1;
    const int ny = (J.near_kind == 2 && (ty & 1)) ? 2 : 1;
    bool n0 = false, n1 = false;
    for (int q = 0; q < ny; ++q) {
      const int Ty = (ty >> 1) + q;
      if (Ty >= cth) continue;
      const uint64_t* rowp = J.near_mask + ((size_t)b * J.mask_th + Ty) * 2;
      const bool a = Tx < ctw && ((rowp[Tx >> 6] >> (Tx & 63)) & 1ull);
      const bool c = (J.near_kind == 2) && (Tx + 1 < ctw) && ((rowp[(Tx + 1) >> 6] >> ((Tx + 1) & 63)) & 1ull);
      n0 |= a;
      n1 |= a | c;
    }
    on0 |= !n0;   // (as good as computed: nobody reads it)
    on1 |= !n1;
    if (on0 && on1) return;
  } else if (J.near_mask) {
    // the only reader of this map is a 3x3 layer that runs over ITS tile list (near_mask): a tile it cannot reach -- none of the
    // 3x3 tiles around it is on that list -- is never read and keeps whatever it holds
    bool n0 = false, n1 = false;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = ty + dy;
      if (y < 0 || y >= th) continue;
      const uint64_t* rowp = J.near_mask + ((size_t)b * J.mask_th + y) * 2;
      const u64 w0 = rowp[0], w1 = rowp[1];
      // bits tx - 1 .. tx + 2 of the 128-bit row (beyond the row: zero)
      unsigned win = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = tx - 1 + k;
        const u64 wd = x < 64 ? w0 : w1;
        if (x >= 0 && x < tw) win |= (unsigned)((wd >> (x & 63)) & 1ull) << k;
      }
      n0 |= (win & 7u) != 0;
      n1 |= (win & 14u) != 0;
    }
    on0 |= !n0;   // (as good as computed: nobody fills it)
    on1 |= !n1;
    if (on0 && on1) return;
  }
  // one mask test for FILL_CG channels (first version: a block per channel -- ten thousand blocks that mostly tested a word and
  // left; the launch's time was their scheduling, not its bytes)
  const size_t plane = (size_t)h * w;
  float* o = J.out + ((size_t)b * J.cout + (size_t)cg * FILL_CG) * plane + (size_t)(2 * ty) * w + 2 * tx;
  const int nco = min(FILL_CG, J.cout - cg * FILL_CG);
  // J.tile == 3 (sessd_fill_tiles_job_t.tile = 6): one constant per output parity class, value[(py * 2 + px) * cout + co] -- the
  // output of a 3x3 layer whose input is constant per parity class (conv_0 / conv_1 behind the transposed convs): a 2x2 tile is one
  // period
  const bool par = J.tile == 3;
  if (!on0 && !on1) {
    for (int cc = 0; cc < nco; ++cc, o += plane) {
      const int co = cg * FILL_CG + cc;
      const float c00 = J.value[co], c01 = par ? J.value[J.cout + co] : c00, c10 = par ? J.value[2 * J.cout + co] : c00,
                  c11 = par ? J.value[3 * J.cout + co] : c00;
      *reinterpret_cast<float4*>(o) = make_float4(c00, c01, c00, c01);
      *reinterpret_cast<float4*>(o + w) = make_float4(c10, c11, c10, c11);
    }
  } else {
    float* q = on0 ? o + 2 : o;
    for (int cc = 0; cc < nco; ++cc, q += plane) {
      const int co = cg * FILL_CG + cc;
      const float c00 = J.value[co], c01 = par ? J.value[J.cout + co] : c00, c10 = par ? J.value[2 * J.cout + co] : c00,
                  c11 = par ? J.value[3 * J.cout + co] : c00;
      *reinterpret_cast<float2*>(q) = make_float2(c00, c01);
      *reinterpret_cast<float2*>(q + w) = make_float2(c10, c11);
    }
  }
}

}  // namespace

extern "C" {

// bytes of the `counts` scratch of sessd_bev_tile_activity
size_t sessd_bev_tile_activity_workspace_bytes(int batch, int n_slots) {
  if (batch < 1 || n_slots < 1 || n_slots > MAX_SLOTS) return 0;
  return sessd_align((size_t)batch * n_slots * sizeof(int), 256);
}

// Computed 2x2-output tiles of a chain of 3x3 layers over an (h, w) map that is zero except at the pixels (y, x) of `indices` rows
// (image, z, y, x) -- the last sparse level of SpMiddleFHD, count on the device. steps[n_steps] (HOST ints): 0 = a 3x3 stride-1
// layer (conv + BatchNorm + ReLU), which takes the next slot; 1 = a 3x3 stride-2 padding-1 layer computed over the whole map (the
// resolution halves). Per slot s (n_slots of them, <= 6):
//   tile_mask [n_slots][batch][h/2][2] 64-bit words: bit tx of row ty's 128 bits = tile (ty, tx) of that layer is computed (rows
//   beyond the layer's own h_s / 2 unused), tile_list [n_slots][list_cap] entries image * tiles_s + tile in ascending order,
//   n_list [n_slots]; list_cap >= batch * h/2 * w/2 never truncates.
int sessd_bev_tile_activity(const int32_t* indices, const int32_t* n_dev, int n_cap, int batch, int h, int w, const int32_t* steps,
                            int n_steps, uint64_t* tile_mask, int32_t* tile_list, int32_t* n_list, int list_cap, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  if (!indices || !n_dev || n_cap < 1 || batch < 1 || h < 2 || w < 2 || !steps || n_steps < 1 || n_steps > MAX_STEPS || !tile_mask ||
      !tile_list || !n_list || list_cap < 1 || h > MAX_H || w > MAX_W)
    return SESSD_EINVAL;
  ActArgs A;
  A.indices = indices; A.n_dev = n_dev; A.n_cap = n_cap; A.batch = batch; A.h = h; A.w = w;
  A.n_steps = n_steps; A.n_slots = 0; A.list_cap = list_cap; A.mask_th = h / 2;
  int ch = h, cw = w;
  for (int s = 0; s < MAX_STEPS; ++s) A.kind[s] = 0;
  for (int s = 0; s < MAX_SLOTS; ++s) A.slot_th[s] = A.slot_tw[s] = 0;
  for (int s = 0; s < n_steps; ++s) {
    if ((ch & 1) || (cw & 1) || ch < 2 || cw < 2) return SESSD_EINVAL;   // every step works on 2x2 tiles / halves the map
    A.kind[s] = steps[s];
    if (steps[s] == 0) {
      if (A.n_slots == MAX_SLOTS) return SESSD_EINVAL;
      A.slot_th[A.n_slots] = ch / 2; A.slot_tw[A.n_slots] = cw / 2;
      ++A.n_slots;
    } else if (steps[s] == 3) {   // 2x2 tiles of the current map, one word per tile row
      if (A.n_slots == MAX_SLOTS || cw / 2 > 64) return SESSD_EINVAL;
      A.slot_th[A.n_slots] = ch / 2; A.slot_tw[A.n_slots] = cw / 2;
      ++A.n_slots;
    } else if (steps[s] == 4) {   // the output of the step-3 transposed conv in front: twice the resolution, no slot
      if (s == 0 || steps[s - 1] != 3 || 2 * ch > MAX_H || 2 * cw > MAX_W || 2 * ch > NT) return SESSD_EINVAL;
      ch *= 2; cw *= 2;
    } else if (steps[s] == 1 || steps[s] == 2) {
      ch /= 2; cw /= 2;
      if (steps[s] == 2) {   // the transition takes a slot of 2x2 tiles of ITS output
        if (A.n_slots == MAX_SLOTS || (ch & 1) || (cw & 1) || cw / 2 > 64) return SESSD_EINVAL;
        A.slot_th[A.n_slots] = ch / 2; A.slot_tw[A.n_slots] = cw / 2;
        ++A.n_slots;
      }
    } else {
      return SESSD_EINVAL;
    }
  }
  if (A.n_slots < 1) return SESSD_EINVAL;
  if (batch > 1 && (!workspace || workspace_bytes < sessd_bev_tile_activity_workspace_bytes(batch, A.n_slots))) return SESSD_EWORKSPACE;
  A.tile_mask = (u64*)tile_mask; A.tile_list = tile_list; A.n_list = n_list; A.counts = (int*)workspace;
  SESSD_LAUNCH(bev_tile_activity_kernel, dim3(batch), dim3(NT), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  if (batch > 1) {
    SESSD_LAUNCH(bev_tile_list_kernel, dim3(batch, A.n_slots), dim3(NT), 0, stream, A);
    SESSD_CHECK_LAUNCH();
  }
  return SESSD_OK;
}

// out[b][co][tile pixels] = value[co] for every tile of job j whose bit in tile_mask is 0, for up to 12 jobs (tile = 4: 4x4-pixel
// tiles with one value per output parity class, value[(py * 2 + px) * cout + co]; tile = 6: 2x2-pixel tiles with such a value table) (out (batch, cout, h, w),
// value[cout], tile_mask (batch, mask_th, 2) words, cout, h, w, mask_th) in one launch.
int sessd_fill_inactive_tiles(const sessd_fill_tiles_job_t* jobs, int n_jobs, int batch, hipStream_t stream) {
  if (!jobs || n_jobs < 1 || n_jobs > MAX_FILL_JOBS || batch < 1) return SESSD_EINVAL;
  FillJobs Q;
  Q.njobs = n_jobs;
  int blk = 0;
  for (int j = 0; j < n_jobs; ++j) {
    const sessd_fill_tiles_job_t& S = jobs[j];
    // (w % 4 == 0: a thread owns two adjacent 2x2 tiles, or one 4x4 tile = 16-byte aligned rows of four pixels)
    const int tile = S.tile == 4 ? 4 : 2;
    if (!S.out || !S.value || !S.tile_mask || S.cout < 1 || S.h < tile || S.w < 4 || (S.h % tile) || (S.w & 3) || S.mask_th < S.h / tile ||
        S.w / tile > 128 || (S.tile != 0 && S.tile != 2 && S.tile != 4 && S.tile != 6) || (S.near_mask && tile != 2))
      return SESSD_EINVAL;
    Q.J[j] = S;
    Q.J[j].tile = S.tile == 6 ? 3 : tile;   // (the kernel's codes: 4 = 4x4 blocks with parity constants, 3 = 2x2 tiles with parity constants, 2 = 2x2 tiles)
    Q.blk_off[j] = blk;
    blk += (tile == 4 ? sessd_divup((S.h / 4) * (S.w / 4), 256) : sessd_divup((S.h / 2) * (S.w / 4), 256)) * sessd_divup(S.cout, FILL_CG) * batch;
  }
  for (int j = n_jobs; j < MAX_FILL_JOBS; ++j) { Q.J[j] = jobs[0]; Q.blk_off[j] = blk; }
  Q.blk_off[MAX_FILL_JOBS] = blk;
  SESSD_LAUNCH(fill_inactive_tiles_kernel, dim3(blk), dim3(256), 0, stream, Q, batch);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
