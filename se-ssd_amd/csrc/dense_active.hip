// Which 2x2-output tiles of the first SSFA layers have anything to compute (gfx950; round 4).
//
// The BEV map that enters the neck (det3d/models/backbones/scn.py:179-183: `.dense()` of the last sparse level, 64 channels x 2
// z-slices per pixel) is ZERO outside the sparse backbone's sites -- 7 % of the 200 x 176 pixels of a KITTI-shaped scan hold one.
// rpn_v1.py:135-160 then runs three 3x3 stride-1 conv + BatchNorm + ReLU layers over the whole map. Away from the sites that is
// arithmetic on constants: conv(0) = 0 -> BatchNorm -> ReLU gives the same value c1[co] in every pixel whose 3x3 window holds no
// site, the next layer maps a window of c1 to c2[co], and so on (the image border, where zero padding enters the window, is not
// constant from the second layer on). A 2x2-output Winograd tile has something to compute iff its 4x4 input patch touches a
// non-constant pixel of the layer's input; on 20 k-point scans that is 18 % / 29 % / 39 % of the tiles of b0.0 / b0.1 / b0.2.
//   bev_tile_activity   per image: site pixels -> bit rows in LDS; per layer: tile bits (4x4 patch test = OR of four pixel rows,
//                       shifted-OR over the patch columns, every second bit; + the border ring from the second layer on), ordered
//                       list of the active tiles (entry image * tiles + tile; ascending: deterministic, no atomics), the next layer's
//                       non-constant rows = the tile rows with every bit doubled. One thread per tile row: a few microseconds. One
//                       launch at batch 1; with more images a second launch numbers the lists (an image's base is the sum of the
//                       earlier images' counts). (First version: byte maps walked by one workgroup -- 104 us.)
//   fill_inactive_tiles the layers' constants into the tiles nobody computes (<= 4 layers per launch).
// The convolutions themselves: conv3x3s1_winograd_sk_kernel<.., LIST = true> (dense_wino_sk.hip) over the list.
// The constants come from the host (float64 over the folded weights: sessd_hip.engine); a computed tile and a filled tile agree
// to float32 rounding of that chain (1e-7 relative), bit-exactly for the first layer (0 * U = 0).
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_H = 256, MAX_W = 192;   // one image's pixel rows as 3 x 64-bit words in LDS; a thread pair per tile row
constexpr int MAX_LAYERS = 4;
typedef unsigned long long u64;

struct ActArgs {
  const int* indices;   // (n, 4) rows (image, z, y, x) of the last sparse level
  const int* n_dev;
  int n_cap, batch, h, w, th, tw, n_layers, list_cap;
  u64* tile_mask;       // [n_layers][batch][th][2]: bit tx of the row's 128-bit word = tile (ty, tx) is computed
  int* tile_list;       // [n_layers][list_cap]
  int* n_list;          // [n_layers]
  int* counts;          // [n_layers][batch]
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

// entries of one 64-bit word of a tile row, ascending, starting at list[at]
__device__ __forceinline__ void emit_word(u64 m, int first, int at, int* list, int list_cap) {
  while (m) {
    const int bit = __ffsll((long long)m) - 1;
    m &= m - 1;
    if (at < list_cap) list[at] = first + bit;
    ++at;
  }
}

// One workgroup per image. Pixel rows and tile rows are bit vectors: a tile row is OR of four pixel rows, shifted-OR over the four
// patch columns, every second bit kept -- one thread per tile row and layer.
__global__ __launch_bounds__(NT) void bev_tile_activity_kernel(ActArgs A) {
  __shared__ u64 nc[MAX_H][3];            // non-constant pixels of the current layer's input
  __shared__ u64 tmb[MAX_H / 2][2];       // computed tiles of the current layer
  __shared__ int s_scan[NT / 64];
  const int b = blockIdx.x, H = A.h, W = A.w, TH = A.th, TW = A.tw;
  for (int i = threadIdx.x; i < H * 3; i += NT) (&nc[0][0])[i] = 0ull;
  __syncthreads();
  const int n = min(A.n_dev[0], A.n_cap);
  for (int i = threadIdx.x; i < n; i += NT) {
    const int4 c = *reinterpret_cast<const int4*>(A.indices + (size_t)i * 4);
    if (c.x == b && c.z >= 0 && c.z < H && c.w >= 0 && c.w < W) atomicOr(&nc[c.z][c.w >> 6], 1ull << (c.w & 63));
  }
  __syncthreads();
  // valid tile bits of a row: TW bits over two words
  const u64 v0 = TW >= 64 ? ~0ull : ((1ull << TW) - 1ull), v1 = TW > 64 ? ((1ull << (TW - 64)) - 1ull) : 0ull;
  for (int l = 0; l < A.n_layers; ++l) {
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;   // thread pair = one tile row; each thread owns one 64-bit word of it
    u64 mine = 0ull;
    if (row < TH) {
      u64 r0 = 0ull, r1 = 0ull, r2 = 0ull;
      for (int y = max(2 * row - 1, 0); y <= min(2 * row + 2, H - 1); ++y) { r0 |= nc[y][0]; r1 |= nc[y][1]; r2 |= nc[y][2]; }
      // patch columns 2 tx - 1 .. 2 tx + 2 at bit 2 tx: r | r << 1 | r >> 1 | r >> 2 over the 192-bit row
      const u64 h0 = r0 | (r0 << 1) | (r0 >> 1) | (r1 << 63) | (r0 >> 2) | (r1 << 62);
      const u64 h1 = r1 | (r1 << 1) | (r0 >> 63) | (r1 >> 1) | (r2 << 63) | (r1 >> 2) | (r2 << 62);
      const u64 h2 = r2 | (r2 << 1) | (r1 >> 63) | (r2 >> 1) | (r2 >> 2);
      u64 t0 = (u64)even_bits(h0) | ((u64)even_bits(h1) << 32), t1 = (u64)even_bits(h2);
      // from the second layer on the input constant is not zero: zero padding makes the border ring a computed region
      if (l > 0) {
        if (row == 0 || row == TH - 1) { t0 = ~0ull; t1 = ~0ull; }
        t0 |= 1ull;
        if (TW - 1 < 64) t0 |= 1ull << (TW - 1); else t1 |= 1ull << (TW - 1 - 64);
      }
      t0 &= v0; t1 &= v1;
      mine = half ? t1 : t0;
      tmb[row][half] = mine;
      A.tile_mask[(((size_t)l * A.batch + b) * TH + row) * 2 + half] = mine;
    }
    int total;
    const int at = sessd_block_exscan<NT>(__popcll(mine), s_scan, &total);   // thread order = (row, word) = ascending tile order
    if (A.batch == 1) {
      if (row < TH) emit_word(mine, row * TW + 64 * half, at, A.tile_list + (size_t)l * A.list_cap, A.list_cap);
      if (threadIdx.x == 0) A.n_list[l] = total;
    } else if (threadIdx.x == 0) {
      A.counts[l * A.batch + b] = total;
    }
    __syncthreads();
    // the layer's output is non-constant exactly in its computed tiles: pixel rows 2 ty, 2 ty + 1 = the tile row, every bit twice
    if (row < TH && half == 0) {
      const u64 t0 = tmb[row][0], t1 = tmb[row][1];
      const u64 p0 = double_bits((unsigned)t0), p1 = double_bits((unsigned)(t0 >> 32)), p2 = double_bits((unsigned)t1);
      nc[2 * row][0] = p0; nc[2 * row][1] = p1; nc[2 * row][2] = p2;
      nc[2 * row + 1][0] = p0; nc[2 * row + 1][1] = p1; nc[2 * row + 1][2] = p2;
    }
    __syncthreads();
  }
}

// batch > 1: number the lists (grid = (batch, n_layers)); an image's entries follow those of the earlier images
__global__ __launch_bounds__(NT) void bev_tile_list_kernel(ActArgs A) {
  __shared__ int s_scan[NT / 64];
  const int b = blockIdx.x, l = blockIdx.y, TH = A.th, TW = A.tw, tiles = TH * TW;
  int base = 0;
  for (int q = 0; q < b; ++q) base += A.counts[l * A.batch + q];
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  const u64 mine = row < TH ? A.tile_mask[(((size_t)l * A.batch + b) * TH + row) * 2 + half] : 0ull;
  int total;
  const int at = base + sessd_block_exscan<NT>(__popcll(mine), s_scan, &total);
  if (row < TH) emit_word(mine, b * tiles + row * TW + 64 * half, at, A.tile_list + (size_t)l * A.list_cap, A.list_cap);
  if (b == A.batch - 1 && threadIdx.x == 0) A.n_list[l] = base + total;
}

struct FillJobs {
  int njobs;
  sessd_fill_tiles_job_t J[MAX_LAYERS];
};

// grid = (tile chunks of 256, cout, batch * njobs): thread = one 2x2 tile of one channel; adjacent threads = adjacent tiles of a row
__global__ __launch_bounds__(256) void fill_inactive_tiles_kernel(FillJobs Q, int batch, int h, int w) {
  const int th = h >> 1, tw = w >> 1, tiles = th * tw;
  const int j = blockIdx.z / batch, b = blockIdx.z - j * batch;
  // static-index copy (a dynamically indexed kernel-argument array goes to scratch)
  sessd_fill_tiles_job_t J = Q.J[0];
#pragma unroll
  for (int q = 1; q < MAX_LAYERS; ++q)
    if (q == j) J = Q.J[q];
  const int co = blockIdx.y;
  if (co >= J.cout) return;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= tiles) return;
  const int ty = t / tw, tx = t - ty * tw;
  if ((J.tile_mask[((size_t)b * th + ty) * 2 + (tx >> 6)] >> (tx & 63)) & 1ull) return;
  const float c = J.value[co];
  float* o = J.out + (((size_t)b * J.cout + co) * h + 2 * ty) * w + 2 * tx;
  *reinterpret_cast<float2*>(o) = make_float2(c, c);
  *reinterpret_cast<float2*>(o + w) = make_float2(c, c);
}

}  // namespace

extern "C" {

// bytes of the `counts` scratch of sessd_bev_tile_activity
size_t sessd_bev_tile_activity_workspace_bytes(int batch, int n_layers) {
  if (batch < 1 || n_layers < 1 || n_layers > MAX_LAYERS) return 0;
  return sessd_align((size_t)batch * n_layers * sizeof(int), 256);
}

// Active 2x2-output tiles of the first `n_layers` 3x3 stride-1 layers over an (h, w) map that is zero except at the pixels
// (y, x) of `indices` rows (image, z, y, x) -- the last sparse level of SpMiddleFHD, count on the device.
//   tile_mask [n_layers][batch][h/2][2] 64-bit words (bit tx of a row's 128 bits = tile (ty, tx) is computed), tile_list [n_layers][list_cap] entries image * (h/2 * w/2) + tile in ascending
//   order, n_list [n_layers]; list_cap >= batch * h/2 * w/2 never truncates.
int sessd_bev_tile_activity(const int32_t* indices, const int32_t* n_dev, int n_cap, int batch, int h, int w, int n_layers,
                            uint64_t* tile_mask, int32_t* tile_list, int32_t* n_list, int list_cap, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  if (!indices || !n_dev || n_cap < 1 || batch < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || n_layers < 1 ||
      n_layers > MAX_LAYERS || !tile_mask || !tile_list || !n_list || list_cap < 1 || h > MAX_H || w > MAX_W)
    return SESSD_EINVAL;
  if (batch > 1 && (!workspace || workspace_bytes < sessd_bev_tile_activity_workspace_bytes(batch, n_layers))) return SESSD_EWORKSPACE;
  ActArgs A;
  A.indices = indices; A.n_dev = n_dev; A.n_cap = n_cap; A.batch = batch; A.h = h; A.w = w; A.th = h / 2; A.tw = w / 2;
  A.n_layers = n_layers; A.list_cap = list_cap; A.tile_mask = (u64*)tile_mask; A.tile_list = tile_list; A.n_list = n_list;
  A.counts = (int*)workspace;
  SESSD_LAUNCH(bev_tile_activity_kernel, dim3(batch), dim3(NT), 0, stream, A);
  SESSD_CHECK_LAUNCH();
  if (batch > 1) {
    SESSD_LAUNCH(bev_tile_list_kernel, dim3(batch, n_layers), dim3(NT), 0, stream, A);
    SESSD_CHECK_LAUNCH();
  }
  return SESSD_OK;
}

// out[b][co][tile pixels] = value[co] for every tile with tile_mask[b][tile] == 0, for up to 4 (out, value, tile_mask, cout) jobs
// over (batch, ., h, w) maps in one launch.
int sessd_fill_inactive_tiles(const sessd_fill_tiles_job_t* jobs, int n_jobs, int batch, int h, int w, hipStream_t stream) {
  if (!jobs || n_jobs < 1 || n_jobs > MAX_LAYERS || batch < 1 || h < 2 || w < 2 || (h & 1) || (w & 1)) return SESSD_EINVAL;
  FillJobs Q;
  Q.njobs = n_jobs;
  int cmax = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!jobs[j].out || !jobs[j].value || !jobs[j].tile_mask || jobs[j].cout < 1) return SESSD_EINVAL;
    Q.J[j] = jobs[j];
    cmax = jobs[j].cout > cmax ? jobs[j].cout : cmax;
  }
  for (int j = n_jobs; j < MAX_LAYERS; ++j) Q.J[j] = jobs[0];
  const int tiles = (h / 2) * (w / 2);
  SESSD_LAUNCH(fill_inactive_tiles_kernel, dim3(sessd_divup(tiles, 256), cmax, batch * n_jobs), dim3(256), 0, stream, Q, batch, h, w);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
