// Which 2x2-output tiles of the first SSFA layers have anything to compute (gfx950; round 4).
//
// The BEV map that enters the neck (det3d/models/backbones/scn.py:179-183: `.dense()` of the last sparse level, 64 channels x 2
// z-slices per pixel) is ZERO outside the sparse backbone's sites -- 7 % of the 200 x 176 pixels of a KITTI-shaped scan hold one.
// rpn_v1.py:135-160 then runs three 3x3 stride-1 conv + BatchNorm + ReLU layers over the whole map. Away from the sites that is
// arithmetic on constants: conv(0) = 0 -> BatchNorm -> ReLU gives the same value c1[co] in every pixel whose 3x3 window holds no
// site, the next layer maps a window of c1 to c2[co], and so on (the image border, where zero padding enters the window, is not
// constant from the second layer on). A 2x2-output Winograd tile has something to compute iff its 4x4 input patch touches a
// non-constant pixel of the layer's input; on 20 k-point scans that is 18 % / 29 % / 39 % of the tiles of b0.0 / b0.1 / b0.2.
//   bev_tile_activity   per image: site pixels -> LDS byte map; per layer: tile mask (4x4 patch test, + the border ring from the
//                       second layer on), ordered list of the active tiles (entry image * tiles + tile; ascending: deterministic, no
//                       atomics), the next layer's non-constant map = the active tiles' pixels. One launch at batch 1; with more
//                       images a second launch numbers the lists (an image's base is the sum of the earlier images' counts).
//   fill_inactive_tiles the layers' constants into the tiles nobody computes (<= 4 layers per launch).
// The convolutions themselves: conv3x3s1_winograd_sk_kernel<.., LIST = true> (dense_wino_sk.hip) over the list.
// The constants come from the host (float64 over the folded weights: sessd_hip.engine); a computed tile and a filled tile agree
// to float32 rounding of that chain (1e-7 relative), bit-exactly for the first layer (0 * U = 0).
#include "common.hpp"
#include "sessd_hip_types.h"

namespace {

constexpr int NT = 1024;
constexpr int MAX_PIX = 40960;    // H * W of one image (LDS byte map)
constexpr int MAX_LAYERS = 4;

struct ActArgs {
  const int* indices;   // (n, 4) rows (image, z, y, x) of the last sparse level
  const int* n_dev;
  int n_cap, batch, h, w, th, tw, n_layers, list_cap;
  unsigned char* tile_mask;   // [n_layers][batch][th * tw]
  int* tile_list;             // [n_layers][list_cap]
  int* n_list;                // [n_layers]
  int* counts;                // [n_layers][batch]
};

// ordered compaction of one image's tile mask (bytes, in LDS or global) into the layer's list starting at `base`
template <typename MaskPtr>
__device__ __forceinline__ int compact_tiles(MaskPtr tm, int tiles, int image, int base, int* list, int list_cap, int* s_scan) {
  const int per = (tiles + NT - 1) / NT;
  const int t0 = threadIdx.x * per, t1 = min(tiles, t0 + per);
  int c = 0;
  for (int t = t0; t < t1; ++t) c += tm[t] ? 1 : 0;
  int total;
  int at = base + sessd_block_exscan<NT>(c, s_scan, &total);
  for (int t = t0; t < t1; ++t)
    if (tm[t]) {
      if (at < list_cap) list[at] = image * tiles + t;
      ++at;
    }
  return total;
}

__global__ __launch_bounds__(NT) void bev_tile_activity_kernel(ActArgs A) {
  __shared__ unsigned char nc[MAX_PIX];        // non-constant pixels of the current layer's input
  __shared__ unsigned char tm[MAX_PIX / 4];    // active tiles of the current layer
  __shared__ int s_scan[NT / 64];
  const int b = blockIdx.x, H = A.h, W = A.w, TH = A.th, TW = A.tw, tiles = TH * TW;
  for (int p = threadIdx.x; p < H * W; p += NT) nc[p] = 0;
  __syncthreads();
  const int n = min(A.n_dev[0], A.n_cap);
  for (int i = threadIdx.x; i < n; i += NT) {
    const int4 c = *reinterpret_cast<const int4*>(A.indices + (size_t)i * 4);
    if (c.x == b && c.z >= 0 && c.z < H && c.w >= 0 && c.w < W) nc[c.z * W + c.w] = 1;
  }
  __syncthreads();
  for (int l = 0; l < A.n_layers; ++l) {
    for (int t = threadIdx.x; t < tiles; t += NT) {
      const int ty = t / TW, tx = t - ty * TW;
      // from the second layer on the input constant is not zero: zero padding makes the border ring a computed region
      bool on = l > 0 && (ty == 0 || ty == TH - 1 || tx == 0 || tx == TW - 1);
      const int ya = max(2 * ty - 1, 0), yb = min(2 * ty + 2, H - 1), xa = max(2 * tx - 1, 0), xb = min(2 * tx + 2, W - 1);
      for (int y = ya; y <= yb && !on; ++y)
        for (int x = xa; x <= xb; ++x) on = on || nc[y * W + x];
      tm[t] = on ? 1 : 0;
    }
    __syncthreads();
    unsigned char* gm = A.tile_mask + ((size_t)l * A.batch + b) * tiles;
    for (int t = threadIdx.x; t < tiles; t += NT) gm[t] = tm[t];
    if (A.batch == 1) {
      const int total = compact_tiles(tm, tiles, 0, 0, A.tile_list + (size_t)l * A.list_cap, A.list_cap, s_scan);
      if (threadIdx.x == 0) A.n_list[l] = total;
    } else {
      int c = 0;
      for (int t = threadIdx.x; t < tiles; t += NT) c += tm[t];
      int total;
      sessd_block_exscan<NT>(c, s_scan, &total);
      if (threadIdx.x == 0) A.counts[l * A.batch + b] = total;
    }
    // the layer's output is non-constant exactly in its active tiles
    for (int p = threadIdx.x; p < H * W; p += NT) {
      const int y = p / W, x = p - y * W;
      nc[p] = tm[(y >> 1) * TW + (x >> 1)];
    }
    __syncthreads();
  }
}

// batch > 1: number the lists (grid = (batch, n_layers)); an image's entries follow those of the earlier images
__global__ __launch_bounds__(NT) void bev_tile_list_kernel(ActArgs A) {
  __shared__ int s_scan[NT / 64];
  const int b = blockIdx.x, l = blockIdx.y, tiles = A.th * A.tw;
  int base = 0;
  for (int q = 0; q < b; ++q) base += A.counts[l * A.batch + q];
  const unsigned char* gm = A.tile_mask + ((size_t)l * A.batch + b) * tiles;
  const int total = compact_tiles(gm, tiles, b, base, A.tile_list + (size_t)l * A.list_cap, A.list_cap, s_scan);
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
  if (t >= tiles || J.tile_mask[(size_t)b * tiles + t]) return;
  const int ty = t / tw, tx = t - ty * tw;
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
//   tile_mask [n_layers][batch][h/2 * w/2] bytes, tile_list [n_layers][list_cap] entries image * (h/2 * w/2) + tile in ascending
//   order, n_list [n_layers]; list_cap >= batch * h/2 * w/2 never truncates.
int sessd_bev_tile_activity(const int32_t* indices, const int32_t* n_dev, int n_cap, int batch, int h, int w, int n_layers,
                            uint8_t* tile_mask, int32_t* tile_list, int32_t* n_list, int list_cap, void* workspace,
                            size_t workspace_bytes, hipStream_t stream) {
  if (!indices || !n_dev || n_cap < 1 || batch < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || h * w > MAX_PIX || n_layers < 1 ||
      n_layers > MAX_LAYERS || !tile_mask || !tile_list || !n_list || list_cap < 1)
    return SESSD_EINVAL;
  if (batch > 1 && (!workspace || workspace_bytes < sessd_bev_tile_activity_workspace_bytes(batch, n_layers))) return SESSD_EWORKSPACE;
  ActArgs A;
  A.indices = indices; A.n_dev = n_dev; A.n_cap = n_cap; A.batch = batch; A.h = h; A.w = w; A.th = h / 2; A.tw = w / 2;
  A.n_layers = n_layers; A.list_cap = list_cap; A.tile_mask = tile_mask; A.tile_list = tile_list; A.n_list = n_list;
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
