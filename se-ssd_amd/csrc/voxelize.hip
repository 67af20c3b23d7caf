// Point cloud -> voxels on gfx950, bit-exact with the reference's serial
// first-come-first-served loop
//   det3d/ops/point_cloud/point_cloud_ops_v2.py:9-62 (_points_to_voxel_reverse_kernel)
//   det3d/core/input/voxel_generator.py:24-32       (VoxelGenerator.generate)
// plus the mean-VFE reader det3d/models/readers/voxel_encoder.py:215-220.
//
// The reference walks points in order, gives a voxel the id "order of first
// appearance", keeps the first `max_points` points of every voxel, and BREAKS
// (drops every later point) when a point would open voxel #max_voxels.
// Parallel restatement (no sort, deterministic final state):
//   K1  per point: cell key -> hash slot (atomicCAS) ; insert the point index into
//       the slot's ascending list of the `max_points` smallest indices with a
//       carry-chain of atomicMin (slot r ends up holding the (r+1)-th smallest
//       index whatever the interleaving, because every value offered to slot r
//       except the final minimum is forwarded exactly once to slot r+1).
//   K2a per point: is_first = (list[slot][0] == i) ; per-block counts.
//   K2b exclusive scan of is_first in point order -> voxel id ; the first point
//       whose id == max_voxels defines `cut` (the reference's break index).
//   K3  per voxel: gather the list entries < cut, write voxels / num_points /
//       coordinates / mean feature with 16-byte stores.
// HBM traffic: points read twice (16 B/pt each), outputs written once.
#include "common.hpp"

namespace {

constexpr int VOX_NT = 256;

struct VoxGrid {
  float lo[3];
  float vs[3];
  int g[3];  // gx, gy, gz
};

__global__ __launch_bounds__(VOX_NT) void vox_insert_kernel(const float* __restrict__ points, int P, int ndim,
                                                              VoxGrid G, uint32_t key_base, uint32_t cells,
                                                              uint32_t* __restrict__ keys, uint32_t mask,
                                                              int* __restrict__ lists, int MP,
                                                              int* __restrict__ ent) {
  // blockIdx.y = frame of a batched launch (sessd_voxelize_frames): its points, its slice of `ent`, its key range
  const int fr = blockIdx.y;
  points += (size_t)fr * P * ndim;
  ent += (size_t)fr * P;
  key_base += (uint32_t)fr * cells;
  int i = blockIdx.x * VOX_NT + threadIdx.x;
  if (i >= P) return;
  const float* p = points + (size_t)i * ndim;
  // fp32 subtract, correctly-rounded fp32 divide, floor: exactly numpy/numba float32 semantics.
  float c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    c[j] = floorf(__fdiv_rn(__fsub_rn(p[j], G.lo[j]), G.vs[j]));
    ok = ok && (c[j] >= 0.0f) && (c[j] < (float)G.g[j]);  // NaN -> dropped
  }
  if (!ok) {
    ent[i] = -1;
    return;
  }
  uint32_t key = key_base + ((uint32_t)c[2] * (uint32_t)G.g[1] + (uint32_t)c[1]) * (uint32_t)G.g[0] + (uint32_t)c[0];
  uint32_t slot = sessd_hash_insert(keys, mask, key);
  if (slot == SESSD_HASH_FULL) {  // cannot happen with capacity >= 2 * points (checked by the entry point)
    ent[i] = -1;
    return;
  }
  ent[i] = (int)slot;
  int* L = lists + (size_t)slot * MP;
  int v = i;
  for (int r = 0; r < MP; ++r) {
    int old = atomicMin(&L[r], v);
    if (old == SESSD_SENT) break;
    v = old > v ? old : v;
  }
}

__global__ __launch_bounds__(VOX_NT) void vox_count_kernel(int P, const int* __restrict__ ent,
                                                             const int* __restrict__ lists, int MP,
                                                             int* __restrict__ blk_cnt, int* __restrict__ meta) {
  __shared__ int sm[VOX_NT / 64];
  const int fr = blockIdx.y;   // frame of a batched launch
  ent += (size_t)fr * P;
  blk_cnt += (size_t)fr * gridDim.x;
  meta += fr * 4;
  int i = blockIdx.x * VOX_NT + threadIdx.x;
  // per-FRAME reset of the break index: the workspace is shared by the frames of a batch, and an engine clears its
  // arena once per batch, so a frame that hit max_voxels must not leave its cut behind for the next frame
  if (i == 0) meta[0] = SESSD_SENT;
  int f = 0;
  if (i < P) {
    int e = ent[i];
    f = (e >= 0 && lists[(size_t)e * MP] == i) ? 1 : 0;
  }
  int s = sessd_wave_sum(f);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < VOX_NT / 64; ++w) t += sm[w];
    blk_cnt[blockIdx.x] = t;
  }
}

// meta[0] = cut (break index, SESSD_SENT when no break), meta[1] = unique voxel count.
__global__ __launch_bounds__(VOX_NT) void vox_assign_kernel(int P, const int* __restrict__ ent,
                                                              const int* __restrict__ lists, int MP,
                                                              const int* __restrict__ blk_cnt, int nblk,
                                                              const uint32_t* __restrict__ keys, uint32_t key_base,
                                                              VoxGrid G, int max_voxels, int batch_index,
                                                              const int* __restrict__ prefix_in,
                                                              int* __restrict__ vals, int* __restrict__ entry_of_vid,
                                                              int* __restrict__ coors, int coors_stride,
                                                              int* __restrict__ meta, int* __restrict__ prefix_out,
                                                              uint32_t cells, int batched) {
  __shared__ int sm[VOX_NT / 64];
  __shared__ int s_base;
  // batched launch (sessd_voxelize_frames): blockIdx.y = frame. The frame's first output row is the sum of the EARLIER frames'
  // voxel counts min(unique cells, max_voxels) -- taken from their block counts by every block itself (a per-frame launch reads
  // it from prefix_in, written by the previous frame's launch)
  const int fr = blockIdx.y;
  int out_base = 0;
  if (batched) {
    for (int f = 0; f < fr; ++f) {
      int part = 0;
      for (int b = threadIdx.x; b < nblk; b += VOX_NT) part += blk_cnt[(size_t)f * nblk + b];
      part = sessd_wave_sum(part);
      __syncthreads();
      if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
      __syncthreads();
      int t = 0;
      for (int w = 0; w < VOX_NT / 64; ++w) t += sm[w];
      out_base += t < max_voxels ? t : max_voxels;
    }
    __syncthreads();
    ent += (size_t)fr * P;
    blk_cnt += (size_t)fr * nblk;
    entry_of_vid += (size_t)fr * max_voxels;
    meta += fr * 4;
    key_base += (uint32_t)fr * cells;
    batch_index += fr;
    prefix_out += fr;
  } else {
    out_base = prefix_in[0];
  }
  // base = sum of counts of all preceding blocks
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += VOX_NT) part += blk_cnt[b];
  part = sessd_wave_sum(part);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < VOX_NT / 64; ++w) t += sm[w];
    s_base = t;
  }
  __syncthreads();
  const int base = s_base;

  int i = blockIdx.x * VOX_NT + threadIdx.x;
  int e = -1, f = 0;
  if (i < P) {
    e = ent[i];
    f = (e >= 0 && lists[(size_t)e * MP] == i) ? 1 : 0;
  }
  int total;
  int vid = base + sessd_block_exscan<VOX_NT>(f, sm, &total);
  if (f) {
    if (vid < max_voxels) {
      vals[e] = out_base + vid;
      entry_of_vid[vid] = e;
      uint32_t lin = keys[e] - key_base;
      int cx = (int)(lin % (uint32_t)G.g[0]);
      uint32_t t = lin / (uint32_t)G.g[0];
      int cy = (int)(t % (uint32_t)G.g[1]);
      int cz = (int)(t / (uint32_t)G.g[1]);
      int* co = coors + (size_t)(out_base + vid) * coors_stride;
      if (coors_stride == 4) {
        co[0] = batch_index; co[1] = cz; co[2] = cy; co[3] = cx;
      } else {
        co[0] = cz; co[1] = cy; co[2] = cx;
      }
    } else {
      vals[e] = SESSD_SENT;  // occupied cell whose voxel was cut by max_voxels: reads as absent
      if (vid == max_voxels) meta[0] = i;  // the point at which the reference loop breaks
    }
  }
  if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
    int uniq = base + total;
    meta[1] = uniq;
    int M = uniq < max_voxels ? uniq : max_voxels;
    prefix_out[0] = out_base + M;
  }
}

template <int NDIM>
__global__ __launch_bounds__(VOX_NT) void vox_gather_kernel(const float* __restrict__ points, int ndim_rt,
                                                              const int* __restrict__ lists, int MP,
                                                              const int* __restrict__ entry_of_vid,
                                                              const int* __restrict__ meta, int max_voxels,
                                                              const int* __restrict__ prefix_in,
                                                              float* __restrict__ voxels, int* __restrict__ num_points,
                                                              float* __restrict__ mean, int P) {
  const int ndim = NDIM > 0 ? NDIM : ndim_rt;
  const int fr = blockIdx.y;   // frame of a batched launch: its points, its meta / entry table, its prefix word
  points += (size_t)fr * P * ndim;
  entry_of_vid += (size_t)fr * max_voxels;
  meta += fr * 4;
  prefix_in += fr;
  int vid = blockIdx.x * VOX_NT + threadIdx.x;
  int uniq = meta[1];
  int M = uniq < max_voxels ? uniq : max_voxels;
  if (vid >= M) return;
  const int cut = meta[0];
  const int out_base = prefix_in[0];
  const int* L = lists + (size_t)entry_of_vid[vid] * MP;
  float* vo = voxels + (size_t)(out_base + vid) * MP * ndim;
  int n = 0;
  if (NDIM == 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < MP; ++r) {
      int idx = L[r];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < cut) {  // SESSD_SENT is never < cut
        v = *reinterpret_cast<const float4*>(points + (size_t)idx * 4);
        ++n;
      }
      *reinterpret_cast<float4*>(vo + r * 4) = v;
      // zero-padded slots take part in the sum exactly as in voxel_encoder.py:219
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (mean) {
      float fn = (float)n;
      *reinterpret_cast<float4*>(mean + (size_t)(out_base + vid) * 4) =
          make_float4(__fdiv_rn(acc.x, fn), __fdiv_rn(acc.y, fn), __fdiv_rn(acc.z, fn), __fdiv_rn(acc.w, fn));
    }
  } else {
    float acc[8];
    for (int d = 0; d < ndim; ++d) acc[d] = 0.f;
    for (int r = 0; r < MP; ++r) {
      int idx = L[r];
      bool ok = idx < cut;
      n += ok ? 1 : 0;
      for (int d = 0; d < ndim; ++d) {
        float v = ok ? points[(size_t)idx * ndim + d] : 0.f;
        vo[r * ndim + d] = v;
        acc[d] += v;
      }
    }
    if (mean) {
      float fn = (float)n;
      for (int d = 0; d < ndim; ++d) mean[(size_t)(out_base + vid) * ndim + d] = __fdiv_rn(acc[d], fn);
    }
  }
  num_points[out_base + vid] = n;
}

__global__ void vfe_mean_kernel(const float* __restrict__ voxels, const int* __restrict__ num_points,
                                const int* __restrict__ n_dev, int n_host, int MP, int ndim, int nfeat,
                                float* __restrict__ out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  int n = n_dev ? n_dev[0] : n_host;
  if (v >= n) return;
  const float* src = voxels + (size_t)v * MP * ndim;
  float fn = (float)num_points[v];
  for (int d = 0; d < nfeat; ++d) {
    float s = 0.f;
    for (int r = 0; r < MP; ++r) s += src[r * ndim + d];  // sequential sum over dim=1 like torch.sum on a short axis
    out[(size_t)v * nfeat + d] = __fdiv_rn(s, fn);
  }
}

// Stage a frame into the engine's fixed-capacity input buffer: rows [0,n) copied, rows [n,cap) set to a
// far-out-of-range sentinel so the voxelizer drops them (keeps a captured hipGraph valid for any point count).
__global__ __launch_bounds__(VOX_NT) void stage_points_kernel(const float4* __restrict__ src, int n, float4* __restrict__ dst,
                                                                int cap) {
  int i = blockIdx.x * VOX_NT + threadIdx.x;
  if (i >= cap) return;
  dst[i] = i < n ? src[i] : make_float4(-1.0e6f, -1.0e6f, -1.0e6f, 0.f);
}

struct VoxWs {
  int* lists;
  int* ent;
  int* blk_cnt;
  int* entry_of_vid;
  int* meta;
};

size_t vox_ws_layout(int hash_cap, int max_pts_total, int MP, int max_voxels, VoxWs* w, char* base, int batch = 1) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = sessd_align(off + bytes, 256);
    return o;
  };
  size_t o_lists = take((size_t)hash_cap * MP * 4);
  size_t o_ent = take((size_t)batch * max_pts_total * 4);
  size_t o_blk = take((size_t)batch * sessd_divup(max_pts_total, VOX_NT) * 4 + 4);
  size_t o_eov = take((size_t)batch * max_voxels * 4);
  size_t o_meta = take((size_t)batch * 16);
  if (w) {
    w->lists = (int*)(base + o_lists);
    w->ent = (int*)(base + o_ent);
    w->blk_cnt = (int*)(base + o_blk);
    w->entry_of_vid = (int*)(base + o_eov);
    w->meta = (int*)(base + o_meta);
  }
  return off;
}

}  // namespace

extern "C" {

// hash capacity (power of two, >= 2x the points it must hold)
uint32_t sessd_hash_capacity(int max_items) {
  uint32_t c = 1024;
  while (c < (uint32_t)max_items * 2u) c <<= 1;
  return c;
}

int sessd_hash_clear(uint32_t* keys, int* vals, uint32_t capacity, hipStream_t stream) {
  SESSD_FILL(keys, SESSD_HASH_EMPTY, capacity, stream);
  SESSD_FILL(vals, SESSD_HASH_EMPTY, capacity, stream);
  return SESSD_OK;
}

size_t sessd_voxelize_workspace_bytes(uint32_t hash_capacity, int max_points_in_frame, int max_points_per_voxel,
                                      int max_voxels) {
  return vox_ws_layout((int)hash_capacity, max_points_in_frame, max_points_per_voxel, max_voxels, nullptr, nullptr);
}

// One frame. `prefix` is a device int[batch+1]; prefix[batch_index] is the row at
// which this frame's voxels start (the caller zeroes prefix[0]); the kernel writes
// prefix[batch_index+1]. The hash (keys/vals) is shared by all frames of a batch and
// maps cell -> global voxel row afterwards (it is the level-0 site index of SpMiddleFHD).
int sessd_voxelize_frame(const float* points, int num_points, int ndim, const float* range6,
                         const float* voxel_size3, const int* grid3, int max_points_per_voxel, int max_voxels,
                         int batch_index, uint32_t* hash_keys, int* hash_vals, uint32_t hash_capacity,
                         float* voxels, int* coors, int coors_stride, int* num_points_per_voxel, float* mean_feat,
                         int* prefix, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (ndim < 3 || ndim > 8 || max_points_per_voxel < 1 || max_voxels < 1) return SESSD_EINVAL;
  if (coors_stride != 3 && coors_stride != 4) return SESSD_EINVAL;
  if ((hash_capacity & (hash_capacity - 1)) != 0) return SESSD_EINVAL;
  VoxGrid G;
  for (int j = 0; j < 3; ++j) {
    G.lo[j] = range6[j];
    G.vs[j] = voxel_size3[j];
    G.g[j] = grid3[j];
  }
  const uint64_t cells = (uint64_t)grid3[0] * grid3[1] * grid3[2];
  if (cells * (uint64_t)(batch_index + 1) >= (uint64_t)SESSD_HASH_EMPTY) return SESSD_EINVAL;
  const uint32_t key_base = (uint32_t)(cells * (uint64_t)batch_index);
  VoxWs w;
  size_t need = vox_ws_layout((int)hash_capacity, num_points, max_points_per_voxel, max_voxels, &w, (char*)workspace);
  if (need > workspace_bytes) return SESSD_EWORKSPACE;
  const int MP = max_points_per_voxel;
  SESSD_FILL_SCRATCH(w.lists, SESSD_HASH_EMPTY, (size_t)hash_capacity * MP, stream);
  SESSD_FILL_SCRATCH(w.meta, SESSD_HASH_EMPTY, 1, stream);  // cut = SESSD_SENT
  const int nblk = sessd_divup(num_points > 0 ? num_points : 1, VOX_NT);
  if (num_points > 0) {
    SESSD_LAUNCH(vox_insert_kernel, dim3(nblk), dim3(VOX_NT), 0, stream, points, num_points, ndim, G, key_base, (uint32_t)cells,
                       hash_keys, hash_capacity - 1, w.lists, MP, w.ent);
    SESSD_CHECK_LAUNCH();
  }
  SESSD_LAUNCH(vox_count_kernel, dim3(nblk), dim3(VOX_NT), 0, stream, num_points, w.ent, w.lists, MP, w.blk_cnt,
                     w.meta);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(vox_assign_kernel, dim3(nblk), dim3(VOX_NT), 0, stream, num_points, w.ent, w.lists, MP, w.blk_cnt,
                     nblk, hash_keys, key_base, G, max_voxels, batch_index, prefix + batch_index, hash_vals,
                     w.entry_of_vid, coors, coors_stride, w.meta, prefix + batch_index + 1, (uint32_t)cells, 0);
  SESSD_CHECK_LAUNCH();
  const int gblk = sessd_divup(max_voxels < num_points ? max_voxels : (num_points > 0 ? num_points : 1), VOX_NT);
  if (ndim == 4) {
    SESSD_LAUNCH(vox_gather_kernel<4>, dim3(gblk), dim3(VOX_NT), 0, stream, points, ndim, w.lists, MP,
                       w.entry_of_vid, w.meta, max_voxels, prefix + batch_index, voxels, num_points_per_voxel,
                       mean_feat, num_points);
  } else {
    SESSD_LAUNCH(vox_gather_kernel<0>, dim3(gblk), dim3(VOX_NT), 0, stream, points, ndim, w.lists, MP,
                       w.entry_of_vid, w.meta, max_voxels, prefix + batch_index, voxels, num_points_per_voxel,
                       mean_feat, num_points);
  }
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_voxelize_frames_workspace_bytes(uint32_t hash_capacity, int batch, int points_per_frame, int max_points_per_voxel,
                                             int max_voxels) {
  if (batch < 1 || points_per_frame < 1) return 0;
  return vox_ws_layout((int)hash_capacity, points_per_frame, max_points_per_voxel, max_voxels, nullptr, nullptr, batch);
}

// ALL frames of a batch in FOUR launches (sessd_voxelize_frame: four per frame, each frame's launches waiting for the previous
// frame's -- 32 dependent launches for the dense-scene batch of 8). points (batch, points_per_frame, ndim): every frame holds
// exactly points_per_frame rows (pad with out-of-range rows, sessd_stage_points). prefix (batch + 1): prefix[0] must be 0 on
// entry; prefix[1 ..] are written. Frame f's voxel rows start at the sum of the earlier frames' voxel counts, which every block
// of the assignment launch takes from the per-block counts itself. Same results as `batch` calls of sessd_voxelize_frame
// with batch_index 0 .. batch - 1, bit for bit.
int sessd_voxelize_frames(const float* points, int batch, int points_per_frame, int ndim, const float* range6,
                          const float* voxel_size3, const int* grid3, int max_points_per_voxel, int max_voxels,
                          uint32_t* hash_keys, int* hash_vals, uint32_t hash_capacity, float* voxels, int* coors, int coors_stride,
                          int* num_points_per_voxel, float* mean_feat, int* prefix, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  if (batch < 1 || batch > 65535 || points_per_frame < 1 || ndim < 3 || ndim > 8 || max_points_per_voxel < 1 || max_voxels < 1)
    return SESSD_EINVAL;
  if (coors_stride != 3 && coors_stride != 4) return SESSD_EINVAL;
  if ((hash_capacity & (hash_capacity - 1)) != 0) return SESSD_EINVAL;
  VoxGrid G;
  for (int j = 0; j < 3; ++j) {
    G.lo[j] = range6[j];
    G.vs[j] = voxel_size3[j];
    G.g[j] = grid3[j];
  }
  const uint64_t cells = (uint64_t)grid3[0] * grid3[1] * grid3[2];
  if (cells * (uint64_t)batch >= (uint64_t)SESSD_HASH_EMPTY) return SESSD_EINVAL;
  if ((uint64_t)hash_capacity < 2ull * (uint64_t)batch * (uint64_t)points_per_frame) return SESSD_EINVAL;
  VoxWs w;
  const size_t need = vox_ws_layout((int)hash_capacity, points_per_frame, max_points_per_voxel, max_voxels, &w, (char*)workspace, batch);
  if (need > workspace_bytes) return SESSD_EWORKSPACE;
  const int MP = max_points_per_voxel, P = points_per_frame;
  SESSD_FILL_SCRATCH(w.lists, SESSD_HASH_EMPTY, (size_t)hash_capacity * MP, stream);
  const int nblk = sessd_divup(P, VOX_NT);
  SESSD_LAUNCH(vox_insert_kernel, dim3(nblk, batch), dim3(VOX_NT), 0, stream, points, P, ndim, G, 0u, (uint32_t)cells, hash_keys,
               hash_capacity - 1, w.lists, MP, w.ent);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(vox_count_kernel, dim3(nblk, batch), dim3(VOX_NT), 0, stream, P, w.ent, w.lists, MP, w.blk_cnt, w.meta);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(vox_assign_kernel, dim3(nblk, batch), dim3(VOX_NT), 0, stream, P, w.ent, w.lists, MP, w.blk_cnt, nblk, hash_keys, 0u, G,
               max_voxels, 0, prefix, hash_vals, w.entry_of_vid, coors, coors_stride, w.meta, prefix + 1, (uint32_t)cells, 1);
  SESSD_CHECK_LAUNCH();
  const int gblk = sessd_divup(max_voxels < P ? max_voxels : P, VOX_NT);
  if (ndim == 4)
    SESSD_LAUNCH(vox_gather_kernel<4>, dim3(gblk, batch), dim3(VOX_NT), 0, stream, points, ndim, w.lists, MP, w.entry_of_vid, w.meta,
                 max_voxels, prefix, voxels, num_points_per_voxel, mean_feat, P);
  else
    SESSD_LAUNCH(vox_gather_kernel<0>, dim3(gblk, batch), dim3(VOX_NT), 0, stream, points, ndim, w.lists, MP, w.entry_of_vid, w.meta,
                 max_voxels, prefix, voxels, num_points_per_voxel, mean_feat, P);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// points (n,4) float32 -> dst (capacity,4): copy + out-of-range padding, one launch on `stream`.
int sessd_stage_points(const float* points, int num_points, float* dst, int capacity, hipStream_t stream) {
  if (num_points < 0 || capacity < num_points) return SESSD_EINVAL;
  if (capacity == 0) return SESSD_OK;
  SESSD_LAUNCH(stage_points_kernel, dim3(sessd_divup(capacity, VOX_NT)), dim3(VOX_NT), 0, stream,
                     (const float4*)points, num_points, (float4*)dst, capacity);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Stand-alone reader (VoxelFeatureExtractorV3.forward): voxels (M,MP,ndim), num_points (M) -> (M,nfeat)
int sessd_vfe_mean(const float* voxels, const int* num_points, const int* num_voxels_dev, int num_voxels_host,
                   int max_points_per_voxel, int ndim, int num_features, float* out, hipStream_t stream) {
  if (num_features > ndim) return SESSD_EINVAL;
  if (num_voxels_host <= 0) return SESSD_OK;
  SESSD_LAUNCH(vfe_mean_kernel, dim3(sessd_divup(num_voxels_host, 256)), dim3(256), 0, stream, voxels, num_points,
                     num_voxels_dev, num_voxels_host, max_points_per_voxel, ndim, num_features, out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
