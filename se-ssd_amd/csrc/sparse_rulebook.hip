// Active-site bookkeeping of the sparse 3-D convolutions of SpMiddleFHD on gfx950.
// Replaces what the reference gets from the third-party spconv package
//   det3d/models/backbones/scn.py:179-183 (SparseConvTensor + SubMConv3d / SparseConv3d
//   rulebook construction: spconv.ops.get_indice_pairs) -- semantics restated in oracle/sparse_conv.py.
//
// Everything is OUTPUT-STATIONARY: a rulebook is a table nbr[k][o] = input row feeding output
// site o through kernel offset k (or -1). That lets the convolution accumulate in registers and
// write every output row exactly once (no scatter-add, no atomics, deterministic).
//   hash_build        indices (N,4) -> open-addressing hash  cell -> row
//   downsample_sites  strided conv: the set of output sites reachable from the active inputs,
//                     numbered in first-touch order of the serial loop (input row asc, offset asc)
//                     -- hash insert + atomicMin of the creator id + block scan; no sort.
//   rulebook          nbr[k][o] by hash lookup of o*s - p + k, plus a per-16-site tile bitmask of
//                     the offsets that have any neighbour (built with wave ballots) so the
//                     convolution skips empty (tile, offset) pairs wave-uniformly.
// Site counts live on the device (n_dev); grids are sized by capacity and exit early.
#include "common.hpp"

namespace {

constexpr int NT = 256;

struct ConvGeom {
  int ks[3], st[3], pd[3];
  int in_dims[3];   // dims used by the INPUT hash keys / bounds
  int out_dims[3];  // output spatial shape
};

__device__ __forceinline__ uint32_t lin_key(int b, int z, int y, int x, const int* d) {
  return (uint32_t)(((b * d[0] + z) * d[1] + y) * d[2] + x);
}

__global__ __launch_bounds__(NT) void hash_build_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                         int n_cap, ConvGeom G, uint32_t* __restrict__ keys,
                                                         int* __restrict__ vals, uint32_t mask) {
  int i = blockIdx.x * NT + threadIdx.x;
  int n = n_dev ? min(n_dev[0], n_cap) : n_cap;
  if (i >= n) return;
  const int4 c = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  uint32_t slot = sessd_hash_insert(keys, mask, lin_key(c.x, c.y, c.z, c.w, G.in_dims));
  vals[slot] = i;
}

// ---- strided conv output sites ------------------------------------------------------------
// candidate id = i * KV + k. first[slot] = min candidate id that produced the cell.
template <bool COUNT_ONLY>
__device__ __forceinline__ bool cand_coord(const int* __restrict__ indices, int id, int KV, const ConvGeom& G, int& b,
                                           int* o) {
  const int i = id / KV, k = id - i * KV;
  const int4 c = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  const int kk[3] = {k / (G.ks[1] * G.ks[2]), (k / G.ks[2]) % G.ks[1], k % G.ks[2]};
  const int ci[3] = {c.y, c.z, c.w};
  b = c.x;
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int t = ci[d] + G.pd[d] - kk[d];
    int q = t / G.st[d];
    ok = ok && t >= 0 && (q * G.st[d] == t) && q < G.out_dims[d];
    o[d] = q;
  }
  return ok;
}

__global__ __launch_bounds__(NT) void down_insert_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                          int n_cap, int KV, ConvGeom G, uint32_t* __restrict__ keys,
                                                          uint32_t mask, int* __restrict__ first,
                                                          int* __restrict__ ent) {
  int id = blockIdx.x * NT + threadIdx.x;
  int n = min(n_dev[0], n_cap);
  if (id >= n_cap * KV) return;
  int e = -1;
  if (id < n * KV) {
    int b, o[3];
    if (cand_coord<false>(indices, id, KV, G, b, o)) {
      uint32_t slot = sessd_hash_insert(keys, mask, lin_key(b, o[0], o[1], o[2], G.out_dims));
      atomicMin(&first[slot], id);
      e = (int)slot;
    }
  }
  ent[id] = e;
}

__global__ __launch_bounds__(NT) void down_count_kernel(int total, const int* __restrict__ ent,
                                                         const int* __restrict__ first, int* __restrict__ blk_cnt) {
  __shared__ int sm[NT / 64];
  int id = blockIdx.x * NT + threadIdx.x;
  int f = 0;
  if (id < total) {
    int e = ent[id];
    f = (e >= 0 && first[e] == id) ? 1 : 0;
  }
  int s = sessd_wave_sum(f);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < NT / 64; ++w) t += sm[w];
    blk_cnt[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(NT) void down_assign_kernel(const int* __restrict__ indices, int total, int KV, ConvGeom G,
                                                          const int* __restrict__ ent, const int* __restrict__ first,
                                                          const int* __restrict__ blk_cnt, int nblk,
                                                          int* __restrict__ vals, int* __restrict__ out_indices,
                                                          int n_out_cap, int* __restrict__ n_out_dev,
                                                          int* __restrict__ err_flag) {
  __shared__ int sm[NT / 64];
  __shared__ int s_base;
  int part = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += NT) part += blk_cnt[b];
  part = sessd_wave_sum(part);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < NT / 64; ++w) t += sm[w];
    s_base = t;
  }
  __syncthreads();
  const int base = s_base;
  int id = blockIdx.x * NT + threadIdx.x;
  int e = -1, f = 0;
  if (id < total) {
    e = ent[id];
    f = (e >= 0 && first[e] == id) ? 1 : 0;
  }
  int tot;
  int row = base + sessd_block_exscan<NT>(f, sm, &tot);
  if (f) {
    if (row < n_out_cap) {
      int b, o[3];
      cand_coord<false>(indices, id, KV, G, b, o);
      vals[e] = row;
      *reinterpret_cast<int4*>(out_indices + (size_t)row * 4) = make_int4(b, o[0], o[1], o[2]);
    } else {
      vals[e] = SESSD_SENT;  // reads as absent
      atomicOr(err_flag, 1);  // capacity overflow: reported, never silent
    }
  }
  if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
    int m = base + tot;
    n_out_dev[0] = m < n_out_cap ? m : n_out_cap;
  }
}

// ---- gather rulebook ------------------------------------------------------------------------
// One wave handles 16 consecutive output sites x 4 kernel offsets per pass: lane = (site&15) + 16*(k&3).
// nbr is [KV][n_cap] (offset-major: the convolution reads 16 consecutive sites of one offset).
__global__ __launch_bounds__(NT) void rulebook_kernel(const int* __restrict__ out_indices, const int* __restrict__ n_dev,
                                                       int n_cap, int KV, ConvGeom G, const uint32_t* __restrict__ keys,
                                                       const int* __restrict__ vals, uint32_t mask,
                                                       int* __restrict__ nbr, uint32_t* __restrict__ tile_mask) {
  const int n = min(n_dev[0], n_cap);
  const int wave = (blockIdx.x * NT + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int tile = wave;
  if (tile * 16 >= n_cap) return;
  const int o = tile * 16 + (lane & 15);
  if (tile * 16 >= n) {  // beyond the live sites: keep the table defined
    if (lane == 0) tile_mask[tile] = 0u;
    return;
  }
  int4 c = make_int4(0, 0, 0, 0);
  const bool live = o < n;
  if (live) c = *reinterpret_cast<const int4*>(out_indices + (size_t)o * 4);
  uint32_t tm = 0;
  // fully unrolled (KV <= 28): the 7 hash probes of a lane are independent, so they are all in flight at once
  // instead of 7 dependent global-memory round trips
#pragma unroll
  for (int k0 = 0; k0 < 28; k0 += 4) {
    if (k0 >= KV) break;
    const int k = k0 + (lane >> 4);
    int found = -1;
    if (live && k < KV) {
      const int kz = k / (G.ks[1] * G.ks[2]), ky = (k / G.ks[2]) % G.ks[1], kx = k % G.ks[2];
      const int z = c.y * G.st[0] - G.pd[0] + kz, y = c.z * G.st[1] - G.pd[1] + ky, x = c.w * G.st[2] - G.pd[2] + kx;
      if (z >= 0 && z < G.in_dims[0] && y >= 0 && y < G.in_dims[1] && x >= 0 && x < G.in_dims[2])
        found = sessd_hash_find(keys, vals, mask, lin_key(c.x, z, y, x, G.in_dims));
    }
    if (k < KV && (lane & 15) + tile * 16 < n_cap) nbr[(size_t)k * n_cap + o] = found;
    unsigned long long bal = __ballot(found >= 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((bal >> (16 * q)) & 0xFFFFull) tm |= 1u << (k0 + q);
  }
  if (lane == 0) tile_mask[tile] = tm;
}

void fill_geom(ConvGeom& G, const int* ks, const int* st, const int* pd, const int* in_dims, const int* out_dims) {
  for (int d = 0; d < 3; ++d) {
    G.ks[d] = ks ? ks[d] : 1;
    G.st[d] = st ? st[d] : 1;
    G.pd[d] = pd ? pd[d] : 0;
    G.in_dims[d] = in_dims ? in_dims[d] : 0;
    G.out_dims[d] = out_dims ? out_dims[d] : 0;
  }
}

struct DownWs {
  int* first;
  int* ent;
  int* blk_cnt;
};

size_t down_ws_layout(int n_in_cap, int kv, uint32_t out_hash_cap, DownWs* w, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = sessd_align(off + bytes, 256);
    return o;
  };
  size_t o_first = take((size_t)out_hash_cap * 4);
  size_t o_ent = take((size_t)n_in_cap * kv * 4);
  size_t o_blk = take((size_t)sessd_divup(n_in_cap * kv, NT) * 4 + 4);
  if (w) {
    w->first = (int*)(base + o_first);
    w->ent = (int*)(base + o_ent);
    w->blk_cnt = (int*)(base + o_blk);
  }
  return off;
}

}  // namespace

extern "C" {

// cell -> row hash of a site list. dims3 = (D,H,W) used for the linear key; capacity a power of two
// >= 2 * n_cap (sessd_hash_capacity). The caller clears the hash first (sessd_hash_clear).
int sessd_sparse_hash_build(const int* indices, const int* n_dev, int n_cap, const int* dims3, uint32_t* keys, int* vals,
                            uint32_t capacity, hipStream_t stream) {
  if (n_cap <= 0 || (capacity & (capacity - 1)) != 0) return SESSD_EINVAL;
  ConvGeom G;
  fill_geom(G, nullptr, nullptr, nullptr, dims3, nullptr);
  hipLaunchKernelGGL(hash_build_kernel, dim3(sessd_divup(n_cap, NT)), dim3(NT), 0, stream, indices, n_dev, n_cap, G, keys,
                     vals, capacity - 1);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_sparse_downsample_workspace_bytes(int n_in_cap, int kernel_volume, uint32_t out_hash_capacity) {
  return down_ws_layout(n_in_cap, kernel_volume, out_hash_capacity, nullptr, nullptr);
}

// Output sites of SparseConv3d(ksize, stride, padding): out_indices (n_out_cap,4) [b,z,y,x], *n_out_dev,
// and the output level's hash (out_keys/out_vals, cleared by this call). err_flag |= 1 on overflow.
int sessd_sparse_downsample_sites(const int* in_indices, const int* n_in_dev, int n_in_cap, const int* ksize3,
                                  const int* stride3, const int* pad3, const int* out_dims3, uint32_t* out_keys,
                                  int* out_vals, uint32_t out_capacity, int* out_indices, int n_out_cap, int* n_out_dev,
                                  int* err_flag, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n_in_cap <= 0 || n_out_cap <= 0 || (out_capacity & (out_capacity - 1)) != 0) return SESSD_EINVAL;
  const int kv = ksize3[0] * ksize3[1] * ksize3[2];
  if ((long long)n_in_cap * kv >= 0x7F000000ll) return SESSD_EINVAL;
  DownWs w;
  if (down_ws_layout(n_in_cap, kv, out_capacity, &w, (char*)workspace) > workspace_bytes) return SESSD_EWORKSPACE;
  ConvGeom G;
  fill_geom(G, ksize3, stride3, pad3, nullptr, out_dims3);
  SESSD_FILL_SCRATCH(out_keys, SESSD_HASH_EMPTY, out_capacity, stream);
  SESSD_FILL_SCRATCH(out_vals, SESSD_HASH_EMPTY, out_capacity, stream);
  SESSD_FILL_SCRATCH(w.first, SESSD_HASH_EMPTY, out_capacity, stream);
  const int total = n_in_cap * kv;
  const int nblk = sessd_divup(total, NT);
  hipLaunchKernelGGL(down_insert_kernel, dim3(nblk), dim3(NT), 0, stream, in_indices, n_in_dev, n_in_cap, kv, G, out_keys,
                     out_capacity - 1, w.first, w.ent);
  SESSD_CHECK_LAUNCH();
  hipLaunchKernelGGL(down_count_kernel, dim3(nblk), dim3(NT), 0, stream, total, w.ent, w.first, w.blk_cnt);
  SESSD_CHECK_LAUNCH();
  hipLaunchKernelGGL(down_assign_kernel, dim3(nblk), dim3(NT), 0, stream, in_indices, total, kv, G, w.ent, w.first,
                     w.blk_cnt, nblk, out_vals, out_indices, n_out_cap, n_out_dev, err_flag);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// nbr[kv][n_out_cap] and tile_mask[ceil(n_out_cap/16)] for the conv (ksize,stride,pad) whose INPUT level is
// hashed in (in_keys,in_vals) with key dims in_dims3. Submanifold conv: stride 1, pad = ksize/2, out == in.
int sessd_sparse_rulebook(const int* out_indices, const int* n_out_dev, int n_out_cap, const int* ksize3,
                          const int* stride3, const int* pad3, const uint32_t* in_keys, const int* in_vals,
                          uint32_t in_capacity, const int* in_dims3, int* nbr, uint32_t* tile_mask, hipStream_t stream) {
  if (n_out_cap <= 0 || (in_capacity & (in_capacity - 1)) != 0) return SESSD_EINVAL;
  const int kv = ksize3[0] * ksize3[1] * ksize3[2];
  if (kv > 28) return SESSD_EINVAL;
  ConvGeom G;
  fill_geom(G, ksize3, stride3, pad3, in_dims3, nullptr);
  const int tiles = sessd_divup(n_out_cap, 16);
  hipLaunchKernelGGL(rulebook_kernel, dim3(sessd_divup(tiles, NT / 64)), dim3(NT), 0, stream, out_indices, n_out_dev,
                     n_out_cap, kv, G, in_keys, in_vals, in_capacity - 1, nbr, tile_mask);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
