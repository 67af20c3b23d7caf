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
  int cd[3];        // candidate outputs per dim of one input site = ceil(ks/st)
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
  // a key at or beyond the empty marker 0x7F7F7F7F (batch * D * H * W too large for 31-bit keys) would alias "empty" or wrap:
  // such a site is left out -- its lookups then miss, as for any absent site -- rather than corrupting the table
  const long long k64 = (((long long)c.x * G.in_dims[0] + c.y) * G.in_dims[1] + c.z) * G.in_dims[2] + c.w;
  if (k64 < 0 || k64 >= 0x7F7F7F7Fll) return;
  uint32_t slot = sessd_hash_insert(keys, mask, lin_key(c.x, c.y, c.z, c.w, G.in_dims));
  if (slot != SESSD_HASH_FULL) vals[slot] = i;  // capacity >= 2 * n_cap: cannot fill up
}

// ---- strided conv output sites ------------------------------------------------------------
// An input site i reaches, per dimension, only the outputs (i + p - k)/s with k = (i+p) mod s, that + s, ...
// (<= ceil(ks/st) of them: 2 for k=3/s=2), so a site has KC = cd0*cd1*cd2 candidates (8 instead of 27 kernel
// offsets). candidate id = i * KC + c, c in ascending-k order; first[slot] = min candidate id that produced the cell,
// which numbers the output sites in first-touch order of the serial loop (input row asc, offset asc).
template <bool COUNT_ONLY>
__device__ __forceinline__ bool cand_coord(const int* __restrict__ indices, int id, int KC, const ConvGeom& G, int& b,
                                           int* o) {
  const int i = id / KC, c = id - i * KC;
  const int4 ci4 = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  const int cc[3] = {c / (G.cd[1] * G.cd[2]), (c / G.cd[2]) % G.cd[1], c % G.cd[2]};
  const int ci[3] = {ci4.y, ci4.z, ci4.w};
  b = ci4.x;
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int ip = ci[d] + G.pd[d];
    const int k = ip % G.st[d] + cc[d] * G.st[d];
    const int t = ip - k;
    const int q = t / G.st[d];
    ok = ok && k < G.ks[d] && t >= 0 && q < G.out_dims[d];
    o[d] = q;
  }
  return ok;
}

__global__ __launch_bounds__(NT) void down_insert_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                          int n_cap, int KV, ConvGeom G, uint32_t* __restrict__ keys,
                                                          uint32_t mask, int* __restrict__ first,
                                                          int* __restrict__ ent, int* __restrict__ err_flag) {
  int id = blockIdx.x * NT + threadIdx.x;
  int n = min(n_dev[0], n_cap);
  if (id >= n_cap * KV) return;
  int e = -1;
  if (id < n * KV) {
    int b, o[3];
    if (cand_coord<false>(indices, id, KV, G, b, o)) {
      uint32_t slot = sessd_hash_insert(keys, mask, lin_key(b, o[0], o[1], o[2], G.out_dims));
      if (slot != SESSD_HASH_FULL) {
        atomicMin(&first[slot], id);
        e = (int)slot;
      } else {
        atomicOr(err_flag, 1);  // more output cells than hash slots: reported like a capacity overflow
      }
    }
  }
  ent[id] = e;
}

// Single-kernel variant: the thread whose CAS creates a cell takes the next output row with an atomic counter.
// Row numbering then depends on scheduling, but nothing downstream does: every output row is computed from its own
// neighbour list in a fixed (offset, cin) order and the last layer scatters by coordinate, so the dense BEV tensor
// and the detections are bit-identical for any numbering. Used by the inference engine (3 launches + 2 scratch
// arrays fewer per level); the ordered 3-kernel version stays the default of the spconv-compatible API.
__global__ __launch_bounds__(NT) void down_insert_unordered_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                                    int n_cap, int KC, ConvGeom G, uint32_t* __restrict__ keys,
                                                                    int* __restrict__ vals, uint32_t mask,
                                                                    int* __restrict__ out_indices, int n_out_cap,
                                                                    int* __restrict__ n_out_dev, int* __restrict__ err_flag) {
  const int id = blockIdx.x * NT + threadIdx.x;
  const int n = min(n_dev[0], n_cap);
  int b = 0, o[3] = {0, 0, 0};
  bool active = id < n * KC;
  if (active) active = cand_coord<false>(indices, id, KC, G, b, o);
  bool created = false;
  uint32_t slot = 0;
  if (active) {
    const uint32_t key = lin_key(b, o[0], o[1], o[2], G.out_dims);
    slot = sessd_hash_home(key, mask);
    uint32_t probes = 0, lap = slot;
    for (; probes <= mask; ++probes) {
      const uint32_t prev = atomicCAS(&keys[slot], SESSD_HASH_EMPTY, key);
      if (prev == SESSD_HASH_EMPTY) {  // this thread created the cell
        created = true;
        break;
      }
      if (prev == key) break;
      SESSD_HASH_ADVANCE(slot, lap, mask)
    }
    if (probes > mask) atomicOr(err_flag, 1);  // table full (a cloud far sparser than the growth factors assume)
  }
  // Row numbers for the cells this WAVE created with ONE atomic on the shared counter (a per-thread atomicAdd on one
  // address serialises: 285 k creators of the dense-scene level cost 0.5 ms).
  const unsigned long long made = __ballot(created);
  if (made == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)made) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(n_out_dev, __popcll(made));
  base = __shfl(base, leader, 64);
  if (created) {
    const int row = base + __popcll(made & ((1ull << lane) - 1ull));
    if (row < n_out_cap) {
      vals[slot] = row;
      *reinterpret_cast<int4*>(out_indices + (size_t)row * 4) = make_int4(b, o[0], o[1], o[2]);
    } else {
      atomicOr(err_flag, 1);  // vals stays SESSD_SENT: reads as absent
    }
  }
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
struct RbJob {
  ConvGeom G;
  const uint32_t* keys;
  const int* vals;
  uint32_t mask;
  int KV;
  int* nbr;
  uint32_t* tile_mask;
};
struct RbJobs {
  RbJob j[2];
};

// blockIdx.y selects the job: up to two rulebooks over the SAME output sites (e.g. the strided conv into a level and
// the submanifold convs on that level) are built by one launch.
__global__ __launch_bounds__(NT) void rulebook_kernel(const int* __restrict__ out_indices, const int* __restrict__ n_dev,
                                                       int n_cap, RbJobs J) {
  const RbJob& Jb = J.j[blockIdx.y];
  const ConvGeom& G = Jb.G;
  const int KV = Jb.KV;
  const uint32_t* __restrict__ keys = Jb.keys;
  const int* __restrict__ vals = Jb.vals;
  const uint32_t mask = Jb.mask;
  int* __restrict__ nbr = Jb.nbr;
  uint32_t* __restrict__ tile_mask = Jb.tile_mask;
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
  // Two phases of INDEPENDENT loads instead of 7 serial probe chains: (1) the home slot's key of all (<= 7) offsets
  // of this lane, (2) the value of every hit. Only a collision (load factor <= 0.5: about one probe in four) falls
  // back to the sequential probe loop.
  constexpr int NP = 7;
  uint32_t key[NP], slot[NP], k0v[NP];
  bool want[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int k = p * 4 + (lane >> 4);
    want[p] = false;
    key[p] = 0; slot[p] = 0;
    if (live && k < KV) {
      const int kz = k / (G.ks[1] * G.ks[2]), ky = (k / G.ks[2]) % G.ks[1], kx = k % G.ks[2];
      const int z = c.y * G.st[0] - G.pd[0] + kz, y = c.z * G.st[1] - G.pd[1] + ky, x = c.w * G.st[2] - G.pd[2] + kx;
      if (z >= 0 && z < G.in_dims[0] && y >= 0 && y < G.in_dims[1] && x >= 0 && x < G.in_dims[2]) {
        want[p] = true;
        key[p] = lin_key(c.x, z, y, x, G.in_dims);
        slot[p] = sessd_hash_home(key[p], mask);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) k0v[p] = want[p] ? keys[slot[p]] : SESSD_HASH_EMPTY;
  int found[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) found[p] = (want[p] && k0v[p] == key[p]) ? vals[slot[p]] : -1;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (want[p] && k0v[p] == key[p]) {
      if (found[p] == SESSD_SENT) found[p] = -1;
    } else if (want[p] && k0v[p] != SESSD_HASH_EMPTY) {
      found[p] = sessd_hash_find(keys, vals, mask, key[p]);  // collision: rare sequential path
    }
    const int k = p * 4 + (lane >> 4);
    if (k < KV && o < n_cap) nbr[(size_t)k * n_cap + o] = found[p];
    const unsigned long long bal = __ballot(found[p] >= 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if ((bal >> (16 * q)) & 0xFFFFull) tm |= 1u << (p * 4 + q);
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
    G.cd[d] = (G.ks[d] + G.st[d] - 1) / G.st[d];
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
  if (n_cap <= 0 || (capacity & (capacity - 1)) != 0 || !dims3) return SESSD_EINVAL;
  // keys are ((b D + z) H + y) W + x in 31 bits below the empty marker: one batch element must fit (the batch index is device
  // data; sites of batch elements beyond the key range are skipped by the kernel, see there)
  if (dims3[0] <= 0 || dims3[1] <= 0 || dims3[2] <= 0 || (long long)dims3[0] * dims3[1] * dims3[2] >= 0x7F7F7F7Fll) return SESSD_EINVAL;
  ConvGeom G;
  fill_geom(G, nullptr, nullptr, nullptr, dims3, nullptr);
  SESSD_LAUNCH(hash_build_kernel, dim3(sessd_divup(n_cap, NT)), dim3(NT), 0, stream, indices, n_dev, n_cap, G, keys,
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
  ConvGeom G;
  fill_geom(G, ksize3, stride3, pad3, nullptr, out_dims3);
  const int kv = G.cd[0] * G.cd[1] * G.cd[2];  // candidates per input site (<= kernel volume)
  if ((long long)n_in_cap * kv >= 0x7F000000ll) return SESSD_EINVAL;
  DownWs w;
  if (down_ws_layout(n_in_cap, kv, out_capacity, &w, (char*)workspace) > workspace_bytes) return SESSD_EWORKSPACE;
  SESSD_FILL_SCRATCH(out_keys, SESSD_HASH_EMPTY, out_capacity, stream);
  SESSD_FILL_SCRATCH(out_vals, SESSD_HASH_EMPTY, out_capacity, stream);
  SESSD_FILL_SCRATCH(w.first, SESSD_HASH_EMPTY, out_capacity, stream);
  const int total = n_in_cap * kv;
  const int nblk = sessd_divup(total, NT);
  SESSD_LAUNCH(down_insert_kernel, dim3(nblk), dim3(NT), 0, stream, in_indices, n_in_dev, n_in_cap, kv, G, out_keys,
                     out_capacity - 1, w.first, w.ent, err_flag);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(down_count_kernel, dim3(nblk), dim3(NT), 0, stream, total, w.ent, w.first, w.blk_cnt);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(down_assign_kernel, dim3(nblk), dim3(NT), 0, stream, in_indices, total, kv, G, w.ent, w.first,
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
  RbJobs J;
  fill_geom(J.j[0].G, ksize3, stride3, pad3, in_dims3, nullptr);
  J.j[0].keys = in_keys; J.j[0].vals = in_vals; J.j[0].mask = in_capacity - 1; J.j[0].KV = kv;
  J.j[0].nbr = nbr; J.j[0].tile_mask = tile_mask;
  J.j[1] = J.j[0];
  const int tiles = sessd_divup(n_out_cap, 16);
  SESSD_LAUNCH(rulebook_kernel, dim3(sessd_divup(tiles, NT / 64), 1), dim3(NT), 0, stream, out_indices, n_out_dev,
                     n_out_cap, J);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Two rulebooks over the same output sites in one launch (A: e.g. the strided conv INTO the level, looked up in the
// previous level's hash; B: the submanifold convs ON the level, looked up in its own hash).
int sessd_sparse_rulebook_pair(const int* out_indices, const int* n_out_dev, int n_out_cap, const int* ksize3_a,
                               const int* stride3_a, const int* pad3_a, const uint32_t* keys_a, const int* vals_a,
                               uint32_t capacity_a, const int* dims3_a, int* nbr_a, uint32_t* tile_mask_a,
                               const int* ksize3_b, const int* stride3_b, const int* pad3_b, const uint32_t* keys_b,
                               const int* vals_b, uint32_t capacity_b, const int* dims3_b, int* nbr_b,
                               uint32_t* tile_mask_b, hipStream_t stream) {
  if (n_out_cap <= 0 || (capacity_a & (capacity_a - 1)) != 0 || (capacity_b & (capacity_b - 1)) != 0) return SESSD_EINVAL;
  RbJobs J;
  fill_geom(J.j[0].G, ksize3_a, stride3_a, pad3_a, dims3_a, nullptr);
  J.j[0].keys = keys_a; J.j[0].vals = vals_a; J.j[0].mask = capacity_a - 1;
  J.j[0].KV = ksize3_a[0] * ksize3_a[1] * ksize3_a[2]; J.j[0].nbr = nbr_a; J.j[0].tile_mask = tile_mask_a;
  fill_geom(J.j[1].G, ksize3_b, stride3_b, pad3_b, dims3_b, nullptr);
  J.j[1].keys = keys_b; J.j[1].vals = vals_b; J.j[1].mask = capacity_b - 1;
  J.j[1].KV = ksize3_b[0] * ksize3_b[1] * ksize3_b[2]; J.j[1].nbr = nbr_b; J.j[1].tile_mask = tile_mask_b;
  if (J.j[0].KV > 28 || J.j[1].KV > 28) return SESSD_EINVAL;
  const int tiles = sessd_divup(n_out_cap, 16);
  SESSD_LAUNCH(rulebook_kernel, dim3(sessd_divup(tiles, NT / 64), 2), dim3(NT), 0, stream, out_indices, n_out_dev,
                     n_out_cap, J);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// Output sites of a strided sparse conv in ONE launch, rows numbered by an atomic counter (order not reproducible,
// results downstream are -- see down_insert_unordered_kernel). The caller has cleared out_keys / out_vals to
// 0x7F7F7F7F and *n_out_dev to 0 (e.g. with its per-frame arena fill); no workspace.
int sessd_sparse_downsample_sites_unordered(const int* in_indices, const int* n_in_dev, int n_in_cap, const int* ksize3,
                                            const int* stride3, const int* pad3, const int* out_dims3, uint32_t* out_keys,
                                            int* out_vals, uint32_t out_capacity, int* out_indices, int n_out_cap,
                                            int* n_out_dev, int* err_flag, hipStream_t stream) {
  if (n_in_cap <= 0 || n_out_cap <= 0 || (out_capacity & (out_capacity - 1)) != 0) return SESSD_EINVAL;
  ConvGeom G;
  fill_geom(G, ksize3, stride3, pad3, nullptr, out_dims3);
  const int kc = G.cd[0] * G.cd[1] * G.cd[2];
  SESSD_LAUNCH(down_insert_unordered_kernel, dim3(sessd_divup(n_in_cap * kc, NT)), dim3(NT), 0, stream, in_indices,
                     n_in_dev, n_in_cap, kc, G, out_keys, out_vals, out_capacity - 1, out_indices, n_out_cap, n_out_dev,
                     err_flag);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
