// Renumbering of a sparse level's sites in grid-row order.
//
// The voxelizer numbers voxels in the order the lidar produced their first point (the reference's order, which its API
// must keep). The sparse convolutions work on tiles of 16 consecutively numbered sites; with scan-order numbering only 44 %
// of the MFMA rows a tile executes carry a rulebook pair and a tile gathers ~75 distinct input rows. Numbering the sites by
// (batch, z, y) grid row instead -- x arbitrary inside a row, rows hold a handful of voxels -- puts spatial neighbours in the
// same tile: 58 % useful rows and ~45 gathered rows (scripts/tile_occupancy_probe.py), and the strided levels below inherit
// it through their first-touch numbering. Convolution results do not depend on the numbering (every output row is the sum
// over the kernel offsets in fixed order; the dense BEV map is addressed by coordinates), so this is an engine-internal
// permutation: three passes over <= ~20 k sites plus a single-workgroup scan of the row counters, no sort.
//
//   count    cnt[(b*D + z)*H + y] += 1
//   scan     exclusive prefix over the B*D*H row counters (one workgroup; 64 k counters per frame)
//   scatter  r = atomicAdd(&cnt[row], 1): out_indices[r], out_feat[r] = the site; hash_vals[slot(site)] = r
//
// STATUS: written in round 1 after the GPU budget was spent -- compiled, not yet run on hardware. The engine keeps it off by
// default (InferenceEngine(sort_sites=False)); tests/test_site_renumber_gpu.py runs only with SESSD_EXPERIMENTAL=1.
#include "common.hpp"

namespace {

constexpr int NT = 256;
constexpr int SCAN_NT = 1024;

__device__ __forceinline__ uint32_t lin_key4(const int4& c, int D, int H, int W) {
  return (uint32_t)(((c.x * D + c.y) * H + c.z) * W + c.w);
}

// slot of an existing key (the key IS in the table: every site was inserted when the level was built); ~0u if absent
__device__ __forceinline__ uint32_t slot_of(const uint32_t* __restrict__ keys, uint32_t mask, uint32_t key) {
  uint32_t slot = sessd_hash_home(key, mask), lap = slot;
  for (uint32_t probes = 0; probes <= mask; ++probes) {
    const uint32_t k = keys[slot];
    if (k == key) return slot;
    if (k == SESSD_HASH_EMPTY) return 0xFFFFFFFFu;
    SESSD_HASH_ADVANCE(slot, lap, mask)
  }
  return 0xFFFFFFFFu;
}

__global__ __launch_bounds__(NT) void row_count_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                        int n_cap, int D, int H, int* __restrict__ cnt) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= min(n_dev[0], n_cap)) return;
  const int4 c = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  atomicAdd(&cnt[(c.x * D + c.y) * H + c.z], 1);
}

// counters -> exclusive prefix, in place; one workgroup, thread t owns the contiguous chunk [t*per, (t+1)*per)
__global__ __launch_bounds__(SCAN_NT) void row_scan_kernel(int* __restrict__ cnt, int nb) {
  __shared__ int smem[SCAN_NT / 64];
  const int per = sessd_divup(nb, SCAN_NT);
  const int j0 = threadIdx.x * per, j1 = min(nb, j0 + per);
  int s = 0;
  for (int j = j0; j < j1; ++j) s += cnt[j];
  int total;
  int base = sessd_block_exscan<SCAN_NT>(s, smem, &total);
  for (int j = j0; j < j1; ++j) {
    const int v = cnt[j];
    cnt[j] = base;
    base += v;
  }
}

__global__ __launch_bounds__(NT) void row_scatter_kernel(const int* __restrict__ indices, const int* __restrict__ n_dev,
                                                          int n_cap, int D, int H, int W, const float* __restrict__ feat,
                                                          int channels, int* __restrict__ next, const uint32_t* __restrict__ keys,
                                                          int* __restrict__ vals, uint32_t mask, int* __restrict__ out_indices,
                                                          float* __restrict__ out_feat) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i >= min(n_dev[0], n_cap)) return;
  const int4 c = *reinterpret_cast<const int4*>(indices + (size_t)i * 4);
  const int r = atomicAdd(&next[(c.x * D + c.y) * H + c.z], 1);
  *reinterpret_cast<int4*>(out_indices + (size_t)r * 4) = c;
  for (int ch = 0; ch < channels; ++ch) out_feat[(size_t)r * channels + ch] = feat[(size_t)i * channels + ch];
  const uint32_t slot = slot_of(keys, mask, lin_key4(c, D, H, W));
  if (slot != 0xFFFFFFFFu) vals[slot] = r;
}

}  // namespace

extern "C" {

size_t sessd_sparse_renumber_workspace_bytes(int batch, const int* dims3) {
  return sessd_align((size_t)batch * dims3[0] * dims3[1] * sizeof(int), 256);
}

// indices (n_cap,4) [b,z,y,x] and feat (n_cap, channels) of a level whose cell -> row hash is (hash_keys, hash_vals) with
// keys ((b*D + z)*H + y)*W + x, dims3 = (D,H,W): writes the same sites to out_indices / out_feat numbered by (b, z, y) grid
// row (order inside a row unspecified) and points the hash at the new rows. *n_dev sites; rows >= *n_dev are not written.
// in and out buffers must not overlap. One stream, no host synchronisation, no allocation.
int sessd_sparse_renumber_sites(const int* indices, const int* n_dev, int n_cap, int batch, const int* dims3,
                                const float* feat, int channels, const uint32_t* hash_keys, int* hash_vals,
                                uint32_t hash_capacity, int* out_indices, float* out_feat, void* workspace,
                                size_t workspace_bytes, hipStream_t stream) {
  if (n_cap <= 0 || batch <= 0 || channels <= 0 || (hash_capacity & (hash_capacity - 1)) != 0) return SESSD_EINVAL;
  if (indices == out_indices || feat == out_feat) return SESSD_EINVAL;
  const long long rows = (long long)batch * dims3[0] * dims3[1];
  if (rows <= 0 || rows > (1ll << 26)) return SESSD_EINVAL;
  if (workspace_bytes < sessd_sparse_renumber_workspace_bytes(batch, dims3)) return SESSD_EWORKSPACE;
  int* cnt = (int*)workspace;
  SESSD_FILL(cnt, 0u, (size_t)rows, stream);
  const int nblk = sessd_divup(n_cap, NT);
  SESSD_LAUNCH(row_count_kernel, dim3(nblk), dim3(NT), 0, stream, indices, n_dev, n_cap, dims3[0], dims3[1], cnt);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(row_scan_kernel, dim3(1), dim3(SCAN_NT), 0, stream, cnt, (int)rows);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(row_scatter_kernel, dim3(nblk), dim3(NT), 0, stream, indices, n_dev, n_cap, dims3[0], dims3[1], dims3[2],
               feat, channels, cnt, hash_keys, hash_vals, hash_capacity - 1, out_indices, out_feat);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
