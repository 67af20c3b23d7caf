// Point-level work of the training data path on the device.
//
// points_in_bodies: membership of P points in M convex bodies given by F inward-facing planes each -- the primitive under
// GT-AUG point removal (box_np_ops.points_in_rbbox), the per-object noise masks and the pyramid masks of the shape-aware
// augmentation (det3d/core/bbox/geometry.py:215-276 points_in_convex_polygon_3d_jit; 6 planes per box, 5 per pyramid).
// The planes (n, d) come from the host (<= ~100 bodies; computed with the reference's float32 arithmetic,
// geometry.py:352-377), the test per point and plane is the reference's   x*nx + y*ny + z*nz + d >= 0  => outside
// evaluated left to right in float32 without contraction, so the masks are bit-identical to the numba loop.
// HBM-bound elementwise work: one thread per point, the planes in LDS, a bit per (point, body) out.
//
// points_rigid_moves: the per-object noise applied to the points (det3d/core/sampler/preprocess.py:544-560 points_transform_):
// every point takes the motion of the FIRST valid box that contains it -- rotation about that box's centre, then the box's
// translation -- membership test fused (same planes, same comparison), float32 steps in the reference's order.
// points_global_transform: flip about the x axis, one yaw rotation and one scale of the whole cloud (preprocess.py:896-945
// random_flip_v2 / global_rotation_v3 / global_scaling_v3) in one pass; optionally snapshots the untransformed cloud first
// (`points_raw`, the teacher's view, pipelines/preprocess.py:130-134).
// points_compact: order-preserving stream compaction by a keep flag (GT-AUG removal of covered points preprocess.py:102-105,
// shape-aware dropout sa_da_v2.py), count left on the device: what feeds the voxelizer without a host round trip.
// The BOX-level decisions (which noise draw survives the collision tests, the flip / angle / scale draws, the database sample)
// stay on the host: a few dozen boxes, and they define the random-number order that makes runs reproducible.
#include "common.hpp"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void points_in_bodies_kernel(const float* __restrict__ points, int num_points, int stride,
                                                               const float* __restrict__ planes, int num_bodies, int faces,
                                                               uint32_t* __restrict__ out_mask, int words) {
  extern __shared__ float lds_planes[];  // [num_bodies][faces][4]
  const int total = num_bodies * faces * 4;
  for (int t = threadIdx.x; t < total; t += NT) lds_planes[t] = planes[t];
  __syncthreads();
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= num_points) return;
  const float x = points[(size_t)p * stride + 0], y = points[(size_t)p * stride + 1], z = points[(size_t)p * stride + 2];
  for (int w = 0; w < words; ++w) {
    uint32_t bits = 0;
    const int m1 = min(num_bodies, (w + 1) * 32);
    for (int m = w * 32; m < m1; ++m) {
      const float* pl = lds_planes + (size_t)m * faces * 4;
      bool inside = true;
      for (int f = 0; f < faces; ++f) {
        const float sign = x * pl[f * 4 + 0] + y * pl[f * 4 + 1] + z * pl[f * 4 + 2] + pl[f * 4 + 3];
        if (sign >= 0.f) {
          inside = false;
          break;
        }
      }
      if (inside) bits |= 1u << (m & 31);
    }
    out_mask[(size_t)p * words + w] = bits;
  }
}

struct MoveArgs {
  const float* planes;   // (num_boxes, 6, 4) inward normals
  const double* centers; // (num_boxes, 3) float64: `points[i, :3] -= centers[j, :3]` is computed in float64 and rounded to float32
  const double* loc;     // (num_boxes, 3) translation, float64 likewise
  const float* sincos;   // (num_boxes, 2) sin, cos of the yaw change (float32 of the float64 values, as numpy casts them)
  const uint8_t* valid;  // (num_boxes)
};

__global__ __launch_bounds__(NT) void points_rigid_moves_kernel(float* __restrict__ points, int num_points, int stride,
                                                                 MoveArgs A, int num_boxes) {
  extern __shared__ double lds_moves[];  // centers [num_boxes][3] | loc [num_boxes][3] (f64) | planes [num_boxes][6][4] | sincos [num_boxes][2]
  double* l_c = lds_moves;
  double* l_t = l_c + num_boxes * 3;
  float* lds_planes = (float*)(l_t + num_boxes * 3);
  float* l_r = lds_planes + num_boxes * 24;
  for (int t = threadIdx.x; t < num_boxes * 24; t += NT) lds_planes[t] = A.planes[t];
  for (int t = threadIdx.x; t < num_boxes * 3; t += NT) { l_c[t] = A.centers[t]; l_t[t] = A.loc[t]; }
  for (int t = threadIdx.x; t < num_boxes * 2; t += NT) l_r[t] = A.sincos[t];
  __syncthreads();
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= num_points) return;
  float* q = points + (size_t)p * stride;
  const float x = q[0], y = q[1], z = q[2];
  for (int m = 0; m < num_boxes; ++m) {
    if (!A.valid[m]) continue;
    const float* pl = lds_planes + (size_t)m * 24;
    bool inside = true;
    for (int f = 0; f < 6; ++f) {
      const float sign = x * pl[f * 4 + 0] + y * pl[f * 4 + 1] + z * pl[f * 4 + 2] + pl[f * 4 + 3];
      if (sign >= 0.f) { inside = false; break; }
    }
    if (!inside) continue;
    // preprocess.py:551-558: p -= c; p = p @ R; p += c; p += t on the float32 row: the float64 operands make each += / -= a
    // float64 operation rounded back to float32; the rotation is float32 (rot_mat_T has the point dtype); no contraction here
    const float s = l_r[m * 2], c = l_r[m * 2 + 1];
    const float dx = (float)((double)x - l_c[m * 3]), dy = (float)((double)y - l_c[m * 3 + 1]), dz = (float)((double)z - l_c[m * 3 + 2]);
    const float rx = dx * c + dy * s, ry = dx * -s + dy * c;
    q[0] = (float)((double)(float)((double)rx + l_c[m * 3]) + l_t[m * 3]);
    q[1] = (float)((double)(float)((double)ry + l_c[m * 3 + 1]) + l_t[m * 3 + 1]);
    q[2] = (float)((double)(float)((double)dz + l_c[m * 3 + 2]) + l_t[m * 3 + 2]);
    break;  // the first valid box that contains the point
  }
}

__global__ __launch_bounds__(NT) void points_global_transform_kernel(float* __restrict__ points, int num_points, int stride,
                                                                      int flip, float s, float c, float scale,
                                                                      float* __restrict__ raw_copy) {
  const int p = blockIdx.x * NT + threadIdx.x;
  if (p >= num_points) return;
  float* q = points + (size_t)p * stride;
  if (raw_copy)
    for (int e = 0; e < stride; ++e) raw_copy[(size_t)p * stride + e] = q[e];
  float x = q[0], y = q[1], z = q[2];
  if (flip) y = -y;
  // points @ [[c, -s, 0], [s, c, 0], [0, 0, 1]] (box_np_ops.py:408-430), float32 products summed left to right
  const float xr = (x * c + y * s) + z * 0.f, yr = (x * -s + y * c) + z * 0.f, zr = (x * 0.f + y * 0.f) + z * 1.f;
  q[0] = xr * scale;
  q[1] = yr * scale;
  q[2] = zr * scale;
}

// order-preserving compaction, two launches: per-block keep counts, then scan of the counts + scatter
__global__ __launch_bounds__(NT) void compact_count_kernel(const uint8_t* __restrict__ keep, int n, int* __restrict__ blk_cnt) {
  __shared__ int sm[NT / 64];
  const int p = blockIdx.x * NT + threadIdx.x;
  int f = (p < n && keep[p]) ? 1 : 0;
  f = sessd_wave_sum(f);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < NT / 64; ++w) t += sm[w];
    blk_cnt[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(NT) void compact_scatter_kernel(const float* __restrict__ points, const uint8_t* __restrict__ keep,
                                                              int n, int stride, const int* __restrict__ blk_cnt, int nblk,
                                                              float* __restrict__ out, int out_cap, int* __restrict__ n_out) {
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
  const int p = blockIdx.x * NT + threadIdx.x;
  const int f = (p < n && keep[p]) ? 1 : 0;
  int tot;
  const int row = base + sessd_block_exscan<NT>(f, sm, &tot);
  if (f && row < out_cap)
    for (int e = 0; e < stride; ++e) out[(size_t)row * stride + e] = points[(size_t)p * stride + e];
  if ((int)blockIdx.x == nblk - 1 && threadIdx.x == 0) n_out[0] = min(base + tot, out_cap);
}

}  // namespace


// Iterative farthest-point sampling of k of n points (sa_da_v2.py's thinning step: `ifp_sample` over all-pairs neighbourhoods,
// restated in det3d/datasets/utils/sa_da_v2.py::ifp_sample): start at point 0; every pick lowers each point's distance-to-selection
// (Euclidean, float64 from the float32 coordinates, as scipy's cKDTree returns it); the next pick is the point with the largest
// remaining distance, lowest index on ties. One workgroup, distances in LDS.
namespace {
constexpr int FPS_MAX = 4096;
__global__ __launch_bounds__(NT) void fps_kernel(const float* __restrict__ pts, int n, int stride, int k, int* __restrict__ out) {
  __shared__ double rem[FPS_MAX];
  __shared__ double s_v[NT / 64];
  __shared__ int s_i[NT / 64];
  __shared__ int s_pick;
  for (int j = threadIdx.x; j < n; j += NT) rem[j] = INFINITY;
  __syncthreads();
  for (int s = 0; s < k; ++s) {
    double bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += NT) {
      const double v = rem[j];
      if (v > bv || (v == bv && j < bi)) { bv = v; bi = j; }   // first maximum of this thread's strided slice
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = bv; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double v = s_v[0];
      int i = s_i[0];
      for (int w = 1; w < NT / 64; ++w)
        if (s_v[w] > v || (s_v[w] == v && s_i[w] < i)) { v = s_v[w]; i = s_i[w]; }
      s_pick = i;
      out[s] = i;
    }
    __syncthreads();
    const int pick = s_pick;
    const double px = pts[(size_t)pick * stride], py = pts[(size_t)pick * stride + 1], pz = pts[(size_t)pick * stride + 2];
    for (int j = threadIdx.x; j < n; j += NT) {
      const double dx = (double)pts[(size_t)j * stride] - px, dy = (double)pts[(size_t)j * stride + 1] - py,
                   dz = (double)pts[(size_t)j * stride + 2] - pz;
      const double d = sqrt(dx * dx + dy * dy + dz * dz);
      rem[j] = j == pick ? -INFINITY : fmin(rem[j], d);
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" {

// points (num_points, point_stride) float32 (x, y, z first), planes (num_bodies, faces, 4) float32 [nx, ny, nz, d] with the
// normals pointing INTO the body -> out_mask (num_points, ceil(num_bodies/32)) uint32: bit (m & 31) of word m/32 is set iff
// the point is strictly inside body m (n.p + d < 0 for every face). num_bodies * faces <= 4096 (the planes live in LDS).
int sessd_points_in_bodies(const float* points, int num_points, int point_stride, const float* planes, int num_bodies,
                           int faces, uint32_t* out_mask, hipStream_t stream) {
  if (num_points < 0 || point_stride < 3 || num_bodies < 1 || faces < 1 || (long long)num_bodies * faces > 4096) return SESSD_EINVAL;
  if (num_points == 0) return SESSD_OK;
  const int words = sessd_divup(num_bodies, 32);
  const size_t lds = (size_t)num_bodies * faces * 4 * sizeof(float);
  SESSD_LAUNCH(points_in_bodies_kernel, dim3(sessd_divup(num_points, NT)), dim3(NT), lds, stream, points, num_points,
               point_stride, planes, num_bodies, faces, out_mask, words);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// In place: every point takes the rigid motion of the first VALID box that contains it. planes (num_boxes, 6, 4) as for
// sessd_points_in_bodies, centers / loc (num_boxes, 3) float64, sincos (num_boxes, 2) [sin, cos] of the yaw change, valid
// (num_boxes) bytes. num_boxes <= 128.
int sessd_points_rigid_moves(float* points, int num_points, int point_stride, const float* planes, const double* centers,
                             const double* loc, const float* sincos, const uint8_t* valid, int num_boxes, hipStream_t stream) {
  if (num_points < 0 || point_stride < 3 || num_boxes < 0 || num_boxes > 128) return SESSD_EINVAL;
  if (num_points == 0 || num_boxes == 0) return SESSD_OK;
  MoveArgs A{planes, centers, loc, sincos, valid};
  const size_t lds = (size_t)num_boxes * ((3 + 3) * sizeof(double) + (24 + 2) * sizeof(float));
  SESSD_LAUNCH(points_rigid_moves_kernel, dim3(sessd_divup(num_points, NT)), dim3(NT), lds, stream, points, num_points,
               point_stride, A, num_boxes);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

// In place: y -> -y if flip, rotation by the angle whose (sin, cos) are given, then the scale; raw_copy (same shape, may be
// NULL) receives the cloud as it was before.
int sessd_points_global_transform(float* points, int num_points, int point_stride, int flip, float sin_angle, float cos_angle,
                                  float scale, float* raw_copy, hipStream_t stream) {
  if (num_points < 0 || point_stride < 3) return SESSD_EINVAL;
  if (num_points == 0) return SESSD_OK;
  SESSD_LAUNCH(points_global_transform_kernel, dim3(sessd_divup(num_points, NT)), dim3(NT), 0, stream, points, num_points,
               point_stride, flip, sin_angle, cos_angle, scale, raw_copy);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

size_t sessd_points_compact_workspace_bytes(int num_points) { return sessd_align((size_t)sessd_divup(num_points > 0 ? num_points : 1, NT) * 4 + 4, 256); }

// out[0 .. *n_out) = the rows of points whose keep byte is non-zero, in order; *n_out (device int) <= out_capacity.
int sessd_points_compact(const float* points, const uint8_t* keep, int num_points, int point_stride, float* out, int out_capacity,
                         int* n_out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (num_points < 0 || point_stride < 1 || out_capacity < 0) return SESSD_EINVAL;
  const int nblk = sessd_divup(num_points > 0 ? num_points : 1, NT);
  if (sessd_points_compact_workspace_bytes(num_points) > workspace_bytes) return SESSD_EWORKSPACE;
  int* blk = (int*)workspace;
  SESSD_LAUNCH(compact_count_kernel, dim3(nblk), dim3(NT), 0, stream, keep, num_points, blk);
  SESSD_CHECK_LAUNCH();
  SESSD_LAUNCH(compact_scatter_kernel, dim3(nblk), dim3(NT), 0, stream, points, keep, num_points, point_stride, blk, nblk, out,
               out_capacity, n_out);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}


// Farthest-point sampling: indices (k, int32, device) of k of the n points (n, point_stride floats, xyz first), starting at point
// 0, ties to the lowest index; k <= n <= 4096.
int sessd_farthest_point_sample(const float* points, int num_points, int point_stride, int k, int* out_indices, hipStream_t stream) {
  if (num_points < 1 || num_points > FPS_MAX || point_stride < 3 || k < 1 || k > num_points) return SESSD_EINVAL;
  SESSD_LAUNCH(fps_kernel, dim3(1), dim3(NT), 0, stream, points, num_points, point_stride, k, out_indices);
  SESSD_CHECK_LAUNCH();
  return SESSD_OK;
}

}  // extern "C"
