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
// STATUS: written in round 1 after the GPU budget was spent -- compiled, not yet run on hardware;
// tests/test_datapath_gpu.py runs only with SESSD_EXPERIMENTAL=1. The host stage (numpy) is what the pipeline uses.
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

}  // extern "C"
