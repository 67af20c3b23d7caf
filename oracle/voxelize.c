/* TEST INFRASTRUCTURE ONLY -- CPU oracle, never on the product path.
 *
 * Plain-C restatement of the reference voxelizer
 *   det3d/ops/point_cloud/point_cloud_ops_v2.py:9-62  (_points_to_voxel_reverse_kernel)
 *   det3d/ops/point_cloud/point_cloud_ops_v2.py:120-194 (points_to_voxel wrapper)
 * and of the mean reader det3d/models/readers/voxel_encoder.py:215-220.
 *
 * Pinned against the reference itself: tests/golden/make_golden.py runs the
 * reference's own Python source (numba's decorator stubbed to identity) and
 * tests/test_oracle_golden.py compares this file with those vectors bit for bit.
 *
 * One deliberate widening: the reference's scratch map is uint16 with 65535 as
 * "empty" (line 6), which caps max_voxels at 65534. An int32 map with -1 behaves
 * identically below that cap and also covers the 64k-voxel stress configuration.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* returns voxel_num, or -1 on allocation failure.
 * voxels: (max_voxels, max_points, ndim) zero-initialised by this function
 * coors:  (max_voxels, 3) int32 [z,y,x]; num_points_per_voxel: (max_voxels) */
int oracle_points_to_voxel(const float *points, int N, int ndim, const float *voxel_size, const float *coors_range,
                           int max_points, int max_voxels, float *voxels, int32_t *coors,
                           int32_t *num_points_per_voxel) {
  int32_t grid[3];
  for (int j = 0; j < 3; ++j) {
    /* float32 arithmetic then round-half-even, as np.round on a float32 array (lines 27-30) */
    float g = (coors_range[3 + j] - coors_range[j]) / voxel_size[j];
    grid[j] = (int32_t)rintf(g);
  }
  size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  /* like the reference's module-global COOR_TO_VOXELIDX2 (line 6): allocated once, kept all-empty between
   * calls by resetting only the touched cells (lines 58-61) */
  static int32_t *map = NULL;
  static size_t map_cells = 0;
  if (map_cells != cells) {
    free(map);
    map = (int32_t *)malloc(cells * sizeof(int32_t));
    if (!map) { map_cells = 0; return -1; }
    memset(map, 0xFF, cells * sizeof(int32_t)); /* -1 == empty */
    map_cells = cells;
  }
  memset(voxels, 0, (size_t)max_voxels * max_points * ndim * sizeof(float));
  memset(coors, 0, (size_t)max_voxels * 3 * sizeof(int32_t));
  memset(num_points_per_voxel, 0, (size_t)max_voxels * sizeof(int32_t));

  int voxel_num = 0;
  for (int i = 0; i < N; ++i) {
    int32_t coor[3]; /* reversed: z, y, x */
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      /* line 37: float32 subtract, float32 divide, floor */
      volatile float d = points[(size_t)i * ndim + j] - coors_range[j];
      volatile float q = d / voxel_size[j];
      float c = floorf(q);
      if (c < 0 || c >= (float)grid[j]) {
        failed = 1;
        break;
      }
      coor[2 - j] = (int32_t)c;
    }
    if (failed) continue;
    size_t lin = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
    int32_t voxelidx = map[lin];
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (voxel_num >= max_voxels) break; /* line 46-47: break, not continue */
      voxel_num += 1;
      map[lin] = voxelidx;
      coors[voxelidx * 3 + 0] = coor[0];
      coors[voxelidx * 3 + 1] = coor[1];
      coors[voxelidx * 3 + 2] = coor[2];
    }
    int32_t num = num_points_per_voxel[voxelidx];
    if (num < max_points) {
      memcpy(voxels + ((size_t)voxelidx * max_points + num) * ndim, points + (size_t)i * ndim, ndim * sizeof(float));
      num_points_per_voxel[voxelidx] += 1;
    }
  }
  for (int v = 0; v < voxel_num; ++v)
    map[((size_t)coors[v * 3] * grid[1] + coors[v * 3 + 1]) * grid[0] + coors[v * 3 + 2]] = -1;
  return voxel_num;
}

/* voxel_encoder.py:219: voxels[:, :, :nfeat].sum(dim=1) / num_points (zeros included in the sum) */
void oracle_vfe_mean(const float *voxels, const int32_t *num_points, int M, int max_points, int ndim, int nfeat,
                     float *out) {
  for (int v = 0; v < M; ++v) {
    for (int d = 0; d < nfeat; ++d) {
      float s = 0.f;
      for (int r = 0; r < max_points; ++r) {
        volatile float t = s + voxels[((size_t)v * max_points + r) * ndim + d];
        s = t;
      }
      out[(size_t)v * nfeat + d] = s / (float)num_points[v];
    }
  }
}
