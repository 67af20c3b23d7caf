"""VoxelGenerator (det3d/core/input/voxel_generator.py:10-48): holds the voxel grid definition in float32 exactly as the
reference does (grid = round((hi - lo) / voxel_size) on float32 arrays) and voxelizes on the MI355X."""
import numpy as np

from det3d.ops.point_cloud.point_cloud_ops_v2 import points_to_voxel


class VoxelGenerator:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        self._vs = np.asarray(voxel_size, dtype=np.float32)
        self._pcr = np.asarray(point_cloud_range, dtype=np.float32)
        extent = self._pcr[3:] - self._pcr[:3]
        self._grid = np.round(extent / self._vs).astype(np.int64)  # (x, y, z) cells
        self._cap_points = int(max_num_points)
        self._cap_voxels = int(max_voxels)

    def generate(self, points, max_voxels=20000):
        """points (P, >=3) float32 -> voxels (M, max_num_points, ndim), coordinates (M, 3) zyx, num_points (M,).
        NB: like the reference, the per-call `max_voxels` argument is ignored in favour of the constructor's."""
        return points_to_voxel(points, self._vs, self._pcr, self._cap_points, True, self._cap_voxels)

    voxel_size = property(lambda self: self._vs)
    point_cloud_range = property(lambda self: self._pcr)
    grid_size = property(lambda self: self._grid)
    max_num_points_per_voxel = property(lambda self: self._cap_points)
