"""mirrors det3d/datasets/pipelines/preprocess.py:178-232 (Voxelization): the pipeline stage that feeds the hot path.
Same result dict (`res["lidar"]["voxels"]` = voxels / coordinates / num_points / num_voxels / shape, plus the `*_raw`
twin when `points_raw` is present), voxelized on the MI355X through VoxelGenerator.generate -> sessd_voxelize_frame.
The training-only ground-truth range filter of the reference (:200-206) belongs to data augmentation and is applied
only if the caller provides `filter_gt_box_outside_range` (out of scope here)."""
import numpy as np

from det3d.core.input.voxel_generator import VoxelGenerator

from ..registry import PIPELINES


def _get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


@PIPELINES.register_module
class Voxelization(object):
    def __init__(self, **kwargs):
        cfg = kwargs.get("cfg", None)
        self.range = _get(cfg, "range")
        self.voxel_size = _get(cfg, "voxel_size")
        self.max_points_in_voxel = _get(cfg, "max_points_in_voxel")
        self.max_voxel_num = _get(cfg, "max_voxel_num")
        self.far_points_first = _get(cfg, "far_points_first", False)
        self.shuffle = False
        self.voxel_generator = VoxelGenerator(point_cloud_range=self.range, voxel_size=self.voxel_size,
                                              max_num_points=self.max_points_in_voxel, max_voxels=self.max_voxel_num)

    def _voxelize(self, points, grid_size):
        voxels, coordinates, num_points_per_voxel = self.voxel_generator.generate(points)
        return dict(voxels=voxels, coordinates=coordinates, num_points=num_points_per_voxel,
                    num_voxels=np.array([voxels.shape[0]], dtype=np.int64), shape=grid_size)

    def __call__(self, res, info):
        grid_size = self.voxel_generator.grid_size
        res["lidar"]["voxels"] = self._voxelize(res["lidar"]["points"], grid_size)
        if "points_raw" in res["lidar"].keys():
            res["lidar"]["voxels_raw"] = self._voxelize(res["lidar"]["points_raw"], grid_size)
        return res, info
