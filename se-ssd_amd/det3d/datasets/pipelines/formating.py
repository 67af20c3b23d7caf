"""mirrors the inference part of det3d/datasets/pipelines/formating.py:14-86 (Reformat): flatten the pipeline result
into the `example` dict that collate_kitti / VoxelNet.forward consume."""
from ..registry import PIPELINES


@PIPELINES.register_module
class Reformat(object):
    def __init__(self, **kwargs):
        pass

    def __call__(self, res, info):
        meta = res["metadata"]
        points = res["lidar"]["points"]
        voxels = res["lidar"]["voxels"]
        data_bundle = dict(metadata=meta, points=points, voxels=voxels["voxels"], shape=voxels["shape"],
                           num_points=voxels["num_points"], num_voxels=voxels["num_voxels"],
                           coordinates=voxels["coordinates"])
        if "voxels_raw" in res["lidar"]:
            vr = res["lidar"]["voxels_raw"]
            data_bundle.update(points_raw=res["lidar"]["points_raw"], voxels_raw=vr["voxels"], shape_raw=vr["shape"],
                               num_points_raw=vr["num_points"], num_voxels_raw=vr["num_voxels"],
                               coordinates_raw=vr["coordinates"])
        if "anchors" in res["lidar"].get("targets", {}):
            data_bundle["anchors"] = res["lidar"]["targets"]["anchors"]
        if "calib" in res:
            data_bundle["calib"] = res["calib"]
        return data_bundle, info
