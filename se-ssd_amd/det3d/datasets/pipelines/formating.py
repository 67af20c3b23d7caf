"""mirrors det3d/datasets/pipelines/formating.py:14-86 (Reformat): flatten the pipeline result into the `example` dict that
collate_kitti / VoxelNet.forward / MultiGroupHead.loss consume (validation keys, and for labelled training samples the
targets of AssignTarget, their `*_raw` twins for the teacher, and the recorded global augmentation)."""
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
        lidar = res["lidar"]
        if "anchors" in lidar.get("targets", {}):
            data_bundle["anchors"] = lidar["targets"]["anchors"]
        if "voxels_raw" in lidar and "anchors" in lidar.get("targets_raw", {}):
            data_bundle["anchors_raw"] = lidar["targets_raw"]["anchors"]
        if "anchors_mask" in lidar.get("targets", {}):
            data_bundle["anchors_mask"] = lidar["targets"]["anchors_mask"]
        if "calib" in res:
            data_bundle["calib"] = res["calib"]
        mode = res.get("mode", "val")
        if mode != "test" and "annotations" in lidar:
            data_bundle["annos"] = lidar["annotations"]
            if "annotations_raw" in lidar:
                data_bundle["annos_raw"] = lidar["annotations_raw"]
        if mode == "train" and res.get("labeled", True):
            t = lidar.get("targets", {})
            for src, dst in (("labels", "labels"), ("reg_targets", "reg_targets"), ("reg_weights", "reg_weights")):
                if t.get(src) is not None:
                    data_bundle[dst] = t[src]
            data_bundle["positive_gt_id"] = dict(positive_gt_id=t.get("positive_gt_id"))
            if "ground_plane" in lidar:
                data_bundle["ground_plane"] = lidar["ground_plane"]
            if "labels" in lidar.get("targets_raw", {}):
                r = lidar["targets_raw"]
                data_bundle.update(labels_raw=r["labels"], reg_targets_raw=r["reg_targets"], reg_weights_raw=r["reg_weights"],
                                   positive_gt_id_raw={"positive_gt_id": r["positive_gt_id"]}, transformation=lidar["transformation"])
        elif mode == "train":
            data_bundle["transformation"] = lidar["transformation"]
        return data_bundle, info
