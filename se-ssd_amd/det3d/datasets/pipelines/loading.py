"""mirrors det3d/datasets/pipelines/loading.py:73-160 for KITTI: the first two stages of both pipelines. Wire format (SURVEY
8f row 4): `velodyne_reduced/*.bin` = float32 x num_point_features per point (image-frustum-reduced cloud, preferred over the
full `velodyne/` file when it exists); annotations in the rectified camera frame inside the `kitti_infos_*.pkl` entries."""
from pathlib import Path

import numpy as np

from det3d.core.bbox import box_np_ops
from det3d.datasets.kitti import kitti_common as kitti

from ..registry import PIPELINES


@PIPELINES.register_module
class LoadPointCloudFromFile(object):
    def __init__(self, dataset="KittiDataset", **kwargs):
        self.type = dataset

    def __call__(self, res, info):
        res["type"] = self.type
        if self.type != "KittiDataset":
            raise NotImplementedError
        path = Path(info["point_cloud"]["velodyne_path"])
        if not path.is_absolute():
            path = Path(res["metadata"]["image_prefix"]) / info["point_cloud"]["velodyne_path"]
        reduced = path.parent.parent / (path.parent.stem + "_reduced") / path.name
        if reduced.exists():
            path = reduced
        res["lidar"]["points"] = np.fromfile(str(path), dtype=np.float32, count=-1).reshape([-1, res["metadata"]["num_point_features"]])
        return res, info


@PIPELINES.register_module
class LoadPointCloudAnnotations(object):
    """calibration (with the image frustum as six inward-facing planes, what the head's post-filter consumes) and the labelled
    boxes: DontCare removed, camera [x,y,z,l,h,w,ry] -> lidar [x,y,z,w,l,h,ry], z moved from the bottom face to the centre."""

    def __init__(self, with_bbox=True, **kwargs):
        self.enable_difficulty_level = kwargs.get("enable_difficulty_level", False)

    def __call__(self, res, info):
        if res["type"] != "KittiDataset":
            raise NotImplementedError
        calib = info["calib"]
        res["calib"] = {"rect": calib["R0_rect"], "Trv2c": calib["Tr_velo_to_cam"], "P2": calib["P2"],
                        "frustum": box_np_ops.get_valid_frustum(calib["R0_rect"], calib["Tr_velo_to_cam"], calib["P2"],
                                                                info["image"]["image_shape"])}
        if "annos" in info:
            annos = kitti.remove_dontcare(info["annos"])
            cam = np.concatenate([annos["location"], annos["dimensions"], annos["rotation_y"][..., np.newaxis]], axis=1).astype(np.float32)
            boxes = box_np_ops.box_camera_to_lidar(cam, calib["R0_rect"], calib["Tr_velo_to_cam"])
            box_np_ops.change_box3d_center_(boxes, [0.5, 0.5, 0], [0.5, 0.5, 0.5])
            res["lidar"]["annotations"] = {"boxes": boxes, "names": annos["name"]}
            if self.enable_difficulty_level:
                res["lidar"]["annotations"]["difficulty"] = annos["difficulty"]
            res["cam"]["annotations"] = {"boxes": annos["bbox"], "names": annos["name"]}
        return res, info
